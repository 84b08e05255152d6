#!/bin/bash
# what sparse_prepare's in-step time is made of: the DCN step with exact TF1 Adam at sweep periods 32 (default) and 96, and with
# LazyAdam (no catch-up, no sweep: the count tiles alone) — rocprofv3 averages of the sparse kernels
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD
run() {  # tag, extra bench args, env
  D=/tmp/prof_ps_$1
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $D -o dcn -- python $R/bench.py --model dcn --steps 200 --warmup 20 --no-cpu-baseline --no-host-fed --no-extra-models --no-kernel-timing --sweep-batches 0 $2 > /tmp/ps_$1.log 2>&1)
  DB=$(find $D -name "*_results.db" | head -1)
  echo "== $1: $(tail -1 /tmp/ps_$1.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step")' 2>/dev/null)"
  python scripts/rocpd_stats.py $DB 40 | grep -E "sparse_(prepare|place|apply)" | head -3 | cut -c1-60,100-170
  rm -rf $D
}
run period32 ""
RECALGO_ADAM_SWEEP_PERIOD=96 run period96 ""
run lazy "--lazy-adam"
