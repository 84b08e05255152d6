#!/bin/bash
# quick GPU check of a kernel change: the dense / model parity tests, then the bare step time of the named models
# usage: scripts/gpu_quick.sh "<pytest args>" "<models>" [tag]
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${3:-q}
if [ -n "${1:-}" ]; then
  timeout 900 python -m pytest $1 -x -q 2>&1 | tail -120 > gpurun_out/${T}_pytest.log
  tail -3 gpurun_out/${T}_pytest.log
fi
for m in ${2:-dcn}; do
  timeout 300 python bench.py --model $m --steps 200 --warmup 20 --no-cpu-baseline --no-host-fed --no-extra-models --no-kernel-timing --sweep-batches 0 > gpurun_out/${T}_bench_$m.json 2> gpurun_out/${T}_bench_$m.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_bench_$m.json").read().strip().splitlines()[-1])
    print("$m", d["value"], d["ms_per_step"])
except Exception as e:
    print("$m", "FAILED", e)
PY
done
