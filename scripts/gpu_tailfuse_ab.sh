#!/bin/bash
# fused DCN tail (recalgo_tail_dense_head_fwd_bwd): parity tests, the step time, and a kernel trace of the captured step
# usage: scripts/gpu_tailfuse_ab.sh [tag]
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${1:-tf}
timeout 1200 python -m pytest tests/test_gpu_tailfuse.py tests/test_gpu_models.py tests/test_gpu_baseline_shapes.py -x -q 2>&1 | tail -60 > gpurun_out/${T}_pytest.log
tail -3 gpurun_out/${T}_pytest.log
for i in 1 2 3; do
  timeout 300 python bench.py --model dcn --steps 400 --warmup 40 --no-cpu-baseline --no-host-fed --no-extra-models --no-kernel-timing --sweep-batches 0 2> gpurun_out/${T}_bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dcn', d['value'], d['ms_per_step'])"
done
export TMPDIR=/tmp
R=$PWD; D=/tmp/prof_${T}
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $D -o dcn -- python $R/bench.py --model dcn --steps 200 --warmup 20 --no-cpu-baseline --no-host-fed --no-extra-models --no-kernel-timing --sweep-batches 0 > $R/gpurun_out/${T}_prof.log 2>&1)
DB=$(find $D -name "*_results.db" | head -1)
if [ -n "$DB" ]; then python scripts/rocpd_stats.py $DB 40 > gpurun_out/${T}_kernel_stats.md; fi
rm -rf $D gpurun_out/${T}_prof.log
head -45 gpurun_out/${T}_kernel_stats.md
