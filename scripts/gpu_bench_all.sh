#!/bin/bash
# Run on the GPU box (via gpurun): bench lines + rocprofv3 kernel traces for every model.
# usage: scripts/gpu_bench_all.sh <tag> [models...]
TAG=${1:-r01}; shift
MODELS=${@:-dcn deepfm xdeepfm din fibinet pnn}
R=$PWD
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
for m in $MODELS; do
  timeout 300 python bench.py --model $m --steps 200 --warmup 20 > $R/gpurun_out/bench_${TAG}_$m.json 2> $R/gpurun_out/bench_${TAG}_$m.err
  echo "== $m: $(head -c 700 $R/gpurun_out/bench_${TAG}_$m.json)"
  tail -2 $R/gpurun_out/bench_${TAG}_$m.err
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_$m -o $m -- python $R/bench.py --model $m --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_${TAG}_$m.log 2>&1)
done
ls $R/gpurun_out
