#!/bin/bash
# A/B of two builds of librecalgo_hip.so on ONE box (RECALGO_HIP_LIB): bench line + in-step sparse kernel times per model.
# usage: scripts/gpu_ab_lib.sh <variant .so> <models...>
V=$1; shift
R=$PWD; O=$R/gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for L in default $V; do
  if [ "$L" == "default" ]; then unset RECALGO_HIP_LIB; else export RECALGO_HIP_LIB=$R/$L; fi
  for m in "$@"; do
    D=/tmp/prof_ab_$m
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $D -o d -- python $R/bench.py --model $m --steps 200 --warmup 20 --no-cpu-baseline --no-host-fed --no-kernel-timing --no-extra-models --sweep-batches 0 > $O/ab.log 2>&1)
    echo "== $(basename $L) $m: $(grep -o '"ms_per_step": [0-9.]*' $O/ab.log | head -1) $(python $R/scripts/rocpd_stats.py $(find $D -name "*_results.db" | head -1) 12 | grep -E "sparse_prepare" | head -1 | cut -c95-130)"
    rm -rf $D $O/ab.log
  done
done
done
