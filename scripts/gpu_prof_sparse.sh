#!/bin/bash
# GPU box: rocprofv3 kernel stats of the sparse micro-benchmark (scripts/bench_sparse.py --advance) for a list of variants:
#   scripts/gpu_prof_sparse.sh <tag> "<env assignments>|<bench_sparse args>" ...
# Leaves gpurun_out/<tag>_<n>_sparse_stats.md per variant.
TAG=$1; shift
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
n=0
for V in "$@"; do
  n=$((n+1))
  ENVS="${V%%|*}"; ARGS="${V#*|}"
  D=/tmp/prof_${TAG}_$n
  (cd /tmp && env $ENVS timeout 200 rocprofv3 --kernel-trace --stats -d $D -o sp -- python $R/scripts/bench_sparse.py --advance --steps 100 $ARGS > $O/${TAG}_${n}.log 2>&1)
  DB=$(find $D -name "*_results.db" | head -1)
  echo "== variant $n: $ENVS | $ARGS"
  tail -1 $O/${TAG}_${n}.log | cut -c1-300
  if [ -n "$DB" ]; then python $R/scripts/rocpd_stats.py $DB 12 > $O/${TAG}_${n}_sparse_stats.md; grep -E "sparse_" $O/${TAG}_${n}_sparse_stats.md | cut -c1-170; fi
  rm -rf $D $O/${TAG}_${n}.log
done
