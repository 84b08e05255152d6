#!/bin/bash
# rocprofv3 kernel stats of the sparse-path micro-benchmark (scripts/bench_sparse.py) in a few configurations
TAG=${1:-r03}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for cfg in "zipf:" "uniform:--uniform" "zipf_advance:--advance" "lazy:--mode lazy" "grad:--mode grad"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  D=/tmp/prof_sparse_$name
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $D -o sp -- python $R/scripts/bench_sparse.py $flags > $O/${TAG}_sparse_$name.log 2>&1)
  DB=$(find $D -name "*_results.db" | head -1)
  if [ -n "$DB" ]; then python $R/scripts/rocpd_stats.py $DB 12 > $O/${TAG}_sparse_${name}_kernel_stats.md; fi
  grep "^{" $O/${TAG}_sparse_$name.log | tail -1
  grep "sparse_" $O/${TAG}_sparse_${name}_kernel_stats.md | cut -c1-200
  rm -rf $D
done
