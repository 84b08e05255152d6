#!/bin/bash
R=$PWD; O=$R/gpurun_out; export TMPDIR=/tmp
for d in 0 1 2 3 4; do
  D=/tmp/prof_dbg_$d
  (cd /tmp && RECALGO_SPARSE_DBG=$d timeout 200 rocprofv3 --kernel-trace --stats -d $D -o sp -- python $R/scripts/bench_sparse.py --mode grad --steps 100 > $O/dbg_$d.log 2>&1)
  DB=$(find $D -name "*_results.db" | head -1)
  echo "dbg=$d"; python $R/scripts/rocpd_stats.py $DB 12 | grep "sparse_" | cut -c1-160
  rm -rf $D
done
