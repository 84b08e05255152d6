#!/bin/bash
# DIN step under rocprofv3 for several bucket counts of the sparse plan (RECALGO_SPARSE_NB_LOG2)
R=$PWD; O=$R/gpurun_out; export TMPDIR=/tmp
for NB in "$@"; do
  D=/tmp/prof_din_$NB
  (cd /tmp && RECALGO_SPARSE_NB_LOG2=$NB timeout 200 rocprofv3 --kernel-trace --stats -d $D -o din -- python $R/bench.py --model din --steps 50 --warmup 10 --no-cpu-baseline --no-host-fed --no-kernel-timing --no-extra-models --sweep-batches 0 > $O/din_nb$NB.log 2>&1)
  DB=$(find $D -name "*_results.db" | head -1)
  echo "== nb_log2 $NB: $(grep -o '"ms_per_step": [0-9.]*' $O/din_nb$NB.log | head -1)"
  python $R/scripts/rocpd_stats.py $DB 8 | grep -E "sparse_" | cut -c1-150 | head -4
  rm -rf $D $O/din_nb$NB.log
done
