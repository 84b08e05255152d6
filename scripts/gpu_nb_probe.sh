#!/bin/bash
# In-step sparse kernel times of a model for several bucket counts of the scatter plan (RECALGO_SPARSE_NB_LOG2), one box.
# usage: scripts/gpu_nb_probe.sh <model> <nb_log2> [<nb_log2> ...]
M=$1; shift
R=$PWD; O=$R/gpurun_out; export TMPDIR=/tmp
for NB in "$@"; do
  D=/tmp/prof_${M}_$NB
  (cd /tmp && RECALGO_SPARSE_NB_LOG2=$NB timeout 200 rocprofv3 --kernel-trace --stats -d $D -o $M -- python $R/bench.py --model $M --steps 100 --warmup 10 --no-cpu-baseline --no-host-fed --no-kernel-timing --no-extra-models --sweep-batches 0 > $O/${M}_nb$NB.log 2>&1)
  DB=$(find $D -name "*_results.db" | head -1)
  echo "== $M nb_log2 $NB: $(grep -o '"ms_per_step": [0-9.]*' $O/${M}_nb$NB.log | head -1)"
  python $R/scripts/rocpd_stats.py $DB 12 | grep -E "sparse_|adam_tf1" | cut -c1-45,95-150 | head -4
  rm -rf $D $O/${M}_nb$NB.log
done
