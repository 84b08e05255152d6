#!/bin/bash
# A/B on ONE box: the 100 M-row DeepFM step under different sweep block sizes (RECALGO_SPARSE_SWEEP_BLOCK_SHIFT), with the
# in-step time of sparse_prepare from rocprofv3.   usage: scripts/gpu_ab_bigtable.sh <shift> [<shift> ...]   ("auto" = default)
R=$PWD; O=$R/gpurun_out; export TMPDIR=/tmp
for S in "$@"; do
  if [ "$S" == "auto" ]; then unset RECALGO_SPARSE_SWEEP_BLOCK_SHIFT; else export RECALGO_SPARSE_SWEEP_BLOCK_SHIFT=$S; fi
  D=/tmp/prof_big_$S
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $D -o d -- python $R/bench.py --model deepfm --big-table-rows 100000000 --steps 100 --warmup 10 --no-cpu-baseline --no-host-fed --no-kernel-timing --no-extra-models --sweep-batches 0 > $O/big_$S.log 2>&1)
  echo "== shift $S: $(grep -o '"ms_per_step": [0-9.]*' $O/big_$S.log | head -1) $(grep -o '"hbm_copy_GBs": [0-9.]*' $O/big_$S.log | head -1)"
  python $R/scripts/rocpd_stats.py $(find $D -name "*_results.db" | head -1) 8 | grep -E "sparse_prepare" | cut -c1-50,90-170 | head -3
  rm -rf $D $O/big_$S.log
done
