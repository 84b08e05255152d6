#!/bin/bash
# SQ / MFMA counters of the dense kernels at one shape (one PMC pass per counter group).
# usage: scripts/gpu_pmc_dense.sh <tag> M K N
TAG=$1; M=$2; K=$3; N=$4
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1)); P=/tmp/pmcd_${TAG}_$i
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $CTRS -f csv -d $P -o d -- python $R/scripts/prof_dense.py $M $K $N 3 > $O/pmcd_${TAG}_$i.log 2>&1)
  python $R/scripts/pmc_summary.py $P > $O/${TAG}_dense_${M}x${K}x${N}_pmc$i.md 2>&1
  tail -c 400 $O/pmcd_${TAG}_$i.log > $O/pmcd_${TAG}_$i.tail; rm -f $O/pmcd_${TAG}_$i.log; rm -rf $P
done
