#!/bin/bash
# SQ stall breakdown (one PMC pass) of every kernel of a short eager run.
# usage: scripts/gpu_pmc_sq.sh <tag> <model> [counters...]
TAG=$1; m=$2; shift 2
CTRS=${@:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
P=/tmp/pmcsq_${TAG}_$m
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -f csv -d $P -o $m -- python $R/bench.py --model $m --steps 4 --warmup 1 --no-graph --no-cpu-baseline --no-host-fed --no-kernel-timing --no-tunable --sweep-batches 0 > $O/pmcsq_${TAG}_$m.log 2>&1)
tail -c 300 $O/pmcsq_${TAG}_$m.log > $O/pmcsq_${TAG}_$m.tail; rm -f $O/pmcsq_${TAG}_$m.log
python $R/scripts/pmc_summary.py $P > $O/${TAG}_${m}_pmc_sq.md 2>&1
rm -rf $P
