#!/bin/bash
# PMC passes over a FULL bench run (training steps + the isolated kernel launches of
# kernel_rooflines, i.e. the same state `roofline.achieved` is measured in).
# usage: scripts/gpu_pmc_bench.sh <tag> <model> [steps]
TAG=$1; m=$2; STEPS=${3:-200}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  P=/tmp/pmcb_${TAG}_${m}_$C
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C -f csv -d $P -o $m -- python $R/bench.py --model $m --steps $STEPS --warmup 20 --no-cpu-baseline --no-host-fed --no-graph --sweep-batches 0 > $O/pmcb_${TAG}_${m}_$C.log 2>&1)
  tail -c 400 $O/pmcb_${TAG}_${m}_$C.log > $O/pmcb_${TAG}_${m}_$C.tail; rm -f $O/pmcb_${TAG}_${m}_$C.log
done
python $R/scripts/pmc_summary.py /tmp/pmcb_${TAG}_${m}_FETCH_SIZE /tmp/pmcb_${TAG}_${m}_WRITE_SIZE > $O/${TAG}_${m}_pmc_fullrun.md 2>&1
rm -rf /tmp/pmcb_${TAG}_${m}_*
head -12 $O/${TAG}_${m}_pmc_fullrun.md | cut -c1-200
