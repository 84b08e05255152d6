#!/bin/bash
# rocprofv3 kernel stats of one model's captured step (bench.py, bare step): usage scripts/gpu_prof_model.sh <model> <tag>
set -u
cd "$(dirname "$0")/.."
M=${1:-dcn}; T=${2:-pm}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD; D=/tmp/prof_${T}
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $D -o $M -- python $R/bench.py --model $M --steps 200 --warmup 20 --no-cpu-baseline --no-host-fed --no-extra-models --no-kernel-timing --sweep-batches 0 > $R/gpurun_out/${T}_prof.log 2>&1)
DB=$(find $D -name "*_results.db" | head -1)
if [ -n "$DB" ]; then python scripts/rocpd_stats.py $DB 40 > gpurun_out/${T}_${M}_kernel_stats.md; fi
tail -1 gpurun_out/${T}_prof.log | cut -c1-200
rm -rf $D gpurun_out/${T}_prof.log
head -24 gpurun_out/${T}_${M}_kernel_stats.md | cut -c1-160
