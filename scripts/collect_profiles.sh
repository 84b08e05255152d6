#!/bin/bash
# copy the text summaries of an evidence run (scripts/gpu_round_final.sh <tag>, merged into gpurun_out/) into profiles/ under the
# names the docs and bench.py quote, and rebuild profiles/pmc_traffic.json.   usage: scripts/collect_profiles.sh <tag>
set -eu
cd "$(dirname "$0")/.."
T=${1:-r06}; O=gpurun_out; P=profiles
for f in $O/bench_${T}_*.json; do b=$(basename $f .json); cp $f $P/${T}_bench_${b#bench_${T}_}.json; done
for f in $O/${T}_*_kernel_stats.md $O/${T}_*pmc*.md $O/${T}_din_lab.md $O/${T}_dropout_cost.md $O/${T}_tailfuse_lab.md \
         $O/${T}_fork_join_lab.md $O/${T}_smoke.log; do [ -f $f ] && cp $f $P/; done
grep -v "^| tests" $O/${T}_pytest_gpu.log | tail -40 > $P/${T}_pytest_gpu.log
N=$(grep -oE "[0-9]+ passed" $O/${T}_pytest_gpu.log | tail -1 | cut -d' ' -f1)
rm -f $P/${T}_strict_parity_all_gpu_tests_*_tests.md
cp $O/${T}_strict_parity_all_gpu_tests.md $P/${T}_strict_parity_all_gpu_tests_${N}_tests.md
python scripts/pmc_traffic_json.py $P/${T}_dcn_pmc_fullrun.md dcn "dense_bwd_kernel<true, false>" "dense_bwd(416->512)" profiles/${T}_dcn_pmc_fullrun.md 32112640
ls $P | grep "^${T}_" | wc -l
