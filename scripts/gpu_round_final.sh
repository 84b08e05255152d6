#!/bin/bash
# Round-end evidence on the GPU box (via gpurun): full GPU test suite, smoke, the default bench line, per-model bench
# lines + rocprofv3 kernel stats, the PMC passes of the default model (HBM traffic over a full run; SQ / MFMA counters)
# and the 100 M-row table run.  Leaves only text summaries under gpurun_out/.
# usage: scripts/gpu_round_final.sh <tag>
TAG=${1:-r02}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q > $O/${TAG}_pytest_gpu_full.log 2>&1; grep -v "^| tests" $O/${TAG}_pytest_gpu_full.log | tail -40 > $O/${TAG}_pytest_gpu.log
cp $O/strict_parity.md $O/${TAG}_strict_parity_all_gpu_tests.md 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${TAG}_smoke.log 2>&1
timeout 400 python bench.py > $O/bench_${TAG}_default.json 2> $O/bench_${TAG}_default.err
# rocprofv3 --kernel-trace --stats of THE SAME default command (agreement with the live HIP-event roofline numbers)
D=/tmp/prof_${TAG}_default
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $D -o dcn -- python $R/bench.py --no-cpu-baseline --no-host-fed > $O/prof_${TAG}_default.log 2>&1)
DB=$(find $D -name "*_results.db" | head -1)
if [ -n "$DB" ]; then python $R/scripts/rocpd_stats.py $DB 40 > $O/${TAG}_default_kernel_stats.md; fi
tail -c 1200 $O/prof_${TAG}_default.log > $O/prof_${TAG}_default.tail; rm -rf $O/prof_${TAG}_default.log $D
bash scripts/gpu_bench_all.sh $TAG > $O/${TAG}_bench_all.log 2>&1
bash scripts/gpu_pmc_bench.sh $TAG dcn 64 > $O/${TAG}_pmc_fullrun.log 2>&1
bash scripts/gpu_pmc_sq.sh $TAG dcn SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE > $O/${TAG}_pmc_sq.log 2>&1
# the CIN kernels' matrix-pipe counters (north_star: "for CIN, MFMA utilisation vs peak")
bash scripts/gpu_pmc_sq.sh $TAG xdeepfm SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE > $O/${TAG}_xdeepfm_pmc_sq.log 2>&1
timeout 400 python bench.py --model deepfm --big-table-rows 100000000 --no-cpu-baseline --no-host-fed --sweep-batches 0 > $O/bench_${TAG}_deepfm_100M.json 2> $O/bench_${TAG}_deepfm_100M.err
timeout 300 python scripts/bench_tfrecord.py --examples 131072 --epochs 60 > $O/bench_${TAG}_tfrecord_e2e.json 2> $O/bench_${TAG}_tfrecord_e2e.err
RECALGO_READER_THREADS=64 timeout 300 python scripts/bench_tfrecord.py --examples 131072 --epochs 60 > $O/bench_${TAG}_tfrecord_e2e_64threads.json 2> $O/bench_${TAG}_tfrecord_e2e_64threads.err
# round 6: the DIN attention lab (plain kernel times + per-phase timeline) and the cost of the reference's default dropout
bash scripts/gpu_din_lab.sh 50 > $O/${TAG}_din_lab.md 2>&1
bash scripts/gpu_dropout_cost.sh deepfm din pnn fibinet > $O/${TAG}_dropout_cost.md 2>&1
# round 6: the fused DCN tail (plain kernel time, per-phase timeline) and the fork / join experiment its riders replace
bash scripts/gpu_tailfuse_lab.sh > $O/${TAG}_tailfuse_lab.md 2>&1
timeout 300 python scripts/lab_fork_join.py 2>/dev/null > $O/${TAG}_fork_join_lab.md
tail -3 $O/${TAG}_pytest_gpu.log; tail -2 $O/${TAG}_smoke.log
for f in $O/bench_${TAG}_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("frac"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
du -sh $O
