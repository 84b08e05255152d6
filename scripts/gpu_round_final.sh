#!/bin/bash
# Round-end evidence on the GPU box (via gpurun): full GPU test suite, smoke, the default bench line,
# per-model bench lines + rocprofv3 kernel stats + PMC passes, the full-run PMC of the default
# model, and the 100 M-row table run.  Leaves only text summaries under gpurun_out/.
# usage: scripts/gpu_round_final.sh <tag>
TAG=${1:-r01}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${TAG}_smoke.log 2>&1
timeout 400 python bench.py > $O/bench_${TAG}_default.json 2> $O/bench_${TAG}_default.err
bash scripts/gpu_bench_all.sh $TAG --pmc dcn deepfm xdeepfm din fibinet pnn > $O/${TAG}_bench_all.log 2>&1
bash scripts/gpu_pmc_bench.sh $TAG dcn 64 > $O/${TAG}_pmc_fullrun.log 2>&1
timeout 400 python bench.py --model deepfm --big-table-rows 100000000 --no-cpu-baseline > $O/bench_${TAG}_deepfm_100M.json 2> $O/bench_${TAG}_deepfm_100M.err
tail -3 $O/${TAG}_pytest_gpu.log; cat $O/${TAG}_smoke.log | tail -2
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
