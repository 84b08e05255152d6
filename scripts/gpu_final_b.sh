#!/bin/bash
# round-end evidence, part B: the full GPU test suite once more on the final tree, then bench line + rocprofv3 kernel stats per model
TAG=${1:-r05b}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/${TAG}_pytest_gpu_full.log 2>&1; grep -v "^| tests" $O/${TAG}_pytest_gpu_full.log | tail -40 > $O/${TAG}_pytest_gpu.log
cp $O/strict_parity.md $O/${TAG}_strict_parity_all_gpu_tests.md 2>/dev/null; rm -f $O/${TAG}_pytest_gpu_full.log
tail -1 $O/${TAG}_pytest_gpu.log
bash scripts/gpu_bench_all.sh $TAG dcn deepfm xdeepfm din fibinet pnn fwfm ffm nfm afm > $O/${TAG}_bench_all.log 2>&1
grep "^==" $O/${TAG}_bench_all.log | cut -c1-40,100-190
