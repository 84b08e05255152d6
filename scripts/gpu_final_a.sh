#!/bin/bash
# round-end evidence, part A: full GPU test suite, smoke, the default bench line, rocprofv3 kernel stats of the same command
TAG=${1:-r05b}
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/${TAG}_pytest_gpu_full.log 2>&1; grep -v "^| tests" $O/${TAG}_pytest_gpu_full.log | tail -40 > $O/${TAG}_pytest_gpu.log
cp $O/strict_parity.md $O/${TAG}_strict_parity_all_gpu_tests.md 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${TAG}_smoke.log 2>&1
timeout 400 python bench.py > $O/bench_${TAG}_default.json 2> $O/bench_${TAG}_default.err
D=/tmp/prof_${TAG}_default
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $D -o dcn -- python $R/bench.py --no-cpu-baseline --no-host-fed > $O/prof_${TAG}_default.log 2>&1)
DB=$(find $D -name "*_results.db" | head -1)
if [ -n "$DB" ]; then python $R/scripts/rocpd_stats.py $DB 40 > $O/${TAG}_default_kernel_stats.md; fi
tail -c 1200 $O/prof_${TAG}_default.log > $O/prof_${TAG}_default.tail; rm -rf $O/prof_${TAG}_default.log $D $O/${TAG}_pytest_gpu_full.log
tail -3 $O/${TAG}_pytest_gpu.log; tail -2 $O/${TAG}_smoke.log; head -c 600 $O/bench_${TAG}_default.json
