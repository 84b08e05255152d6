#!/bin/bash
# Run on the GPU box (via gpurun): bench lines + rocprofv3 kernel-trace stats (+ optional PMC passes)
# for every model.  Only small text summaries are left under gpurun_out/ (64 MiB merge limit).
# usage: scripts/gpu_bench_all.sh <tag> [--pmc] [models...]
TAG=${1:-r01}; shift
PMC=0; if [ "$1" == "--pmc" ]; then PMC=1; shift; fi
MODELS=${@:-dcn deepfm xdeepfm din fibinet pnn fwfm nfm afm ffm}
R=$PWD
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
for m in $MODELS; do
  timeout 300 python bench.py --model $m --steps 400 --warmup 20 --no-cpu-baseline --no-host-fed --no-extra-models --sweep-batches 0 > $O/bench_${TAG}_$m.json 2> $O/bench_${TAG}_$m.err
  echo "== $m: $(head -c 300 $O/bench_${TAG}_$m.json)"
  tail -2 $O/bench_${TAG}_$m.err
  D=/tmp/prof_${TAG}_$m
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $D -o $m -- python $R/bench.py --model $m --steps 50 --warmup 10 --no-cpu-baseline --no-host-fed --no-kernel-timing --no-extra-models --sweep-batches 0 > $O/prof_${TAG}_$m.log 2>&1)
  DB=$(find $D -name "*_results.db" | head -1)
  if [ -n "$DB" ]; then python $R/scripts/rocpd_stats.py $DB 60 > $O/${TAG}_${m}_kernel_stats.md; fi
  tail -c 1500 $O/prof_${TAG}_$m.log > $O/prof_${TAG}_$m.tail; rm -f $O/prof_${TAG}_$m.log
  if [ $PMC == 1 ]; then
    for C in FETCH_SIZE WRITE_SIZE; do
      P=/tmp/pmc_${TAG}_${m}_$C
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -f csv -d $P -o $m -- python $R/bench.py --model $m --steps 4 --warmup 1 --no-graph --no-cpu-baseline --no-host-fed --no-kernel-timing --no-extra-models --sweep-batches 0 > $O/pmc_${TAG}_${m}_$C.log 2>&1)
      tail -c 600 $O/pmc_${TAG}_${m}_$C.log > $O/pmc_${TAG}_${m}_$C.tail; rm -f $O/pmc_${TAG}_${m}_$C.log
    done
    python $R/scripts/pmc_summary.py /tmp/pmc_${TAG}_${m}_FETCH_SIZE /tmp/pmc_${TAG}_${m}_WRITE_SIZE > $O/${TAG}_${m}_pmc.md 2>&1
  fi
  rm -rf $D /tmp/pmc_${TAG}_${m}_*
done
du -sh $O; ls $O
