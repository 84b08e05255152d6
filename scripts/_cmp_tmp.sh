O=$PWD/gpurun_out; export TMPDIR=/tmp
for m in fibinet pnn; do
  for mode in hip blas; do
    RECALGO_DENSE=$mode timeout 300 python bench.py --model $m --steps 200 --warmup 20 --no-cpu-baseline --sweep-batches 0 --no-kernel-timing 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m $mode', round(d['value']/1e6,3), d['ms_per_step'])"
  done
  D=/tmp/prof_$m
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $D -o $m -- python $PWD/../$(basename $PWD)/bench.py --model $m --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --sweep-batches 0 > /dev/null 2>&1)
  DB=$(find $D -name "*_results.db" | head -1)
  if [ -n "$DB" ]; then python scripts/rocpd_stats.py $DB 25 > $O/r02o_${m}_kernel_stats.md; head -34 $O/r02o_${m}_kernel_stats.md; fi
done
