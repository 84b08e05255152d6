#!/bin/bash
# host-fed loop (TFRecord bytes -> step) at several reader thread counts, same box
O=gpurun_out; mkdir -p $O
for t in ${@:-32 48 64 96 128}; do
  RECALGO_READER_THREADS=$t timeout 300 python scripts/bench_tfrecord.py --examples 131072 --epochs 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('threads $t: end to end', round(d['value']/1e6,2), 'M ex/s,', d['ms_per_step'], 'ms/step; reader alone', round(d['host']['reader_ex_s']/1e6,2), 'M ex/s')"
done
