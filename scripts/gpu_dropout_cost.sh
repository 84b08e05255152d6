#!/bin/bash
# step time of the models with a dropout flag at rate 0 (parity configuration) and 0.1 (the reference's default)
O=gpurun_out; mkdir -p $O
for m in ${@:-deepfm din pnn fibinet}; do
  for r in 0 0.1; do
    timeout 300 python bench.py --model $m --dropout-rate $r --steps 300 --warmup 20 --no-cpu-baseline --no-host-fed --no-extra-models --no-kernel-timing --sweep-batches 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m rate $r:', d['ms_per_step'], 'ms', round(d['value']/1e6,3), 'M ex/s')"
  done
done
