#!/bin/bash
# A/B of an environment switch on ONE box: scripts/gpu_ab.sh VAR "valA valB" "models" [reps]
set -u
cd "$(dirname "$0")/.."
VAR=$1; VALS=$2; MODELS=${3:-dcn}; REPS=${4:-2}
for r in $(seq 1 $REPS); do
 for m in $MODELS; do
  for v in $VALS; do
    env $VAR=$v timeout 300 python bench.py --model $m --steps 400 --warmup 20 --no-cpu-baseline --no-host-fed --no-extra-models --no-kernel-timing --sweep-batches 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m $VAR=$v', d['ms_per_step'])"
  done
 done
done
