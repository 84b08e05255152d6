#!/bin/bash
# A/B of a module hook on ONE box: bench.py's bare step time with <module>.<attr> = each value, interleaved, `reps` times
# usage: scripts/gpu_ab_hook.sh recalgorithm_amd.ops LAZY_GATHER "True False" "dcn" 3
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
MOD=$1; ATTR=$2; VALS=$3; MODELS=${4:-dcn}; REPS=${5:-3}
for r in $(seq 1 $REPS); do
  for v in $VALS; do
    for m in $MODELS; do
      timeout 300 python - <<PY > gpurun_out/abh_${ATTR}_${v}_${m}_$r.json 2> gpurun_out/abh_${ATTR}_${v}_${m}_$r.err
import importlib, runpy, sys
setattr(importlib.import_module("$MOD"), "$ATTR", $v)
sys.argv = ["bench.py", "--model", "$m", "--steps", "300", "--warmup", "30", "--no-cpu-baseline", "--no-host-fed", "--no-extra-models",
            "--no-kernel-timing", "--sweep-batches", "0"]
runpy.run_path("bench.py", run_name="__main__")
PY
      python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/abh_${ATTR}_${v}_${m}_$r.json").read().strip().splitlines()[-1])
    print("$ATTR=$v", "$m", "rep$r", d["value"], d["ms_per_step"])
except Exception as e:
    print("$ATTR=$v", "$m", "rep$r", "FAILED", e)
PY
    done
  done
done
