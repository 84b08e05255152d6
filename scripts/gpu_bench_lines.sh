#!/bin/bash
# Run on the GPU box (via gpurun): one bench line per model (no rocprof, no CPU baseline, no optimizer-state sweep).
# usage: scripts/gpu_bench_lines.sh <tag> [models...]
TAG=${1:-r02}; shift
MODELS=${@:-dcn deepfm xdeepfm din fibinet pnn fwfm nfm afm ffm}
O=$PWD/gpurun_out; mkdir -p $O
for m in $MODELS; do
  timeout 300 python bench.py --model $m --steps 300 --warmup 20 --no-cpu-baseline --no-host-fed --sweep-batches 0 > $O/bench_${TAG}_$m.json 2> $O/bench_${TAG}_$m.err
  python - "$O/bench_${TAG}_$m.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["config"]["workload"][:40], "|", round(d["value"] / 1e6, 3), "M ex/s |", d["ms_per_step"], "ms |", d["config"].get("launch"))
    for k in d.get("kernels", []):
        print("   %-34s %9.1f us  %-5s frac %.3f" % (k["kernel"], k["avg_us"], k["bound"], k["frac"]))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
  tail -2 $O/bench_${TAG}_$m.err
done
