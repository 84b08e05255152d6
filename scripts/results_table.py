#!/usr/bin/env python
"""Markdown results table of one round-end evidence run: profiles/<tag>_bench_<model>.json (+ the previous round's, for the
comparison column) and the largest kernels of profiles/<tag>_<model>_kernel_stats.md.
    python scripts/results_table.py r04zz r03z"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load(tag, m):
    try:
        return json.loads(open(os.path.join(ROOT, "profiles", f"{tag}_bench_{m}.json")).read().strip().splitlines()[-1])
    except OSError:
        return None


def top_kernels(tag, m, n=5):
    path = os.path.join(ROOT, "profiles", f"{tag}_{m}_kernel_stats.md")
    if not os.path.exists(path):
        return "", 0.0
    rows = []
    for line in open(path):
        if line.startswith("per launch shape"):
            break
        mm = re.match(r"\| `(.*?)` \| (\d+) \| (\d+) \| (\d+) \|", line)
        if mm:
            rows.append((mm.group(1), int(mm.group(2)), int(mm.group(4))))
    steps = next((c for nme, c, _ in rows if "adam_tf1_step_kernel" in nme), 0)
    if not steps:
        return "", 0.0

    def short(x):
        x = re.sub(r"\(anonymous namespace\)::", "", x)
        x = re.sub(r"^void ", "", x)
        x = x.split("(")[0].split("<")[0].replace("_kernel", "")
        if x.startswith("Cijk_"):
            return "hipBLASLt " + x[:18] + "…"
        return x.replace("at::native::", "aten ")
    per = [(short(nme), c / steps, a / 1e3) for nme, c, a in rows if c >= 0.9 * steps]
    per.sort(key=lambda t: -t[1] * t[2])
    disp = sum(c for _, c, _ in per)
    return ", ".join(f"`{k}` {c:.0f}×{a:.1f}" if c > 1.5 else f"`{k}` {a:.1f}" for k, c, a in per[:n]), disp


def main(tag, prev):
    print("| model (`bench.py --model`) | ex/s | ms/step | previous round | dispatches / step | largest kernels in the step (rocprofv3 averages, µs) |")
    print("|---|---:|---:|---:|---:|---|")
    for m in ("dcn", "deepfm", "xdeepfm", "din", "fibinet", "pnn", "fwfm", "nfm", "afm", "ffm", "deepfm_100M"):
        d, p = load(tag, m), load(prev, m)
        if d is None:
            continue
        ks, disp = top_kernels(tag, m if m != "deepfm_100M" else "deepfm_100M")
        print(f"| {m} | {d['value'] / 1e6:.2f} M | {d['ms_per_step']:.4f} | {'' if p is None else '%.4f' % p['ms_per_step']} | "
              f"{'' if not disp else '%.0f' % disp} | {ks} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "r03z")
