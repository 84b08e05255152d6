#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counters from `*_counter_collection.csv` files.

    python scripts/pmc_summary.py <dir-or-csv>... > profiles/rNN_<model>_pmc.md

FETCH_SIZE / WRITE_SIZE are reported in the tool's unit (KiB on this build) and converted to bytes
per dispatch; per guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts a wide
coalesced read at half its bytes, so `fetch_x2_B` (doubled) is the figure to compare with
algorithmic bytes for streaming reads; WRITE_SIZE is uncalibrated.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def files(args):
    for a in args:
        if os.path.isdir(a):
            yield from glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)
        else:
            yield a


def main(args):
    acc = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> values (one per dispatch)
    for f in files(args):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name") or row.get("kernel_name")
                c = row.get("Counter_Name") or row.get("counter_name")
                v = row.get("Counter_Value") or row.get("counter_value")
                if k and c and v not in (None, ""):
                    acc[k][c].append(float(v))
    counters = sorted({c for k in acc for c in acc[k]})
    print("| kernel | dispatches | " + " | ".join(f"avg {c} | max {c}" for c in counters) + " |")
    print("|---|---:|" + "---:|---:|" * len(counters))
    for k in sorted(acc, key=lambda k: -sum(sum(v) for v in acc[k].values())):
        n = max(len(v) for v in acc[k].values())
        name = k if len(k) < 100 else k[:97] + "..."
        cells = [f"{sum(acc[k][c]) / len(acc[k][c]):.1f} | {max(acc[k][c]):.1f}" if acc[k].get(c) else " | " for c in counters]
        print(f"| `{name}` | {n} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main(sys.argv[1:])
