#!/usr/bin/env python
"""Record the PMC traffic of one kernel (max over the dispatches of a full-run summary written by
scripts/pmc_summary.py) in profiles/pmc_traffic.json, the file bench.py reads `roofline.traffic` from.

    python scripts/pmc_traffic_json.py <fullrun.md> <model> <kernel-substring> <bench kernel name> <source path> <alg_bytes>

alg_bytes = the algorithmic bytes per launch bench.py reports for that kernel in the measured state; bench.py only
quotes the traffic while its own figure is within 10 % of it (same shapes / same optimizer state).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(md, model, needle, bench_name, source, alg_bytes=0):
    hdr, row = None, None
    for line in open(md):
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if cells and cells[0] == "kernel":
            hdr = cells
        elif hdr and needle in cells[0]:
            row = cells
            break
    if row is None:
        raise SystemExit(f"{needle} not found in {md}")
    col = {h: i for i, h in enumerate(hdr)}
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data.setdefault(model, {})[bench_name] = {"fetch_size_kb": float(row[col["max FETCH_SIZE"]]),
                                               "write_size_kb": float(row[col["max WRITE_SIZE"]]), "source": source,
                                               "alg_bytes": int(float(alg_bytes))}
    json.dump(data, open(path, "w"), indent=1)
    print(model, bench_name, data[model][bench_name])


if __name__ == "__main__":
    main(*sys.argv[1:7])
