#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (`*_results.db`) into the per-kernel table that
`rocprofv3 --kernel-trace --stats` prints in CSV mode: calls, total / average / min / max
duration (ns) and share of GPU kernel time.

    python scripts/rocpd_stats.py gpurun_out/prof_dcn/dcn_results.db > profiles/r01_dcn_kernel_stats.md
"""
import sqlite3
import sys


def main(path, top=40):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {name} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"source: {path}")
    print(f"total kernel time: {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total_ns | avg_ns | min_ns | max_ns | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for r in rows[:top]:
        n = r[0] if len(r[0]) < 110 else r[0][:107] + "..."
        print(f"| `{n}` | {r[1]} | {r[2]} | {r[3]:.0f} | {r[4]} | {r[5]} | {100.0 * r[2] / tot:.2f} |")
    # one kernel name can serve several shapes (the dense kernels run the three MLP layers): the same statistics per
    # (kernel, grid size), so that a per-shape bench row can be compared with the dispatches of exactly that shape
    gx = next((c_ for c_ in cols if c_.lower() in ("grid_size_x", "grid_x", "grid_size")), None)
    if gx is None:
        return
    rows = c.execute(f"select {name}, {gx}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels where {name} like '%dense_%' or {name} like '%cin_%' or {name} like '%sparse_%' group by {name}, {gx} "
                     f"order by 1, 2").fetchall()
    if rows:
        print("\nper launch shape (kernel, grid size):\n")
        print("| kernel | grid | calls | avg_ns | min_ns | max_ns |")
        print("|---|---:|---:|---:|---:|---:|")
        for r in rows:
            n = r[0] if len(r[0]) < 90 else r[0][:87] + "..."
            print(f"| `{n}` | {r[1]} | {r[2]} | {r[4]:.0f} | {r[5]} | {r[6]} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
