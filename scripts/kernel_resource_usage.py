#!/usr/bin/env python
"""Static resource usage of every kernel of librecalgo_hip.so (no GPU needed):

    python scripts/kernel_resource_usage.py > profiles/rNN_kernel_resource_usage.md

Runs `hipcc -Rpass-analysis=kernel-resource-usage` over recalgorithm_amd/csrc/*.hip with the build's flags and tabulates
VGPRs / AGPRs / scratch / static LDS / the compiler's occupancy bound per kernel instantiation."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def main():
    rows = []
    for f in sorted(glob.glob(os.path.join(ROOT, "recalgorithm_amd", "csrc", "*.hip"))):
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-munsafe-fp-atomics", "-std=c++17", "-c", f,
                            "-I", os.path.join(ROOT, "include"), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                           capture_output=True, text=True)
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r"remark: (?:([^:]+):\d+:\d+: )?Function Name: (\S+)", line)
            if m:
                cur = {"file": os.path.basename(f), "name": m.group(2)}
                rows.append(cur)
                continue
            if cur is None:
                continue
            for key, pat in (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                             ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
                mm = re.search(pat, line)
                if mm:
                    cur[key] = int(mm.group(1))
    names = demangle([r["name"] for r in rows])
    print("# Static resource usage of every kernel in librecalgo_hip.so\n")
    print("`hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage` over "
          "`recalgorithm_amd/csrc/*.hip` (`scripts/kernel_resource_usage.py`).  Occupancy is the compiler's waves/SIMD bound from "
          "registers and static LDS only; kernels with dynamic LDS (scatter aggregators, DIN, bilinear backward, the dense "
          "tile engine's 3-slot ring, the fused loss tail) are further limited at launch.\n")
    print("| file | kernel | VGPRs | AGPRs | scratch B/lane | static LDS B | occupancy (waves/SIMD) |")
    print("|---|---|---:|---:|---:|---:|---:|")
    for r, n in zip(rows, names):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)
        n = n.split("(")[0]
        print(f"| {r['file']} | `{n}` | {r.get('vgpr', '')} | {r.get('agpr', '')} | {r.get('scratch', '')} | {r.get('lds', '')} | {r.get('occ', '')} |")
    spills = [r for r in rows if r.get("scratch", 0)]
    print(f"\n{len(rows)} kernel instantiations; {len(spills)} use scratch.")


if __name__ == "__main__":
    main()
