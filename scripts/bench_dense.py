#!/usr/bin/env python
"""Per-layer timing of the fp32-MFMA dense kernels (csrc/dense.hip) against the library GEMMs they replace
(torch -> hipBLASLt + the relu_bwd_bias / colsum glue), at the BASELINE MLP shapes.  HIP events around
hipGraph replays (bench.event_time_ms).  usage: python scripts/bench_dense.py [M]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from recalgorithm_amd import ops  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    dev = torch.device("cuda:0")
    rows = []
    for K, N in [(416, 512), (512, 256), (256, 128), (416, 1024), (9600, 512)]:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(K, N, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        g = torch.randn(M, N, device=dev)
        y = ops.dense_fwd(x, w, b, True)
        dw, db = torch.empty_like(w), torch.empty_like(b)
        fl = 2.0 * M * K * N
        t = {}
        t["fwd mfma"] = bench.event_time_ms(lambda: ops.dense_fwd(x, w, b, True))
        t["fwd blas"] = bench.event_time_ms(lambda: torch._addmm_activation(b, x, w))
        t["dgrad mfma"] = bench.event_time_ms(lambda: ops.dense_bwd_input(g, y, w))
        t["wgrad mfma"] = bench.event_time_ms(lambda: ops.dense_bwd_weights(x, g, y, dw, db))

        def blas_bwd():
            g2 = ops.relu_bwd_bias_(g, y, db)
            torch.mm(x.t(), g2, out=dw)
            return g2 @ w.t()
        t["bwd blas (mask+bias glue, wgrad, dgrad)"] = bench.event_time_ms(blas_bwd)
        t["bwd mfma (wgrad + dgrad)"] = t["dgrad mfma"] + t["wgrad mfma"]
        for k, ms in t.items():
            nf = 2 if k.startswith("bwd") else 1
            rows.append(f"| {M}x{K}x{N} | {k} | {ms * 1e3:.1f} | {nf * fl / (ms * 1e-3) / 1e12:.1f} |")
    print("| M x K x N | kernel | us | TFLOP/s |\n|---|---|---:|---:|")
    print("\n".join(rows))


if __name__ == "__main__":
    main()
