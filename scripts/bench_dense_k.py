#!/usr/bin/env python
"""Fixed cost vs per-chunk cost of the dense kernels: time at K = 32 .. 1024 for M = 4096, N = 512."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from recalgorithm_amd import ops
dev = torch.device("cuda:0")
M = 4096
print("| N | K | fwd us | dgrad us | wgrad us |\n|---|---|---:|---:|---:|")
for N in (512, 128):
    for K in (32, 64, 128, 256, 416, 512, 1024):
        x = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
        g = torch.randn(M, N, device=dev); y = ops.dense_fwd(x, w, b, True); dw, db = torch.empty_like(w), torch.empty_like(b)
        f = bench.event_time_ms(lambda: ops.dense_fwd(x, w, b, True))
        d = bench.event_time_ms(lambda: ops.dense_bwd_input(g, y, w))
        wg = bench.event_time_ms(lambda: ops.dense_bwd_weights(x, g, y, dw, db))
        print(f"| {N} | {K} | {f*1e3:.1f} | {d*1e3:.1f} | {wg*1e3:.1f} |")
