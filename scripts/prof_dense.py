#!/usr/bin/env python
"""A few eager launches of the dense kernels at one shape, for rocprofv3 (--kernel-trace / --pmc) runs.
usage: python scripts/prof_dense.py M K N [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recalgorithm_amd import ops  # noqa: E402

M, K, N = (int(a) for a in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev)
w = torch.randn(K, N, device=dev) / K ** 0.5
b = torch.randn(N, device=dev)
g = torch.randn(M, N, device=dev)
dw, db = torch.empty_like(w), torch.empty_like(b)
for _ in range(reps):
    y = ops.dense_fwd(x, w, b, True)
    ops.dense_bwd_input(g, y, w)
    ops.dense_bwd_weights(x, g, y, dw, db)
    torch._addmm_activation(b, x, w)
torch.cuda.synchronize()
