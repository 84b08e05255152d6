#!/usr/bin/env python
"""Time of the deferred-Adam sweep share of `prepare` alone (recalgo_scatter_prepare with RECALGO_PREPARE_SWEEP only) on a
table of --rows rows x 16 floats, for a given fraction of rows that carry optimizer state (and lag).
    python scripts/bench_sweep.py --rows 100000000 [--live 0.001] [--k 16]"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--live", type=float, default=0.0)
    ap.add_argument("--period", type=int, default=32)
    a = ap.parse_args()
    from recalgorithm_amd import _lib, sparse as sp
    lib = _lib.load()
    dev = torch.device("cuda:0")
    w = torch.zeros(a.rows, a.k, device=dev)
    m = torch.zeros(a.rows, a.k, device=dev)
    v = torch.zeros(a.rows, a.k, device=dev)
    last = torch.zeros(a.rows, dtype=torch.int32, device=dev)
    if a.live > 0:
        idx = torch.randint(0, a.rows, (int(a.rows * a.live),), device=dev)
        last[idx] = 1
        m[idx] = 0.01
        v[idx] = 0.0001
    ring = torch.full((sp.LR_RING,), 0.001, device=dev)
    step = torch.full((1,), 40, dtype=torch.int64, device=dev)
    d = sp._CDeferred(w.data_ptr(), m.data_ptr(), v.data_ptr(), last.data_ptr(), ring.data_ptr(), 0.9, 0.999, 1e-8)
    nb = 10
    ws = torch.zeros(int(lib.recalgo_scatter_plan_workspace_bytes(256, nb, a.k)), dtype=torch.uint8, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def once():
        _lib.check(lib.recalgo_scatter_prepare(None, a.k, ctypes.c_void_p(ws.data_ptr()), 256, nb, 0, sp.PREPARE_SWEEP, ctypes.byref(d),
                                               None, a.rows, 0, a.period, ctypes.c_void_p(step.data_ptr()), 0, st), "prepare")
    res = {}
    for rep in range(2):
        ts = []
        for i in range(a.period):                  # one whole period: every share once
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); once(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
            step.add_(1)
        res[f"pass{rep}_us_min_med_max"] = [round(min(ts), 1), round(sorted(ts)[len(ts) // 2], 1), round(max(ts), 1)]
    print(json.dumps({"rows": a.rows, "live": a.live, "shift": os.environ.get("RECALGO_SPARSE_SWEEP_BLOCK_SHIFT", "auto"), **res}))


if __name__ == "__main__":
    main()
