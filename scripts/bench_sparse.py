#!/usr/bin/env python
"""Micro-benchmark of the owner-computes scatter / sparse optimizer (csrc/sparse.hip) at the BASELINE shape:
B = 4096, F = 26, K = 16, the bench's Zipf id distribution (or --uniform), an arena of the bench's size.

    python scripts/bench_sparse.py [--uniform] [--mode adam|lazy|grad] [--steps 200] [--advance]

Prints one JSON line: average HIP-event time per step of `prepare` alone and of `prepare + place + apply`.  Run it
under `rocprofv3 --kernel-trace --stats` for the per-kernel split (sparse_prepare / sparse_place / sparse_apply)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--uniform", action="store_true")
    ap.add_argument("--mode", default="adam", choices=["adam", "lazy", "grad"])
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--fields", type=int, default=26)
    ap.add_argument("--emb", type=int, default=16)
    ap.add_argument("--advance", action="store_true", help="advance the step counter every step (real catch-up / sweep work)")
    ap.add_argument("--data-batches", type=int, default=32, help="distinct batches rotated through (the bench's 32)")
    ap.add_argument("--split", action="store_true",
                    help="issue the parts of `prepare` as THREE launches (count | catch-up | sweep) instead of one grid: under "
                         "rocprofv3 the per-grid-size table of scripts/rocpd_stats.py then shows each part's time")
    a = ap.parse_args()
    from recalgorithm_amd import sparse as sp
    from recalgorithm_amd.io import synth
    from recalgorithm_amd.variables import EmbeddingArena
    dev = torch.device("cuda:0")
    spec = synth.SynthSpec(n_fields=a.fields, max_vocab=1_000_000)
    ar = EmbeddingArena("t", a.emb, dev, seed=1)
    for n, v in zip(spec.names, spec.vocabs):
        ar.add_table(n, v)
    ar.materialize()
    rb = torch.tensor([ar.tables[n][0] for n in sorted(spec.names)], dtype=torch.int64, device=dev)
    B, F, K = a.batch, a.fields, a.emb
    batches = []
    for i in range(a.data_batches):
        feats, _, _ = synth.device_features(spec, B, dev, batch_index=i)
        ids = torch.stack([feats[n] for n in sorted(spec.names)], 1).contiguous()
        if a.uniform:
            voc = torch.tensor([ar.tables[n][1] for n in sorted(spec.names)], device=dev)
            ids = (torch.rand(B, F, device=dev) * voc).long()
        batches.append(ids)
    g = torch.randn(B, F * K, device=dev)

    class Store:
        opt_state = {"step": torch.ones(1, dtype=torch.int64, device=dev), "lr_t": torch.zeros(1, device=dev)}
        arenas = {"t": ar}
    store = Store()
    step = store.opt_state["step"]

    def split_lookup(ids):
        """begin_lookup's launch as three: count | catch-up | sweep (same work, same state afterwards)."""
        import ctypes
        from recalgorithm_amd import _lib
        lib = _lib.load()
        plan = sp.plan_of(ar)
        src = sp.Source(ids, None, rb, 0, B, F)
        src.arena = ar
        plan._ensure_ws(src.slots)
        d = plan._deferred_struct()
        cs = src.c_struct(K)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        ws, stp = ctypes.c_void_p(plan.ws.data_ptr()), ctypes.c_void_p(step.data_ptr())
        rows = ar.weight.shape[0]
        lib.recalgo_scatter_prepare(ctypes.byref(cs), K, ws, plan.capacity, plan.nb_log2, 0, sp.PREPARE_COUNT, None, None, 0, 0, 1, None, 0, st)
        if d is not None:
            lib.recalgo_scatter_prepare(ctypes.byref(cs), K, ws, plan.capacity, plan.nb_log2, 0, sp.PREPARE_CATCHUP, ctypes.byref(d), None,
                                        rows, 0, sp.sweep_period(), stp, 0, st)
            lib.recalgo_scatter_prepare(None, K, ws, plan.capacity, plan.nb_log2, 0, sp.PREPARE_SWEEP, ctypes.byref(d), None, rows, 0,
                                        sp.sweep_period(), stp, 0, st)
            plan.swept = True
        plan.sources.append(src)
        plan.counted = plan.counted[:2] + (plan.counted[2] + (id(src),),)
        return src

    def one(i, apply=True):
        ids = batches[i % len(batches)]
        src = split_lookup(ids) if (a.split and sp.plan_of(ar) is not None) else sp.begin_lookup(ar, store, ids, None, rb, 0, B, F, True)
        if not apply:
            sp.plan_of(ar).sources = []
            return
        src.set_grad(g)
        if a.advance:
            step.add_(1)
        if a.mode == "grad":
            sp.materialize_arena(ar)
            sp.new_forward(store)
        else:
            sp.apply(ar, a.mode == "lazy", step, 0.001, 0.9, 0.999, 1e-8)

    for i in range(max(16, 2 * a.data_batches)):
        one(i)
    torch.cuda.synchronize()

    def timed(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    full = timed(lambda i: one(i), a.steps)
    sp.plan_of(ar).counted = None
    prep = timed(lambda i: one(i, apply=False), a.steps)
    distinct = int(torch.unique((batches[0] + rb.unsqueeze(0))[batches[0] >= 0]).numel())
    print(json.dumps({"mode": a.mode, "uniform": a.uniform, "advance": a.advance, "requests": B * F, "distinct_rows_batch0": distinct,
                      "prepare_us_eager": round(prep, 2), "prepare_place_apply_us_eager": round(full, 2),
                      "note": "eager launches: includes ~10-20 us of Python / ctypes per call; use the rocprofv3 kernel stats for kernel time"}))


if __name__ == "__main__":
    main()
