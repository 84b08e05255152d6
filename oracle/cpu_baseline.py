"""oracle/cpu_baseline.py — TEST/BENCH INFRASTRUCTURE ONLY (see oracle/__init__.py).

Times the unfused, op-for-op torch-CPU restatement (oracle/ref_models.py) of one full training
step — forward, autograd backward, TF1 dense Adam on every variable including all embedding
tables — on the host cores of the box that runs it.  This is the stand-in for "the TF1 CPU
path" (TensorFlow 1.14 cannot be installed here); it is reported as
`cpu_baseline.kind = "port"`, never as TensorFlow.
"""
from __future__ import annotations

import os
import time
from types import SimpleNamespace

import numpy as np
import torch

from . import ref_models as M
from . import ref_ops as R


def _glorot(shape, gen):
    fan_in, fan_out = shape[0], shape[-1]
    lim = (6.0 / (fan_in + fan_out)) ** 0.5
    return (torch.rand(*shape, generator=gen) * 2 - 1) * lim


def _zipf(rng, n, vocab, s=1.05):
    ranks = np.arange(1, vocab + 1, dtype=np.float64)
    cdf = np.cumsum(ranks ** (-s))
    cdf /= cdf[-1]
    return np.searchsorted(cdf, rng.random(n), side="left").astype(np.int64)


def build_dcn(fields, emb, vocabs, hidden=(512, 256, 128), L=3, seed=42):
    gen = torch.Generator().manual_seed(seed)
    names = [f"f{i:02d}" for i in range(fields)]
    cols = [SimpleNamespace(name=f"{n}_embedding", key=n, dimension=emb, shared_name=None) for n in names]
    P = {}
    for c, v in zip(cols, vocabs):
        P[f"category_input/input_layer/{c.name}/embedding_weights"] = torch.randn(v, emb, generator=gen) / emb ** 0.5
    d = fields * emb
    for i in range(L):
        P[f"cross_part/wl_{i}"] = _glorot((d, 1), gen)
        P[f"cross_part/bl_{i}"] = _glorot((d, 1), gen)
    prev = d
    for i, h in enumerate(hidden):
        P[f"dnn_part/dnn_dense_{i}/kernel"] = _glorot((prev, h), gen)
        P[f"dnn_part/dnn_dense_{i}/bias"] = torch.zeros(h)
        prev = h
    P["output_part/dense/kernel"] = _glorot((d + prev, 1), gen)
    P["output_part/dense/bias"] = torch.zeros(1)
    params = {"category_feature_columns": cols, "dense_feature_columns": [], "hidden_units": list(hidden),
              "num_cross_layer": L}
    return P, params, names


def run(model: str, batch=4096, fields=26, emb=16, max_vocab=1_000_000, seconds=15.0, threads=None):
    if model != "dcn":
        return {"value": None, "unit": "examples/s", "cores": 0, "kind": "port",
                "sample": f"{model}: cpu baseline not wired yet"}
    # threads actually used: the cores this process may run on, capped at 32 (the per-op work of
    # a 4096-example batch does not scale past that; TF1's default intra-op pool has the same
    # problem on many-core hosts)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = threads or max(1, min(avail, 32))
    torch.set_num_threads(threads)
    real = [20000, 106444, 2, 18789, 25159, 17500]
    extra = fields - len(real)
    vocabs = real[:fields] + ([int(round(10 ** x)) for x in np.linspace(2.0, np.log10(max_vocab), extra)] if extra > 0 else [])
    P, params, names = build_dcn(fields, emb, vocabs)
    for t in P.values():
        t.requires_grad_(True)
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    v2 = {k: torch.zeros_like(v) for k, v in P.items()}
    rng = np.random.default_rng(1234)
    feats = {n: torch.from_numpy(_zipf(rng, batch, v)) for n, v in zip(names, vocabs)}
    labels = {"read_comment": torch.from_numpy((rng.random((batch, 1)) < 0.0356).astype(np.float32))}

    def step(t):
        for p in P.values():
            p.grad = None
        out = M.dcn(P, feats, labels, params, training=True)
        out["loss"].backward()
        with torch.no_grad():
            for k, p in P.items():
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                R.adam_tf1_step(p, g, m[k], v2[k], t, 0.005)       # dense: all rows decay (TF1)
        return float(out["loss"].detach())

    step(1)                                            # warm-up (allocations, MKL init)
    times, t, t_start = [], 2, time.perf_counter()
    while True:
        t0 = time.perf_counter()
        step(t)
        times.append(time.perf_counter() - t0)
        t += 1
        if time.perf_counter() - t_start > seconds:
            break
        if len(times) >= 50:
            break
    med = float(np.median(times))
    return {"value": round(batch / med, 1), "unit": "examples/s", "cores": threads, "kind": "port",
            "sample": f"{len(times)} full train steps (fwd+bwd+dense TF1-Adam over {sum(vocabs)} rows) "
                      f"of the DCN workload at batch {batch}, median {med * 1e3:.1f} ms/step, "
                      f"torch-CPU fp32 op-for-op restatement (TF 1.14 not installable)"}


if __name__ == "__main__":
    import argparse
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dcn")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--fields", type=int, default=26)
    ap.add_argument("--emb", type=int, default=16)
    ap.add_argument("--max-vocab", type=int, default=1_000_000)
    ap.add_argument("--seconds", type=float, default=15.0)
    ap.add_argument("--threads", type=int, default=None)
    a = ap.parse_args()
    print(json.dumps(run(a.model, a.batch, a.fields, a.emb, a.max_vocab, a.seconds, a.threads)), flush=True)
