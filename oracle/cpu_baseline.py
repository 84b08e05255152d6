"""oracle/cpu_baseline.py — TEST/BENCH INFRASTRUCTURE ONLY (see oracle/__init__.py).

Times the unfused, op-for-op torch-CPU restatement (oracle/ref_models.py) of one full training
step — forward, autograd backward, TF1 dense Adam on every variable including all embedding
tables — on the host cores of the box that runs it.  This is the stand-in for "the TF1 CPU
path" (TensorFlow 1.14 cannot be installed here); it is reported as
`cpu_baseline.kind = "port"`, never as TensorFlow.

Run as a child process by bench.py (`python -m oracle.cpu_baseline --model dcn ...`): the model's
variables are created by the mirrored model_fn's launch-free registration pass on the CPU (same
names, shapes and initialisers as the GPU run), then handed to the oracle as plain tensors.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

from . import ref_models as M
from . import ref_ops as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(model: str, batch=4096, fields=26, emb=16, max_vocab=1_000_000, seconds=15.0, threads=None, also_threads=None):
    # threads actually used: the cores this process may run on, capped at 32 (the per-op work of
    # a 4096-example batch does not scale past that; TF1's default intra-op pool has the same
    # problem on many-core hosts)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = threads or max(1, min(avail, 32))
    torch.set_num_threads(threads)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    from recalgorithm_amd.feature_column import Ragged

    # xDeepFM's reference graph materialises the (B, D, Hk*m) outer product (872 MB at B=4096):
    # the CPU sample uses a quarter batch to stay inside a bounded memory / time budget
    # (AFM: the pair tensor + attention net over B x 325 rows; FFM: the oracle walks bags in Python — bounded samples)
    cpu_batch = {"xdeepfm": min(batch, 1024), "afm": min(batch, 512), "ffm": min(batch, 64)}.get(model, batch)
    args = bench.parse_args(["--model", model, "--batch", str(cpu_batch), "--fields", str(fields), "--emb", str(emb),
                             "--max-vocab", str(max_vocab)])
    est, spec, feats, labels, workload = bench.build_estimator(args, torch.device("cpu"))
    P = {k: v.detach().clone().requires_grad_(True) for k, v in est.store.named_arrays().items()}
    cf = {k: ((v.values, v.offsets) if isinstance(v, Ragged) else v) for k, v in feats.items()}
    fn = getattr(M, model)
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    v2 = {k: torch.zeros_like(v) for k, v in P.items()}
    rows = int(sum(a.weight.shape[0] for a in est.store.arenas.values()))
    lr = float(est.params["learning_rate"])

    extra = {}
    if model == "nfm":          # its hard-coded dropout (nfm.py:170): a fixed keep mask stands in for TF's random stream
        gen = torch.Generator().manual_seed(1)
        K = int(est.params["category_feature_columns"][0].dimension)
        extra["dropout_masks"] = [(torch.rand(cpu_batch, K, generator=gen) < 0.9).float()]

    def step(t):
        for p in P.values():
            p.grad = None
        out = fn(P, cf, labels, est.params, training=True, **extra)
        out["loss"].backward()
        with torch.no_grad():
            for k, p in P.items():
                if "moving_" in k:
                    continue
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                R.adam_tf1_step(p, g, m[k], v2[k], t, lr)          # dense: all rows decay (TF1)
        return float(out["loss"].detach())

    t0 = time.perf_counter()
    step(1)                                            # warm-up (allocations, MKL init)
    warm = time.perf_counter() - t0
    t = 2

    def sample(secs, max_steps=50, min_steps=1):
        nonlocal t
        times, t_start = [], time.perf_counter()
        while True:
            t0 = time.perf_counter()
            step(t)
            times.append(time.perf_counter() - t0)
            t += 1
            if (time.perf_counter() - t_start > secs and len(times) >= min_steps) or len(times) >= max_steps:
                break
        return times
    times = sample(seconds)
    med = float(np.median(times))
    out = {"value": round(cpu_batch / med, 1), "unit": "examples/s", "cores": threads, "kind": "port",
           "sample": f"{len(times)} full train steps (fwd+bwd+dense TF1-Adam over {rows} embedding rows) of "
                     f"[{workload.replace(f'batch {cpu_batch}/GPU', f'batch {cpu_batch}')}], median "
                     f"{med * 1e3:.1f} ms/step (warm-up {warm:.1f} s), torch-CPU fp32 op-for-op restatement of the "
                     f"reference graph (TF 1.14 not installable)"}
    # the primary sample leaves first: a parent that has to cut the child off still has it (bench.py reads the last complete line)
    import json
    print(json.dumps(out), flush=True)
    others = []
    quota = None
    try:
        qv, pv = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if qv == "max" else max(1, int(int(qv) / int(pv)))
    except (OSError, ValueError):
        pass
    more = [also_threads // 2, also_threads] if also_threads and also_threads > 2 * threads else ([also_threads] if also_threads else [])
    if quota and quota < threads:
        more.insert(0, quota)                          # one thread per CPU of the cgroup's quota
    for n in more:
        if n == threads or n < 1:
            continue
        # the same process, model and state at another intra-op thread count ("N = all host cores"; half of them =
        # the physical cores of an SMT host): no second import / build / warm-up, a bounded number of steps — an oversubscribed
        # pool can take seconds per step
        torch.set_num_threads(n)
        t0 = time.perf_counter()
        step(t); t += 1                                # (the pool's threads are spawned here, not inside the timed steps)
        first = time.perf_counter() - t0
        budget = max(3.0, seconds / 3)
        t2 = sample(budget, max_steps=20, min_steps=1) if first < 2 * budget else [first]
        m2 = float(np.median(t2))
        others.append({"cores": n, "value": round(cpu_batch / m2, 1), "unit": "examples/s",
                       "note": f"{len(t2)} steps of the same process at {n} intra-op threads, median {m2 * 1e3:.1f} ms/step"})
        out["other_samples"] = others
        print(json.dumps(out), flush=True)
    return out


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
    ap.add_argument("--also-threads", type=int, default=None)
    a = ap.parse_args()
    print(json.dumps(run(a.model, a.batch, a.fields, a.emb, a.max_vocab, a.seconds, a.threads, a.also_threads)), flush=True)
