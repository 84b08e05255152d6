"""Diagnostic (GPU): where do the strict-bound counts of a model step come from?  For every gradient of a bench model:
elements outside the strict bound for the HIP path and for the fp32 oracle, and the ratio of the RMS errors."""
import sys

import torch

sys.path.insert(0, ".")
import bench
from oracle import ref_models as M
from recalgorithm_amd import ops
from recalgorithm_amd.estimator import ModeKeys
from recalgorithm_amd.variables import named_grads
from tests.util import strict_violations

dev = torch.device("cuda:0")


def model(name):
    args = bench.parse_args(["--model", name, "--batch", "4096", "--max-vocab", "100000"])
    est, spec, feats, labels, _ = bench.build_estimator(args, dev)
    fn = getattr(M, name)

    def inputs(dt):
        P = {k: v.detach().cpu().to(dt).requires_grad_(True) for k, v in est.store.named_arrays().items()}
        cf = {k: (v.cpu() if isinstance(v, torch.Tensor) else (v.values.cpu(), v.offsets.cpu())) for k, v in feats.items()}
        cf = {k: (v.to(dt) if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in cf.items()}
        return P, cf, {k: v.cpu().to(dt) for k, v in labels.items()}
    P, cf, cl = inputs(torch.float64)
    fn(P, cf, cl, est.params, training=True)["loss"].backward()
    P32, cf32, cl32 = inputs(torch.float32)
    fn(P32, cf32, cl32, est.params, training=True)["loss"].backward()
    spec_ = est._call_model_fn(feats, labels, ModeKeys.TRAIN)
    spec_.loss.backward()
    grads = named_grads(est.store)
    print(f"== {name}: tensor | n | hip outside | fp32-oracle outside | rms err hip / fp32 | max|ref|")
    for k, p in P.items():
        if p.grad is None or "embedding" in k:
            continue
        ref = p.grad.reshape(-1)
        h = grads[k].detach().cpu().double().reshape(-1)
        o = P32[k].grad.double().reshape(-1)
        eh, eo = (h - ref).pow(2).mean().sqrt(), (o - ref).pow(2).mean().sqrt()
        print(f"{k} | {ref.numel()} | {strict_violations(h, ref)[0]} | {strict_violations(o, ref)[0]} | {float(eh / eo.clamp(min=1e-300)):.2f} | {float(ref.abs().max()):.3g}")


def dense_case(Mr, K, N, gscale):
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(Mr, K, generator=gen) * 0.25
    g = torch.randn(Mr, N, generator=gen) * gscale
    y = torch.randn(Mr, N, generator=gen)
    g2 = (g * (y > 0)).double()
    ref = x.double().t() @ g2
    o = (x.t() @ g2.float()).double()
    dw = torch.empty(K, N, device=dev)
    db = torch.empty(N, device=dev)
    ops.dense_bwd_weights(x.to(dev), g.to(dev), y.to(dev), dw, db)
    h = dw.cpu().double()
    eh, eo = (h - ref).pow(2).mean().sqrt(), (o - ref).pow(2).mean().sqrt()
    print(f"dense wgrad {Mr}x{K}x{N} gscale {gscale}: hip outside {strict_violations(h.reshape(-1), ref.reshape(-1))[0]}, fp32 torch outside "
          f"{strict_violations(o.reshape(-1), ref.reshape(-1))[0]} of {ref.numel()}; rms err ratio {float(eh / eo):.2f}")


for sh in [(4096, 416, 512), (4096, 416, 1024), (4096, 352, 1024), (4096, 1024, 512)]:
    for gs in (1.0, 1e-5):
        dense_case(*sh, gs)
for name in sys.argv[1:] or ["pnn", "fibinet", "dcn"]:
    model(name)
