"""Shared parity helpers.

Tolerance (BASELINE.json north_star): index gather bit-exact; fp32 interaction outputs and
gradients within 1e-5 relative.  "Relative" is evaluated per element against the fp64 oracle
with an absolute floor tied to the tensor's own RMS, so that elements that are small only
through cancellation are judged at the scale of the terms that produced them:
    |hip - ref64| <= rtol * (|ref64| + rms(ref64))
Outputs that are sums over the batch (scatter-added row gradients of hot Zipf rows, weight
gradients: up to ~1e5 fp32 terms whose sum cancels) pass `reduced=True`, which adds the fp32
accumulation floor 1e-6 * max|ref64| — the summation-order error of any fp32 implementation
(TF1-CPU included) is relative to sum|terms|, not to the cancelled result.  `floor` adds a measured
absolute floor (callers pass 4x the fp32-vs-fp64 deviation of the oracle itself on that tensor).
"""
import torch

RTOL = 1e-5


def assert_close(a, ref, rtol=RTOL, what="", reduced=False, floor=0.0):
    a = a.detach().double().cpu().reshape(-1)
    ref = ref.detach().double().cpu().reshape(-1)
    assert a.shape == ref.shape, f"{what}: shape {a.shape} vs {ref.shape}"
    assert torch.isfinite(a).all(), f"{what}: non-finite values"
    rms = ref.pow(2).mean().sqrt() if ref.numel() else ref.new_zeros(())
    tol = rtol * (ref.abs() + rms)
    if reduced and ref.numel():
        tol = tol + 1e-6 * ref.abs().max()
    if floor:
        tol = tol + floor
    err = (a - ref).abs()
    bad = err > tol
    if bad.any():
        i = int((err / tol.clamp(min=1e-300)).argmax())
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{a.numel()} elements outside rtol={rtol}; worst idx {i}: "
            f"got {a[i].item():.9g} ref {ref[i].item():.9g} err {err[i].item():.3g} tol {tol[i].item():.3g}")


def assert_bit_exact(a, ref, what=""):
    a = a.detach().cpu()
    ref = ref.detach().cpu()
    assert a.dtype == ref.dtype and a.shape == ref.shape, what
    assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a,
                       ref.view(torch.int32) if ref.dtype == torch.float32 else ref), \
        f"{what}: not bit-exact"


def zipf_ids(gen, B, vocab, oov_frac=0.01, s=1.05):
    """Truncated Zipf(s) ids in [0, vocab), with a fraction of -1 (OOV) entries."""
    ranks = torch.arange(1, vocab + 1, dtype=torch.float64)
    p = ranks.pow(-s)
    p = p / p.sum()
    ids = torch.multinomial(p, B, replacement=True, generator=gen)
    if oov_frac > 0:
        oov = torch.rand(B, generator=gen) < oov_frac
        ids = torch.where(oov, torch.full_like(ids, -1), ids)
    return ids
