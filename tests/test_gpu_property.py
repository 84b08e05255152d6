"""Property sweeps (Hypothesis, derandomised: the same examples every run) over the shapes and id distributions the
fixed parametrisations of the other GPU tests do not visit: the owner-computes scatter (csrc/sparse.hip) in GRAD mode
against the fp64 scatter-add for any (rows, K, n_ex, F, hot-row share, OOV share), its bit-reproducibility, the deferred
Adam against the dense TF1 pass at random sweep periods, CrossNet and the fused DeepFM sparse path against the oracle."""
import math

import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import ref_ops as R
from recalgorithm_amd import ops
from recalgorithm_amd.variables import EmbeddingArena, Variable, VariableStore
from tests.util import assert_bit_exact, assert_close

pytestmark = pytest.mark.gpu
SWEEP = settings(max_examples=20, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))


class _Store:
    def __init__(self, dev):
        self.opt_state = {"step": torch.zeros(1, dtype=torch.int64, device=dev), "lr_t": torch.zeros(1, device=dev)}
        self.arenas = {}


def _arena(dev, rows, K, seed=1, name="t"):
    ar = EmbeddingArena(name, K, dev, seed=seed)
    ar.add_table("t0", rows)
    ar.materialize()
    return ar


@SWEEP
@given(rows=st.integers(1, 6000), K=st.sampled_from([1, 2, 3, 4, 8, 12, 16, 32, 64]), n_ex=st.integers(1, 1500),
       F=st.integers(1, 6), hot=st.floats(0.0, 0.9), oov=st.floats(0.0, 0.5), seed=st.integers(0, 4))
def test_scatter_grad_any_shape(dev, rows, K, n_ex, F, hot, oov, seed):
    """sum over the requests of a row == fp64 scatter-add, whatever the shape, the width (float4 and scalar lanes, staged
    and unstaged tiles), the share of one hot row (tile partial sums, whole-workgroup rows, oversize buckets) and of
    missing ids; and the result is bit-identical when the same plan runs again from scratch."""
    from recalgorithm_amd import sparse
    gen = torch.Generator().manual_seed(seed * 7919 + rows)
    ids = torch.randint(0, rows, (n_ex, F), generator=gen)
    ids[torch.rand(n_ex, F, generator=gen) < hot] = rows // 2
    ids[torch.rand(n_ex, F, generator=gen) < oov] = -1
    g = torch.randn(n_ex, F * K, generator=gen)
    ok = ids.reshape(-1) >= 0
    ref = torch.zeros(rows, K, dtype=torch.float64).index_add_(0, ids.reshape(-1)[ok], g.reshape(-1, K).double()[ok])
    outs = []
    for _ in range(2):
        ar = _arena(dev, rows, K)
        store = _Store(dev)
        store.arenas["t"] = ar
        with torch.enable_grad():
            src = sparse.begin_lookup(ar, store, ids.to(dev), None, None, 0, n_ex, F)
        assert src is not None
        src.set_grad(g.to(dev))
        sparse.materialize_grads(store)
        outs.append(ar.grad.clone())
    assert_bit_exact(outs[0], outs[1], "scatter is bit-reproducible")
    assert_close(outs[0], ref, what=f"scatter GRAD rows={rows} K={K} n_ex={n_ex} F={F} hot={hot:.2f}", reduced=True)


@settings(max_examples=8, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(rows=st.integers(50, 3000), K=st.sampled_from([1, 4, 8, 16]), period=st.integers(1, 9), seed=st.integers(0, 4))
def test_deferred_adam_equals_dense_adam_any_period(dev, rows, K, period, seed):
    """Deferred exact Adam (replay on catch-up / sweep / flush) == the dense TF1 Adam pass over every row, bit for bit, for
    any sweep period and table size."""
    import os
    from recalgorithm_amd import sparse
    old = os.environ.get("RECALGO_ADAM_SWEEP_PERIOD")
    os.environ["RECALGO_ADAM_SWEEP_PERIOD"] = str(period)
    try:
        gen = torch.Generator().manual_seed(seed * 31 + rows)
        A, Bn = _arena(dev, rows, K, seed=9, name="a"), _arena(dev, rows, K, seed=9, name="b")
        stA, stB = _Store(dev), _Store(dev)
        stA.arenas["a"], stB.arenas["b"] = A, Bn
        lr, F, n_ex = 0.01, 2, 90
        for step in range(1, 2 * period + 4):
            lo = (step * 53) % max(rows - 40, 1)
            ids = (lo + torch.randint(0, 40, (n_ex, F), generator=gen)).clamp_(max=rows - 1)
            far = torch.randint(0, rows, (n_ex, F), generator=gen)
            ids = torch.where(torch.rand(n_ex, F, generator=gen) < 0.15, far, ids)
            g = torch.randn(n_ex, F * K, generator=gen).to(dev)
            with torch.enable_grad():
                sB = sparse.begin_lookup(Bn, stB, ids.to(dev), None, None, 0, n_ex, F)
                sA = sparse.begin_lookup(A, stA, ids.to(dev), None, None, 0, n_ex, F)
            sA.set_grad(g)
            sB.set_grad(g)
            sparse.materialize_grads(stA)
            sparse.new_forward(stA)
            ops.adam_tf1_advance_(stA.opt_state["step"], stA.opt_state["lr_t"], lr)
            ops.adam_tf1_(A.weight.view(-1), A.grad.view(-1), A.m.view(-1), A.v.view(-1), step=-1, lr=lr,
                          lr_t_dev=stA.opt_state["lr_t"])
            stB.opt_state["step"] += 1
            sparse.apply(Bn, False, stB.opt_state["step"], lr, 0.9, 0.999, 1e-8)
        sparse.sync_store(stB)
        for a, b, nm in ((A.weight, Bn.weight, "w"), (A.m, Bn.m, "m"), (A.v, Bn.v, "v")):
            assert_bit_exact(b, a, f"deferred vs dense TF1 Adam (period {period}): {nm}")
    finally:
        if old is None:
            os.environ.pop("RECALGO_ADAM_SWEEP_PERIOD", None)
        else:
            os.environ["RECALGO_ADAM_SWEEP_PERIOD"] = old


@SWEEP
@given(B=st.integers(1, 700), d4=st.integers(1, 130), L=st.integers(1, 5), seed=st.integers(0, 4))
def test_cross_stack_any_shape(dev, B, d4, L, seed):
    d = 4 * d4
    gen = torch.Generator().manual_seed(seed * 101 + d + L)
    x0 = torch.randn(B, d, generator=gen)
    w = torch.randn(L, d, generator=gen) / math.sqrt(d)
    b = torch.randn(L, d, generator=gen) * 0.1
    store = VariableStore(dev)
    wv, bv = Variable("w", w.to(dev)), Variable("b", b.to(dev))
    x0d = x0.to(dev).requires_grad_(True)
    out = ops.cross_stack(store, x0d, wv, bv)
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x0, w, b))
    ref = R.cross_stack(x64, [w64[l].unsqueeze(1) for l in range(L)], [b64[l].unsqueeze(1) for l in range(L)])
    assert_close(out, ref, what=f"cross fwd B={B} d={d} L={L}")
    g = torch.randn(B, d, generator=gen)
    out.backward(g.to(dev))
    ops.flush_dense_splits()
    ref.backward(g.double())
    assert_close(x0d.grad, x64.grad, what="cross dx0")
    assert_close(wv.grad, w64.grad, what="cross dw", reduced=True)
    assert_close(bv.grad, b64.grad, what="cross db", reduced=True)


@SWEEP
@given(B=st.integers(1, 600), F=st.integers(1, 30), K=st.sampled_from([4, 8, 16, 32]), seed=st.integers(0, 4))
def test_deepfm_sparse_forward_any_shape(dev, B, F, K, seed):
    """emb is the bit-exact gather; FM second order == the reference's sum-square form (deepfm.py:184-200) and the
    brute-force pair sum; first order == the indicator-dense(1) sum."""
    gen = torch.Generator().manual_seed(seed * 13 + B + F)
    vocabs = [int(v) for v in torch.randint(2, 60, (F,), generator=gen)]
    ar = EmbeddingArena("e", K, dev, seed=3)
    w1 = EmbeddingArena("w", 1, dev, seed=4)
    for i, v in enumerate(vocabs):
        ar.add_table(f"t{i}", v)
        w1.add_table(f"t{i}", v)
    ar.materialize()
    w1.materialize()
    rb = torch.tensor([ar.tables[f"t{i}"][0] for i in range(F)], dtype=torch.int64, device=dev)
    ids = torch.stack([torch.randint(0, v, (B,), generator=gen) for v in vocabs], 1)
    ids[torch.rand(B, F, generator=gen) < 0.1] = -1
    store = VariableStore(dev)
    bias = Variable("b", torch.full((1,), 0.25, device=dev))
    with torch.no_grad():
        emb, fm1, fm2 = ops.deepfm_sparse(store, ids.to(dev), ar, w1, bias, rb)
    rows = (ids + rb.cpu().unsqueeze(0)).clamp(min=0)
    E = torch.where((ids >= 0).unsqueeze(-1), ar.weight.cpu()[rows], torch.zeros(1, K))
    assert_bit_exact(emb.cpu().view(B, F, K), E, "deep_input is the gathered rows")
    E64 = E.double()
    ref2 = R.fm_second_order([E64[:, f] for f in range(F)])
    # 0.5 * ((sum e)^2 - sum e^2): exactly zero for F = 1 — judged at the scale of the squares that cancel
    scale = float(E64.abs().sum(1).pow(2).sum(-1).max())
    assert_close(fm2.view(-1), ref2.view(-1), what=f"FM second order B={B} F={F} K={K}", floor=2e-6 * scale)
    W = torch.where(ids >= 0, w1.weight.cpu()[rows].squeeze(-1), torch.zeros(1)).double()
    assert_close(fm1.view(-1), W.sum(1) + 0.25, what="FM first order")
