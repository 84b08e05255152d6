"""-m gpu parity of the PNN product-layer kernels (K6) against the oracle's op-for-op restatement
of /root/reference algorithm/PNN/pnn.py:133-181 (the D-iteration loop, `pnn_product`)."""
import ctypes

import pytest
import torch

from oracle import ref_ops as R
from recalgorithm_amd import _lib
from tests.util import assert_close

pytestmark = pytest.mark.gpu


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("method", ["IPNN", "OPNN"])
@pytest.mark.parametrize("B,F,K,D", [(129, 26, 16, 64), (40, 8, 8, 33), (7, 3, 4, 5), (66, 70, 12, 16)])
def test_pnn_product_layer_pieces(dev, method, B, F, K, D):
    """lp = phi @ omega reproduces the reference loop; the feature / weight backward kernels
    reproduce autograd through the loop."""
    lib = _lib.load()
    m = {"IPNN": 0, "OPNN": 1}[method]
    gen = torch.Generator().manual_seed(B + D)
    emb = torch.randn(B, F * K, generator=gen) * 0.5
    lw = torch.randn(F * K, D, generator=gen) * 0.1
    pw = torch.randn(*((D, F) if method == "IPNN" else (D, K, K)), generator=gen) * 0.3
    bias = torch.randn(D, generator=gen) * 0.1
    g = torch.randn(B, D, generator=gen)
    ed, lwd, pwd, bd = (t.double().requires_grad_(True) for t in (emb, lw, pw, bias))
    _, lp_ref, out_ref = R.pnn_product(ed, lwd, pwd, bd, F, K, method)        # the literal D-iteration loop
    lp_ref.backward(g.double())
    ef, lwf, pwf, bf = (t.clone().requires_grad_(True) for t in (emb, lw, pw, bias))        # the oracle in float32
    _, lp32, _ = R.pnn_product(ef, lwf, pwf, bf, F, K, method)
    lp32.backward(g)

    T = lib.recalgo_pnn_feature_count(F, K, m)
    assert T == (F * (F + 1) // 2 if method == "IPNN" else K * (K + 1) // 2)
    eg, pg, gg = emb.to(dev), pw.to(dev), g.to(dev)
    # phi with a padded row stride (what ops._PnnProductFn does: float4-addressable GEMM operand): the kernel zero-fills
    # the padding columns, the backward skips them
    ld = (T + 3) // 4 * 4 + 4
    phi_pad = torch.full((B, ld), float("nan"), device=dev)
    omega = torch.empty(T, D, device=dev)
    _lib.check(lib.recalgo_pnn_features_fwd(_p(eg), B, F, K, m, _p(phi_pad), ld, _st()), "features fwd")
    assert float(phi_pad[:, T:].abs().max()) == 0.0
    phi = phi_pad[:, :T].contiguous()
    phi_t = torch.empty(B, T, device=dev)
    _lib.check(lib.recalgo_pnn_features_fwd(_p(eg), B, F, K, m, _p(phi_t), T, _st()), "features fwd (dense rows)")
    assert torch.equal(phi, phi_t)
    _lib.check(lib.recalgo_pnn_weights_fwd(_p(pg), D, F, K, m, _p(omega), _st()), "weights fwd")
    lp = phi.double() @ omega.double()
    assert_close(lp, lp_ref, what=f"{method} lp = phi @ omega", ref32=lp32)
    # backward pieces, chained in fp64 around the kernels
    dphi = (gg.double() @ omega.double().t()).float()
    domega = (phi.double().t() @ gg.double()).float()
    d_emb = torch.full((B, F * K), 0.25, device=dev)
    dphi_pad = torch.full((B, ld), float("nan"), device=dev)
    dphi_pad[:, :T] = dphi
    _lib.check(lib.recalgo_pnn_features_bwd(_p(eg), _p(dphi_pad), ld, B, F, K, m, _p(d_emb), 1, _st()), "features bwd")
    assert_close(d_emb - 0.25, ed.grad, what=f"{method} d_emb (accumulate)", reduced=True)
    d_emb2 = torch.empty_like(d_emb)
    _lib.check(lib.recalgo_pnn_features_bwd(_p(eg), _p(dphi), T, B, F, K, m, _p(d_emb2), 0, _st()), "features bwd")
    assert_close(d_emb2, ed.grad, what=f"{method} d_emb", reduced=True, ref32=ef.grad)
    dpw = torch.empty_like(pg)
    _lib.check(lib.recalgo_pnn_weights_bwd(_p(pg), _p(domega), D, F, K, m, _p(dpw), _st()), "weights bwd")
    assert_close(dpw, pwd.grad, what=f"{method} d_product_w", reduced=True, ref32=pwf.grad)
    if method == "OPNN":
        assert float(torch.tril(dpw, diagonal=-1).abs().max()) == 0.0        # quirk B-10


def test_ipnn_identity(dev):
    """lp_i == sum_{f,g} theta_if theta_ig <e_f, e_g>  (SURVEY.md §8c property 6)."""
    lib = _lib.load()
    B, F, K, D = 5, 6, 4, 3
    gen = torch.Generator().manual_seed(0)
    E = torch.randn(B, F, K, generator=gen)
    th = torch.randn(D, F, generator=gen)
    phi = torch.empty(B, F * (F + 1) // 2, device=dev)
    omega = torch.empty(F * (F + 1) // 2, D, device=dev)
    _lib.check(lib.recalgo_pnn_features_fwd(_p(E.to(dev)), B, F, K, 0, _p(phi), phi.shape[1], _st()), "f")
    _lib.check(lib.recalgo_pnn_weights_fwd(_p(th.to(dev)), D, F, K, 0, _p(omega), _st()), "w")
    gram = torch.einsum("bfk,bgk->bfg", E.double(), E.double())
    brute = torch.einsum("if,ig,bfg->bi", th.double(), th.double(), gram)
    assert_close(phi.double() @ omega.double(), brute, what="ipnn identity")


@pytest.mark.parametrize("B,F,K", [(4096, 26, 16), (37, 6, 8), (5, 2, 4), (64, 3, 12)])
def test_field_pair_logit_against_reference_double_loop(dev, B, F, K):
    """FwFM second order (reference algorithm/FwFM/fwfm.py:146-158): ops.field_pair_logit (IPNN Gram features +
    one-unit head over the scattered pair strengths) == the reference's double loop over i < j with
    index_from_upper_triangular ordering, values and gradients (fp64 loop as the reference)."""
    from recalgorithm_amd import ops
    from recalgorithm_amd.variables import Variable, VariableStore
    gen = torch.Generator().manual_seed(B * 31 + F)
    emb = torch.randn(B, F * K, generator=gen) * 0.5
    n = F * (F - 1) // 2
    r = torch.randn(n, generator=gen) * 0.3
    g = torch.randn(B, 1, generator=gen)
    ed, rd = emb.double().requires_grad_(True), r.double().requires_grad_(True)
    fields = [ed[:, f * K:(f + 1) * K] for f in range(F)]
    ref = torch.zeros(B, 1, dtype=torch.float64)
    index = 0
    for i in range(F - 1):
        for j in range(i + 1, F):
            ref = ref + rd[index] * (fields[i] * fields[j]).sum(1, keepdim=True)
            index += 1
    ref.backward(g.double())
    ef, rf = emb.clone().requires_grad_(True), r.clone().requires_grad_(True)              # the same loop in float32
    r32 = torch.zeros(B, 1)
    index = 0
    for i in range(F - 1):
        for j in range(i + 1, F):
            r32 = r32 + rf[index] * (ef[:, i * K:(i + 1) * K] * ef[:, j * K:(j + 1) * K]).sum(1, keepdim=True)
            index += 1
    r32.backward(g)
    store = VariableStore(dev)
    rv = Variable("fields_pair_strength/fields_pair_strength_weight", r.to(dev))
    x = emb.to(dev).requires_grad_(True)
    out = ops.field_pair_logit(store, x, rv, F, K)
    assert_close(out, ref.detach(), what="fwfm second-order logit", ref32=r32)
    out.backward(g.to(dev))
    assert_close(x.grad, ed.grad, what="fwfm d(embeddings)", ref32=ef.grad)
    assert_close(rv.grad, rd.grad, what="fwfm d(pair strengths)", reduced=True, ref32=rf.grad)


@pytest.mark.parametrize("B,F,K", [(4096, 26, 16), (37, 6, 8), (5, 2, 4)])
def test_field_pair_logit_in_the_fused_loss_tail(dev, B, F, K):
    """Inside a TRAIN step (ops.loss_seed) the pair head is a nn.LazyLogit: the loss launch forms the weighted pair sum, the
    loss and both gradients; d(pair strengths) reaches r.grad as F - 1 jobs of the deferred-sum launch.  Compared with
    mean sigmoid-CE(first + second) written as the reference's double loop, fp64."""
    from recalgorithm_amd import nn, ops
    from recalgorithm_amd.variables import Variable, VariableStore
    gen = torch.Generator().manual_seed(B * 17 + F)
    emb = torch.randn(B, F * K, generator=gen) * 0.5
    n = F * (F - 1) // 2
    r = torch.randn(n, generator=gen) * 0.3
    first = torch.randn(B, 1, generator=gen) * 0.2
    y = (torch.rand(B, 1, generator=gen) < 0.3).float()

    def oracle(dt):
        e, rr, f1 = (t.clone().to(dt).requires_grad_(True) for t in (emb, r, first))
        second, index = torch.zeros(B, 1, dtype=dt), 0
        for i in range(F - 1):
            for j in range(i + 1, F):
                second = second + rr[index] * (e[:, i * K:(i + 1) * K] * e[:, j * K:(j + 1) * K]).sum(1, keepdim=True)
                index += 1
        x = f1 + second
        loss = torch.nn.functional.binary_cross_entropy_with_logits(x, y.to(dt))
        loss.backward()
        return loss.detach(), e.grad, rr.grad, f1.grad
    l64, ge, gr, gf = oracle(torch.float64)
    l32, ge32, gr32, gf32 = oracle(torch.float32)
    store = VariableStore(dev)
    rv = Variable("fields_pair_strength/fields_pair_strength_weight", r.to(dev))
    x = emb.to(dev).requires_grad_(True)
    f1 = first.to(dev).requires_grad_(True)
    with ops.loss_seed(1.0):
        lazy = f1 + ops.field_pair_logit(store, x, rv, F, K)
        assert isinstance(lazy, nn.LazyLogit) and lazy.fusable()
        heads = [(k, len(ps)) for k, _, ps in lazy.heads]
        parts = [t for _, _, ps in lazy.heads for t in ps]
        loss, prob, logit = ops.logit_loss(store, y.to(dev), heads, None, parts, lazy.tensors)
        loss.backward(torch.ones((), device=dev))
    ops.flush_dense_splits()
    assert_close(loss.reshape(1), l64.reshape(1), what="fwfm fused loss", reduced=True, ref32=l32.reshape(1))
    assert_close(x.grad, ge, what="fwfm fused d(embeddings)", ref32=ge32)
    assert_close(f1.grad, gf, what="fwfm fused d(first order)", ref32=gf32)
    assert_close(rv.grad, gr, what="fwfm fused d(pair strengths)", reduced=True, ref32=gr32)
    # the same head outside the tail (materialize): the separate one-unit head kernels
    rv2 = Variable("fields_pair_strength/fields_pair_strength_weight", r.to(dev))
    x2 = emb.to(dev).requires_grad_(True)
    from recalgorithm_amd.variables import use_store
    with ops.loss_seed(1.0), use_store(store):
        t = ops.field_pair_logit(store, x2, rv2, F, K).materialize()
    t.backward(torch.ones_like(t))
    fields = emb.double().reshape(B, F, K)
    want = torch.stack([(fields[:, i] * fields[:, j]).sum() for i in range(F - 1) for j in range(i + 1, F)])
    assert_close(rv2.grad.reshape(-1), want, what="fwfm materialized d(pair strengths)", reduced=True)
