"""-m gpu parity of the FiBiNET kernels (K7 SENET, K8 bilinear interaction) against the oracle
restatement of /root/reference algorithm/FiBiNET/{senet,bilinear_interaction_layer}.py."""
import ctypes
import itertools

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


@pytest.mark.parametrize("B,F,K,ratio", [(257, 26, 16, 2), (64, 8, 8, 2), (33, 5, 4, 2), (19, 70, 32, 4)])
def test_senet_fwd_bwd(dev, B, F, K, ratio):
    lib = _lib.load()
    gen = torch.Generator().manual_seed(B + F)
    Rd = K // ratio
    E = torch.randn(B, F, K, generator=gen)
    w1 = torch.randn(F, Rd, generator=gen) * 0.4
    w2 = torch.randn(Rd, F, generator=gen) * 0.4
    g = torch.randn(B, F, K, generator=gen)
    Ed, w1d, w2d = (t.double().requires_grad_(True) for t in (E, w1, w2))
    ref = R.senet(Ed, w1d, w2d)
    ref.backward(g.double())
    # the same oracle in float32: the reference arithmetic's own rounding (strict-bound regression guard, tests/util.py)
    Ef, w1f, w2f = (t.clone().requires_grad_(True) for t in (E, w1, w2))
    r32 = R.senet(Ef, w1f, w2f)
    r32.backward(g)
    Eg, w1g, w2g, gg = (t.to(dev) for t in (E, w1, w2, g))
    v = torch.empty_like(Eg)
    a = torch.empty(B, F, device=dev)
    _lib.check(lib.recalgo_senet_fwd(_p(Eg), _p(w1g), _p(w2g), B, F, K, Rd, _p(v), _p(a), _st()), "senet fwd")
    assert_close(v, ref, what="senet out", ref32=r32, strict_slack=4 * K)       # (one gate value scales K outputs)
    ws = torch.empty(lib.recalgo_senet_bwd_workspace_bytes(B, F, K, Rd), dtype=torch.uint8, device=dev)
    dE, dw1, dw2 = torch.empty_like(Eg), torch.empty_like(w1g), torch.empty_like(w2g)
    _lib.check(lib.recalgo_senet_bwd(_p(Eg), _p(w1g), _p(w2g), _p(gg), B, F, K, Rd, _p(dE), 0, _p(dw1), _p(dw2),
                                     _p(ws), _st()), "senet bwd")
    assert_close(dE, Ed.grad, what="senet dE", ref32=Ef.grad, strict_slack=4 * K)
    assert_close(dw1, w1d.grad, what="senet dw1", reduced=True, ref32=w1f.grad)
    assert_close(dw2, w2d.grad, what="senet dw2", reduced=True, ref32=w2f.grad)
    # accumulate flag
    dE2 = torch.ones_like(Eg)
    _lib.check(lib.recalgo_senet_bwd(_p(Eg), _p(w1g), _p(w2g), _p(gg), B, F, K, Rd, _p(dE2), 1, _p(dw1), _p(dw2),
                                     _p(ws), _st()), "senet bwd acc")
    assert_close(dE2, Ed.grad + 1.0, what="senet dE accumulate")


def _weights(gen, F, K, btype):
    n = {"all": None, "each": F - 1, "interaction": F * (F - 1) // 2}[btype]
    shape = (K, K) if n is None else (n, K, K)
    return torch.randn(*shape, generator=gen) * 0.3


@pytest.mark.parametrize("btype", ["all", "each", "interaction"])
@pytest.mark.parametrize("B,F,K,nv", [(130, 26, 16, 2), (37, 8, 8, 2), (21, 8, 8, 1), (9, 5, 4, 1), (11, 12, 32, 2),
                                      (5, 4, 64, 2), (7, 28, 16, 2), (6, 40, 8, 1)])   # (351 / 741 pairs: the wave-per-example form)
def test_bilinear_fwd_bwd(dev, btype, B, F, K, nv):
    lib = _lib.load()
    T = {"all": 0, "each": 1, "interaction": 2}[btype]
    gen = torch.Generator().manual_seed(B * 7 + F + nv)
    xs = [torch.randn(B, F, K, generator=gen) for _ in range(nv)]
    ws_ = [_weights(gen, F, K, btype) for _ in range(nv)]
    P = (F - 1) * (F - 2) // 2
    g = torch.randn(B, P, nv * K, generator=gen)
    xd = [x.double().requires_grad_(True) for x in xs]
    wd = [w.double().requires_grad_(True) for w in ws_]
    ref = torch.cat([R.bilinear_interaction(x, w, btype) for x, w in zip(xd, wd)], dim=-1)
    assert ref.shape == (B, P, nv * K)
    ref.backward(g.double())
    xf = [x.clone().requires_grad_(True) for x in xs]
    wf = [w.clone().requires_grad_(True) for w in ws_]
    r32 = torch.cat([R.bilinear_interaction(x, w, btype) for x, w in zip(xf, wf)], dim=-1)
    r32.backward(g)
    xg = [x.to(dev) for x in xs] + [None] * (2 - nv)
    wg = [w.to(dev) for w in ws_] + [None] * (2 - nv)
    gg = g.to(dev)
    out = torch.empty(B, P, nv * K, device=dev)
    _lib.check(lib.recalgo_bilinear_fwd(_p(xg[0]), _p(wg[0]), _p(xg[1]), _p(wg[1]), B, F, K, T, _p(out), nv * K, 0,
                                        _st()), "bilinear fwd")
    assert_close(out, ref, what=f"bilinear[{btype}] out", ref32=r32)
    wsb = torch.empty(lib.recalgo_bilinear_bwd_workspace_bytes(B, F, K, nv, T), dtype=torch.uint8, device=dev)
    dx = [torch.empty_like(xg[0])] + ([torch.empty_like(xg[1])] if nv == 2 else [None])
    dw = [torch.zeros_like(wg[0])] + ([torch.zeros_like(wg[1])] if nv == 2 else [None])
    _lib.check(lib.recalgo_bilinear_bwd(_p(xg[0]), _p(wg[0]), _p(xg[1]), _p(wg[1]), _p(gg), nv * K, 0, B, F, K, T,
                                        _p(dx[0]), _p(dw[0]), _p(dx[1]), _p(dw[1]), _p(wsb), _st()), "bilinear bwd")
    for v in range(nv):
        assert_close(dx[v], xd[v].grad, what=f"bilinear[{btype}] dx{v}", ref32=xf[v].grad)
        assert_close(dw[v], wd[v].grad, what=f"bilinear[{btype}] dw{v}", reduced=True, ref32=wf[v].grad)
        # quirk B-3: the last field never participates -> exactly zero gradient
        assert float(dx[v][:, F - 1].abs().max()) == 0.0


def test_bilinear_pairs_and_last_field_quirk(dev):
    """Pair count (F-1)(F-2)/2 in combinations(range(F-1), 2) order; output independent of field F-1."""
    lib = _lib.load()
    B, F, K = 6, 7, 8
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, F, K, generator=gen).to(dev)
    w = torch.eye(K).to(dev)                       # W = I -> p_ij = e_i * e_j
    P = (F - 1) * (F - 2) // 2
    out = torch.empty(B, P, K, device=dev)
    _lib.check(lib.recalgo_bilinear_fwd(_p(x), _p(w), None, None, B, F, K, 0, _p(out), K, 0, _st()), "fwd")
    pairs = list(itertools.combinations(range(F - 1), 2))
    assert len(pairs) == P
    for p, (i, j) in enumerate(pairs):
        assert torch.equal(out[:, p], x[:, i] * x[:, j])
    x2 = x.clone()
    x2[:, F - 1] = 123.0
    out2 = torch.empty_like(out)
    _lib.check(lib.recalgo_bilinear_fwd(_p(x2), _p(w), None, None, B, F, K, 0, _p(out2), K, 0, _st()), "fwd")
    assert torch.equal(out, out2)


def test_bilinear_rejects_bad_arguments(dev):
    lib = _lib.load()
    x = torch.zeros(2, 5, 8, device=dev)
    w = torch.zeros(8, 8, device=dev)
    out = torch.zeros(2, 6, 8, device=dev)
    assert lib.recalgo_bilinear_fwd(_p(x), _p(w), None, None, 2, 5, 8, 3, _p(out), 8, 0, _st()) != 0    # bad type
    assert lib.recalgo_bilinear_fwd(_p(x), _p(w), None, None, 2, 5, 10, 0, _p(out), 8, 0, _st()) != 0   # bad K
    assert lib.recalgo_senet_fwd(_p(x), _p(w), _p(w), 2, 5, 8, 8, _p(x), None, _st()) != 0              # Rd !< K
