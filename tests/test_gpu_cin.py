"""-m gpu parity of the CIN MFMA kernels vs the op-for-op oracle (einsum + matmul restatement
of cin_layer.py) — forward, pooled output, all three gradients, at several shapes incl. the
reference default (m=8, D=8, maps 50) and the benchmark shape (m=26, D=16, maps 128)."""
import pytest
import torch

from oracle import ref_ops as R
from recalgorithm_amd import ops
from recalgorithm_amd.variables import Variable, VariableStore
from tests.util import assert_close

pytestmark = pytest.mark.gpu


def _oracle(x0, xk, w, go, gp, dtype):
    """R.cin_layer (cin_layer.py:17-28) and its autograd in `dtype` -> (out, [dx0, dxk, dW])."""
    a = [t.to(dtype).requires_grad_(True) for t in (x0, xk, w)]
    r = R.cin_layer(a[0], a[1], a[2])
    torch.autograd.backward([r, r.sum(-1)], [go.to(dtype), gp.to(dtype)])
    return r.detach(), [t.grad for t in a]


@pytest.mark.parametrize("B,m,Hk,N,D", [
    (3, 4, 4, 5, 4), (17, 8, 8, 50, 8), (33, 8, 50, 50, 8), (64, 26, 26, 128, 16), (40, 26, 128, 128, 16),
    (9, 5, 7, 33, 32), (130, 26, 100, 100, 16),
    # the round-5 kernels (cin_contract2 / cin_filter_grad2 / cin_input_grad2) at their edges: 32 fields (16 steps per slab),
    # a partly filled second row block and a partly filled workgroup; 64 filter columns (two column tiles); three row blocks
    # (a second pass of one); few fields (the general forward / filter-gradient kernels beside the new input-gradient one)
    (5, 32, 33, 128, 16), (21, 17, 70, 64, 16), (12, 20, 96, 128, 16), (7, 9, 40, 128, 16),
])
def test_cin_layer_fwd_bwd(dev, B, m, Hk, N, D):
    gen = torch.Generator().manual_seed(B * 7 + N)
    x0 = torch.randn(B, m, D, generator=gen)
    xk = torch.randn(B, Hk, D, generator=gen)
    w = torch.randn(1, Hk * m, N, generator=gen) / (Hk * m) ** 0.5
    store = VariableStore(dev)
    wv = Variable("f", w.to(dev))
    x0d, xkd = x0.to(dev).requires_grad_(True), xk.to(dev).requires_grad_(True)
    out, pooled = ops.cin_layer(store, x0d, xkd, wv)
    go = torch.randn(B, N, D, generator=gen)
    gp = torch.randn(B, N, generator=gen)
    ref, a = _oracle(x0, xk, w, go, gp, torch.float64)
    r32, a32 = _oracle(x0, xk, w, go, gp, torch.float32)               # the reference arithmetic's own fp32 rounding
    # every output element is a contraction over Hk*m (fwd), N*m (dxk) or Hk*N (dx0) fp32 terms —
    # up to 1.6e4 — accumulated by the MFMA in two levels: judged with the reduction floor of tests/util.py,
    # and (ref32) the kernels may not leave materially more elements outside the strict bound than the fp32 oracle
    assert_close(out, ref, what="cin fwd", reduced=True, ref32=r32)
    assert_close(pooled, ref.sum(-1), what="cin pooled", reduced=True, ref32=r32.sum(-1))
    torch.autograd.backward([out, pooled], [go.to(dev), gp.to(dev)])
    assert_close(x0d.grad, a[0], what="cin dx0", reduced=True, ref32=a32[0])
    assert_close(xkd.grad, a[1], what="cin dxk", reduced=True, ref32=a32[1])
    assert_close(wv.grad, a[2], what="cin dW", reduced=True, ref32=a32[2])


def test_cin_one_hot_filter_selects_pair(dev):
    """A one-hot filter (i,j)->n gives out[:, n, :] = xk[:, i, :] * x0[:, j, :]  (SURVEY §8c (3))."""
    gen = torch.Generator().manual_seed(0)
    B, m, Hk, N, D = 8, 6, 5, 4, 16
    x0, xk = torch.randn(B, m, D, generator=gen), torch.randn(B, Hk, D, generator=gen)
    w = torch.zeros(1, Hk * m, N)
    picks = [(0, 0), (4, 5), (2, 3), (1, 4)]
    for n, (i, j) in enumerate(picks):
        w[0, i * m + j, n] = 1.0
    store = VariableStore(dev)
    out, _ = ops.cin_layer(store, x0.to(dev), xk.to(dev), Variable("f", w.to(dev)))
    for n, (i, j) in enumerate(picks):
        assert torch.equal(out[:, n, :].cpu(), xk[:, i, :] * x0[:, j, :])


def test_cin_only_pool_gradient(dev):
    """xdeepfm uses only the pooled output of the last layer: g_out is None there."""
    gen = torch.Generator().manual_seed(3)
    B, m, Hk, N, D = 20, 8, 8, 16, 8
    x0 = torch.randn(B, m, D, generator=gen)
    w = torch.randn(1, Hk * m, N, generator=gen) * 0.1
    store = VariableStore(dev)
    wv = Variable("f", w.to(dev))
    x0d = x0.to(dev).requires_grad_(True)
    out, pooled = ops.cin_layer(store, x0d, x0d, wv)      # first layer: xk is x0
    pooled.sum().backward()
    a0 = x0.double().requires_grad_(True)
    R.cin_layer(a0, a0, w.double()).sum().backward()
    assert_close(x0d.grad, a0.grad, what="cin dx0 (xk is x0, pool grad only)")


@pytest.mark.parametrize("B,m,Hk,N,D", [(24, 8, 8, 200, 8), (24, 8, 200, 200, 8), (10, 26, 150, 130, 16)])
def test_cin_layers_wider_than_one_launch(dev, B, m, Hk, N, D):
    """--cin_layer_feature_maps=200,200: more than 128 maps on either side is tiled over several launches."""
    gen = torch.Generator().manual_seed(B + Hk + N)
    x0 = torch.randn(B, m, D, generator=gen)
    xk = torch.randn(B, Hk, D, generator=gen)
    w = torch.randn(1, Hk * m, N, generator=gen) / (Hk * m) ** 0.5
    store = VariableStore(dev)
    wv = Variable("f", w.to(dev))
    x0d, xkd = x0.to(dev).requires_grad_(True), xk.to(dev).requires_grad_(True)
    out, pooled = ops.cin_layer(store, x0d, xkd, wv)
    go, gp = torch.randn(B, N, D, generator=gen), torch.randn(B, N, generator=gen)
    ref, a = _oracle(x0, xk, w, go, gp, torch.float64)
    r32, a32 = _oracle(x0, xk, w, go, gp, torch.float32)
    assert_close(out, ref, what="wide cin fwd", reduced=True, ref32=r32)
    assert_close(pooled, ref.sum(-1), what="wide cin pooled", reduced=True, ref32=r32.sum(-1))
    torch.autograd.backward([out, pooled], [go.to(dev), gp.to(dev)])
    assert_close(x0d.grad, a[0], what="wide cin dx0", reduced=True, ref32=a32[0])
    assert_close(xkd.grad, a[1], what="wide cin dxk", reduced=True, ref32=a32[1])
    assert_close(wv.grad, a[2], what="wide cin dW", reduced=True, ref32=a32[2])
