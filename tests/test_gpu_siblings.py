"""-m gpu parity of the sibling-model kernels (csrc/siblings.hip) against the oracle restatements of
nfm.py / afm.py / ffm.py: NFM bi-interaction pooling, AFM attention pooling, FFM field-aware pair dots."""
import pytest
import torch

from recalgorithm_amd import ops
from tests.util import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,F,K", [(3, 2, 4), (257, 7, 8), (4096, 26, 16), (50, 5, 3)])
def test_bi_interaction(dev, B, F, K):
    gen = torch.Generator().manual_seed(B + F)
    e = torch.randn(B, F * K, generator=gen)
    ed = e.to(dev).requires_grad_(True)
    out = ops.bi_interaction(ed, F, K)
    g = torch.randn(B, K, generator=gen)

    def oracle(dtype):
        a = e.to(dtype).requires_grad_(True)
        f = a.reshape(B, F, K)
        r = 0.5 * (f.sum(1) ** 2 - (f ** 2).sum(1))                    # nfm.py:163-167
        r.backward(g.to(dtype))
        return r.detach(), a.grad
    ref, gref = oracle(torch.float64)
    r32, g32 = oracle(torch.float32)                                   # the reference arithmetic's own fp32 rounding
    assert_close(out, ref, what="bi-interaction fwd", ref32=r32)
    out.backward(g.to(dev))
    assert_close(ed.grad, gref, what="bi-interaction bwd", ref32=g32)


@pytest.mark.parametrize("B,P,K", [(2, 1, 4), (130, 21, 8), (700, 325, 16), (33, 10, 5)])
def test_attention_pool(dev, B, P, K):
    gen = torch.Generator().manual_seed(P + K)
    pairs, att = torch.randn(B, P, K, generator=gen), torch.randn(B, P, generator=gen) * 3
    pd, ad = pairs.to(dev).requires_grad_(True), att.to(dev).requires_grad_(True)
    out = ops.attention_pool(pd, ad)
    g = torch.randn(B, K, generator=gen)

    def oracle(dtype):
        a, b = pairs.to(dtype).requires_grad_(True), att.to(dtype).requires_grad_(True)
        r = (a * torch.softmax(b, dim=1).unsqueeze(-1)).sum(1)         # afm.py:184-188
        r.backward(g.to(dtype))
        return r.detach(), a.grad, b.grad
    ref, ga, gb = oracle(torch.float64)
    r32, ga32, gb32 = oracle(torch.float32)
    assert_close(out, ref, what="attention pool fwd", ref32=r32)
    out.backward(g.to(dev))
    assert_close(pd.grad, ga, what="attention pool d pairs", ref32=ga32)
    assert_close(ad.grad, gb, what="attention pool d att", reduced=True, ref32=gb32)


@pytest.mark.parametrize("B,F,K", [(2, 2, 4), (200, 7, 8), (64, 26, 16), (31, 4, 3)])
def test_ffm_pairs(dev, B, F, K):
    gen = torch.Generator().manual_seed(F * K)
    x = torch.randn(B, F, F - 1, K, generator=gen)
    xd = x.reshape(B, -1).to(dev).requires_grad_(True)
    out = ops.ffm_pairs(xd, F, K)
    g = torch.randn(B, 1, generator=gen)

    def oracle(dtype):
        a = x.to(dtype).requires_grad_(True)
        r = sum((a[:, i, j - 1, :] * a[:, j, i, :]).sum(-1, keepdim=True) for i in range(F - 1) for j in range(i + 1, F))   # ffm.py:146-160
        r.backward(g.to(dtype))
        return r.detach(), a.grad.reshape(B, -1)
    ref, gref = oracle(torch.float64)
    r32, g32 = oracle(torch.float32)
    assert_close(out, ref, what="ffm pairs fwd", reduced=True, ref32=r32)
    out.backward(g.to(dev))
    assert_close(xd.grad, gref, what="ffm pairs bwd", ref32=g32)
