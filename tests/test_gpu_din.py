"""-m gpu parity of the DIN attention kernel vs the op-for-op oracle (tile/concat/dense x3/mask/
softmax restatement of din_attention.py), both branches, plus the reference demo's edge cases."""
import pytest
import torch

from oracle import ref_ops as R
from recalgorithm_amd import ops
from recalgorithm_amd.variables import Variable, VariableStore
from tests.util import assert_close

pytestmark = pytest.mark.gpu


def make(B, T, H, gen, dev, bias_scale=0.1):
    q = torch.randn(B, H, generator=gen)
    lens = torch.randint(0, T + 1, (B,), generator=gen)
    lens[0] = 0
    if B > 1:
        lens[1] = T
    keys = torch.randn(B, T, H, generator=gen)
    keys = keys * (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).unsqueeze(-1)     # zero padded
    ws = [torch.randn(4 * H, 64, generator=gen) / (4 * H) ** 0.5, torch.randn(64, generator=gen) * bias_scale,
          torch.randn(64, 32, generator=gen) / 8.0, torch.randn(32, generator=gen) * bias_scale,
          torch.randn(32, 1, generator=gen) / 32 ** 0.5, torch.randn(1, generator=gen) * bias_scale]
    vs = [Variable(f"v{i}", w.to(dev)) for i, w in enumerate(ws)]
    return q, keys, lens, ws, vs


@pytest.mark.parametrize("is_softmax", [False, True])
@pytest.mark.parametrize("B,T,H", [(2, 3, 4), (37, 50, 16), (260, 64, 16), (19, 7, 8)])
def test_din_attention(dev, B, T, H, is_softmax):
    gen = torch.Generator().manual_seed(B + T)
    q, keys, lens, ws, vs = make(B, T, H, gen, dev)
    store = VariableStore(dev)
    qd, kd = q.to(dev).requires_grad_(True), keys.to(dev).requires_grad_(True)
    out = ops.din_attention(store, qd, kd, lens.to(dev), vs, is_softmax)
    g = torch.randn(B, H, generator=gen)
    a = [t.double().requires_grad_(True) for t in [q, keys] + ws]
    ref = R.din_attention(a[0], a[1], lens, *a[2:], is_softmax=is_softmax)
    ref.backward(g.double())
    a32 = [t.clone().requires_grad_(True) for t in [q, keys] + ws]     # the oracle in float32 (strict-bound guard)
    r32 = R.din_attention(a32[0], a32[1], lens, *a32[2:], is_softmax=is_softmax)
    r32.backward(g)
    assert_close(out, ref, what="din fwd", ref32=r32)
    out.backward(g.to(dev))
    assert_close(qd.grad, a[0].grad, what="din dq", ref32=a32[0].grad)
    assert_close(kd.grad, a[1].grad, what="din dkeys", ref32=a32[1].grad)
    for i, nm in enumerate(["f1_w", "f1_b", "f2_w", "f2_b", "f3_w", "f3_b"]):
        if nm == "f3_b" and is_softmax:
            # softmax is shift invariant: d/d(f3 bias) = sum_t ds_t is exactly 0 in exact arithmetic
            # whenever a row has no padded position, so the reference value is pure cancellation
            # noise; judge it at the scale of its terms (|d f3_w| is sum_t ds_t * h2, same size)
            scale = float(a[6].grad.abs().max())
            assert float((vs[i].grad.cpu().double() - a[7].grad).abs().max()) <= 1e-5 * max(scale, 1e-30)
            continue
        assert_close(vs[i].grad, a[2 + i].grad, what=f"din d{nm}", reduced=True, ref32=a32[2 + i].grad)


def test_din_length_zero_edge_cases(dev):
    """keys_length = 0: non-softmax -> exact zeros (the reference demo, din_attention.py:52);
    softmax -> uniform weights over the zero-padded keys, i.e. zeros too (quirk B-6)."""
    gen = torch.Generator().manual_seed(0)
    q, keys, lens, ws, vs = make(4, 5, 16, gen, dev)
    store = VariableStore(dev)
    for sm in (False, True):
        out = ops.din_attention(store, q.to(dev), keys.to(dev), lens.to(dev), vs, sm)
        assert torch.equal(out[0].cpu(), torch.zeros(16)), sm
    # softmax with length 0 and NON-zero "padding": uniform mean of the keys
    k2 = torch.randn(1, 5, 16, generator=gen)
    out = ops.din_attention(store, q[:1].to(dev), k2.to(dev), torch.zeros(1, dtype=torch.int32, device=dev), vs, True)
    assert_close(out, k2.mean(dim=1), what="uniform softmax over pads")


def test_din_permutation_invariance(dev):
    """The pooling is invariant to permuting the valid history positions."""
    gen = torch.Generator().manual_seed(2)
    B, T, H = 8, 20, 16
    q, keys, lens, ws, vs = make(B, T, H, gen, dev)
    lens[:] = T
    keys = torch.randn(B, T, H, generator=gen)
    perm = torch.randperm(T, generator=gen)
    store = VariableStore(dev)
    for sm in (False, True):
        o1 = ops.din_attention(store, q.to(dev), keys.to(dev), lens.to(dev), vs, sm)
        o2 = ops.din_attention(store, q.to(dev), keys[:, perm].contiguous().to(dev), lens.to(dev), vs, sm)
        assert_close(o2, o1.double(), what="permutation")
