"""-m gpu parity tests: HIP kernels (through the C-ABI / ctypes) vs oracle/ref_ops.py on the
same seeded inputs.  Gradients of the oracle come from torch.autograd in float64."""
import math

import pytest
import torch

from oracle import ref_ops as R
from recalgorithm_amd import ops
from recalgorithm_amd.variables import EmbeddingArena, Variable, VariableStore
from tests.util import assert_bit_exact, assert_close, zipf_ids

pytestmark = pytest.mark.gpu


def make_arena(vocabs, K, dev, seed=0):
    ar = EmbeddingArena("t", K, dev, seed=seed)
    for i, v in enumerate(vocabs):
        ar.add_table(f"t{i}", v)
    ar.materialize()
    rb = torch.tensor([ar.tables[f"t{i}"][0] for i in range(len(vocabs))], dtype=torch.int64, device=dev)
    return ar, rb


def make_ids(gen, B, vocabs, oov=0.02):
    return torch.stack([zipf_ids(gen, B, v, oov) for v in vocabs], dim=1).contiguous()


@pytest.mark.parametrize("B,vocabs,K", [
    (1, [5], 16), (37, [11, 2, 301], 8), (512, [1000] * 26, 16), (4096, [20000, 106444, 2, 18789] + [997] * 22, 16),
])
def test_gather_fwd_bit_exact_and_bwd(dev, B, vocabs, K):
    gen = torch.Generator().manual_seed(B)
    ar, rb = make_arena(vocabs, K, dev)
    ids = make_ids(gen, B, vocabs)
    store = VariableStore(dev)
    out = ops.embedding_gather(store, ids.to(dev), ar, rb)
    w = ar.weight.cpu()
    ref = torch.cat([R.embedding_lookup_single(ids[:, f], w[rb[f].item():rb[f].item() + vocabs[f]])
                     for f in range(len(vocabs))], dim=1)
    assert_bit_exact(out, ref, "gather fwd")
    # backward: dense-equivalent scatter-add
    g = torch.randn(B, len(vocabs) * K, generator=gen)
    out.backward(g.to(dev))
    def table_grad(dt):                # autograd of the lookup = dense scatter-add of the gradient rows, in `dt`
        wt = w.to(dt).requires_grad_(True)
        torch.cat([R.embedding_lookup_single(ids[:, f], wt[rb[f].item():rb[f].item() + vocabs[f]])
                   for f in range(len(vocabs))], dim=1).backward(g.to(dt))
        return wt.grad
    assert_close(ar.grad, table_grad(torch.float64), what="gather bwd", reduced=True, ref32=table_grad(torch.float32))


def test_gather_empty_batch(dev):
    ar, rb = make_arena([7, 9], 16, dev)
    store = VariableStore(dev)
    out = ops.embedding_gather(store, torch.zeros(0, 2, dtype=torch.int64, device=dev), ar, rb)
    assert out.shape == (0, 32)


def _bags(gen, B, vocab, maxlen, oov=0.1):
    lens = torch.randint(0, maxlen + 1, (B,), generator=gen)
    lens[0] = 0
    offsets = torch.zeros(B + 1, dtype=torch.int64)
    offsets[1:] = lens.cumsum(0)
    values = zipf_ids(gen, int(offsets[-1]), vocab, oov)
    return values, offsets


@pytest.mark.parametrize("B,vocab,K,maxlen", [(5, 13, 4, 3), (300, 350, 16, 12)])
def test_bag_mean(dev, B, vocab, K, maxlen):
    gen = torch.Generator().manual_seed(7)
    ar, _ = make_arena([3, vocab], K, dev)
    values, offsets = _bags(gen, B, vocab, maxlen)
    store = VariableStore(dev)
    out = ops.embedding_bag_mean(store, values.to(dev), offsets.to(dev), ar, "t1")
    tab = ar.table_view("t1").cpu()
    ref = R.embedding_lookup_mean(values, offsets, tab)
    assert_close(out, R.embedding_lookup_mean(values, offsets, tab.double()), what="bag mean fwd", ref32=ref)
    # bags of exactly one valid id are exact row copies
    lens = offsets[1:] - offsets[:-1]
    for b in torch.nonzero(lens == 1).flatten().tolist():
        assert torch.equal(out[b].cpu(), ref[b])
    g = torch.randn(B, K, generator=gen)
    out.backward(g.to(dev))
    t64, t32 = tab.double().requires_grad_(True), tab.clone().requires_grad_(True)
    R.embedding_lookup_mean(values, offsets, t64).backward(g.double())
    R.embedding_lookup_mean(values, offsets, t32).backward(g)
    rb, v = ar.tables["t1"]
    assert_close(ar.grad[rb:rb + v], t64.grad, what="bag mean bwd", reduced=True, ref32=t32.grad)
    assert float(ar.grad[:rb].abs().sum()) == 0.0


@pytest.mark.parametrize("B,T", [(4, 3), (129, 50)])
def test_sequence_gather(dev, B, T):
    gen = torch.Generator().manual_seed(3)
    K, vocab = 16, 211
    ar, _ = make_arena([vocab], K, dev)
    values, offsets = _bags(gen, B, vocab, T)
    store = VariableStore(dev)
    out, sl = ops.sequence_gather(store, values.to(dev), offsets.to(dev), ar, "t0", T)
    tab = ar.table_view("t0").cpu()
    ref, lens = R.sequence_lookup(values, offsets, tab, T)
    assert_bit_exact(out, ref, "sequence gather")
    assert torch.equal(sl.cpu().long(), lens)
    g = torch.randn(B, T, K, generator=gen)
    out.backward(g.to(dev))
    t64, t32 = tab.double().requires_grad_(True), tab.clone().requires_grad_(True)
    R.sequence_lookup(values, offsets, t64, T)[0].backward(g.double())
    R.sequence_lookup(values, offsets, t32, T)[0].backward(g)
    assert_close(ar.grad, t64.grad, what="sequence gather bwd", reduced=True, ref32=t32.grad)


@pytest.mark.parametrize("B,F,K", [(3, 2, 4), (130, 6, 8), (1024, 26, 16), (4096, 26, 16)])
def test_deepfm_sparse(dev, B, F, K):
    gen = torch.Generator().manual_seed(B + F)
    vocabs = [max(2, 1000 // (f + 1)) for f in range(F)]
    ar, rb = make_arena(vocabs, K, dev)
    w1 = EmbeddingArena("w1", 1, dev, seed=9)
    for i, v in enumerate(vocabs):
        w1.add_table(f"t{i}", v)
    w1.materialize()
    ids = make_ids(gen, B, vocabs, oov=0.03)
    bias = Variable("b", torch.tensor([0.37], device=dev))
    store = VariableStore(dev)
    emb, fm1, fm2 = ops.deepfm_sparse(store, ids.to(dev), ar, w1, bias, rb)

    def oracle(dt):
        W = ar.weight.cpu().to(dt).requires_grad_(True)
        W1 = w1.weight.cpu().to(dt).reshape(-1).requires_grad_(True)
        bb = bias.data.cpu().to(dt).requires_grad_(True)
        fields = [R.embedding_lookup_single(ids[:, f], W[rb[f].item():rb[f].item() + vocabs[f]]) for f in range(F)]
        o1 = R.indicator_first_order([ids[:, f] for f in range(F)],
                                     [W1[rb[f].item():rb[f].item() + vocabs[f]] for f in range(F)], bb[0])
        o2 = R.fm_second_order(fields)
        return W, W1, bb, torch.cat(fields, 1), o1, o2

    Wf, W1f, bbf, e32, o1f, o2f = oracle(torch.float32)
    assert_bit_exact(emb, e32.detach(), "deep_input")
    W, W1, bb, e64, o1, o2 = oracle(torch.float64)
    assert_close(fm1, o1, what="fm first order", ref32=o1f)
    assert_close(fm2, o2, what="fm second order", ref32=o2f)
    ge = torch.randn(B, F * K, generator=gen)
    g1 = torch.randn(B, 1, generator=gen)
    g2 = torch.randn(B, 1, generator=gen)
    torch.autograd.backward([emb, fm1, fm2], [ge.to(dev), g1.to(dev), g2.to(dev)])
    torch.autograd.backward([e64, o1, o2], [ge.double(), g1.double(), g2.double()])
    torch.autograd.backward([e32, o1f, o2f], [ge, g1, g2])
    assert_close(ar.grad, W.grad, what="deepfm d(table)", reduced=True, ref32=Wf.grad)
    assert_close(w1.grad.reshape(-1), W1.grad, what="deepfm d(w1)", reduced=True, ref32=W1f.grad)
    assert_close(bias.grad, bb.grad, what="deepfm d(bias)", reduced=True, ref32=bbf.grad)


def test_fm_identity_bruteforce(dev):
    """0.5*sum_k[(sum e)^2 - sum e^2] == sum_{i<j} <e_i, e_j>  (SURVEY.md §8c (1))."""
    gen = torch.Generator().manual_seed(1)
    B, F, K = 64, 7, 16
    vocabs = [50] * F
    ar, rb = make_arena(vocabs, K, dev)
    w1 = EmbeddingArena("w1", 1, dev)
    for i, v in enumerate(vocabs):
        w1.add_table(f"t{i}", v)
    w1.materialize()
    ids = make_ids(gen, B, vocabs, oov=0.0)
    store = VariableStore(dev)
    emb, _, fm2 = ops.deepfm_sparse(store, ids.to(dev), ar, w1, Variable("b", torch.zeros(1, device=dev)), rb)
    E = emb.detach().cpu().double().view(B, F, K)
    brute = torch.zeros(B, dtype=torch.float64)
    for i in range(F):
        for j in range(i + 1, F):
            brute += (E[:, i] * E[:, j]).sum(-1)
    assert_close(fm2.view(-1), brute, what="FM identity")


@pytest.mark.parametrize("B,d,L", [(1, 4, 1), (5, 48, 2), (257, 416, 3), (4096, 416, 3), (64, 432, 6), (33, 1024, 4)])
def test_cross_stack(dev, B, d, L):
    gen = torch.Generator().manual_seed(d + L)
    x0 = torch.randn(B, d, generator=gen)
    w = torch.randn(L, d, generator=gen) / math.sqrt(d)
    b = torch.randn(L, d, generator=gen) * 0.1
    store = VariableStore(dev)
    wv, bv = Variable("w", w.to(dev)), Variable("b", b.to(dev))
    x0d = x0.to(dev).requires_grad_(True)
    out = ops.cross_stack(store, x0d, wv, bv)
    g = torch.randn(B, d, generator=gen)

    def oracle(dt):
        xx, ww, bb = (t.to(dt).requires_grad_(True) for t in (x0, w, b))
        r = R.cross_stack(xx, [ww[l].unsqueeze(1) for l in range(L)], [bb[l].unsqueeze(1) for l in range(L)])
        r.backward(g.to(dt))
        return r.detach(), xx.grad, ww.grad, bb.grad
    ref, gx, gw, gb = oracle(torch.float64)
    r32, gx32, gw32, gb32 = oracle(torch.float32)
    assert_close(out, ref, what="cross fwd", ref32=r32)
    out.backward(g.to(dev))
    ops.flush_dense_splits()          # dw / db: column sums of the partial rows, finished by the step's deferred-sum launch
    assert_close(x0d.grad, gx, what="cross dx0", ref32=gx32)
    assert_close(wv.grad, gw, what="cross dw", reduced=True, ref32=gw32)
    assert_close(bv.grad, gb, what="cross db", reduced=True, ref32=gb32)


@pytest.mark.parametrize("B,vocabs,K,L", [(1, [5], 16, 1), (37, [11, 2, 301], 8, 2), (4096, [20000, 106444, 2, 18789] + [997] * 22, 16, 3),
                                          (513, [1000] * 27, 16, 6), (64, [50] * 64, 16, 3), (4096, [31] * 5, 4, 3)])
def test_gather_left_to_the_cross_kernel(dev, B, vocabs, K, L):
    """recalgo_gather_cross_fwd (ops.gather_feeds_cross): x0 and the cross output are BIT-identical to gather-then-cross,
    and so is every gradient of the step (the backward is the same two kernels either way)."""
    gen = torch.Generator().manual_seed(B + L)
    d = len(vocabs) * K
    ids = make_ids(gen, B, vocabs).to(dev)
    w = (torch.randn(L, d, generator=gen) / math.sqrt(d)).to(dev)
    b = (torch.randn(L, d, generator=gen) * 0.1).to(dev)
    g = torch.randn(B, d, generator=gen).to(dev)

    def run(fused):
        ar, rb = make_arena(vocabs, K, dev)
        store = VariableStore(dev)
        wv, bv = Variable("w", w.clone()), Variable("b", b.clone())
        if fused:
            with ops.gather_feeds_cross() as lz:
                x0 = ops.embedding_gather(store, ids, ar, rb)
                lz.keep(x0)
            assert getattr(x0, "_recalgo_lazy_gather", None) is not None     # not launched yet
        else:
            x0 = ops.embedding_gather(store, ids, ar, rb)
        out = ops.cross_stack(store, x0, wv, bv)
        assert getattr(x0, "_recalgo_lazy_gather", None) is None and not ops._lazy_gathers
        (out * g).sum().backward()
        ops.flush_dense_splits()
        return x0.detach().clone(), out.detach().clone(), ar.grad.clone(), wv.grad.clone(), bv.grad.clone()
    for what, a, r in zip(("x0", "cross out", "arena grad", "dw", "db"), run(True), run(False)):
        assert_bit_exact(a, r, what)


def test_gather_left_pending_is_launched_when_nobody_takes_it(dev):
    """The safety net of ops.gather_feeds_cross: not kept (two gathers / a different tensor) or never consumed -> plain gather."""
    gen = torch.Generator().manual_seed(3)
    vocabs = [100] * 4
    ids = make_ids(gen, 33, vocabs).to(dev)
    ar, rb = make_arena(vocabs, 16, dev)
    store = VariableStore(dev)
    ref = ops.embedding_gather(store, ids, ar, rb).detach().clone()
    with ops.gather_feeds_cross() as lz:                       # not kept -> launched at exit
        a = ops.embedding_gather(store, ids, ar, rb)
    assert_bit_exact(a, ref, "not kept")
    with ops.gather_feeds_cross() as lz:                       # two gathers -> cannot both be x0
        a = ops.embedding_gather(store, ids, ar, rb)
        c = ops.embedding_gather(store, ids, ar, rb)
        lz.keep(a)
    assert_bit_exact(a, ref, "two gathers a"); assert_bit_exact(c, ref, "two gathers c")
    with ops.gather_feeds_cross() as lz:                       # kept, then the consumer never comes
        a = ops.embedding_gather(store, ids, ar, rb)
        lz.keep(a)
    ops.flush_lazy_gathers()
    assert_bit_exact(a, ref, "flushed")
    with torch.no_grad():                                      # inference lookups take the same route
        with ops.gather_feeds_cross() as lz:
            a = ops.embedding_gather(store, ids, ar, rb)
            lz.keep(a)
        wv, bv = Variable("w", torch.zeros(1, 64, device=dev)), Variable("b", torch.zeros(1, 64, device=dev))
        out = ops.cross_stack(store, a, wv, bv)
    assert_bit_exact(a, ref, "no_grad x0"); assert_bit_exact(out, ref, "no_grad out (w = b = 0 -> x1 = x0)")


def test_cross_layer_separate_xl_and_identities(dev):
    gen = torch.Generator().manual_seed(5)
    B, d = 77, 96
    x0, xl = torch.randn(B, d, generator=gen), torch.randn(B, d, generator=gen)
    w, b = torch.randn(d, 1, generator=gen) * 0.1, torch.randn(d, 1, generator=gen)
    store = VariableStore(dev)
    wv, bv = Variable("w", w.to(dev)), Variable("b", b.to(dev))
    x0d, xld = x0.to(dev).requires_grad_(True), xl.to(dev).requires_grad_(True)
    out = ops.cross_layer(store, x0d, xld, wv, bv)
    a64 = [t.double().requires_grad_(True) for t in (x0, xl, w, b)]
    a32 = [t.clone().requires_grad_(True) for t in (x0, xl, w, b)]
    ref, r32 = R.cross_layer(*a64), R.cross_layer(*a32)
    assert_close(out, ref, what="cross_layer fwd", ref32=r32)
    g = torch.randn(B, d, generator=gen)
    out.backward(g.to(dev))
    ref.backward(g.double())
    r32.backward(g)
    for got, k, nm in [(x0d.grad, 0, "dx0"), (xld.grad, 1, "dxl"), (wv.grad, 2, "dw"), (bv.grad, 3, "db")]:
        assert_close(got, a64[k].grad, what=f"cross_layer {nm}", reduced=True, ref32=a32[k].grad)
    # w = 0  =>  out = xl + b    — exact
    wz = Variable("wz", torch.zeros(d, 1, device=dev))
    outz = ops.cross_layer(store, x0.to(dev), xl.to(dev), wz, bv)
    assert torch.equal(outz.cpu(), (x0 * 0.0 + b.t()) + xl)


@pytest.mark.parametrize("B", [1, 100, 4096, 5000])
def test_sigmoid_ce(dev, B):
    gen = torch.Generator().manual_seed(B)
    x = torch.randn(B, 1, generator=gen) * 4
    x[0] = 30.0
    if B > 1:
        x[1] = -30.0
    z = (torch.rand(B, 1, generator=gen) < 0.3).float()
    # saturated logits with the label that does NOT cancel (for x=30, z=1 the fp64 autograd
    # reference itself loses 3 digits forming (1 - r) - 1; the kernel's ((1-z) - r) is exact)
    z[0] = 0.0
    if B > 1:
        z[1] = 1.0
    xd = x.to(dev).requires_grad_(True)
    loss, prob = ops.sigmoid_cross_entropy(xd, z.to(dev))
    x64, x32 = x.double().requires_grad_(True), x.clone().requires_grad_(True)
    ref, r32 = R.ce_loss(z.double(), x64), R.ce_loss(z, x32)
    assert_close(loss, ref, what="loss", ref32=r32)
    assert_close(prob, torch.sigmoid(x64), what="prob", ref32=torch.sigmoid(x32))
    (loss * 2.5).backward()
    (ref * 2.5).backward()
    (r32 * 2.5).backward()
    assert_close(xd.grad, x64.grad, what="dlogit", ref32=x32.grad)


@pytest.mark.parametrize("n", [1, 7, 4096, 1_000_003])
def test_adam_tf1(dev, n):
    gen = torch.Generator().manual_seed(n)
    p, g = torch.randn(n, generator=gen), torch.randn(n, generator=gen)
    g[::3] = 0.0
    m, v = torch.zeros(n), torch.zeros(n)
    pd, gd, md, vd = (t.clone().to(dev) for t in (p, g, m, v))
    p64, m64, v64 = p.double(), m.double(), v.double()
    step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
    lr_dev = torch.zeros(1, device=dev)
    for step in (1, 2, 3):
        gs = g * step
        gd.copy_(gs)
        if step < 3:
            ops.adam_tf1_(pd, gd, md, vd, step, 0.005)
            ops.adam_tf1_advance_(step_dev, lr_dev, 0.005)
        else:  # device-side step counter path (hipGraph replayable)
            ops.adam_tf1_advance_(step_dev, lr_dev, 0.005)
            ops.adam_tf1_(pd, gd, md, vd, -1, 0.005, lr_t_dev=lr_dev)
        R.adam_tf1_step(p64, gs.double(), m64, v64, step, 0.005)
        assert float(gd.abs().sum()) == 0.0, "zero_grad"
    assert int(step_dev) == 3
    assert_close(pd, p64, what="adam p")
    assert_close(md, m64, what="adam m")
    assert_close(vd, v64, what="adam v")


@pytest.mark.parametrize("rows,C", [(300, 200), (4096, 128), (65, 68), (130, 203), (1, 4)])
@pytest.mark.parametrize("kind", ["prelu", "dice"])
def test_activation(dev, kind, rows, C):
    gen = torch.Generator().manual_seed(11 + rows + C)
    x = torch.randn(rows, C, generator=gen) * 2
    a = torch.rand(C, generator=gen) + 0.5
    store = VariableStore(dev)
    av = Variable("alpha", a.to(dev))
    xd = x.to(dev).requires_grad_(True)
    y = ops.activation(store, xd, av, kind)
    x64, a64 = x.double().requires_grad_(True), a.double().requires_grad_(True)
    x32, a32 = x.clone().requires_grad_(True), a.clone().requires_grad_(True)
    fn = R.prelu if kind == "prelu" else R.dice
    ref, r32 = fn(x64, a64), fn(x32, a32)
    assert_close(y, ref, what=kind, ref32=r32)
    g = torch.randn(rows, C, generator=gen)
    y.backward(g.to(dev))
    ref.backward(g.double())
    r32.backward(g)
    assert_close(xd.grad, x64.grad, what=f"{kind} dx", ref32=x32.grad)
    assert_close(av.grad, a64.grad, what=f"{kind} dalpha", reduced=True, ref32=a32.grad)


def test_cpu_tensor_is_rejected():
    """The product path has no CPU fallback."""
    from recalgorithm_amd._lib import RecalgoError
    store = VariableStore("cpu")
    ar = EmbeddingArena("t", 16, "cpu")
    ar.add_table("t0", 4)
    ar.materialize()
    with pytest.raises(RecalgoError):
        ops.embedding_gather(store, torch.zeros(2, 1, dtype=torch.int64), ar, torch.zeros(1, dtype=torch.int64))


@pytest.mark.parametrize("rows,C", [(4096, 512), (300, 128), (33, 16), (1000, 1024), (7, 4), (65, 200), (129, 2052), (64, 68)])
def test_mlp_glue_relu_bwd_bias_and_batchnorm(dev, rows, C):
    """csrc/mlp.hip vs a float64 torch restatement of tf.layers.dense's ReLU/bias backward and of
    tf.layers.batch_normalization(training=True) [TF-ext A-8]."""
    from recalgorithm_amd import ops
    assert ops.mlp_width_supported(C) and not ops.mlp_width_supported(82) and not ops.mlp_width_supported(0)
    gen = torch.Generator().manual_seed(rows + C)
    g = torch.randn(rows, C, generator=gen)
    y = torch.relu(torch.randn(rows, C, generator=gen))
    dbias = torch.empty(C, device=dev)
    g2 = ops.relu_bwd_bias_(g.to(dev), y.to(dev), dbias)
    ref = torch.where(y > 0, g, torch.zeros_like(g)).double()
    assert_bit_exact(g2, ref.float(), "relu mask")
    assert_close(dbias, ref.sum(0), what="dbias", reduced=True)
    g3 = ops.relu_bwd_bias_(g.to(dev), None, dbias)
    assert_close(dbias, g.double().sum(0), what="dbias (no relu)", reduced=True)
    assert g3.data_ptr() != 0
    # batch norm, with a column offset much larger than the spread (stresses the variance formula)
    x = torch.randn(rows, C, generator=gen) * 0.5 + torch.linspace(-20, 20, C)
    gamma, beta = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen)
    mm, mv = torch.randn(C, generator=gen), torch.rand(C, generator=gen) + 0.5
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    mean = xd.mean(0)
    var = ((xd - mean) ** 2).mean(0)
    yref = (xd - mean) * torch.rsqrt(var + 1e-3) * gd + bd
    yref.backward(g.double())
    mmg, mvg = mm.to(dev), mv.to(dev)
    yh, smean, srstd = ops.batchnorm_train_fwd(x.to(dev), gamma.to(dev), beta.to(dev), mmg, mvg, 0.99, 1e-3)
    assert_close(yh, yref, what="bn y")
    assert_close(smean, mean, what="bn mean")
    assert_close(srstd, torch.rsqrt(var + 1e-3), what="bn rstd")
    assert_close(mmg, mm.double() * 0.99 + mean.detach() * 0.01, what="moving_mean")
    assert_close(mvg, mv.double() * 0.99 + var.detach() * 0.01, what="moving_variance")
    dgamma, dbeta = torch.empty(C, device=dev), torch.empty(C, device=dev)
    dx = ops.batchnorm_train_bwd(x.to(dev), gamma.to(dev), smean, srstd, g.to(dev), dgamma, dbeta)
    assert_close(dbeta, bd.grad, what="bn dbeta", reduced=True)
    assert_close(dgamma, gd.grad, what="bn dgamma", reduced=True, floor=1e-6 * float(g.abs().sum(0).max()))
    assert_close(dx, xd.grad, what="bn dx", reduced=True, floor=2e-6 * float(xd.grad.abs().max()))


@pytest.mark.parametrize("rows,K", [(5000, 16), (777, 8), (300, 64), (129, 4)])
def test_adam_rows_is_bit_identical_to_dense(dev, rows, K):
    """recalgo_adam_tf1_rows (skips rows no gradient has touched yet) == recalgo_adam_tf1_dense, bit
    for bit, over several steps with sparse row gradients; the liveness bytes track the touched rows."""
    gen = torch.Generator().manual_seed(rows)
    w0 = torch.randn(rows, K, generator=gen)
    st_a = [w0.clone().to(dev), torch.zeros(rows, K, device=dev), torch.zeros(rows, K, device=dev), torch.zeros(rows, K, device=dev)]
    st_b = [t.clone() for t in st_a]
    live = torch.zeros(rows, dtype=torch.uint8, device=dev)
    step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
    lr_t = torch.zeros(1, device=dev)
    touched = torch.zeros(rows, dtype=torch.bool)
    for step in range(1, 6):
        idx = torch.randint(0, rows, (rows // 10,), generator=gen)
        g = torch.zeros(rows, K)
        g[idx] = torch.randn(idx.numel(), K, generator=gen)
        touched[idx] = True
        ops.adam_tf1_advance_(step_dev, lr_t, 0.005)
        st_a[1].copy_(g.to(dev)); st_b[1].copy_(g.to(dev))
        ops.adam_tf1_(st_a[0].view(-1), st_a[1].view(-1), st_a[2].view(-1), st_a[3].view(-1), step=-1, lr=0.005, lr_t_dev=lr_t)
        ops.adam_tf1_rows_(st_b[0], st_b[1], st_b[2], st_b[3], live, lr_t)
        for a, b, nm in zip(st_a, st_b, ("p", "g", "m", "v")):
            assert_bit_exact(b, a, f"adam rows vs dense: {nm} at step {step}")
        assert torch.equal(live.cpu().bool(), touched)
    assert float(st_b[1].abs().sum()) == 0.0       # gradients consumed


@pytest.mark.parametrize("rows,K,F", [(5003, 16, 3), (700, 8, 1), (260, 64, 2), (1500, 1, 2), (900, 6, 1)])
def test_live_row_list_adam_is_bit_identical_to_dense(dev, rows, K, F):
    """recalgo_mark_live_rows + recalgo_adam_tf1_list == recalgo_adam_tf1_dense, bit for bit, over
    several steps of sparse row gradients; every touched row enters the list exactly once."""
    import ctypes
    from recalgorithm_amd import _lib
    from recalgorithm_amd.variables import EmbeddingArena
    lib = _lib.load()
    gen = torch.Generator().manual_seed(rows + K)
    per = rows // F
    ar = EmbeddingArena("t", K, dev, seed=1)
    for f in range(F):
        ar.add_table(f"t{f}", per if f < F - 1 else rows - per * (F - 1))
    ar.materialize()
    assert ar.tracks_live_rows
    rb = torch.tensor([ar.tables[f"t{f}"][0] for f in range(F)], dtype=torch.int64, device=dev)
    dense = [ar.weight.clone(), torch.zeros_like(ar.weight), torch.zeros_like(ar.weight), torch.zeros_like(ar.weight)]
    step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
    lr_t = torch.zeros(1, device=dev)
    touched = torch.zeros(rows, dtype=torch.bool)
    for step in range(1, 6):
        B = 200
        ids = torch.stack([torch.randint(0, ar.tables[f"t{f}"][1], (B,), generator=gen) for f in range(F)], 1)
        ids[torch.rand(B, F, generator=gen) < 0.1] = -1                       # OOV
        ids[:, 0] = torch.where(torch.rand(B, generator=gen) < 0.3, torch.zeros(B, dtype=torch.int64), ids[:, 0])  # hot row
        grows = ids + rb.cpu()
        g = torch.zeros(rows, K)
        ok = ids >= 0
        g.index_add_(0, grows[ok], torch.randn(int(ok.sum()), K, generator=gen))
        touched[grows[ok]] = True
        ar.grad.copy_(g.to(dev)); dense[1].copy_(g.to(dev))
        ops.mark_live_rows(ar, ids.to(dev).contiguous(), rb, F)
        ops.adam_tf1_advance_(step_dev, lr_t, 0.005)
        ops.adam_tf1_(dense[0].view(-1), dense[1].view(-1), dense[2].view(-1), dense[3].view(-1), step=-1, lr=0.005, lr_t_dev=lr_t)
        ops.adam_tf1_list_(ar, lr_t)
        for a, b, nm in zip(dense, (ar.weight, ar.grad, ar.m, ar.v), ("p", "g", "m", "v")):
            assert_bit_exact(b, a, f"adam list vs dense: {nm} at step {step}")
        live, lst, cnt = ar.live_state()
        n = int(cnt.item())
        assert n == int(touched.sum())
        assert sorted(lst[:n].tolist()) == torch.nonzero(touched).squeeze(1).tolist()     # each row exactly once
        assert torch.equal(live[:rows].cpu().bool(), touched)
        # housekeeping between steps: the list is rebuilt in address order from the liveness bytes
        if step % 2 == 0:
            ar.order_live_list()
            assert int(cnt.item()) == n
            assert lst[:n].tolist() == torch.nonzero(touched).squeeze(1).tolist()
    # rebuilt from the moments (restore / re-shard path) it is the same set
    ar.live = None
    _, lst2, cnt2 = ar.live_state()
    assert sorted(lst2[:int(cnt2.item())].tolist()) == torch.nonzero(touched).squeeze(1).tolist()


@pytest.mark.parametrize("rows,density", [(1, 1.0), (4095, 0.5), (4097, 0.01), (300001, 0.12), (5_000_003, 0.1), (5_000_003, 0.0)])
def test_order_live_list_matches_nonzero(dev, rows, density):
    """recalgo_order_live_list == torch.nonzero(row_live) (ascending), count included; covers partial
    words, chunk boundaries and more than one round of the chunk scan (> 1024 chunks of 4096 rows)."""
    import ctypes
    from recalgorithm_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(rows)
    live = torch.zeros((rows + 3) // 4 * 4, dtype=torch.uint8)
    live[:rows] = (torch.rand(rows, generator=g) < density).to(torch.uint8) * torch.randint(1, 255, (rows,), generator=g).to(torch.uint8)
    live = live.to(dev)
    lst = torch.full((rows,), -7, dtype=torch.int32, device=dev)
    cnt = torch.full((1,), -1, dtype=torch.int32, device=dev)
    ws = torch.empty(max(int(lib.recalgo_order_live_list_workspace_bytes(rows)), 4), dtype=torch.uint8, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.recalgo_order_live_list(p(live), rows, p(lst), p(cnt), p(ws), st), "recalgo_order_live_list")
    want = torch.nonzero(live[:rows]).squeeze(1).to(torch.int32)
    n = int(cnt.item())
    assert n == want.numel()
    assert torch.equal(lst[:n], want)
    assert bool((lst[n:] == -7).all())


@pytest.mark.parametrize("B,widths,bias,skip_dx", [(4096, (416, 128), True, None), (1000, (82,), True, None), (15, (1,), False, None),
                                                   (257, (7, 130, 3, 64), True, 1), (1, (300, 20), False, 0), (0, (16, 8), True, None),
                                                   (513, (2048,), True, None), (100_003, (128,), True, None),
                                                   (70_000, (33, 4), False, 1)])
def test_dense1_head_against_torch(dev, B, widths, bias, skip_dx):
    """recalgo_dense1_{fwd,bwd} == concat + matmul (fp64 reference), incl. unaligned widths, missing
    bias, a part that needs no input gradient, an empty batch, and batches beyond 32 K rows (AFM runs this head over
    B x pairs rows: a workgroup then walks several 32-row tiles so that at most 1024 partial rows are left)."""
    g = torch.Generator().manual_seed(B + sum(widths))
    parts = [torch.randn(B, w, generator=g).to(dev) for w in widths]
    C = sum(widths)
    w = torch.randn(C, 1, generator=g).to(dev)
    b = torch.randn(1, generator=g).to(dev) if bias else None
    out = ops.dense1_fwd(parts, w, b)
    x64 = torch.cat([p.double() for p in parts], 1) if B else torch.zeros(0, C, dtype=torch.float64, device=dev)
    ref = x64 @ w.double() + (b.double() if bias else 0.0)
    assert out.shape == (B, 1)
    assert_close(out, ref, what="dense1 fwd")
    gl = torch.randn(B, 1, generator=g).to(dev)
    dxs = [None if i == skip_dx else torch.full_like(p, float("nan")) for i, p in enumerate(parts)]
    dw = torch.full((C, 1), float("nan"), device=dev)
    db = torch.full((1,), float("nan"), device=dev) if bias else None
    ops.dense1_bwd(parts, w, gl, dxs, dw, db)
    assert_close(dw, x64.t() @ gl.double(), what="dense1 dw", reduced=True)
    if bias:
        assert_close(db, gl.double().sum().reshape(1), what="dense1 dbias", reduced=True)
    off = 0
    for i, (p, dx) in enumerate(zip(parts, dxs)):
        if dx is not None:
            assert_close(dx, gl.double() * w.double()[off:off + p.shape[1]].t(), what=f"dense1 dx[{i}]")
        off += p.shape[1]


@pytest.mark.gpu
@pytest.mark.parametrize("nbytes,phase", [(0, 0), (1, 0), (15, 3), (16, 0), (4096 * 26 * 8 + 4096 * 4, 0), (1_000_003, 5), (64 << 20, 0)])
def test_copy_bytes_is_exact(dev, nbytes, phase):
    """recalgo_copy_bytes (the batch -> static-input copy of a captured step): every byte, any length, any common
    alignment phase; bytes outside [0, nbytes) stay untouched."""
    from recalgorithm_amd import ops
    g = torch.Generator().manual_seed(nbytes + phase)
    src_all = torch.randint(0, 256, (nbytes + 64,), dtype=torch.uint8, generator=g).to(dev)
    dst_all = torch.full((nbytes + 64,), 7, dtype=torch.uint8, device=dev)
    src, dst = src_all[phase:phase + nbytes], dst_all[phase:phase + nbytes]
    ops.copy_bytes(dst, src)
    torch.cuda.synchronize()
    assert torch.equal(dst, src)
    assert bool((dst_all[:phase] == 7).all()) and bool((dst_all[phase + nbytes:] == 7).all())
    # different phases on the two sides: the runtime-copy fallback
    if nbytes > 32:
        dst2 = dst_all[phase + 1:phase + 1 + nbytes - 1]
        ops.copy_bytes(dst2, src[:nbytes - 1])
        torch.cuda.synchronize()
        assert torch.equal(dst2, src[:nbytes - 1])
