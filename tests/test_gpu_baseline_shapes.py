"""-m gpu parity AT THE BASELINE.json SHAPES THEMSELVES (configs[1..3]: DCN / xDeepFM CIN [128,128] / DIN T=50,
all at batch 4096, 26 fields x emb 16) — kernel level and model level (forward, loss, every gradient, one
TF1-Adam step).  The split-K partial counts, workspace sizes and grid shapes of `cin_filter_grad`,
`din_attention_bwd`, the scatter kernels and the MLP glue at B = 4096 differ from the small shapes of the other
test files; this file runs exactly what bench.py runs (the estimators come from bench.build_estimator).

Every comparison also records how many elements lie outside SURVEY.md §8c's strict per-element bound
|a - b| <= 1e-5 * max(|a|, |b|, eps) — for the HIP result AND for the oracle itself evaluated in float32 (the
reference arithmetic's own rounding against the same fp64 anchor); tests/conftest.py prints the table."""
import argparse

import pytest
import torch

import bench
from oracle import ref_models as M
from oracle import ref_ops as R
from recalgorithm_amd import ops
from recalgorithm_amd.estimator import ModeKeys
from recalgorithm_amd.variables import Variable, VariableStore, named_grads
from tests.util import assert_adam_update, assert_bit_exact, assert_close

pytestmark = pytest.mark.gpu

B, F, K, T = 4096, 26, 16, 50


# ------------------------------------------------------------------------------------------------------
# kernel level
# ------------------------------------------------------------------------------------------------------
def _cin_oracle(x0, xk, w, go, gp, dtype, chunk=256):
    """R.cin_layer over batch chunks (examples are independent; the unchunked einsum of cin_layer.py:21
    materialises (B, D, Hk, m) = 1.7 GB in fp64 at B = 4096, Hk = 128), backward per chunk."""
    a = [t.to(dtype).requires_grad_(True) for t in (x0, xk, w)]
    outs = []
    for s in range(0, x0.shape[0], chunk):
        sl = slice(s, s + chunk)
        o = R.cin_layer(a[0][sl], a[1][sl], a[2])
        torch.autograd.backward([o, o.sum(-1)], [go[sl].to(dtype), gp[sl].to(dtype)])
        outs.append(o.detach())
    return torch.cat(outs), a[0].grad, a[1].grad, a[2].grad


@pytest.mark.parametrize("Hk", [26, 128])
def test_cin_layer_at_config3_shape(dev, Hk):
    """xDeepFM CIN layer 1 (Hk = m = 26) and layer 2 (Hk = 128), N = 128, B = 4096 (configs[2])."""
    m, N, D = F, 128, K
    gen = torch.Generator().manual_seed(100 + Hk)
    x0 = torch.randn(B, m, D, generator=gen)
    xk = torch.randn(B, Hk, D, generator=gen)
    w = torch.randn(1, Hk * m, N, generator=gen) / (Hk * m) ** 0.5
    go = torch.randn(B, N, D, generator=gen)
    gp = torch.randn(B, N, generator=gen)
    store = VariableStore(dev)
    wv = Variable("f", w.to(dev))
    x0d, xkd = x0.to(dev).requires_grad_(True), xk.to(dev).requires_grad_(True)
    out, pooled = ops.cin_layer(store, x0d, xkd, wv)
    torch.autograd.backward([out, pooled], [go.to(dev), gp.to(dev)])
    ref = _cin_oracle(x0, xk, w, go, gp, torch.float64)
    r32 = _cin_oracle(x0, xk, w, go, gp, torch.float32)
    assert_close(out, ref[0], what=f"cin(Hk={Hk}) fwd", reduced=True, ref32=r32[0])
    assert_close(pooled, ref[0].sum(-1), what=f"cin(Hk={Hk}) pooled", reduced=True, ref32=r32[0].sum(-1))
    assert_close(x0d.grad, ref[1], what=f"cin(Hk={Hk}) dx0", reduced=True, ref32=r32[1])
    assert_close(xkd.grad, ref[2], what=f"cin(Hk={Hk}) dxk", reduced=True, ref32=r32[2])
    # dW sums B*D = 65 536 fp32 products per element in split-K partials: the fp32 oracle's own deviation is
    # the measured floor (x4)
    noise = float((r32[3].double() - ref[3]).abs().max())
    assert_close(wv.grad, ref[3], what=f"cin(Hk={Hk}) dW", reduced=True, floor=4 * noise, ref32=r32[3])


@pytest.mark.parametrize("is_softmax", [False, True])
def test_din_attention_at_config4_shape(dev, is_softmax):
    """DIN attention over a 50-long history, H = 16, B = 4096 (configs[3]), ragged lengths 0..50."""
    from tests.test_gpu_din import make
    H = K
    gen = torch.Generator().manual_seed(4096 + int(is_softmax))
    q, keys, lens, ws, vs = make(B, T, H, gen, dev)
    store = VariableStore(dev)
    qd, kd = q.to(dev).requires_grad_(True), keys.to(dev).requires_grad_(True)
    out = ops.din_attention(store, qd, kd, lens.to(dev), vs, is_softmax)
    g = torch.randn(B, H, generator=gen)
    out.backward(g.to(dev))

    def oracle(dtype):
        a = [t.to(dtype).requires_grad_(True) for t in [q, keys] + ws]
        r = R.din_attention(a[0], a[1], lens, *a[2:], is_softmax=is_softmax)
        r.backward(g.to(dtype))
        return r.detach(), [t.grad for t in a]
    ref, gr = oracle(torch.float64)
    r32, g32 = oracle(torch.float32)
    assert_close(out, ref, what="din fwd", ref32=r32)
    assert_close(qd.grad, gr[0], what="din dq", ref32=g32[0])
    assert_close(kd.grad, gr[1], what="din dkeys", ref32=g32[1])
    for i, nm in enumerate(["f1_w", "f1_b", "f2_w", "f2_b", "f3_w", "f3_b"]):
        noise = float((g32[2 + i].double() - gr[2 + i]).abs().max())
        if nm == "f3_b" and is_softmax:       # shift invariance of softmax: an analytic zero (tests/test_gpu_din.py)
            scale = float(gr[6].abs().max())
            assert float((vs[i].grad.cpu().double() - gr[7]).abs().max()) <= 1e-5 * max(scale, 1e-30) + 4 * noise
            continue
        assert_close(vs[i].grad, gr[2 + i], what=f"din d{nm}", reduced=True, floor=4 * noise, ref32=g32[2 + i])


def test_gather_and_sequence_gather_bit_exact_at_config_shapes(dev):
    """Index work is bit-exact at the full shapes: [4096, 26] ids -> [4096, 416]; [4096, <=50] histories."""
    from tests.util import zipf_ids
    gen = torch.Generator().manual_seed(7)
    vocab = 50_000
    table = torch.randn(vocab, K, generator=gen)
    ids = torch.stack([zipf_ids(gen, B, vocab) for _ in range(F)], 1).contiguous()
    from recalgorithm_amd.variables import EmbeddingArena
    ar = EmbeddingArena("t", K, dev)
    ar.add_table("tab", vocab, table)
    ar.materialize()
    store = VariableStore(dev)
    rb = torch.zeros(F, dtype=torch.int64, device=dev)
    out = ops.embedding_gather(store, ids.to(dev), ar, rb)
    want = torch.cat([R.embedding_lookup_single(ids[:, f], table) for f in range(F)], 1)
    assert_bit_exact(out, want, "gather [4096, 26] x 16")
    lens = torch.randint(0, T + 1, (B,), generator=gen)
    offs = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)])
    vals = zipf_ids(gen, int(offs[-1]), vocab)
    seq, sl = ops.sequence_gather(store, vals.to(dev), offs.to(dev), ar, "tab", T)
    want_seq, want_len = R.sequence_lookup(vals, offs, table, T)
    assert_bit_exact(seq, want_seq, "sequence gather [4096, 50, 16]")
    assert torch.equal(sl.cpu().long(), want_len)


# ------------------------------------------------------------------------------------------------------
# model level: exactly bench.py's estimators (configs[1], [2], [3]; DeepFM = configs[4]'s model with the companion
# first-order arena at full size; FiBiNET and PNN = the two remaining north_star models at 26 x 16 x 4096)
# ------------------------------------------------------------------------------------------------------
def _bench_estimator(model, dev):
    args = bench.parse_args(["--model", model, "--batch", str(B), "--fields", str(F), "--emb", str(K),
                             "--max-vocab", "100000"])
    est, spec, feats, labels, workload = bench.build_estimator(args, dev)
    return est, feats, labels, workload


def _oracle_inputs(est, feats, labels, dtype):
    P = {k: v.detach().cpu().to(dtype).requires_grad_(True) for k, v in est.store.named_arrays().items()}
    cf = {k: (v.cpu() if isinstance(v, torch.Tensor) else (v.values.cpu(), v.offsets.cpu())) for k, v in feats.items()}
    cf = {k: (v.to(dtype) if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in cf.items()}
    cl = {k: v.cpu().to(dtype) for k, v in labels.items()}
    return P, cf, cl


class _ReluPattern:
    """A ReLU whose pre-activation lies within fp32 rounding of 0 is a kink ANY fp32 evaluation can land on either side of:
    one flipped mask bit changes the gradients upstream of it by O(|g|) — 1e5 x the fp32 rounding noise the strict guard
    compares against, whoever's arithmetic (the fp32 oracle's as much as the kernels').  At B = 4096 with 512 + 256 + 128
    (+ 1024 / 9600-wide) units per example a batch holds dozens of pre-activations within 1e-5 * rms of the kink, whatever
    the batch.  For the two models with the wide first layers the gradients are therefore compared CONDITIONAL ON THE
    ACTIVATION PATTERN of the run under test: the masks [y > 0] of the HIP forward's ReLU layers (tf.layers.dense(...,
    relu) and PNN's product layer) are recorded, and both oracles (fp64 and fp32) evaluate those ReLUs as x * mask.  The
    pattern itself is checked separately: wherever it differs from the free-running fp64 oracle's, the fp64
    pre-activation must be within 1e-4 * rms of zero (i.e. the HIP forward put no unit on the wrong side of the kink by
    more than rounding)."""

    def __init__(self):
        self.masks, self.flips = [], []
        self.kept = {}           # id(mask) -> keep mask of the dropout FUSED into that layer's epilogue: y > 0 <=> relu > 0 AND kept, so
                                 # the recorded pattern says nothing about the pre-activation of a dropped unit (and need not: it is dropped)

    def record_hip(self, call):
        from recalgorithm_amd import nn, ops
        real_dense, real_pnn = nn.dense, ops.pnn_product_layer

        def dense(x, units, activation=None, *a, **k):
            y = real_dense(x, units, activation, *a, **k)
            if activation == "relu" and isinstance(y, torch.Tensor):
                self.masks.append((y.detach() > 0).cpu())
            return y

        def pnn(*a, **k):
            y = real_pnn(*a, **k)
            self.masks.append((y.detach() > 0).cpu())
            return y
        nn.dense, ops.pnn_product_layer = dense, pnn
        try:
            return call()
        finally:
            nn.dense, ops.pnn_product_layer = real_dense, real_pnn

    def oracle(self, call, check=False):
        """Run `call` with torch.relu replaced by the recorded pattern (matched by shape, in order)."""
        real, queue = torch.relu, list(self.masks)

        def relu(x):
            for i, m in enumerate(queue):
                if m.shape == x.shape:
                    queue.pop(i)
                    if check:
                        flip = (x.detach() > 0) != m
                        if id(m) in self.kept:
                            flip = flip & (self.kept[id(m)] > 0)
                        if bool(flip.any()):
                            rms = float(x.detach().pow(2).mean().sqrt())
                            self.flips.append((tuple(x.shape), int(flip.sum()), float(x.detach().abs()[flip].max()) / rms))
                    return x * m.to(x.dtype)
            return real(x)
        torch.relu = relu
        try:
            out = call()
        finally:
            torch.relu = real
        assert not queue or not self.masks, f"{len(queue)} recorded ReLU layers were not met by the oracle"
        return out


@pytest.mark.parametrize("model", ["dcn", "xdeepfm", "din", "deepfm", "fibinet", "pnn",
                                   # the reference's DEFAULT dropout_rate 0.1 (deepfm.py:39, din.py:41, fibinet.py:42, pnn.py:39)
                                   "din+dropout", "deepfm+dropout", "fibinet+dropout", "pnn+dropout"])
def test_model_step_at_baseline_config(dev, model):
    model, _, drop = model.partition("+")
    est, feats, labels, workload = _bench_estimator(model, dev)
    params = est.params
    if drop:
        params["dropout_rate"] = 0.1
    if model == "din":       # alpha = 1 makes Dice the identity (activations.py:31): move it so the kernel is exercised
        g = torch.Generator().manual_seed(99)
        for name, v in est.store.vars.items():
            if "alpha" in name:
                v.data.copy_((0.25 + 0.5 * torch.rand(v.data.shape, generator=g)).to(dev))
    fn = {"dcn": M.dcn, "xdeepfm": M.xdeepfm, "din": M.din, "deepfm": M.deepfm, "fibinet": M.fibinet, "pnn": M.pnn}[model]
    P, cf, cl = _oracle_inputs(est, feats, labels, torch.float64)
    P32, cf32, cl32 = _oracle_inputs(est, feats, labels, torch.float32)
    before = {k: v.detach().cpu().double().clone() for k, v in est.store.named_arrays().items()}
    pattern = _ReluPattern()
    # every model: the gradients are compared conditional on the HIP forward's ReLU pattern (see _ReluPattern; rounds 3-4 did
    # this for the two models with the 9600 / 1024-wide first layers only — the other four passed by the luck of the batch:
    # round 5's main loop contracts the reduction in another order and put ONE unit of DeepFM's first layer, 3e-7 * rms from
    # the kink, on the other side: one column of d(dense/kernel) off by 1e-2, everything behind xDeepFM's BatchNorm by
    # 200 x fp32 noise).  The flips themselves are asserted to be within rounding of the kink below.
    from recalgorithm_amd import nn
    nn.DROPOUT_SPECS[:] = []
    spec = pattern.record_hip(lambda: est._call_model_fn(feats, labels, ModeKeys.TRAIN))
    extra = {}
    if drop:
        # the keep masks the library's hash stream stood for in this step (the step counter has not moved), for both oracles
        dspecs = list(nn.DROPOUT_SPECS)
        assert len(dspecs) == 3 and all(d.mask is None for d in dspecs)
        masks = [ops.dropout_keep_mask((B, w), d, dev).cpu() for d, w in zip(dspecs, (512, 256, 128))]
        assert all(0.88 < float(m.mean()) < 0.92 for m in masks)
        if model != "din":       # dense(relu) -> dropout fused into the dense epilogue: the recorded outputs are the dropped tensors
            for pm in pattern.masks:
                for km in masks:
                    if pm.shape == km.shape:
                        pattern.kept[id(pm)] = km
        extra = {"dropout_masks": masks}
    fn_ = fn
    fn = (lambda *a, **k: fn_(*a, **k, dropout_masks=[m.clone() for m in extra["dropout_masks"]])) if drop else fn_
    if model in ("fibinet", "pnn"):
        assert len(pattern.masks) == (4 if model == "pnn" else 3)
    ref = pattern.oracle(lambda: fn(P, cf, cl, params, training=True), check=True)
    ref["loss"].backward()
    r32 = pattern.oracle(lambda: fn(P32, cf32, cl32, params, training=True))
    r32["loss"].backward()
    for shape, n_flip, dist in pattern.flips:
        print(f"[{model}] ReLU layer {shape}: {n_flip} units on the other side of the kink than the fp64 oracle's, the farthest "
              f"{dist:.2e} * rms from it")
        assert dist < 1e-4 and n_flip < 64, f"{model}: activation pattern differs beyond rounding at layer {shape}"
    assert_close(spec.loss, ref["loss"], what=f"{model} loss", ref32=r32["loss"])
    assert_close(spec.predictions["probabilities"], ref["prob"], what=f"{model} prob", ref32=r32["prob"])
    spec.loss.backward()
    grads = named_grads(est.store)
    tol_gs = {}
    # strict guard: 1.5 x the fp32 oracle's count per tensor.  Two compositions get more room, measured and explained
    # (profiles/r04*_strict*.md): FiBiNET's forward goes through the 9600-wide first layer on hipBLASLt, whose one long
    # accumulation chain leaves ~3 x the rounding error of the oracle's blocked GEMM in every activation downstream (all
    # gradients 3.0-3.5 x, uniformly); PNN's embedding gradients sum two K = 1024 dgrads (1.6 x)
    factor = {"fibinet": 4.0, "pnn": 2.0}.get(model, 1.5)
    for name, p in P.items():
        if p.grad is None:
            continue
        g32 = P32[name].grad
        noise = float((g32.double() - p.grad).abs().max())
        gref = p.grad.abs()
        tol_gs[name] = 1e-5 * (gref + gref.pow(2).mean().sqrt()) + 1e-6 * gref.max() + 4 * noise
        sib = name.replace("bias", "kernel")
        if name.endswith("/bias") and sib in P and P[sib].grad is not None:
            # a bias whose batch-summed gradient cancels analytically (in front of a training-mode BatchNorm;
            # f3_att/bias): judged at the scale of its sibling kernel's gradient (same upstream terms)
            scale = float(P[sib].grad.abs().max())
            err = float((grads[name].cpu().double() - p.grad).abs().max())
            assert err <= 1e-5 * scale + 4 * noise, f"{model} d({name}): err {err} scale {scale} noise {noise}"
            tol_gs[name] = tol_gs[name] + 1e-5 * scale
            continue
        # floor: 4x the deviation of the reference arithmetic itself in fp32 (batch sums of 4096 x up to 26 terms)
        assert_close(grads[name], p.grad, what=f"{model} d({name})", reduced=True, floor=4 * noise, ref32=g32, strict_factor=factor)

    spec.train_op.optimizer.apply_gradients(est.store)
    after = est.store.named_arrays()
    lr = params["learning_rate"]
    for name, p in P.items():
        if p.grad is None:
            continue
        pp, m_, v_ = before[name].clone(), torch.zeros_like(before[name]), torch.zeros_like(before[name])
        R.adam_tf1_step(pp, p.grad, m_, v_, 1, lr)
        upd = after[name].detach().cpu().double() - before[name]
        assert_adam_update(upd, pp - before[name], before[name], p.grad, tol_gs[name], lr,
                           what=f"{model} adam update {name}")
    assert float(est.store.flat_grad.abs().sum()) == 0.0
    for ar in est.store.arenas.values():
        assert float(ar.grad.abs().sum()) == 0.0
