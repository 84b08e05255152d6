"""-m gpu: tf.layers.dropout in training mode (the reference's DEFAULT dropout_rate is 0.1: deepfm.py:39,208-209; din.py:41,
235-236; fibinet.py:42,193-194; pnn.py:39,188-189).  The keep decisions are a counter-based hash of (seed, call, device step
counter, element) — csrc/dropout.h — or an explicit mask; kernel level, model level against the oracle with the SAME masks,
and a captured step against eager."""
import pytest
import torch

from oracle import ref_models as M
from recalgorithm_amd import feature_column as fc
from recalgorithm_amd import nn, ops
from recalgorithm_amd.estimator import Estimator, GraphedTrainStep, ModeKeys, RunConfig
from recalgorithm_amd.io import synth
from recalgorithm_amd.variables import named_grads
from tests.util import assert_bit_exact, assert_close

pytestmark = pytest.mark.gpu


def _spec(rate, dev, mask=None, seed=5, call=0, step=0):
    st = torch.tensor([step], dtype=torch.int64, device=dev)
    return ops.DropSpec(rate, mask, seed, call, st)


@pytest.mark.parametrize("shape", [(1, 4), (37, 20), (4096, 512), (3, 5, 7)])
def test_dropout_with_an_explicit_mask_is_the_masked_scaling(dev, shape):
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=gen)
    keep = (torch.rand(*shape, generator=gen) >= 0.3).float()
    g = torch.randn(*shape, generator=gen)
    xd = x.to(dev).requires_grad_(True)
    y = ops.dropout(xd, _spec(0.3, dev, keep.to(dev)))
    y.backward(g.to(dev))
    s = torch.tensor(1.0 / (1.0 - 0.3), dtype=torch.float32)
    assert_bit_exact(y, x * (keep * s), "dropout fwd")            # keep * s is exact (0 or s): one rounding, as in the kernel
    assert_bit_exact(xd.grad, g * (keep * s), "dropout bwd")


@pytest.mark.parametrize("rate", [0.1, 0.5, 0.9])
def test_dropout_hash_stream(dev, rate):
    shape = (4096, 256)
    x = torch.randn(*shape, device=dev)
    d = _spec(rate, dev, step=3)
    keep = ops.dropout_keep_mask(shape, d, dev)
    assert set(keep.unique().tolist()) <= {0.0, 1.0}
    n = keep.numel()
    frac = float(keep.mean())
    assert abs(frac - (1 - rate)) < 5 * (rate * (1 - rate) / n) ** 0.5 + 1e-4, frac
    # rows and columns are balanced too (no stripe pattern of the index hash)
    assert float((keep.mean(0) - (1 - rate)).abs().max()) < 0.05 and float((keep.mean(1) - (1 - rate)).abs().max()) < 0.15
    xd = x.clone().requires_grad_(True)
    y = ops.dropout(xd, d)
    y.backward(torch.ones_like(y))
    s = torch.tensor(1.0 / (1.0 - rate), dtype=torch.float32, device=dev)
    assert_bit_exact(y, x * (keep * s), "hash dropout fwd == its keep mask")
    assert_bit_exact(xd.grad, keep * s, "hash dropout bwd == its keep mask")
    assert_bit_exact(ops.dropout_keep_mask(shape, _spec(rate, dev, step=3), dev), keep, "same key, same mask")
    for other in (_spec(rate, dev, step=4), _spec(rate, dev, call=1, step=3), _spec(rate, dev, seed=6, step=3)):
        k2 = ops.dropout_keep_mask(shape, other, dev)
        agree = float((k2 == keep).float().mean())
        want = (1 - rate) ** 2 + rate ** 2                       # independent masks
        assert abs(agree - want) < 0.01, (agree, want)


def _make(model, dev, batch_norm=True, rate=0.1, B=300):
    spec = synth.SynthSpec(n_fields=8, max_vocab=400, seed=11, oov_frac=0.05, with_history=(model == "din"))
    cats = [fc.categorical_column_with_identity(n, v) for n, v in zip(spec.names, spec.vocabs)]
    common = {"hidden_units": ["64", "32"], "dropout_rate": rate, "batch_norm": batch_norm, "learning_rate": 0.005}
    if model == "deepfm":
        from recalgorithm_amd.algorithm.DeepFM.deepfm import deepfm_model_fn as fn
        params = dict(common, first_order_feature_columns=[fc.indicator_column(c) for c in cats],
                      second_order_feature_columns=[fc.embedding_column(c, 16) for c in cats])
        oracle = M.deepfm
    elif model == "pnn":
        from recalgorithm_amd.algorithm.PNN.pnn import pnn_model_fn as fn
        params = dict(common, category_feature_columns=[fc.embedding_column(c, 16) for c in cats], output_dimension=24,
                      product_method="IPNN", weight_regularizer=0.0, embedding_dim=16)
        oracle = M.pnn
    elif model == "fibinet":
        from recalgorithm_amd.algorithm.FiBiNET.fibinet import fibinet_model_fn as fn
        params = dict(common, category_feature_columns=[fc.embedding_column(c, 16) for c in cats], dense_feature_columns=[],
                      embedding_dim=16, reduction_ratio=2, bilinear_interaction_type="all")
        oracle = M.fibinet
    else:
        from recalgorithm_amd.algorithm.DIN.din import din_model_fn as fn
        cmap = dict(zip(spec.names, cats))
        his = fc.categorical_column_with_identity("his_read_comment_7d_seq", cmap["feedid"].num_buckets)
        his.is_sequence = True
        feed = cmap.pop("feedid")
        feed.is_sequence = True
        tgt, seq = fc.shared_embedding_columns([feed, his], 16, combiner="mean")
        cat = [fc.embedding_column(c, 16) for c in cmap.values()]
        params = dict(common, dense_feature_columns=[], category_feature_columns=cat, sequence_feature_columns=[seq],
                      target_feedid_feature_columns=[tgt], activation="dice", mini_batch_aware_regularization=True,
                      l2_lambda=0.2, use_softmax=False, sequence_max_length=50)
        oracle = M.din
    est = Estimator(fn, params, RunConfig(device=dev, seed=5))
    feats, labels, _ = synth.device_features(spec, B, dev)
    est.build(feats, labels)
    return est, params, feats, labels, oracle


def _oracle_inputs(est, feats, labels, dtype=torch.float64):
    P = {k: v.detach().cpu().to(dtype).requires_grad_(True) for k, v in est.store.named_arrays().items()}
    cf = {k: (v.cpu() if isinstance(v, torch.Tensor) else (v.values.cpu(), v.offsets.cpu())) for k, v in feats.items()}
    cl = {k: v.cpu().to(dtype) for k, v in labels.items()}
    return P, cf, cl


@pytest.mark.parametrize("model,batch_norm", [("deepfm", True), ("deepfm", False), ("pnn", True), ("fibinet", True),
                                              ("din", True), ("din", False)])
def test_model_with_hash_dropout_matches_the_oracle_on_the_same_masks(dev, model, batch_norm):
    """TRAIN forward + backward with the library's own random stream; the masks the hash stood for are read back
    (ops.dropout_keep_mask, the step counter has not moved) and handed to the oracle."""
    est, params, feats, labels, oracle = _make(model, dev, batch_norm)
    nn.DROPOUT_SPECS[:] = []
    spec = est._call_model_fn(feats, labels, ModeKeys.TRAIN)
    specs = list(nn.DROPOUT_SPECS)
    assert len(specs) == 2 and [d.call for d in specs] == [0, 1] and all(d.mask is None for d in specs)
    widths = [64, 32]
    B = labels["read_comment"].shape[0]
    masks = [ops.dropout_keep_mask((B, w), d, dev).cpu().double() for d, w in zip(specs, widths)]
    assert all(0.8 < float(m.mean()) < 0.97 for m in masks)
    P, cf, cl = _oracle_inputs(est, feats, labels)
    ref = oracle(P, cf, cl, params, training=True, dropout_masks=masks)
    ref["loss"].backward()
    P32, cf32, cl32 = _oracle_inputs(est, feats, labels, torch.float32)          # the reference arithmetic's own fp32 noise
    oracle(P32, cf32, cl32, params, training=True, dropout_masks=[m.float() for m in masks])["loss"].backward()
    assert_close(spec.loss, ref["loss"], what=f"{model} loss")
    assert_close(spec.predictions["probabilities"], ref["prob"], what=f"{model} prob")
    spec.loss.backward()
    grads = named_grads(est.store)
    for name, p in P.items():
        if p.grad is None:
            continue
        noise = float((P32[name].grad.double() - p.grad).abs().max())
        sib = name.replace("/bias", "/kernel")
        floor = 4 * noise + (1e-5 * float(P[sib].grad.abs().max()) if name.endswith("/bias") and sib in P and P[sib].grad is not None else 0.0)
        assert_close(grads[name], p.grad, what=f"{model} d({name})", reduced=True, floor=floor)


@pytest.mark.parametrize("model", ["deepfm", "din", "pnn"])
def test_captured_step_with_dropout_equals_eager(dev, model):
    """The hash is keyed by the DEVICE step counter, which the captured step advances itself: replay k draws the masks eager
    step k draws, so the two estimators stay together (and the masks do change from step to step)."""
    estA, params, feats, labels, _ = _make(model, dev, B=512)
    estB, _, _, _, _ = _make(model, dev, B=512)
    losses = [float(estA.train_step(feats, labels)) for _ in range(5)]
    g = GraphedTrainStep(estB.train_step, feats, labels, warmup=3)   # 3 eager + capture
    g()
    lb = g()
    torch.cuda.synchronize()
    assert len({round(l, 9) for l in losses}) == 5
    a, b = estA.store.named_arrays(), estB.store.named_arrays()
    for k in a:
        assert_close(b[k], a[k], rtol=1e-4, what=f"graph vs eager {k}", reduced=True)
    assert_close(lb, torch.tensor(losses[-1]), rtol=1e-5, what="graph vs eager loss")
    assert int(estB.store.opt_state["step"]) == 5
    # a different store seed is a different stream
    nn.DROPOUT_SPECS[:] = []
    estA._call_model_fn(feats, labels, ModeKeys.TRAIN)
    d0 = nn.DROPOUT_SPECS[0]
    k5 = ops.dropout_keep_mask((512, 64), d0, dev)
    estA.store.opt_state["step"] += 1
    k6 = ops.dropout_keep_mask((512, 64), d0, dev)
    assert 0.7 < float((k5 == k6).float().mean()) < 0.9


@pytest.mark.parametrize("M,K,N", [(4096, 416, 512), (300, 82, 52), (65, 48, 8)])
@pytest.mark.parametrize("explicit", [False, True])
def test_dropout_in_the_dense_epilogue_and_the_batchnorm_kernels_equals_the_separate_launches(dev, M, K, N, explicit):
    """recalgo_dense_fwd_drop / recalgo_batchnorm_apply_drop / recalgo_batchnorm_train_bwd_drop against the same layers with the
    dropout as launches of its own (ops.dropout): bit-identical outputs, tile moments of the DROPPED tensor, and gradients."""
    gen = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=gen).to(dev)
    w = (torch.randn(K, N, generator=gen) / K ** 0.5).to(dev)
    b = (torch.randn(N, generator=gen) * 0.1).to(dev)
    gamma, beta = (torch.rand(N, generator=gen) + 0.5).to(dev), torch.randn(N, generator=gen).to(dev)
    g = torch.randn(M, N, generator=gen).to(dev)
    mask = (torch.rand(M, N, generator=gen) >= 0.2).float().to(dev) if explicit else None
    d = _spec(0.2, dev, mask, call=3, step=7)
    s = torch.tensor(d.scale, dtype=torch.float32, device=dev)
    keep = mask if explicit else ops.dropout_keep_mask((M, N), d, dev)
    # ---- dense(relu) -> dropout -> batch_norm (deepfm.py:207-211) ----
    nb = ops.bn_partial_rows(M)
    part = torch.full((nb, 2 * N), float("nan"), device=dev)
    y = ops.dense_fwd(x, w, b, True, bn_partials=part, drop=d)
    y0 = ops.dense_fwd(x, w, b, True)
    assert_bit_exact(y, y0 * (keep * s), "dense epilogue dropout")
    if N % 4 == 0:
        want = torch.empty(nb, 2 * N, device=dev)
        from recalgorithm_amd import _lib
        _lib.check(_lib.load().recalgo_batchnorm_moments(ops._p(y), M, N, ops._p(want), ops._stream(y)), "moments")
        assert_close(part[:, :N], want[:, :N].double(), what="tile means of the dropped tensor")
        assert_close(part[:, N:], want[:, N:].double(), what="tile M2 of the dropped tensor", reduced=True)
        mm, mv = torch.zeros(N, device=dev), torch.ones(N, device=dev)
        o, mean, rstd = ops.batchnorm_train_fwd(y, gamma, beta, mm, mv, 0.99, 1e-3, partials=part)
        # BatchNorm backward: masks dx with x (= the dropped ReLU output) and scales it by 1 / (1 - rate) ...
        dg, db_ = torch.empty(N, device=dev), torch.empty(N, device=dev)
        dx = ops.batchnorm_train_bwd(y, gamma, mean, rstd, g, dg, db_, relu_x=True, relu_scale=d.scale)
        # ... == the plain BatchNorm backward followed by the dropout's and the ReLU's
        dg0, db0 = torch.empty(N, device=dev), torch.empty(N, device=dev)
        dx0 = ops.batchnorm_train_bwd(y, gamma, mean, rstd, g, dg0, db0)
        assert_bit_exact(dx, torch.where(y > 0, dx0 * s, torch.zeros_like(dx0)), "dx through the fused dropout + ReLU")
        assert torch.equal(dg, dg0) and torch.equal(db_, db0)
        # ---- batch_norm -> dropout (din.py:233-236) ----
        mm2, mv2 = torch.zeros(N, device=dev), torch.ones(N, device=dev)
        od, mean2, rstd2 = ops.batchnorm_train_fwd(y0, gamma, beta, mm2, mv2, 0.99, 1e-3, out_drop=d)
        mm3, mv3 = torch.zeros(N, device=dev), torch.ones(N, device=dev)
        o3, mean3, rstd3 = ops.batchnorm_train_fwd(y0, gamma, beta, mm3, mv3, 0.99, 1e-3)
        assert_bit_exact(od, o3 * (keep * s), "BatchNorm store dropout")
        assert torch.equal(mean2, mean3) and torch.equal(rstd2, rstd3) and torch.equal(mm2, mm3)
        dg1, db1 = torch.empty(N, device=dev), torch.empty(N, device=dev)
        dxa = ops.batchnorm_train_bwd(y0, gamma, mean3, rstd3, g, dg1, db1, g_drop=d)
        dg2, db2 = torch.empty(N, device=dev), torch.empty(N, device=dev)
        dxb = ops.batchnorm_train_bwd(y0, gamma, mean3, rstd3, g * (keep * s), dg2, db2)
        assert_bit_exact(dxa, dxb, "BatchNorm backward reading g through the dropout")
        assert torch.equal(dg1, dg2) and torch.equal(db1, db2)
