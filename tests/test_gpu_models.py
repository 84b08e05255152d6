"""-m gpu model-level parity: the mirrored model_fns (HIP kernels + hipBLASLt MLP) vs
oracle/ref_models.py on the same weights and batch: logits, loss, every gradient, and the
weights after one TF1-Adam step.  Also: hipGraph replay == eager."""
import re

import pytest
import torch

from oracle import ref_models as M
from oracle import ref_ops as R
from recalgorithm_amd import feature_column as fc
from recalgorithm_amd.estimator import Estimator, GraphedTrainStep, ModeKeys, RunConfig
from recalgorithm_amd.io import synth
from recalgorithm_amd.variables import named_grads
from tests.util import assert_adam_update, assert_bit_exact, assert_close

pytestmark = pytest.mark.gpu

ORACLE = {"dcn": M.dcn, "deepfm": M.deepfm}


def make(model, dev, n_fields=8, K=16, B=300, hidden=("64", "32"), max_vocab=400, **extra):
    spec = synth.SynthSpec(n_fields=n_fields, max_vocab=max_vocab, seed=11, oov_frac=0.05)
    cats = [fc.categorical_column_with_identity(n, v) for n, v in zip(spec.names, spec.vocabs)]
    if model == "dcn":
        from recalgorithm_amd.algorithm.DCN.dcn import dcn_model_fn as fn
        params = {"category_feature_columns": [fc.embedding_column(c, K) for c in cats],
                  "dense_feature_columns": [], "hidden_units": list(hidden), "num_cross_layer": 3,
                  "learning_rate": 0.005}
    elif model == "deepfm":
        from recalgorithm_amd.algorithm.DeepFM.deepfm import deepfm_model_fn as fn
        params = {"first_order_feature_columns": [fc.indicator_column(c) for c in cats],
                  "second_order_feature_columns": [fc.embedding_column(c, K) for c in cats],
                  "hidden_units": list(hidden), "dropout_rate": 0.0, "batch_norm": True,
                  "learning_rate": 0.005}
    params.update(extra)
    est = Estimator(fn, params, RunConfig(device=dev, seed=5))
    feats, labels, _ = synth.device_features(spec, B, dev)
    est.build(feats, labels)
    return est, params, feats, labels


def oracle_inputs(est, feats, labels, dtype=torch.float64):
    P = {k: v.detach().cpu().to(dtype).requires_grad_(True) for k, v in est.store.named_arrays().items()}
    cf = {k: (v.cpu() if isinstance(v, torch.Tensor) else (v.values.cpu(), v.offsets.cpu())) for k, v in feats.items()}
    cf = {k: (v.to(dtype) if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in cf.items()}
    cl = {k: v.cpu().to(dtype) for k, v in labels.items()}
    return P, cf, cl


@pytest.mark.parametrize("model", ["dcn", "deepfm"])
def test_model_forward_backward_adam(dev, model):
    est, params, feats, labels = make(model, dev)
    P, cf, cl = oracle_inputs(est, feats, labels)
    ref = ORACLE[model](P, cf, cl, params, training=True)
    ref["loss"].backward()

    spec = est._call_model_fn(feats, labels, ModeKeys.TRAIN)
    assert_close(spec.loss, ref["loss"], what=f"{model} loss")
    assert_close(spec.predictions["probabilities"], ref["prob"], what=f"{model} prob")
    spec.loss.backward()
    grads = named_grads(est.store)
    skipped = []
    for name, p in P.items():
        if p.grad is None:
            skipped.append(name)
            continue
        assert_close(grads[name], p.grad, what=f"{model} d({name})", reduced=True)
    assert all("moving_" in s or s.endswith("/wl") or s.endswith("/bl") for s in skipped), skipped

    # one TF1-Adam step on both sides
    before = {k: v.detach().cpu().double().clone() for k, v in est.store.named_arrays().items()}
    spec.train_op.optimizer.apply_gradients(est.store)
    after = est.store.named_arrays()
    for name, p in P.items():
        if p.grad is None:
            continue
        pp, m, v = before[name].clone(), torch.zeros_like(before[name]), torch.zeros_like(before[name])
        R.adam_tf1_step(pp, p.grad, m, v, 1, params["learning_rate"])
        # compare the UPDATE (p_after - p_before), not p: at step 1 it is lr*g/(|g| + eps'), ~lr*sign(g), and
        # ill-conditioned in g only where |g| ~ eps' = 3e-7 (tests/util.py assert_adam_update)
        lr = params["learning_rate"]
        gref = p.grad.abs()
        tol_g = 1e-5 * (gref + gref.pow(2).mean().sqrt()) + 1e-6 * gref.max()
        upd = after[name].detach().cpu().double() - before[name]
        assert_adam_update(upd, pp - before[name], before[name], gref, tol_g, lr, what=f"{model} adam update {name}")
    # gradients were consumed and zeroed by the fused optimizer
    assert float(est.store.flat_grad.abs().sum()) == 0.0
    for ar in est.store.arenas.values():
        assert float(ar.grad.abs().sum()) == 0.0


def test_dcn_eval_and_predict_modes(dev):
    est, params, feats, labels = make("dcn", dev)
    P, cf, cl = oracle_inputs(est, feats, labels)
    ref = M.dcn(P, cf, cl, params)
    ev = est._call_model_fn(feats, labels, ModeKeys.EVAL)
    assert set(ev.eval_metric_ops) == {"eval_accuracy", "eval_auc"}
    assert_close(ev.loss, ref["loss"], what="eval loss")
    pr = est._call_model_fn(feats, None, ModeKeys.PREDICT)
    assert set(pr.predictions) == {"logit", "probabilities"}
    assert_close(pr.predictions["logit"], ref["logit"], what="predict logit")
    for k, (m, _) in ev.eval_metric_ops.items():
        m.update()
    auc = ev.eval_metric_ops["eval_auc"][0].result()
    assert abs(auc - R.tf_metrics_auc(cl["read_comment"], ref["prob"])) < 1e-6


def test_deepfm_prediction_keys(dev):
    est, params, feats, labels = make("deepfm", dev)
    pr = est._call_model_fn(feats, None, ModeKeys.PREDICT)
    assert set(pr.predictions) == {"probabilities", "fm_first_order_logit", "fm_second_order_logit", "deep_logit"}


def test_string_hyperparameters_accepted(dev):
    """hidden_units arrive as strings from FLAGS.hidden_units.split(',') (quirk B-2)."""
    est, params, feats, labels = make("dcn", dev, hidden=("32", "16"))
    assert est.store.vars["dnn_part/dnn_dense_0/kernel"].shape == (8 * 16, 32)


@pytest.mark.parametrize("model", ["dcn", "deepfm", "dcn_tail"])
def test_hipgraph_replay_matches_eager(dev, model):
    # dcn_tail: hidden units for which the step runs the fused last layer + head + loss kernel and its riders (csrc/tailfuse.hip)
    kw = dict(hidden=("256", "256", "128")) if model == "dcn_tail" else {}
    model = "dcn" if model == "dcn_tail" else model
    estA, params, feats, labels = make(model, dev, B=512, **kw)
    estB, _, _, _ = make(model, dev, B=512, **kw)
    for _ in range(5):
        la = estA.train_step(feats, labels)
    g = GraphedTrainStep(estB.train_step, feats, labels, warmup=3)   # 3 eager + capture
    g()                                                               # replay #1 -> 4 steps
    lb = g()                                                          # replay #2 -> 5 steps
    torch.cuda.synchronize()
    a, b = estA.store.named_arrays(), estB.store.named_arrays()
    for k in a:
        assert_close(b[k], a[k], rtol=1e-4, what=f"graph vs eager {k}", reduced=True)
    assert_close(lb, la, rtol=1e-5, what="graph vs eager loss")
    assert int(estB.store.opt_state["step"]) == 5


def _make_xdeepfm(dev, n_fields=8, K=8, B=160, maps=("12", "10")):
    from recalgorithm_amd.algorithm.xDeepFM.xdeepfm import xdeepfm_model_fn
    spec = synth.SynthSpec(n_fields=n_fields, max_vocab=300, seed=13, oov_frac=0.05)
    cats = [fc.categorical_column_with_identity(n, v) for n, v in zip(spec.names, spec.vocabs)]
    params = {"category_feature_columns": [fc.embedding_column(c, K) for c in cats],
              "dense_feature_columns": [], "hidden_units": ["32", "16"], "learning_rate": 0.005,
              "embedding_dim": K, "cin_layer_feature_maps": list(maps)}
    est = Estimator(xdeepfm_model_fn, params, RunConfig(device=dev, seed=5))
    feats, labels, _ = synth.device_features(spec, B, dev)
    est.build(feats, labels)
    return est, params, feats, labels


def test_xdeepfm_forward_backward(dev):
    est, params, feats, labels = _make_xdeepfm(dev)
    P, cf, cl = oracle_inputs(est, feats, labels)
    ref = M.xdeepfm(P, cf, cl, params, training=True)
    ref["loss"].backward()
    spec = est._call_model_fn(feats, labels, ModeKeys.TRAIN)
    assert_close(spec.loss, ref["loss"], what="xdeepfm loss")
    assert_close(spec.predictions["probabilities"], ref["prob"], what="xdeepfm prob")
    spec.loss.backward()
    grads = named_grads(est.store)
    for name, p in P.items():
        if p.grad is not None:
            assert_close(grads[name], p.grad, what=f"xdeepfm d({name})", reduced=True)


@pytest.mark.parametrize("use_softmax,activation", [(False, "dice"), (True, "prelu")])
def test_din_forward_backward(dev, use_softmax, activation):
    from recalgorithm_amd.algorithm.DIN.din import din_model_fn
    spec = synth.SynthSpec(n_fields=6, max_vocab=300, seed=17, oov_frac=0.05, with_history=True, with_dense=True)
    cats = {n: fc.categorical_column_with_identity(n, v) for n, v in zip(spec.names, spec.vocabs)}
    his = fc.categorical_column_with_identity("his_read_comment_7d_seq", cats["feedid"].num_buckets)
    his.is_sequence = True
    feed = cats.pop("feedid")
    feed.is_sequence = True
    shared = fc.shared_embedding_columns([feed, his], 16, combiner="mean")
    dims = {"userid": 16, "device": 2, "authorid": 4, "bgm_song_id": 4, "bgm_singer_id": 4}
    from recalgorithm_amd.algorithm._common import DENSE_FEATURES
    params = {"dense_feature_columns": [fc.numeric_column(k, default_value=0.0) for k in DENSE_FEATURES],
              "category_feature_columns": [fc.embedding_column(cats[k], d) for k, d in dims.items()],
              "target_feedid_feature_columns": [shared[0]], "sequence_feature_columns": [shared[1]],
              "hidden_units": ["32", "16"], "dropout_rate": 0.0, "batch_norm": True, "learning_rate": 0.005,
              "activation": activation, "mini_batch_aware_regularization": True, "l2_lambda": 0.2,
              "use_softmax": use_softmax}
    est = Estimator(din_model_fn, params, RunConfig(device=dev, seed=5))
    feats, labels, _ = synth.device_features(spec, 200, dev)
    est.build(feats, labels)
    # alpha starts at 1.0, which makes PReLU and Dice the identity (activations.py:13-17,31) and the
    # whole fcn stack affine: move it away so that the activations are exercised for real
    g = torch.Generator().manual_seed(99)
    for name, v in est.store.vars.items():
        if "alpha" in name:
            v.data.copy_((0.25 + 0.5 * torch.rand(v.data.shape, generator=g)).to(dev))
    P, cf, cl = oracle_inputs(est, feats, labels)
    ref = M.din(P, cf, cl, params, training=True)
    ref["loss"].backward()
    # fp32 evaluation noise of the reference arithmetic itself (the same op-for-op restatement run
    # in float32, i.e. what the TF1-CPU graph would do): batch-summed gradients behind a
    # BatchNorm cancel heavily (sum_b g_b = 0), so their error scales with sum|terms|, which only
    # the fp32 run of the same sum can tell
    P32, cf32, cl32 = oracle_inputs(est, feats, labels, dtype=torch.float32)
    M.din(P32, cf32, cl32, params, training=True)["loss"].backward()
    spec_ = est._call_model_fn(feats, labels, ModeKeys.TRAIN)
    assert_close(spec_.loss, ref["loss"], what="din loss")
    assert_close(spec_.predictions["probabilities"], ref["prob"], what="din prob")
    spec_.loss.backward()
    grads = named_grads(est.store)
    for name, p in P.items():
        if p.grad is None:
            continue
        noise = float((P32[name].grad.double() - p.grad).abs().max())
        if name.endswith("/bias") and name.replace("bias", "kernel") in P:
            # biases whose gradient cancels analytically (f3_att/bias = sum_t ds_t is exactly 0 under
            # softmax; a dense bias feeding a training-mode BatchNorm): judged at the scale of the
            # sibling kernel gradient (same upstream terms, no cancellation)
            scale = float(P[name.replace("bias", "kernel")].grad.abs().max())
            err = float((grads[name].cpu().double() - p.grad).abs().max())
            assert err <= 1e-5 * scale + 4 * noise, f"{name}: err {err} scale {scale} noise {noise}"
            continue
        assert_close(grads[name], p.grad, what=f"din d({name})", reduced=True, floor=4 * noise)


def _grad_parity(model, est, P, grads):
    for name, p in P.items():
        if p.grad is None:
            continue
        if re.search(r"/dense(_\d+)?/bias$", name) and name.replace("bias", "kernel") in P \
                and any(k.startswith(name.rsplit("/", 2)[0] + "/batch_normalization") for k in P):
            # a bias feeding a training-mode BatchNorm (dense -> BN order of deepfm/pnn/fibinet, quirk B-7,
            # through a ReLU here so it does not cancel exactly, but stays tiny): judged at the scale
            # of the sibling kernel gradient
            scale = float(P[name.replace("bias", "kernel")].grad.abs().max())
            err = float((grads[name].cpu().double() - p.grad).abs().max())
            assert err <= 1e-5 * scale + 1e-5 * float(p.grad.abs().max()), f"{model} d({name}) err {err}"
            continue
        assert_close(grads[name], p.grad, what=f"{model} d({name})", reduced=True)


@pytest.mark.parametrize("btype", ["all", "each", "interaction"])
def test_fibinet_forward_backward(dev, btype):
    from recalgorithm_amd.algorithm.FiBiNET.fibinet import fibinet_model_fn
    K = 8
    spec = synth.SynthSpec(n_fields=7, max_vocab=300, seed=21, oov_frac=0.05, with_dense=True)
    cats = [fc.categorical_column_with_identity(n, v) for n, v in zip(spec.names, spec.vocabs)]
    from recalgorithm_amd.algorithm._common import DENSE_FEATURES
    params = {"category_feature_columns": [fc.embedding_column(c, K) for c in cats],
              "dense_feature_columns": [fc.numeric_column(k, default_value=0.0) for k in DENSE_FEATURES],
              "hidden_units": ["32", "16"], "dropout_rate": 0.0, "batch_norm": True, "learning_rate": 0.005,
              "embedding_dim": K, "reduction_ratio": 2, "bilinear_interaction_type": btype}
    est = Estimator(fibinet_model_fn, params, RunConfig(device=dev, seed=5))
    feats, labels, _ = synth.device_features(spec, 150, dev)
    est.build(feats, labels)
    names = set(est.store.named_arrays())
    assert f"bilinear_interaction_part/orginal_w_{btype}" in names and "senet_part/senet_w1" in names
    P, cf, cl = oracle_inputs(est, feats, labels)
    # eval / predict modes share the forward (before TRAIN, whose forward updates the BN moving stats)
    ev = est._call_model_fn(feats, labels, ModeKeys.EVAL)
    ref_e = M.fibinet(P, cf, cl, params, training=False)
    assert_close(ev.loss, ref_e["loss"], what="fibinet eval loss")
    pr = est._call_model_fn(feats, None, ModeKeys.PREDICT)
    assert set(pr.predictions) == {"logit", "probabilities"}
    assert_close(pr.predictions["logit"], ref_e["logit"], what="fibinet logit")
    ref = M.fibinet(P, cf, cl, params, training=True)
    ref["loss"].backward()
    spec_ = est._call_model_fn(feats, labels, ModeKeys.TRAIN)
    assert_close(spec_.loss, ref["loss"], what="fibinet loss")
    assert_close(spec_.predictions["probabilities"], ref["prob"], what="fibinet prob")
    spec_.loss.backward()
    _grad_parity("fibinet", est, P, named_grads(est.store))


@pytest.mark.parametrize("method,wr", [("IPNN", 0.0), ("OPNN", 0.0), ("IPNN", 0.01), ("OPNN", 0.02)])
def test_pnn_forward_backward(dev, method, wr):
    from recalgorithm_amd.algorithm.PNN.pnn import pnn_model_fn
    K, D = 8, 48
    spec = synth.SynthSpec(n_fields=7, max_vocab=300, seed=23, oov_frac=0.05)
    cats = [fc.categorical_column_with_identity(n, v) for n, v in zip(spec.names, spec.vocabs)]
    params = {"category_feature_columns": [fc.embedding_column(c, K) for c in cats],
              "hidden_units": ["32", "16"], "dropout_rate": 0.0, "batch_norm": True, "learning_rate": 0.005,
              "output_dimension": D, "product_method": method, "weight_regularizer": wr, "embedding_dim": K}
    est = Estimator(pnn_model_fn, params, RunConfig(device=dev, seed=5))
    feats, labels, _ = synth.device_features(spec, 170, dev, sorted_layout=False)
    est.build(feats, labels)
    names = set(est.store.named_arrays())
    assert "linear_part/linear_w" in names and "bias" in names
    assert ("product_part/inner_product_w" if method == "IPNN" else "product_part/outer_product_w") in names
    P, cf, cl = oracle_inputs(est, feats, labels)
    ref = M.pnn(P, cf, cl, params, training=True)
    ref["loss"].backward()
    spec_ = est._call_model_fn(feats, labels, ModeKeys.TRAIN)
    assert_close(spec_.loss, ref["loss"], what="pnn loss")
    assert_close(spec_.predictions["probabilities"], ref["prob"], what="pnn prob")
    spec_.loss.backward()
    grads = named_grads(est.store)
    _grad_parity("pnn", est, P, grads)
    if method == "OPNN" and wr == 0.0:      # quirk B-10: only the upper triangle of each (K,K) weight is used
        gw = grads["product_part/outer_product_w"]
        assert float(torch.tril(gw, diagonal=-1).abs().max()) == 0.0


@pytest.mark.parametrize("model", ["dcn", "deepfm", "dcn_tail"])
def test_checkpoint_resume_equals_uninterrupted_training(dev, model, tmp_path):
    """4 training steps in one process == 2 steps, save_checkpoint, a NEW estimator restoring from model_dir,
    2 more steps: variables, Adam moments and losses (the live-row list is rebuilt from the restored moments in
    a different order, which must not matter).  Not bit for bit: two runs of the same steps already differ in
    the last bits because the row-gradient scatter adds with float atomics."""
    from recalgorithm_amd.algorithm.DCN.dcn import dcn_model_fn
    from recalgorithm_amd.algorithm.DeepFM.deepfm import deepfm_model_fn
    kw = dict(hidden=("256", "256", "128")) if model == "dcn_tail" else {}      # (the fused tail and its riders, csrc/tailfuse.hip)
    model = "dcn" if model == "dcn_tail" else model
    fn = {"dcn": dcn_model_fn, "deepfm": deepfm_model_fn}[model]
    ref, params, feats, labels = make(model, dev, **kw)
    spec = synth.SynthSpec(n_fields=8, max_vocab=400, seed=11, oov_frac=0.05)
    batches = [(feats, labels)] + [synth.device_features(spec, 300, dev, batch_index=i)[:2] for i in (1, 2, 3)]
    losses_ref = [float(ref.train_step(*b)) for b in batches]
    md = str(tmp_path / "model_dir")
    a = Estimator(fn, params, RunConfig(device=dev, seed=5, model_dir=md, use_hip_graph=False))
    a.build(feats, labels)
    la = [float(a.train_step(*b)) for b in batches[:2]]
    a.global_step = 2
    a.save_checkpoint()
    b = Estimator(fn, params, RunConfig(device=dev, seed=77, model_dir=md, use_hip_graph=False))   # other seed: state comes from the file
    b.build(feats, labels)
    assert b.global_step == 2
    lb = [float(b.train_step(*bt)) for bt in batches[2:]]
    for got, want in zip(la + lb, losses_ref):
        assert abs(got - want) <= 1e-6 * abs(want), (la + lb, losses_ref)
    for k, v in ref.store.named_arrays().items():
        assert_close(b.store.named_arrays()[k], v.double(), rtol=1e-4, what=f"{model} {k} after resume", reduced=True)
    for n, ar in ref.store.arenas.items():
        assert_close(b.store.arenas[n].m, ar.m.double(), rtol=1e-4, what=f"{model} arena {n}.m", reduced=True)
        assert_close(b.store.arenas[n].v, ar.v.double(), rtol=1e-4, what=f"{model} arena {n}.v", reduced=True)
    assert_close(b.store.flat_m, ref.store.flat_m.double(), rtol=1e-4, what="dense m", reduced=True)
    assert_close(b.store.flat_v, ref.store.flat_v.double(), rtol=1e-4, what="dense v", reduced=True)


@pytest.mark.parametrize("model", ["dcn", "deepfm", "xdeepfm", "din"])
def test_fused_train_step_matches_unfused_graph(dev, model):
    """Estimator.train_step knows the loss-gradient seed before the forward, so the one-unit head(s), the sigmoid-CE and
    their backward run as ONE kernel, the step counter / weight-gradient split sums / loss value are finished by one
    deferred launch and the two gradients of a shared input are joined inside a kernel.  The same step driven op by op
    (model_fn, loss.backward(), apply_gradients: separate head, loss and optimizer launches) must give the same
    losses and variables."""
    if model in ("dcn", "deepfm"):
        mk = lambda: make(model, dev)
    elif model == "xdeepfm":
        mk = lambda: _make_xdeepfm(dev)
    else:
        def mk():
            import bench
            args = bench.parse_args(["--model", "din", "--batch", "256", "--fields", "8", "--max-vocab", "500"])
            est, spec, feats, labels, _ = bench.build_estimator(args, dev)
            return est, est.params, feats, labels
    a, params, feats, labels = mk()
    b, _, _, _ = mk()
    if model == "din":
        # alpha = 1 makes Dice the identity: every dense -> BN pair is then affine and the gradients of the biases / BN
        # shifts in between cancel analytically (noise that Adam turns into O(lr) moves) — move alpha away from 1
        for est in (a, b):
            g = torch.Generator().manual_seed(99)
            for name, v in est.store.vars.items():
                if "alpha" in name:
                    v.data.copy_((0.25 + 0.5 * torch.rand(v.data.shape, generator=g)).to(dev))
    la, lb = [], []
    for _ in range(3):
        la.append(float(a.train_step(feats, labels)))                       # fused tail
        spec = b._call_model_fn(feats, labels, ModeKeys.TRAIN)               # op by op
        spec.loss.backward()
        spec.train_op.optimizer.apply_gradients(b.store)
        lb.append(float(spec.loss))
    torch.cuda.synchronize()
    for x, y in zip(la, lb):
        assert abs(x - y) <= 2e-6 * abs(y), (la, lb)
    A, B_ = a.store.named_arrays(), b.store.named_arrays()
    for k in A:
        if k.endswith("/bias") and model in ("deepfm", "din"):
            # a dense bias in front of a training-mode BatchNorm: its gradient cancels analytically, what is left is
            # rounding noise, and Adam's g / (|g| + eps') turns noise into O(lr) moves — not comparable run to run
            continue
        assert_close(A[k], B_[k].double(), rtol=1e-4, what=f"{model} {k}: fused vs op-by-op after 3 steps", reduced=True)
    assert int(a.store.opt_state["step"]) == int(b.store.opt_state["step"]) == 3


def test_graphed_step_copies_into_private_buffers(dev):
    """GraphedTrainStep captures on PRIVATE static input buffers (storage-preserving clones of the first batch): the
    caller's tensors are never written, and a batch tensor that comes round again is copied again, not skipped."""
    est, params, feats, labels = make("dcn", dev, B=256)
    ref, _, _, _ = make("dcn", dev, B=256)
    spec = synth.SynthSpec(n_fields=8, max_vocab=400, seed=11, oov_frac=0.05)
    b1 = synth.device_features(spec, 256, dev, batch_index=1)[:2]
    keep0 = {k: v.clone() for k, v in feats.items()}
    keep1 = {k: v.clone() for k, v in b1[0].items()}
    g = GraphedTrainStep(est.train_step, feats, labels, warmup=2)      # two eager steps (lazy state is built), then capture
    for _ in range(2):
        ref.train_step(feats, labels)                                  # the two warm-up steps (the capture pass executes nothing)
    seq = [(feats, labels), b1, (feats, labels), b1, (feats, labels)]
    la = [float(g(*b)) for b in seq]
    lr_ = [float(ref.train_step(*b)) for b in seq]
    torch.cuda.synchronize()
    assert all(torch.equal(feats[k], keep0[k]) for k in keep0) and all(torch.equal(b1[0][k], keep1[k]) for k in keep1)
    for x, y in zip(la, lr_):
        assert abs(x - y) <= 5e-6 * abs(y), (la, lr_)
    assert any(v.data_ptr() != g.static_f[k].data_ptr() for k, v in feats.items())


@pytest.mark.parametrize("model", ["dcn", "deepfm", "dcn_tail"])
def test_steps_and_resume_are_bit_reproducible(dev, model, tmp_path):
    """No kernel of the step issues a float atomic (the row-gradient scatter is owner-computes, every split sum has a fixed
    order): (a) two runs of the same steps and (b) a run interrupted by save_checkpoint / restore in a NEW estimator are
    BIT-identical — variables, tables, Adam moments."""
    from recalgorithm_amd.algorithm.DCN.dcn import dcn_model_fn
    from recalgorithm_amd.algorithm.DeepFM.deepfm import deepfm_model_fn
    kw = dict(hidden=("256", "256", "128")) if model == "dcn_tail" else {}      # (the fused tail and its riders, csrc/tailfuse.hip)
    model = "dcn" if model == "dcn_tail" else model
    fn = {"dcn": dcn_model_fn, "deepfm": deepfm_model_fn}[model]
    ref, params, feats, labels = make(model, dev, **kw)
    spec = synth.SynthSpec(n_fields=8, max_vocab=400, seed=11, oov_frac=0.05)
    batches = [(feats, labels)] + [synth.device_features(spec, 300, dev, batch_index=i)[:2] for i in (1, 2, 3)]
    losses_ref = [float(ref.train_step(*b)) for b in batches]
    again, _, _, _ = make(model, dev, **kw)
    assert [float(again.train_step(*b)) for b in batches] == losses_ref
    md = str(tmp_path / "model_dir")
    a = Estimator(fn, params, RunConfig(device=dev, seed=5, model_dir=md, use_hip_graph=False))
    a.build(feats, labels)
    la = [float(a.train_step(*b)) for b in batches[:2]]
    a.global_step = 2
    a.save_checkpoint()
    b = Estimator(fn, params, RunConfig(device=dev, seed=77, model_dir=md, use_hip_graph=False))
    b.build(feats, labels)
    lb = [float(b.train_step(*bt)) for bt in batches[2:]]
    assert la + lb == losses_ref
    for other in (again, b):
        for k, v in ref.store.named_arrays().items():
            assert_bit_exact(other.store.named_arrays()[k], v, f"{model} {k}")
        for n, ar in ref.store.arenas.items():
            assert_bit_exact(other.store.arenas[n].m, ar.m, f"{model} arena {n}.m")
            assert_bit_exact(other.store.arenas[n].v, ar.v, f"{model} arena {n}.v")
        assert_bit_exact(other.store.flat_m, ref.store.flat_m, "dense m")
        assert_bit_exact(other.store.flat_v, ref.store.flat_v, "dense v")


@pytest.mark.parametrize("layout", ["mixed_dims", "with_tags", "indicator", "one_width"])
def test_dcn_without_dense_columns_on_every_input_layer_path(dev, layout):
    """dcn_model_fn with NO dense columns promises the cross kernel the gather (ops.gather_feeds_cross).  Only the
    one-width, single-valued input_layer keeps that promise; mixed embedding widths (the script's own 16/2/4/4/4,
    dcn.py:97-107), a multi-valued column or an indicator column take input_layer's per-column path + concat, where a
    pending gather must have been launched before the concat reads it: bit-equal to LAZY_GATHER off."""
    from recalgorithm_amd import ops
    from recalgorithm_amd.algorithm.DCN.dcn import dcn_model_fn
    spec = synth.SynthSpec(n_fields=6, max_vocab=300, seed=3, oov_frac=0.05, with_tags=(layout == "with_tags"))
    cats = [fc.categorical_column_with_identity(n, v) for n, v in zip(spec.names, spec.vocabs)]
    dims = {"mixed_dims": [16, 2, 4, 4, 4, 16], "with_tags": [8] * 6, "indicator": [8] * 6, "one_width": [8] * 6}[layout]
    cols = [fc.embedding_column(c, k) for c, k in zip(cats, dims)]
    if layout == "with_tags":
        cols.append(fc.embedding_column(fc.categorical_column_with_identity("manual_tag_list", spec.tag_vocab), 8, combiner="mean"))
    if layout == "indicator":
        cols = cols[:1] + [fc.indicator_column(cats[2])]          # ONE gather, then a concat with a multi-hot block
    params = {"category_feature_columns": cols, "dense_feature_columns": [], "hidden_units": ["32", "16"],
              "num_cross_layer": 2, "learning_rate": 0.005}

    def run(lazy):
        prev, ops.LAZY_GATHER = ops.LAZY_GATHER, lazy
        try:
            est = Estimator(dcn_model_fn, params, RunConfig(device=dev, seed=5))
            feats, labels, _ = synth.device_features(spec, 257, dev)
            est.build(feats, labels)
            spec_ = est._call_model_fn(feats, labels, ModeKeys.TRAIN)
            assert not ops._lazy_gathers
            spec_.loss.backward()
            ops.flush_dense_splits()
            g = {k: v.detach().clone() for k, v in named_grads(est.store).items()}
            return spec_.loss.detach().clone(), spec_.predictions["probabilities"].detach().clone(), g
        finally:
            ops.LAZY_GATHER = prev
    la, pa, ga = run(True)
    lb, pb, gb = run(False)
    assert_bit_exact(la, lb, f"{layout} loss")
    assert_bit_exact(pa, pb, f"{layout} probabilities")
    assert set(ga) == set(gb)
    for k in ga:
        assert_bit_exact(ga[k], gb[k], f"{layout} d({k})")
    assert torch.isfinite(pa).all()


def test_din_batched_lookups_are_bit_identical_to_separate_prepares(dev):
    """DIN issues its three lookups (profile fields, target item, history) inside sparse.batch_lookups(): ONE `prepare` launch for
    the arena (recalgo_scatter_prepare_multi) and the three forward kernels behind it, against one launch per lookup in front of
    each forward kernel.  The counts, the catch-up and the sweep are the same work on the same rows: every variable, table and
    optimizer slot bit-identical after 4 steps."""
    import bench
    from recalgorithm_amd import sparse
    runs = {}
    for batched in (True, False):
        sparse.BATCH_LOOKUPS = batched
        try:
            args = bench.parse_args(["--model", "din", "--batch", "256", "--fields", "8", "--max-vocab", "500"])
            est, spec, feats, labels, _ = bench.build_estimator(args, dev)
            merged0, ml0 = sparse.prepare_stats["merged"], sparse.prepare_stats["merged_lookups"]
            losses = [float(est.train_step(feats, labels)) for _ in range(4)]
            merged = sparse.prepare_stats["merged"] - merged0
            assert sparse.prepare_stats["merged_lookups"] - ml0 == (8 if batched else 0)   # (three forward kernels -> one, four steps)
            torch.cuda.synchronize()
            sparse.sync_store(est.store)
            arrays = {k: v.detach().clone() for k, v in est.store.named_arrays().items()}
            arrays["flat_m"], arrays["flat_v"] = est.store.flat_m.clone(), est.store.flat_v.clone()
            for n, a in est.store.arenas.items():
                if a.weight is not None:
                    arrays[f"{n}/m"], arrays[f"{n}/v"] = a.m.clone(), a.v.clone()
            runs[batched] = (losses, arrays, merged)
        finally:
            sparse.BATCH_LOOKUPS = True
    assert runs[True][2] >= 4 and runs[False][2] == 0, "the batched run was expected to save `prepare` launches, the other none"
    assert runs[True][0] == runs[False][0]
    for k in runs[False][1]:
        assert_bit_exact(runs[True][1][k], runs[False][1][k], what=f"din {k}: batched lookups vs separate prepares")


def test_nfm_batched_lookups_with_an_unused_lookup_are_bit_identical(dev):
    """NFM looks the category columns up twice — once unused (nfm.py:150-151) — inside sparse.batch_lookups().  The unused lookup is
    counted into the plan's workspace by the block's one launch but takes no part in `apply`: the plan's book-keeping has to say so
    (a workspace re-sized between the two registrations used to leave totals that `apply` trusted: buckets overflowed).  Steps with
    the block against steps with one `prepare` launch per lookup: bit-identical."""
    import bench
    from recalgorithm_amd import sparse
    runs = {}
    for batched in (True, False):
        sparse.BATCH_LOOKUPS = batched
        try:
            args = bench.parse_args(["--model", "nfm", "--batch", "1024", "--fields", "12", "--max-vocab", "3000"])
            est, spec, feats, labels, _ = bench.build_estimator(args, dev)
            losses = [float(est.train_step(feats, labels)) for _ in range(4)]
            torch.cuda.synchronize()
            sparse.sync_store(est.store)
            arrays = {k: v.detach().clone() for k, v in est.store.named_arrays().items()}
            for n, a in est.store.arenas.items():
                if a.weight is not None:
                    arrays[f"{n}/m"], arrays[f"{n}/v"] = a.m.clone(), a.v.clone()
            runs[batched] = (losses, arrays)
        finally:
            sparse.BATCH_LOOKUPS = True
    assert runs[True][0] == runs[False][0]
    for k in runs[False][1]:
        assert_bit_exact(runs[True][1][k], runs[False][1][k], what=f"nfm {k}: batched lookups vs separate prepares")
