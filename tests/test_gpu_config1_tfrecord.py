"""-m gpu: BASELINE.json configs[0] — DeepFM, 26 sparse fields, emb 16, batch 512, from TFRecord
bytes on disk to the loss: vocabulary files + tf.train.Example records (synthetic, WeChat-shaped)
-> train_input_fn / example_parser (string keys) -> vocabulary encoding -> fused DeepFM sparse
kernel + MLP -> loss.  Checked against the oracle fed with the ids the generator drew."""
import pytest
import torch

from oracle import ref_models as M
from recalgorithm_amd import feature_column as fc
from recalgorithm_amd.algorithm.DeepFM.deepfm import deepfm_model_fn
from recalgorithm_amd.algorithm.utils import eval_input_fn, parse_example, train_input_fn
from recalgorithm_amd.estimator import Estimator, ModeKeys, RunConfig
from recalgorithm_amd.io import synth
from tests.util import assert_close

pytestmark = pytest.mark.gpu


def test_deepfm_from_tfrecord_batch512(dev, tmp_path):
    F, K, B, N = 26, 16, 512, 1200
    spec = synth.SynthSpec(n_fields=F, max_vocab=5000, seed=41, oov_frac=0.02)
    vocab_dir = str(tmp_path / "vocabulary") + "/"
    synth.write_vocabularies(spec, vocab_dir)
    path = str(tmp_path / "train.tfrecord")
    assert synth.write_tfrecord(spec, path, N, chunk=B) == N
    cats = [fc.categorical_column_with_vocabulary_file(n, vocab_dir + n + ".txt") for n in spec.names]
    label_cols = [fc.numeric_column("read_comment", default_value=0.0)]
    first = [fc.indicator_column(c) for c in cats]
    second = [fc.embedding_column(c, K) for c in cats]

    def example_parser(serialized):
        f = parse_example(serialized, fc.make_parse_example_spec(first + second + label_cols))
        y = f.pop("read_comment")
        return f, {"read_comment": y}
    params = {"first_order_feature_columns": first, "second_order_feature_columns": second,
              "hidden_units": ["512", "256", "128"], "dropout_rate": 0.0, "batch_norm": True, "learning_rate": 0.005}
    est = Estimator(deepfm_model_fn, params, RunConfig(device=dev, seed=11))

    batches = list(eval_input_fn(path, example_parser, B))
    assert [b[1]["read_comment"].shape[0] for b in batches] == [512, 512, 176]
    feats, labels = batches[0]
    assert isinstance(feats["userid"][0][0], bytes)                      # raw vocabulary keys reach the model
    est.build(feats, labels)
    dfeats, dlabels = est._to_device(feats, labels)

    # oracle on the ids the generator drew for chunk 0 (independent of the codec and the vocab lookup)
    ids, ylab, *_ = synth.make_id_batch(spec, B, 0)
    cf = {n: torch.from_numpy(ids[:, j].copy()) for j, n in enumerate(spec.names)}
    P = {k: v.detach().cpu().double().requires_grad_(True) for k, v in est.store.named_arrays().items()}
    ref_eval = M.deepfm(P, cf, {"read_comment": torch.from_numpy(ylab).double()}, params, training=False)
    ev = est._call_model_fn(dfeats, dlabels, ModeKeys.EVAL)
    assert_close(ev.loss, ref_eval["loss"], what="config[0] eval loss from TFRecord")
    ref = M.deepfm(P, cf, {"read_comment": torch.from_numpy(ylab).double()}, params, training=True)
    tr = est._call_model_fn(dfeats, dlabels, ModeKeys.TRAIN)
    assert_close(tr.loss, ref["loss"], what="config[0] train loss from TFRecord")
    assert_close(tr.predictions["probabilities"], ref["prob"], what="config[0] probabilities")

    # the Estimator drivers over the file: train (eager + hipGraph + eager for the partial batch), evaluate, predict
    est2 = Estimator(deepfm_model_fn, params, RunConfig(device=dev, seed=11))
    est2.train(lambda: train_input_fn(path, example_parser, B, num_epochs=2, shuffle_buffer_size=0), log_every=0)
    assert est2.global_step == 5        # repeat(2) THEN batch(512): 2400 records -> 4 full batches + one of 352 (utils.py:20-21)
    metrics = est2.evaluate(lambda: eval_input_fn(path, example_parser, B))
    assert set(metrics) >= {"eval_accuracy", "eval_auc", "loss", "global_step"} and 0.0 <= metrics["eval_auc"] <= 1.0
    preds = list(est2.predict(lambda: eval_input_fn(path, example_parser, B)))
    assert len(preds) == N and set(preds[0]) == {"probabilities", "fm_first_order_logit", "fm_second_order_logit", "deep_logit"}

    # ---- export + serving (deepfm.py:307-321: parsing receiver + BestExporter) ----------------
    import os
    from recalgorithm_amd import export as E
    from recalgorithm_amd.io import tfrecord
    recv = E.build_parsing_serving_input_receiver_fn(fc.make_parse_example_spec(first + second))     # features only
    exporter = E.BestExporter(name="best_exporter", serving_input_receiver_fn=recv, exports_to_keep=5)
    export_dir = exporter.export(est2, str(tmp_path / "model_dir" / "export" / "best_exporter"), None, metrics, True)
    assert export_dir and sorted(os.listdir(export_dir)) == ["serving.json", "variables.npz"]
    served = E.ServingModel(deepfm_model_fn, params, export_dir, device=dev)
    records = list(tfrecord.read_records(path))[:300]                          # raw serialized tf.train.Example protos
    out = served.predict(records)
    assert set(out) == set(preds[0]) and out["probabilities"].shape[0] == 300
    want = torch.tensor([float(p["probabilities"].reshape(-1)[0]) for p in preds[:300]])
    got = torch.from_numpy(out["probabilities"]).reshape(-1)
    assert torch.equal(got, want), "served probabilities differ from the trained estimator's on the same records"
    # the exported variables carry the reference's names and shapes: a fresh estimator loads them like a TF dump
    import numpy as np
    dumped = dict(np.load(os.path.join(export_dir, "variables.npz")))
    assert "fm_first_order/fm_first_order_dense/kernel" in dumped
    assert dumped["fm_first_order/fm_first_order_dense/kernel"].shape == (sum(spec.vocabs), 1)


@pytest.mark.gpu
def test_reader_batch_reaches_the_device_as_one_allocation(dev):
    """A batch of the native reader (io/native.py PackedBatch: the features are the columns of ONE [B, F] id matrix) with
    [B, 1] labels goes to the device in one staged copy and arrives as views of one allocation — id matrix first, labels
    behind it — with the values unchanged; anything else takes the general path."""
    from recalgorithm_amd.io.native import PackedBatch
    est = Estimator(deepfm_model_fn, {}, RunConfig(device="cuda"))
    B, F = 300, 5
    mat = torch.randint(-1, 1000, (B, F), dtype=torch.int64)
    keys = [f"c{j}" for j in range(F)]
    feats = PackedBatch({k: mat[:, j] for j, k in enumerate(keys)})
    feats.packed_ids = (mat, keys)
    labels = {"read_comment": torch.rand(B, 1), "like": torch.rand(B, 1)}
    f, l = est._to_device(feats, labels)
    torch.cuda.synchronize()
    assert all(f[k].is_cuda and torch.equal(f[k].cpu(), mat[:, j]) for j, k in enumerate(keys))
    assert all(l[k].is_cuda and l[k].shape == (B, 1) and torch.equal(l[k].cpu(), labels[k]) for k in labels)
    ptrs = {t.untyped_storage().data_ptr() for t in list(f.values()) + list(l.values())}
    assert len(ptrs) == 1
    assert f[keys[1]].data_ptr() == f[keys[0]].data_ptr() + 8 and l["read_comment"].data_ptr() == f[keys[0]].data_ptr() + B * F * 8
    # a batch with one more host feature is not that shape: general path, same values
    feats2 = PackedBatch(dict(feats))
    feats2["dense0"] = torch.rand(B, 1)
    feats2.packed_ids = (mat, keys)
    f2, l2 = est._to_device(feats2, labels)
    assert torch.equal(f2["c3"].cpu(), mat[:, 3]) and torch.equal(f2["dense0"].cpu(), feats2["dense0"])


@pytest.mark.gpu
def test_captured_step_loads_a_reader_batch_with_one_span_copy(dev):
    """GraphedTrainStep.load on DeviceBatch inputs (Estimator._to_device_one_copy): the static buffers end up holding exactly
    the new batch (ids and labels), through the one-span fast path — and a batch of another layout still takes the general one."""
    from recalgorithm_amd.estimator import GraphedTrainStep
    from recalgorithm_amd.io.native import PackedBatch
    est = Estimator(deepfm_model_fn, {}, RunConfig(device="cuda"))
    B, F = 256, 4
    keys = [f"c{j}" for j in range(F)]

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        mat = torch.randint(-1, 1000, (B, F), dtype=torch.int64, generator=g)
        feats = PackedBatch({k: mat[:, j] for j, k in enumerate(keys)})
        feats.packed_ids = (mat, keys)
        return mat, feats, {"y": torch.rand(B, 1, generator=g)}

    seen = []

    def step(f, l):
        seen.append((torch.stack([f[k] for k in keys], 1).clone(), l["y"].clone()))
        return l["y"].sum()

    m0, f0, l0 = batch(0)
    g = GraphedTrainStep(step, *est._to_device(f0, l0), warmup=1)
    assert g._span_plan is not None
    m1, f1, l1 = batch(1)
    d1 = est._to_device(f1, l1)
    g.load(*d1)
    torch.cuda.synchronize()
    assert torch.equal(torch.stack([g.static_f[k] for k in keys], 1).cpu(), m1) and torch.equal(g.static_l["y"].cpu(), l1["y"])
    out = g()
    torch.cuda.synchronize()
    assert abs(float(out) - float(l1["y"].sum())) < 1e-3
    # same values through the general path (plain dicts: no span)
    m2, f2, l2 = batch(2)
    d2 = est._to_device(f2, l2)
    g.load(dict(d2[0]), d2[1])
    torch.cuda.synchronize()
    assert torch.equal(torch.stack([g.static_f[k] for k in keys], 1).cpu(), m2) and torch.equal(g.static_l["y"].cpu(), l2["y"])


@pytest.mark.gpu
def test_feed_step_copies_a_host_batch_straight_into_the_captured_inputs(dev):
    """Estimator.feed_step: a reader batch of the captured layout goes from pinned staging into the graph's static input span
    (no device tensor in between) and the replay sees exactly that batch; more batches than the staging ring has slots keep
    their order; a batch of another size falls back to the general path (which refuses it: the captured shape is fixed)."""
    from recalgorithm_amd.estimator import GraphedTrainStep
    from recalgorithm_amd.io.native import PackedBatch
    est = Estimator(deepfm_model_fn, {}, RunConfig(device="cuda"))
    B, F = 256, 4
    keys = [f"c{j}" for j in range(F)]

    def batch(seed, n=B):
        g = torch.Generator().manual_seed(seed)
        mat = torch.randint(-1, 1000, (n, F), dtype=torch.int64, generator=g)
        feats = PackedBatch({k: mat[:, j] for j, k in enumerate(keys)})
        feats.packed_ids = (mat, keys)
        return mat, feats, {"y": torch.rand(n, 1, generator=g)}

    def step(f, l):
        return torch.stack([f[k] for k in keys], 1).to(torch.float32).sum() + l["y"].sum()

    m0, f0, l0 = batch(0)
    g = GraphedTrainStep(step, *est._to_device(f0, l0), warmup=1)
    assert g._span_plan is not None
    outs, want = [], []
    for s in range(1, 2 * est._STAGING_RING + 4):
        m, f, l = batch(s)
        outs.append(est.feed_step(g, f, l).clone())          # (g.out is overwritten by the next replay)
        want.append(float(m.to(torch.float64).sum() + l["y"].to(torch.float64).sum()))
    torch.cuda.synchronize()
    assert torch.equal(torch.stack([g.static_f[k] for k in keys], 1).cpu(), m) and torch.equal(g.static_l["y"].cpu(), l["y"])
    for o, w in zip(outs, want):
        assert abs(float(o) - w) <= 1e-6 * abs(w) + 1e-2
    m2, f2, l2 = batch(99, n=B - 7)
    with pytest.raises(ValueError):
        est.feed_step(g, f2, l2)
