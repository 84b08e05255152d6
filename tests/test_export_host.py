"""Host logic of the export / serving path (recalgorithm_amd/export.py; the reference's BestExporter block,
/root/reference algorithm/DeepFM/deepfm.py:307-321): exporter bookkeeping, the parsing receiver, the feature-spec
round trip.  The numbers of a served model are checked on the GPU (tests/test_gpu_config1_tfrecord.py)."""
import json
import os
import types

import numpy as np
import pytest

from recalgorithm_amd import export as E
from recalgorithm_amd import feature_column as fc
from recalgorithm_amd.io import tfrecord


def _stub(step=0, value=1.0):
    return types.SimpleNamespace(global_step=step, export_variables=lambda: {"a/kernel": np.full((2, 3), value, np.float32)})


def _spec(tmp_path):
    vocab = tmp_path / "userid.txt"
    vocab.write_bytes(b"userid_0\nuserid_1\n")
    cols = [fc.numeric_column("videoplayseconds", default_value=0.0),
            fc.embedding_column(fc.categorical_column_with_vocabulary_file("userid", str(vocab)), 4)]
    return fc.make_parse_example_spec(cols)


def test_best_exporter_exports_only_improvements_and_keeps_n(tmp_path, monkeypatch):
    spec = _spec(tmp_path)
    recv = E.build_parsing_serving_input_receiver_fn(spec)
    ex = E.BestExporter(name="best_exporter", serving_input_receiver_fn=recv, exports_to_keep=2)
    base = str(tmp_path / "export" / ex.name)
    clock = [1_700_000_000]
    monkeypatch.setattr(E.time, "time", lambda: clock[0])
    p1 = ex.export(_stub(1, 1.0), base, None, {"loss": 0.5, "eval_auc": 0.6, "global_step": 1}, True)
    assert p1 and os.path.basename(p1) == "1700000000" and sorted(os.listdir(p1)) == ["serving.json", "variables.npz"]
    assert ex.export(_stub(2, 2.0), base, None, {"loss": 0.7}, True) is None            # worse: no export
    assert ex.export(_stub(2, 2.0), base, None, {"loss": 0.5}, True) is None            # equal: TF's compare is strict
    p2 = ex.export(_stub(3, 3.0), base, None, {"loss": 0.4}, True)                       # same second: timestamp bumped
    assert os.path.basename(p2) == "1700000001"
    clock[0] += 10
    p3 = ex.export(_stub(4, 4.0), base, None, {"loss": 0.3}, True)
    assert [os.path.basename(p) for p in sorted(E.list_exports(base))] == ["1700000001", "1700000010"]   # keeps 2
    assert E.latest_export(base) == p3
    assert float(np.load(os.path.join(p3, "variables.npz"))["a/kernel"][0, 0]) == 4.0
    meta = json.load(open(os.path.join(p3, "serving.json")))
    assert meta["global_step"] == 4 and meta["eval_result"]["loss"] == 0.3
    assert not [d for d in os.listdir(base) if d.startswith("temp-")]
    # a new process (new exporter object) remembers the best loss from the newest export
    ex2 = E.BestExporter(name="best_exporter", serving_input_receiver_fn=recv, exports_to_keep=2)
    assert ex2.export(_stub(5, 5.0), base, None, {"loss": 0.35}, True) is None
    assert ex2.export(_stub(6, 6.0), base, None, {"loss": 0.2}, True) is not None
    with pytest.raises(ValueError):
        ex2.export(_stub(), base, None, {"eval_auc": 0.9}, True)                          # no loss to compare
    with pytest.raises(ValueError):
        E.BestExporter(serving_input_receiver_fn=None)


def test_parsing_receiver_and_spec_round_trip(tmp_path):
    spec = _spec(tmp_path)
    assert set(spec) == {"videoplayseconds", "userid"}
    back = E._spec_from_json(json.loads(json.dumps(E._spec_to_json(spec))))
    assert back == spec
    recs = [tfrecord.encode_example({"userid": ("bytes", [b"userid_1"]), "videoplayseconds": ("float", [2.5])}),
            tfrecord.encode_example({"userid": ("bytes", [b"nobody"])})]                 # dense feature absent -> default
    f = E.build_parsing_serving_input_receiver_fn(back)(recs)
    assert f["userid"] == [[b"userid_1"], [b"nobody"]]
    assert f["videoplayseconds"].tolist() == [[2.5], [0.0]]


def test_train_and_evaluate_runs_the_eval_specs_exporters(tmp_path):
    """tf.estimator.train_and_evaluate hands the evaluation result to EvalSpec.exporters (deepfm.py:309-321):
    <model_dir>/export/<exporter name>/<timestamp>/."""
    from recalgorithm_amd.estimator import EvalSpec, TrainSpec, train_and_evaluate
    calls = []
    est = _stub(7, 7.0)
    est.config = types.SimpleNamespace(model_dir=str(tmp_path / "model_dir"))
    est._ckpt_path = lambda: os.path.join(est.config.model_dir, "model.ckpt.pt")
    est.train = lambda input_fn, max_steps=None: calls.append(("train", max_steps))
    est.evaluate = lambda input_fn, steps=None: {"loss": 0.25, "eval_auc": 0.7, "global_step": 7}
    recv = E.build_parsing_serving_input_receiver_fn(_spec(tmp_path))
    ex = E.BestExporter(name="best_exporter", serving_input_receiver_fn=recv, exports_to_keep=5)
    out = train_and_evaluate(est, TrainSpec(lambda: iter(()), max_steps=100), EvalSpec(lambda: iter(()), exporters=[ex]))
    assert out["loss"] == 0.25 and calls == [("train", 100)]
    base = os.path.join(est.config.model_dir, "export", "best_exporter")
    exports = E.list_exports(base)
    assert len(exports) == 1 and json.load(open(os.path.join(exports[0], "serving.json")))["eval_result"]["eval_auc"] == 0.7
    # a single exporter (not a list) is accepted like tf does; no exporters -> nothing written
    train_and_evaluate(est, TrainSpec(lambda: iter(())), EvalSpec(lambda: iter(()), exporters=ex))     # same loss: not better
    assert len(E.list_exports(base)) == 1
    train_and_evaluate(est, TrainSpec(lambda: iter(())), EvalSpec(lambda: iter(())))
