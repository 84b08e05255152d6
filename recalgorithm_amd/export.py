"""Export + serving: the counterpart of the reference's BestExporter block (/root/reference
algorithm/DeepFM/deepfm.py:307-321; the same block in every model script):

    feature_spec = tf.feature_column.make_parse_example_spec(total_feature_columns)
    serving_input_receiver_fn = tf.estimator.export.build_parsing_serving_input_receiver_fn(feature_spec)
    exporters = [tf.estimator.BestExporter(name="best_exporter", serving_input_receiver_fn=..., exports_to_keep=5)]
    eval_spec = tf.estimator.EvalSpec(..., exporters=exporters)

TF writes a SavedModel (graph + variables) whose serving signature takes a batch of serialized tf.train.Example
protos.  Here a model is its model_fn + params (code) and its variables: an export is a directory

    <model_dir>/export/<exporter name>/<unix seconds>/variables.npz      {reference TF variable name: array}
                                                       serving.json       receiver feature spec, prediction keys, eval result

and `ServingModel(model_fn, params, export_dir)` is the serving side: `predict(serialized_examples)` parses the protos
with the exported feature spec (the parsing receiver) and runs the PREDICT graph on the HIP kernels.  The variable file
uses the reference's names and shapes (Estimator.export_variables), so it is interchangeable with a dump of a
reference-trained TF checkpoint (scripts/tf_ckpt_to_npz.py) — either loads into either.
"""
from __future__ import annotations

import json
import os
import shutil
import time
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np



def build_parsing_serving_input_receiver_fn(feature_spec: dict) -> Callable[[Sequence[bytes]], dict]:
    """tf.estimator.export.build_parsing_serving_input_receiver_fn: the returned receiver turns a batch of serialized
    tf.train.Example protos into the features dict the model_fn takes (tf.parse_example semantics)."""
    def receiver(serialized_examples: Sequence[bytes]) -> dict:
        from .algorithm.utils import parse_example
        return parse_example(list(serialized_examples), feature_spec)
    receiver.feature_spec = feature_spec
    return receiver


def _spec_to_json(feature_spec: dict) -> dict:
    out = {}
    for key, s in feature_spec.items():
        kind = s[0]
        if kind == "fixed":
            _, dtype, shape, default = s
            out[key] = {"kind": "fixed", "dtype": np.dtype(dtype).name, "shape": list(shape),
                        "default": None if default is None else float(default)}
        else:
            out[key] = {"kind": kind, "dtype": "string" if len(s) < 2 or s[1] in (bytes, str, np.bytes_, None) else np.dtype(s[1]).name}
    return out


def _spec_from_json(js: dict) -> dict:
    out = {}
    for key, s in js.items():
        if s["kind"] == "fixed":
            out[key] = ("fixed", np.dtype(s["dtype"]).type, tuple(s["shape"]), s["default"])
        else:
            out[key] = (s["kind"], np.bytes_ if s["dtype"] == "string" else np.dtype(s["dtype"]).type)
    return out


def export_model(estimator, export_dir_base: str, serving_input_receiver_fn, eval_result: Optional[dict] = None) -> str:
    """Estimator.export_saved_model: one timestamped directory under export_dir_base; returns its path."""
    os.makedirs(export_dir_base, exist_ok=True)
    stamp = int(time.time())
    while os.path.exists(os.path.join(export_dir_base, str(stamp))):       # TF also bumps a colliding timestamp
        stamp += 1
    final = os.path.join(export_dir_base, str(stamp))
    tmp = os.path.join(export_dir_base, f"temp-{stamp}")
    os.makedirs(tmp)
    np.savez(os.path.join(tmp, "variables.npz"), **estimator.export_variables())
    meta = {"feature_spec": _spec_to_json(getattr(serving_input_receiver_fn, "feature_spec", {})),
            "global_step": int(getattr(estimator, "global_step", 0)),
            "eval_result": {k: float(v) for k, v in (eval_result or {}).items()},
            "signature": "serialized tf.train.Example protos -> model_fn(mode=PREDICT).predictions"}
    with open(os.path.join(tmp, "serving.json"), "w") as f:
        json.dump(meta, f, indent=1)
    os.replace(tmp, final)                                                  # readers never see a half-written export
    return final


def _loss_smaller(best_eval_result: dict, current_eval_result: dict) -> bool:
    """tf.estimator.BestExporter's default compare_fn: a strictly smaller `loss` is better."""
    for r in (best_eval_result, current_eval_result):
        if not r or "loss" not in r:
            raise ValueError("BestExporter: the evaluation result has no 'loss'")
    return float(current_eval_result["loss"]) < float(best_eval_result["loss"])


class BestExporter:
    """tf.estimator.BestExporter: exports after an evaluation only when it is the best so far, keeps the newest
    `exports_to_keep` exports.  The best result is remembered across runs (TF re-reads it from the eval event files;
    here from the newest export's serving.json)."""

    def __init__(self, name: str = "best_exporter", serving_input_receiver_fn=None, exports_to_keep: Optional[int] = 5,
                 compare_fn: Callable[[dict, dict], bool] = _loss_smaller, **_ignored):
        if serving_input_receiver_fn is None:
            raise ValueError("BestExporter: serving_input_receiver_fn is required")
        if exports_to_keep is not None and exports_to_keep <= 0:
            raise ValueError("BestExporter: exports_to_keep must be positive or None")
        self.name, self.receiver, self.exports_to_keep, self.compare_fn = name, serving_input_receiver_fn, exports_to_keep, compare_fn
        self._best: Optional[dict] = None

    def _recover_best(self, export_path: str) -> None:
        if self._best is not None or not os.path.isdir(export_path):
            return
        for d in sorted(list_exports(export_path), key=_stamp, reverse=True):
            try:
                with open(os.path.join(d, "serving.json")) as f:
                    r = json.load(f).get("eval_result")
                if r and "loss" in r:
                    self._best = r
                    return
            except (OSError, ValueError):
                continue

    def export(self, estimator, export_path: str, checkpoint_path: Optional[str], eval_result: dict,
               is_the_final_export: bool = False) -> Optional[str]:
        self._recover_best(export_path)
        if self._best is not None and not self.compare_fn(self._best, eval_result):
            return None
        self._best = dict(eval_result)
        out = export_model(estimator, export_path, self.receiver, eval_result)
        if self.exports_to_keep is not None:
            for old in sorted(list_exports(export_path), key=_stamp)[:-self.exports_to_keep]:
                shutil.rmtree(old, ignore_errors=True)
        return out


def list_exports(export_path: str) -> List[str]:
    """The timestamped export directories under export_path (temp-* leftovers are not exports)."""
    if not os.path.isdir(export_path):
        return []
    return [os.path.join(export_path, d) for d in os.listdir(export_path)
            if d.isdigit() and os.path.isfile(os.path.join(export_path, d, "variables.npz"))]


def _stamp(path: str) -> int:
    return int(os.path.basename(path))


def latest_export(export_path: str) -> Optional[str]:
    ex = sorted(list_exports(export_path), key=_stamp)
    return ex[-1] if ex else None


class ServingModel:
    """The serving side of an export: model_fn + params (the code) + an export directory (the variables and the
    receiver's feature spec).  predict(serialized tf.train.Example protos) -> {prediction key: array [n, ...]}."""

    def __init__(self, model_fn, params: dict, export_dir: str, device=None):
        from .estimator import Estimator, RunConfig
        with open(os.path.join(export_dir, "serving.json")) as f:
            self.meta = json.load(f)
        self.feature_spec = _spec_from_json(self.meta["feature_spec"])
        self.receiver = build_parsing_serving_input_receiver_fn(self.feature_spec)
        self.estimator = Estimator(model_fn, params, RunConfig(device=device))
        self._variables = dict(np.load(os.path.join(export_dir, "variables.npz")))
        self._loaded = False

    def predict(self, serialized_examples: Sequence[bytes]) -> Dict[str, np.ndarray]:
        from .estimator import ModeKeys
        est = self.estimator
        features = self.receiver(serialized_examples)
        features, _ = est._to_device(features, None)
        if not self._loaded:
            est._build(features, None, ModeKeys.PREDICT)
            est.load_variables(self._variables)
            self._loaded, self._variables = True, None
        spec = est._call_model_fn(features, None, ModeKeys.PREDICT)
        return {k: v.detach().cpu().numpy() for k, v in spec.predictions.items()}
