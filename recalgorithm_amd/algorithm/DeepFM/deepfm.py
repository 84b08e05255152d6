"""DeepFM entry point — MI355X drop-in for /root/reference algorithm/DeepFM/deepfm.py: same
flags, `create_feature_columns`, `example_parser`, `deepfm_model_fn(features, labels, mode,
params)`, `main`, same prediction keys (`probabilities`, `fm_first_order_logit`,
`fm_second_order_logit`, `deep_logit`).

The whole sparse side of the model — the per-field embedding lookup, the FM first-order term
(indicator -> dense(1), deepfm.py:179-181), the FM second-order sum-square term (:184-200) and
the concat that feeds the MLP (:204) — is ONE hand-written HIP kernel
(`recalgo_deepfm_sparse_fwd/bwd`).  The `fm_first_order_dense` kernel of shape (sum V, 1) is
stored as a width-1 arena whose rows mirror the embedding arena, so a single id gathers both.
"""
from __future__ import annotations

import os
from typing import Tuple

from ... import feature_column as fc
from ... import flags, nn
from ...estimator import Estimator, EvalSpec, ModeKeys, RunConfig, TrainSpec, train_and_evaluate
from ...model_tail import finish_model_fn
from ...variables import variable_scope
from .. import _common as common
from ..utils import eval_input_fn, parse_example, train_input_fn

# flags: /root/reference algorithm/DeepFM/deepfm.py:14-41
flags.DEFINE_string("model_dir", "./model_dir", "Directory where model parameters, graph, etc are saved")
flags.DEFINE_string("output_dir", "./output_dir", "Directory where pb file are saved")
flags.DEFINE_string("train_data", "../../dataset/wechat_algo_data1/tfrecord/train.tfrecord", "Path to the train data")
flags.DEFINE_string("eval_data", "../../dataset/wechat_algo_data1/tfrecord/test.tfrecord", "Path to the evaluation data")
flags.DEFINE_string("vocabulary_dir", "../../dataset/wechat_algo_data1/vocabulary/", "Folder where the vocabulary file is stored")
flags.DEFINE_integer("num_epochs", 1, "Epoch of training phase")
flags.DEFINE_integer("train_steps", 10000, "Number of (global) training steps to perform")
flags.DEFINE_integer("shuffle_buffer_size", 10000, "Dataset shuffle buffer size")
flags.DEFINE_integer("num_parallel_readers", -1, "Number of parallel readers for training data")
flags.DEFINE_integer("save_checkpoints_steps", 1000, "Save checkpoints every this many steps")
flags.DEFINE_integer("batch_size", 1024, "Training batch size")
flags.DEFINE_float("learning_rate", 0.005, "Learning rate")
flags.DEFINE_integer("embedding_dim", 8, "Embedding dimension")
flags.DEFINE_string("hidden_units", "512,256,128", "Comma-separated list of number of units in each hidden layer")
flags.DEFINE_boolean("batch_norm", True, "Perform batch normalization (True or False)")
flags.DEFINE_float("dropout_rate", 0.1, "Dropout rate")
FLAGS = flags.FLAGS

CATEGORICAL = ["userid", "feedid", "device", "authorid", "bgm_song_id", "bgm_singer_id"]


def create_feature_columns() -> Tuple[list, list, list]:
    """-> (first_order_feature_columns, second_order_feature_columns, label_feature_columns)."""
    cats = [fc.categorical_column_with_vocabulary_file(k, os.path.join(FLAGS.vocabulary_dir, k + ".txt"))
            for k in CATEGORICAL]
    first = [fc.indicator_column(c) for c in cats]
    second = [fc.embedding_column(c, FLAGS.embedding_dim) for c in cats]
    label = [fc.numeric_column("read_comment", default_value=0.0)]
    return first, second, label


total_feature_columns: list = []
label_feature_columns: list = []


def example_parser(serialized_example):
    spec = fc.make_parse_example_spec(total_feature_columns + label_feature_columns)
    features = parse_example(serialized_example, spec)
    read_comment = features.pop("read_comment")
    return features, {"read_comment": read_comment}


example_parser.columns_getter = lambda: (total_feature_columns, label_feature_columns)   # native decoder hook


def deepfm_model_fn(features, labels, mode, params):
    """deepfm.py:165-273."""
    deep_input, fm_first_order_logit, fm_second_order_logit = common.fused_fm_sparse_part(
        features, params, "fm_first_order", "fm_first_order_dense", "deepfm_model_fn")
    training = mode == ModeKeys.TRAIN
    with variable_scope("fm_deep"):
        net = deep_input
        for unit in params["hidden_units"]:
            # dense(relu) -> [dropout] -> [batch_normalization], deepfm.py:207-211 (nn.dense_relu_dropout_bn: in a training step the
            # dropout rides in the dense layer's epilogue and the BatchNorm's backward)
            net = nn.dense_relu_dropout_bn(net, unit, params["dropout_rate"] if "dropout_rate" in params else None,
                                           bool(params["batch_norm"]), training)
        deep_logit = nn.dense(net, 1)
    # deepfm.py:214 (fm_first_order_logit + fm_second_order_logit + deep_logit).  Grouped from the right so that both FM terms
    # join the lazily evaluated head as addends of the fused logit / loss launch (no elementwise add launch of their own)
    total_logit = fm_first_order_logit + (fm_second_order_logit + deep_logit)
    return finish_model_fn(
        mode, total_logit, labels, params,
        predictions=lambda prob: {"probabilities": prob, "fm_first_order_logit": fm_first_order_logit,
                                  "fm_second_order_logit": fm_second_order_logit, "deep_logit": deep_logit})


def main(unused_argv):
    global total_feature_columns, label_feature_columns
    first, second, label_feature_columns = create_feature_columns()
    total_feature_columns = first + second
    params = {
        "first_order_feature_columns": first,
        "second_order_feature_columns": second,
        "hidden_units": FLAGS.hidden_units.split(","),
        "dropout_rate": FLAGS.dropout_rate,
        "batch_norm": FLAGS.batch_norm,
        "learning_rate": FLAGS.learning_rate,
    }
    print(params)
    print(FLAGS.embedding_dim, FLAGS.num_epochs)
    estimator = Estimator(model_fn=deepfm_model_fn, params=params,
                          config=RunConfig(model_dir=FLAGS.model_dir,
                                           save_checkpoints_steps=FLAGS.save_checkpoints_steps))
    train_spec = TrainSpec(
        input_fn=lambda: train_input_fn(filepath=FLAGS.train_data, example_parser=example_parser,
                                        batch_size=FLAGS.batch_size, num_epochs=FLAGS.num_epochs,
                                        shuffle_buffer_size=FLAGS.shuffle_buffer_size),
        max_steps=FLAGS.train_steps)
    # deepfm.py: feature_spec / build_parsing_serving_input_receiver_fn / BestExporter(exports_to_keep=5)
    from ...export import BestExporter, build_parsing_serving_input_receiver_fn
    feature_spec = fc.make_parse_example_spec(total_feature_columns)
    exporters = [BestExporter(name="best_exporter",
                              serving_input_receiver_fn=build_parsing_serving_input_receiver_fn(feature_spec),
                              exports_to_keep=5)]
    eval_spec = EvalSpec(
        input_fn=lambda: eval_input_fn(filepath=FLAGS.eval_data, example_parser=example_parser,
                                       batch_size=FLAGS.batch_size),
        throttle_secs=600, steps=None, exporters=exporters)
    train_and_evaluate(estimator, train_spec, eval_spec)
    metrics = estimator.evaluate(input_fn=lambda: eval_input_fn(
        filepath=FLAGS.eval_data, example_parser=example_parser, batch_size=FLAGS.batch_size))
    for key in sorted(metrics):
        print("%s: %s" % (key, metrics[key]))
    from ..DCN.dcn import write_predictions
    write_predictions(estimator, example_parser)
    print("after evaluate")


if __name__ == "__main__":
    flags.run(main)
