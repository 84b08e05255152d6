"""DCN (Deep & Cross Network) entry point — MI355X drop-in for /root/reference
algorithm/DCN/dcn.py: same flags, `create_feature_columns`, `example_parser`,
`dcn_model_fn(features, labels, mode, params)`, `main`, same variable scopes and prediction
keys (`logit`, `probabilities`).  The embedding gather, the fused CrossNet and the loss tail are
hand-written HIP kernels; the DNN branch is plain fp32 GEMMs.

    python -m recalgorithm_amd.algorithm.DCN.dcn --num_cross_layer=3 --batch_size=4096
"""
from __future__ import annotations

import os
from typing import Tuple

import torch

from ... import feature_column as fc
from ... import flags, nn, ops
from ...estimator import (Estimator, EvalSpec, ModeKeys, RunConfig, TrainSpec, train_and_evaluate)
from ...model_tail import finish_model_fn
from ...variables import variable_scope
from ..utils import eval_input_fn, parse_example, train_input_fn
from .cross_layer import cross_network

# flags: /root/reference algorithm/DCN/dcn.py:17-42
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
flags.DEFINE_string("hidden_units", "512,256,128", "Comma-separated list of number of units in each hidden layer")
flags.DEFINE_integer("num_cross_layer", 1, "Number of cross layers")
FLAGS = flags.FLAGS

DENSE_FEATURES = [  # dcn.py:59-76 (list order; input_layer sorts by name)
    "videoplayseconds", "u_read_comment_7d_sum", "u_like_7d_sum", "u_click_avatar_7d_sum",
    "u_forward_7d_sum", "u_comment_7d_sum", "u_follow_7d_sum", "u_favorite_7d_sum",
    "i_read_comment_7d_sum", "i_like_7d_sum", "i_click_avatar_7d_sum", "i_forward_7d_sum",
    "i_comment_7d_sum", "i_follow_7d_sum", "i_favorite_7d_sum", "c_user_author_read_comment_7d_sum",
]


def create_feature_columns() -> Tuple[list, list, list]:
    """-> (dense_feature_columns, category_feature_columns, label_feature_columns); dcn.py:45-113."""
    vd = FLAGS.vocabulary_dir
    dense_cols = [fc.numeric_column(k, default_value=0.0) for k in DENSE_FEATURES]

    def vocab(key, fname=None):
        return fc.categorical_column_with_vocabulary_file(key, os.path.join(vd, (fname or key) + ".txt"))

    userid, feedid, device = vocab("userid"), vocab("feedid"), vocab("device")
    authorid, bgm_song_id, bgm_singer_id = vocab("authorid"), vocab("bgm_song_id"), vocab("bgm_singer_id")
    manual_tag_list = vocab("manual_tag_list", "manual_tag_id")
    his_read_comment_7d_seq = vocab("his_read_comment_7d_seq", "feedid")

    feedid_emb = fc.shared_embedding_columns([feedid, his_read_comment_7d_seq], 16, combiner="mean")
    category_cols = [
        fc.embedding_column(userid, 16), fc.embedding_column(device, 2), fc.embedding_column(authorid, 4),
        fc.embedding_column(bgm_song_id, 4), fc.embedding_column(bgm_singer_id, 4),
        fc.embedding_column(manual_tag_list, 4, combiner="mean"),
    ] + feedid_emb
    label_cols = [fc.numeric_column("read_comment", default_value=0.0)]
    return dense_cols, category_cols, label_cols


total_feature_columns: list = []
label_feature_columns: list = []


def example_parser(serialized_example):
    """Batch of serialized tf.train.Example -> (features, {"read_comment": (B,1)}); dcn.py:116-131."""
    spec = fc.make_parse_example_spec(total_feature_columns + label_feature_columns)
    features = parse_example(serialized_example, spec)
    read_comment = features.pop("read_comment")
    return features, {"read_comment": read_comment}


def dcn_model_fn(features, labels, mode, params):
    """dcn.py:134-213."""
    with variable_scope("dense_input"):
        dense_cols = params.get("dense_feature_columns") or []
        dense_input = fc.input_layer(features, dense_cols) if dense_cols else None
    with variable_scope("category_input"):
        if dense_input is None and int(params["num_cross_layer"]) > 0:
            # the cross network is the first reader of the embeddings: its forward kernel does the gather (ops.gather_feeds_cross)
            with ops.gather_feeds_cross() as lz:
                category_input = fc.input_layer(features, params["category_feature_columns"])
                lz.keep(category_input)
        else:
            category_input = fc.input_layer(features, params["category_feature_columns"])
    concat_all = category_input if dense_input is None else torch.cat([dense_input, category_input], dim=-1)

    # concat_all feeds both branches: their two input gradients are summed inside cross_bwd (nn.GradJoin)
    join = nn.GradJoin() if int(params["num_cross_layer"]) > 0 and len(params["hidden_units"]) > 0 else None
    with variable_scope("cross_part"):
        cross_vec = cross_network(concat_all, params["num_cross_layer"], grad_join=join)
    ops.flush_lazy_gathers()                 # (nothing is pending once the cross network has run)

    with variable_scope("dnn_part"):
        dnn_vec = concat_all
        for i, unit in enumerate(params["hidden_units"]):
            dnn_vec = nn.dense(dnn_vec, unit, activation="relu", name=f"dnn_dense_{i}", grad_join=join if i == 0 else None,
                               last_hidden=(i == len(params["hidden_units"]) - 1))

    with variable_scope("output_part"):
        output = nn.concat([cross_vec, dnn_vec], axis=-1)       # read in place by the one-unit head (nn.LazyConcat)
        logit = nn.dense(output, 1, activation=None)

    return finish_model_fn(mode, logit, labels, params,
                           predictions=lambda prob: {"logit": logit, "probabilities": prob})


def main(unused_argv):
    global total_feature_columns, label_feature_columns
    dense_cols, category_cols, label_feature_columns = create_feature_columns()
    total_feature_columns = dense_cols + category_cols
    params = {
        "category_feature_columns": category_cols,
        "dense_feature_columns": dense_cols,
        "hidden_units": FLAGS.hidden_units.split(","),
        "num_cross_layer": FLAGS.num_cross_layer,
        "learning_rate": FLAGS.learning_rate,
    }
    print(params)
    estimator = Estimator(model_fn=dcn_model_fn, params=params,
                          config=RunConfig(model_dir=FLAGS.model_dir,
                                           save_checkpoints_steps=FLAGS.save_checkpoints_steps))
    train_spec = TrainSpec(
        input_fn=lambda: train_input_fn(filepath=FLAGS.train_data, example_parser=example_parser,
                                        batch_size=FLAGS.batch_size, num_epochs=FLAGS.num_epochs,
                                        shuffle_buffer_size=FLAGS.shuffle_buffer_size),
        max_steps=FLAGS.train_steps)
    # dcn.py: feature_spec / build_parsing_serving_input_receiver_fn / BestExporter(exports_to_keep=5)
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
    write_predictions(estimator, example_parser)


def write_predictions(estimator, example_parser, out_csv="predictions.csv"):
    """dcn.py predict tail: probabilities (+ labels when dataframe/test.csv exists; quirk B-13)."""
    import csv
    results = estimator.predict(input_fn=lambda: eval_input_fn(
        filepath=FLAGS.eval_data, example_parser=example_parser, batch_size=FLAGS.batch_size))
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["", "probabilities"])
        for i, r in enumerate(results):
            w.writerow([i, float(r["probabilities"].reshape(-1)[0])])


if __name__ == "__main__":
    flags.run(main)
