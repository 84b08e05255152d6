"""Pieces every reference model script repeats (flags, WeChat feature columns, the Estimator
driver of `main`): shared here so that each algorithm/<MODEL>/<model>.py only holds what is
specific to its model.  Reference: the skeleton of /root/reference algorithm/DeepFM/deepfm.py
:14-41 (flags), algorithm/DCN/dcn.py:45-113 (columns), deepfm.py:276-343 (main)."""
from __future__ import annotations

import csv
import math
import os
import torch
from typing import Callable, Dict, Optional, Tuple

from .. import feature_column as fc
from .. import flags, ops
from ..variables import EmbeddingArena, current_store, variable_scope, zeros
from ..estimator import Estimator, EvalSpec, RunConfig, TrainSpec, train_and_evaluate
from .utils import eval_input_fn, parse_example, train_input_fn

FLAGS = flags.FLAGS

DENSE_FEATURES = [  # list order of dcn.py:59-76; fc.input_layer sorts by name
    "videoplayseconds", "u_read_comment_7d_sum", "u_like_7d_sum", "u_click_avatar_7d_sum",
    "u_forward_7d_sum", "u_comment_7d_sum", "u_follow_7d_sum", "u_favorite_7d_sum",
    "i_read_comment_7d_sum", "i_like_7d_sum", "i_click_avatar_7d_sum", "i_forward_7d_sum",
    "i_comment_7d_sum", "i_follow_7d_sum", "i_favorite_7d_sum", "c_user_author_read_comment_7d_sum",
]


def define_common_flags(batch_size=1024, learning_rate=0.005):
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
    flags.DEFINE_integer("batch_size", batch_size, "Training batch size")
    flags.DEFINE_float("learning_rate", learning_rate, "Learning rate")


def dense_columns() -> list:
    return [fc.numeric_column(k, default_value=0.0) for k in DENSE_FEATURES]


def label_columns() -> list:
    return [fc.numeric_column("read_comment", default_value=0.0)]


def vocab_column(key: str, fname: Optional[str] = None, sequence: bool = False):
    path = os.path.join(FLAGS.vocabulary_dir, (fname or key) + ".txt")
    if sequence:
        return fc.sequence_categorical_column_with_vocabulary_file(key, path)
    return fc.categorical_column_with_vocabulary_file(key, path)


def wechat_category_columns(dims: Dict[str, int], sequence_feed: bool = False):
    """The eight categorical embedding columns of dcn.py:84-107 / xdeepfm.py:85-112 /
    fibinet.py:89-116: six single-valued ids, the `manual_tag_list` bag and the table shared by
    `feedid` and `his_read_comment_7d_seq`.  Returns (category_columns_without_feedid, feedid_emb
    [feedid, his_seq]) — callers add `feedid_emb` as the reference does."""
    userid, device = vocab_column("userid"), vocab_column("device")
    authorid, bgm_song_id, bgm_singer_id = vocab_column("authorid"), vocab_column("bgm_song_id"), vocab_column("bgm_singer_id")
    manual_tag_list = vocab_column("manual_tag_list", "manual_tag_id")
    feedid = vocab_column("feedid", sequence=sequence_feed)
    his = vocab_column("his_read_comment_7d_seq", "feedid", sequence=sequence_feed)
    feedid_emb = fc.shared_embedding_columns([feedid, his], dims["feedid"], combiner="mean")
    cols = [
        fc.embedding_column(userid, dims["userid"]), fc.embedding_column(device, dims["device"]),
        fc.embedding_column(authorid, dims["authorid"]), fc.embedding_column(bgm_song_id, dims["bgm_song_id"]),
        fc.embedding_column(bgm_singer_id, dims["bgm_singer_id"]),
        fc.embedding_column(manual_tag_list, dims["manual_tag_list"], combiner="mean"),
    ]
    return cols, feedid_emb


def make_example_parser(columns_getter: Callable[[], Tuple[list, list]]):
    def example_parser(serialized_example):
        total, label = columns_getter()
        spec = fc.make_parse_example_spec(total + label)
        features = parse_example(serialized_example, spec)
        read_comment = features.pop("read_comment")
        return features, {"read_comment": read_comment}
    example_parser.columns_getter = columns_getter        # lets the input fns use the native decoder
    return example_parser


def write_predictions(estimator, example_parser, out_csv="predictions.csv"):
    """The predict tail of every script (deepfm.py:331-337): probabilities per test row; the
    label join with dataframe/test.csv is applied only when that file exists (quirk B-13)."""
    results = estimator.predict(input_fn=lambda: eval_input_fn(
        filepath=FLAGS.eval_data, example_parser=example_parser, batch_size=FLAGS.batch_size))
    labels = None
    test_csv = "../../dataset/wechat_algo_data1/dataframe/test.csv"
    if os.path.exists(test_csv):
        with open(test_csv) as f:
            labels = [row.get("read_comment") for row in csv.DictReader(f)]
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["", "probabilities"] + (["read_comment"] if labels else []))
        for i, r in enumerate(results):
            row = [i, float(r["probabilities"].reshape(-1)[0])]
            if labels and i < len(labels):
                row.append(labels[i])
            w.writerow(row)


def run_estimator(model_fn, params, example_parser):
    """main() body shared by the scripts: train_and_evaluate, evaluate, predict."""
    print(params)
    estimator = Estimator(model_fn=model_fn, params=params,
                          config=RunConfig(model_dir=FLAGS.model_dir,
                                           save_checkpoints_steps=FLAGS.save_checkpoints_steps))
    train_spec = TrainSpec(
        input_fn=lambda: train_input_fn(filepath=FLAGS.train_data, example_parser=example_parser,
                                        batch_size=FLAGS.batch_size, num_epochs=FLAGS.num_epochs,
                                        shuffle_buffer_size=FLAGS.shuffle_buffer_size),
        max_steps=FLAGS.train_steps)
    # deepfm.py:307-314: the parsing serving receiver over the model's feature columns + BestExporter (keeps 5)
    from ..export import BestExporter, build_parsing_serving_input_receiver_fn
    exporters = []
    getter = getattr(example_parser, "columns_getter", None)
    if getter is not None:
        total_feature_columns, _ = getter()
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
    return estimator


def fused_fm_sparse_part(features, params, first_scope: str, dense_name: str, who: str):
    """gather + first-order (indicator -> dense(1)) + FM second-order + concatenated embeddings through the
    fused kernel (`recalgo_deepfm_sparse_*`); shared by DeepFM (deepfm.py:179-204) and FwFM (fwfm.py:135-143).
    The (sum V, 1) kernel `<first_scope>/<dense_name>/kernel` is a width-1 arena mirroring the embedding arena's
    rows.  -> (embeddings [B, F*K], first_order_logit [B, 1], fm_second_order_logit [B, 1])."""
    store = current_store()
    first, second = params["first_order_feature_columns"], params["second_order_feature_columns"]
    keys = [c.key for c in second]
    K = second[0].dimension
    if sorted(c.key for c in first) != sorted(keys) or any(c.dimension != K for c in second) or K % 4 or K > 64:
        raise NotImplementedError(
            f"{who}: the fused sparse kernel needs the first- and second-order columns to "
            "cover the same categorical keys with one embedding width (multiple of 4, <= 64)")
    # second-order tables: one fc.input_layer call per column at top level (deepfm.py:187-190)
    tables = []
    for i, c in enumerate(second):
        layer = store.auto_name("input_layer")
        tables.append(fc._table_for(store, c, store.full_name(layer)))
    arena = tables[0][0]
    # first-order (sum V, 1) kernel as a width-1 arena with the same row layout
    w1_name = f"{first_scope}_w1"
    kprefix = f"{first_scope}/{dense_name}/kernel/"
    w1 = store.arenas.get(w1_name)
    if w1 is None:
        w1 = store.arenas[w1_name] = EmbeddingArena(w1_name, 1, store.device, seed=store.seed + 77)
    if w1.weight is None:
        total_v = sum(c.categorical_column.num_buckets for c in second)
        limit = math.sqrt(6.0 / (total_v + 1))              # glorot-uniform of the (sum V, 1) kernel
        for c in second:
            v = c.categorical_column.num_buckets
            init = (torch.rand(v, 1, generator=w1._gen) * 2 - 1) * limit
            w1.add_table(kprefix + c.key, v, init)
    with variable_scope(first_scope):
        with variable_scope(dense_name):
            bias = store.get_variable("bias", (1,), zeros)
    B = fc._batch_size(features, second[0])
    if store.building:
        z = torch.zeros(B, 1, device=store.device)
        return torch.zeros(B, len(second) * K, device=store.device), z, z
    if [n for n in w1.tables] != [kprefix + k for k in keys] or \
            [w1.tables[n][0] for n in w1.tables] != [arena.tables[t][0] for _, t in tables]:
        raise RuntimeError("first-order arena rows do not mirror the embedding arena")
    ids = [c.categorical_column.ids(features, store.device) for c in second]
    if not all(isinstance(i, torch.Tensor) for i in ids):
        raise NotImplementedError(f"{who}: multi-valued fields are not supported by the fused sparse kernel")
    rb = store.row_base_tensor(arena, [t for _, t in tables])
    return ops.deepfm_sparse(store, fc._as_matrix(ids), arena, w1, bias, rb)
