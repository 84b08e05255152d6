"""NFM entry point — MI355X drop-in for /root/reference algorithm/NFM/nfm.py (He & Chua, SIGIR 2017): same flags,
`create_feature_columns`, `example_parser`, `nfm_model_fn(features, labels, mode, params)`, `main`, same variable scopes
(`dense_input/dense_logit`, `category_input`, `bi_interaction_part/bi_interaction_bn`, `dnn_part`) and prediction keys
(`logit`, `probabilities`).

SURVEY.md §8f-3 sibling: the per-field embedding lookups are the gather / bag-mean kernels, the bi-interaction pooling
0.5 * ((sum_f e_f)^2 - sum_f e_f^2) (nfm.py:155-167: the FM second-order term without the sum over k) is
`recalgo_bi_interaction_*`, the MLP runs on the fp32-MFMA dense kernels, the tail on the fused logit / loss kernel.
Quirks kept: the dropout after the pooling has the HARD-CODED rate 0.1 (nfm.py:170, independent of the dropout_rate
flag); `category_input = fc.input_layer(...)` (:150-151) creates a second, unused set of tables.
"""
from __future__ import annotations

from typing import Tuple

from ... import feature_column as fc
from ... import flags, nn, ops
from ...estimator import ModeKeys
from ...model_tail import finish_model_fn
from ...variables import variable_scope
from .. import _common as common

# flags: /root/reference algorithm/NFM/nfm.py:17-44
common.define_common_flags(batch_size=1024, learning_rate=0.005)
flags.DEFINE_integer("embedding_dim", 8, "Embedding dimension")
flags.DEFINE_string("hidden_units", "512,256,128", "Comma-separated list of number of units in each hidden layer")
flags.DEFINE_boolean("batch_norm", True, "Perform batch normalization (True or False)")
flags.DEFINE_float("dropout_rate", 0.1, "Dropout rate")
FLAGS = flags.FLAGS

CATEGORICAL = ["userid", "feedid", "device", "authorid", "bgm_song_id", "bgm_singer_id"]


def sibling_category_columns(embedding_dim: int) -> list:
    """The seven embedding columns of nfm.py:85-107 / afm.py:82-104: six single-valued ids + the `manual_tag_list` bag."""
    cols = [fc.embedding_column(common.vocab_column(k), embedding_dim) for k in CATEGORICAL]
    cols.append(fc.embedding_column(common.vocab_column("manual_tag_list", "manual_tag_id"), embedding_dim, combiner="mean"))
    return cols


def create_feature_columns() -> Tuple[list, list, list]:
    """-> (dense_feature_columns, category_feature_columns, label_feature_columns); nfm.py:47-113."""
    return common.dense_columns(), sibling_category_columns(FLAGS.embedding_dim), common.label_columns()


total_feature_columns: list = []
label_feature_columns: list = []
example_parser = common.make_example_parser(lambda: (total_feature_columns, label_feature_columns))


def nfm_model_fn(features, labels, mode, params):
    """nfm.py:133-229."""
    training = mode == ModeKeys.TRAIN
    cols = params["category_feature_columns"]
    F, K = len(cols), int(cols[0].dimension)
    # the lookups are issued together (sparse.batch_lookups: one `prepare` launch per arena, the forward kernels behind it);
    # same calls, same order, same variable names as the reference — only `dense_logit` is computed after the block
    from recalgorithm_amd import sparse as _sparse
    with _sparse.batch_lookups():
        with variable_scope("dense_input"):
            dense_input = fc.input_layer(features, params["dense_feature_columns"])
        with variable_scope("category_input"):
            fc.input_layer(features, params["category_feature_columns"])            # unused by the model (nfm.py:150-151)
        with variable_scope("bi_interaction_part"):
            fields = fc.input_layers_concat(features, cols)                           # one input_layer per column, list order
    with variable_scope("dense_input"):
        dense_logit = nn.dense(dense_input, 1, name="dense_logit")
    with variable_scope("bi_interaction_part"):
        x = fields.new_zeros(fields.shape[0], K) if _building() else ops.bi_interaction(fields.contiguous(), F, K)
        x = nn.batch_normalization(x, training=training, name="bi_interaction_bn")
        x = nn.dropout(x, 0.1, training=training)                                 # hard-coded rate (nfm.py:170)

    with variable_scope("dnn_part"):
        net = x
        for unit in params["hidden_units"]:
            net = nn.dense(net, unit, activation="relu", bn_stats=bool(params["batch_norm"]) and training)
            if params["batch_norm"]:
                net = nn.batch_normalization(net, training=training)
            if "dropout_rate" in params and 0.0 < params["dropout_rate"] < 1.0:
                net = nn.dropout(net, params["dropout_rate"], training=training)
        nfm_logit = nn.dense(net, 1)
    total_logit = dense_logit + nfm_logit
    return finish_model_fn(mode, total_logit, labels, params,
                           predictions=lambda prob: {"logit": total_logit, "probabilities": prob})


def _building() -> bool:
    from ...variables import current_store
    return current_store().building


def main(unused_argv):
    global total_feature_columns, label_feature_columns
    dense, cat, label_feature_columns = create_feature_columns()
    total_feature_columns = dense + cat
    params = {"dense_feature_columns": dense, "category_feature_columns": cat, "hidden_units": FLAGS.hidden_units.split(","),
              "dropout_rate": FLAGS.dropout_rate, "batch_norm": FLAGS.batch_norm, "learning_rate": FLAGS.learning_rate}
    common.run_estimator(nfm_model_fn, params, example_parser)


if __name__ == "__main__":
    flags.run(main)
