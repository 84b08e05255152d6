"""FiBiNET entry point — MI355X drop-in for /root/reference algorithm/FiBiNET/fibinet.py: same
flags, `create_feature_columns`, `example_parser`, `fibinet_model_fn(features, labels, mode,
params)`, `main`, scopes (`linear_part`, `senet_part`, `bilinear_interaction_part` with the
reference's variable names `orginal_w_*` [sic] / `senet_w_*`, `dnn_part`) and prediction keys
(`logit`, `probabilities`).

    python -m recalgorithm_amd.algorithm.FiBiNET.fibinet --embedding_dim=16 --reduction_ratio=2
"""
from __future__ import annotations

from typing import Tuple

from ... import feature_column as fc
from ... import flags, nn
from ...estimator import ModeKeys
from ...model_tail import finish_model_fn
from ...variables import variable_scope
from .. import _common as common
from .bilinear_interaction_layer import bilinear_interaction_pair
from .senet import senet

common.define_common_flags()
flags.DEFINE_string("hidden_units", "512,256,128", "Comma-separated list of number of units in each hidden layer of the deep part")
flags.DEFINE_boolean("batch_norm", True, "Perform batch normalization (True or False)")
flags.DEFINE_float("dropout_rate", 0.1, "Dropout rate")
flags.DEFINE_integer("embedding_dim", 8, "Embedding dimension")
flags.DEFINE_integer("reduction_ratio", 2, "Reduction ratio defined in SENET, must be greater than 1, smaller than number of category features")
flags.DEFINE_enum("bilinear_interaction_type", "all", ["all", "each", "interaction"], "Bilinear interaction type")
FLAGS = flags.FLAGS


def create_feature_columns() -> Tuple[list, list, list]:
    """-> (dense_feature_columns, category_feature_columns, label_feature_columns); every
    categorical column uses FLAGS.embedding_dim (fibinet.py:106-116)."""
    K = FLAGS.embedding_dim
    dims = {k: K for k in ("userid", "device", "authorid", "bgm_song_id", "bgm_singer_id", "manual_tag_list", "feedid")}
    cols, feedid_emb = common.wechat_category_columns(dims)
    return common.dense_columns(), cols + feedid_emb, common.label_columns()


total_feature_columns: list = []
label_feature_columns: list = []
example_parser = common.make_example_parser(lambda: (total_feature_columns, label_feature_columns))


def fibinet_model_fn(features, labels, mode, params):
    """fibinet.py:143-221."""
    with variable_scope("dense_input"):
        dense_cols = params.get("dense_feature_columns") or []
        dense_input = fc.input_layer(features, dense_cols) if dense_cols else None
    with variable_scope("category_input"):
        category_input = fc.input_layer(features, params["category_feature_columns"])     # (batch, F*K)
        # the reference reshapes with FLAGS.embedding_dim (fibinet.py:163); params carries the same value
        category_input = category_input.reshape(-1, len(params["category_feature_columns"]),
                                                int(params["embedding_dim"]))             # (batch, F, K)

    linear_logit = None
    if dense_input is not None:
        with variable_scope("linear_part"):
            linear_logit = nn.dense(dense_input, 1, activation=None, use_bias=True)       # fibinet.py:168

    with variable_scope("senet_part"):
        senet_output = senet(category_input, embedding_dim=params["embedding_dim"],
                             reduction_ratio=params["reduction_ratio"])

    with variable_scope("bilinear_interaction_part"):
        # bilinear(original) ++ bilinear(senet) on the last axis, then flatten (fibinet.py:177-187)
        bi_total = bilinear_interaction_pair(category_input, "orginal", senet_output, "senet",
                                             embedding_dim=params["embedding_dim"],
                                             type=params["bilinear_interaction_type"])
        bi_total = bi_total.reshape(bi_total.shape[0], -1)

    training = mode == ModeKeys.TRAIN
    with variable_scope("dnn_part"):
        net = bi_total
        for unit in params["hidden_units"]:
            # dense(relu) -> [dropout] -> [batch_normalization], fibinet.py:192-196 (nn.dense_relu_dropout_bn: in a training step the
            # dropout rides in the dense layer's epilogue and the BatchNorm's backward)
            net = nn.dense_relu_dropout_bn(net, unit, params["dropout_rate"] if "dropout_rate" in params else None,
                                           bool(params["batch_norm"]), training)
        fibinet_logit = nn.dense(net, 1)

    total_logit = fibinet_logit if linear_logit is None else linear_logit + fibinet_logit
    return finish_model_fn(mode, total_logit, labels, params,
                           predictions=lambda prob: {"logit": total_logit, "probabilities": prob})


def main(unused_argv):
    global total_feature_columns, label_feature_columns
    dense_cols, category_cols, label_feature_columns = create_feature_columns()
    total_feature_columns = dense_cols + category_cols
    params = {
        "category_feature_columns": category_cols,
        "dense_feature_columns": dense_cols,
        "hidden_units": FLAGS.hidden_units.split(","),
        "dropout_rate": FLAGS.dropout_rate,
        "batch_norm": FLAGS.batch_norm,
        "learning_rate": FLAGS.learning_rate,
        "embedding_dim": FLAGS.embedding_dim,
        "reduction_ratio": FLAGS.reduction_ratio,
        "bilinear_interaction_type": FLAGS.bilinear_interaction_type,
    }
    common.run_estimator(fibinet_model_fn, params, example_parser)


if __name__ == "__main__":
    flags.run(main)
