"""xDeepFM entry point — MI355X drop-in for /root/reference algorithm/xDeepFM/xdeepfm.py: same
flags, `create_feature_columns`, `example_parser`, `xdeepfm_model_fn(features, labels, mode,
params)`, `main`, scopes (`linear_part`, `cin_part`, `dnn_part`) and prediction keys (`logit`,
`probabilities`).  The CIN stack runs on the fp32 matrix cores (cin_layer.py).

    python -m recalgorithm_amd.algorithm.xDeepFM.xdeepfm --cin_layer_feature_maps=128,128
"""
from __future__ import annotations

from typing import Tuple

import torch

from ... import feature_column as fc
from ... import flags, nn
from ...model_tail import finish_model_fn
from ...variables import variable_scope
from .. import _common as common
from .cin_layer import cin_network

common.define_common_flags()
flags.DEFINE_string("hidden_units", "512,256,128", "Comma-separated list of number of units in each hidden layer of the dnn part")
flags.DEFINE_integer("embedding_dim", 8, "Embedding dimension")
flags.DEFINE_string("cin_layer_feature_maps", "50,50,50", "Comma-separated list of number of feature map in each CIN layer")
FLAGS = flags.FLAGS


def create_feature_columns() -> Tuple[list, list, list]:
    """-> (dense_feature_columns, category_feature_columns, label_feature_columns); every
    categorical column uses FLAGS.embedding_dim (xdeepfm.py:102-112)."""
    K = FLAGS.embedding_dim
    dims = {k: K for k in ("userid", "device", "authorid", "bgm_song_id", "bgm_singer_id", "manual_tag_list", "feedid")}
    cols, feedid_emb = common.wechat_category_columns(dims)
    return common.dense_columns(), cols + feedid_emb, common.label_columns()


total_feature_columns: list = []
label_feature_columns: list = []
example_parser = common.make_example_parser(lambda: (total_feature_columns, label_feature_columns))


def xdeepfm_model_fn(features, labels, mode, params):
    """xdeepfm.py:139-207."""
    with variable_scope("dense_input"):
        dense_cols = params.get("dense_feature_columns") or []
        dense_input = fc.input_layer(features, dense_cols) if dense_cols else None
    with variable_scope("category_input"):
        category_input = fc.input_layer(features, params["category_feature_columns"])   # (batch, m*D)

    with variable_scope("linear_part"):
        linear_vec = category_input if dense_input is None else torch.cat([dense_input, category_input], dim=-1)
        linear_logit = nn.dense(linear_vec, 1, activation=None, use_bias=True)

    with variable_scope("cin_part"):
        m = len(params["category_feature_columns"])
        D = int(params["embedding_dim"])
        x0 = category_input.reshape(-1, m, D)
        _, p_plus = cin_network(x0, params["cin_layer_feature_maps"])
        cin_logit = nn.dense(p_plus, 1, activation=None, use_bias=False)

    with variable_scope("dnn_part"):
        dnn_vec = linear_vec
        for i, unit in enumerate(params["hidden_units"]):
            dnn_vec = nn.dense(dnn_vec, unit, activation="relu", name=f"dense_{i}")
        dnn_logit = nn.dense(dnn_vec, 1, activation=None, use_bias=False)

    total_logit = linear_logit + cin_logit + dnn_logit
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
        "learning_rate": FLAGS.learning_rate,
        "embedding_dim": FLAGS.embedding_dim,
        "cin_layer_feature_maps": FLAGS.cin_layer_feature_maps.split(","),
    }
    common.run_estimator(xdeepfm_model_fn, params, example_parser)


if __name__ == "__main__":
    flags.run(main)
