"""PNN entry point — MI355X drop-in for /root/reference algorithm/PNN/pnn.py: same flags,
`create_feature_columns`, `example_parser`, `pnn_model_fn(features, labels, mode, params)`,
`main`, scopes / variables (`linear_part/linear_w` (F*K, D), `product_part/inner_product_w`
(D, F) | `product_part/outer_product_w` (D, K, K), `bias` (D), `fcn/...`) and the prediction key
`probabilities`.

The reference builds lp with a Python loop of `output_dimension` (1024) sub-graphs
(pnn.py:152-158 / :167-172).  Here the product layer is: one HIP kernel for the per-example
second-order statistics, one for the batch-constant weight expansion, and one hipBLASLt GEMM
(`ops.pnn_product_layer`; recalgorithm_amd/csrc/pnn.hip).

    python -m recalgorithm_amd.algorithm.PNN.pnn --product_method=OPNN --embedding_dim=16
"""
from __future__ import annotations

from typing import Tuple

from ... import feature_column as fc
from ... import flags, nn, ops
from ...estimator import ModeKeys
from ...model_tail import finish_model_fn
from ...variables import current_store, variable_scope
from .. import _common as common

common.define_common_flags()
flags.DEFINE_integer("embedding_dim", 8, "Embedding dimension")
flags.DEFINE_string("hidden_units", "512,256,128", "Comma-separated list of number of units in each hidden layer of the deep part")
flags.DEFINE_boolean("batch_norm", True, "Perform batch normalization (True or False)")
flags.DEFINE_float("dropout_rate", 0.1, "Dropout rate")
flags.DEFINE_integer("output_dimension", 1024, "Output dimension of linear part and product part")
flags.DEFINE_string("product_method", "IPNN", "product_method, supported strings are in {'IPNN', 'OPNN'}")
flags.DEFINE_float("weight_regularizer", 0.0, "linear and product weight variable regularizer")
FLAGS = flags.FLAGS


def create_feature_columns() -> Tuple[list, list]:
    """-> (category_feature_columns, label_feature_columns), list order of pnn.py:83-85."""
    K = FLAGS.embedding_dim
    dims = {k: K for k in ("userid", "device", "authorid", "bgm_song_id", "bgm_singer_id", "manual_tag_list", "feedid")}
    cols, feedid_emb = common.wechat_category_columns(dims)
    return cols + feedid_emb, common.label_columns()


total_feature_columns: list = []
label_feature_columns: list = []
example_parser = common.make_example_parser(lambda: (total_feature_columns, label_feature_columns))


def pnn_model_fn(features, labels, mode, params):
    """pnn.py:112-250."""
    store = current_store()
    cols = params["category_feature_columns"]
    F = len(cols)
    # the reference sizes linear_w / reshapes with FLAGS.embedding_dim (pnn.py:135,143)
    K = int(params["embedding_dim"]) if "embedding_dim" in params else int(cols[0].dimension)
    D = int(params["output_dimension"])
    fields_embeddings = fc.input_layers_concat(features, cols)                     # (batch, F*K), pnn.py:126-130

    with variable_scope("linear_part"):
        linear_w = store.get_variable("linear_w", (F * K, D))
    with variable_scope("product_part"):
        if params["product_method"] == "IPNN":
            product_w = store.get_variable("inner_product_w", (D, F))
        else:  # OPNN
            product_w = store.get_variable("outer_product_w", (D, K, K))
    bias = store.get_variable("bias", (D,))                                        # glorot default (pnn.py:178)

    product_final = ops.pnn_product_layer(store, fields_embeddings.contiguous(), linear_w, product_w, bias,
                                          F, K, params["product_method"])          # relu(lz + lp + bias)

    training = mode == ModeKeys.TRAIN
    with variable_scope("fcn"):
        net = product_final
        for unit in params["hidden_units"]:
            # dense(relu) -> [dropout] -> [batch_normalization], pnn.py:187-191 (nn.dense_relu_dropout_bn: in a training step the
            # dropout rides in the dense layer's epilogue and the BatchNorm's backward)
            net = nn.dense_relu_dropout_bn(net, unit, params["dropout_rate"] if "dropout_rate" in params else None,
                                           bool(params["batch_norm"]), training)
        logit = nn.dense(net, 1)

    wr = float(params.get("weight_regularizer") or 0.0)

    def reg_loss():
        # tf.contrib.layers.l2_regularizer(scale)(w) = scale * sum(w^2) / 2 on linear_w and the
        # product weight, added to the loss (pnn.py:138,151,164,209-211)
        if wr <= 0.0:
            return None
        return nn.l2_regularization(wr, [linear_w, product_w])

    return finish_model_fn(mode, logit, labels, params, extra_loss=reg_loss)


def main(unused_argv):
    global total_feature_columns, label_feature_columns
    category_cols, label_feature_columns = create_feature_columns()
    total_feature_columns = category_cols
    params = {
        "category_feature_columns": total_feature_columns,
        "hidden_units": FLAGS.hidden_units.split(","),
        "dropout_rate": FLAGS.dropout_rate,
        "batch_norm": FLAGS.batch_norm,
        "learning_rate": FLAGS.learning_rate,
        "output_dimension": FLAGS.output_dimension,
        "product_method": FLAGS.product_method,
        "weight_regularizer": FLAGS.weight_regularizer,
        "embedding_dim": FLAGS.embedding_dim,
    }
    common.run_estimator(pnn_model_fn, params, example_parser)


if __name__ == "__main__":
    flags.run(main)
