"""DIN (Deep Interest Network) entry point — MI355X drop-in for /root/reference
algorithm/DIN/din.py: same flags, `create_feature_columns` (5 lists), `example_parser`,
`din_model_fn(features, labels, mode, params)`, `main`, scopes (`target_input`, `his_seq_input`,
`attention_part`, `fcn`), prediction key `probabilities`, Dice / PReLU MLP and the
mini-batch-aware regulariser of din.py:254-257 (an L2 on the *activations* — quirk B-11).

    python -m recalgorithm_amd.algorithm.DIN.din --use_softmax=True --activation=dice
"""
from __future__ import annotations

from typing import Tuple

import torch

from ... import feature_column as fc
from ... import flags, nn
from ...estimator import ModeKeys
from ...model_tail import finish_model_fn
from ...variables import current_store, variable_scope
from .. import _common as common
from .activations import dice, prelu
from .din_attention import din_attention

common.define_common_flags()
flags.DEFINE_string("hidden_units", "512,256,128", "Comma-separated list of number of units in each hidden layer of the deep part")
flags.DEFINE_boolean("batch_norm", True, "Perform batch normalization (True or False)")
flags.DEFINE_float("dropout_rate", 0.1, "Dropout rate")
flags.DEFINE_string("activation", "dice", "Dense layer activation, supported strings are in {'prelu', dice'}")
flags.DEFINE_boolean("mini_batch_aware_regularization", True, "Whether to use mini_batch_aware_regularization")
flags.DEFINE_float("l2_lambda", 0.2, "Coefficient when using mini_batch_aware_regularization")
flags.DEFINE_boolean("use_softmax", False, "Whether to use softmax on attention score")
FLAGS = flags.FLAGS


def create_feature_columns() -> Tuple[list, list, list, list, list]:
    """-> (dense, category, target_feedid, sequence, label) feature columns; din.py:50-120.
    feedid / his_read_comment_7d_seq are *sequence* categorical columns sharing one 16-wide table;
    [0] is the target feed, [1] the history (shared_embedding_columns keeps input order)."""
    dims = {"userid": 16, "device": 2, "authorid": 4, "bgm_song_id": 4, "bgm_singer_id": 4,
            "manual_tag_list": 4, "feedid": 16}
    cols, feedid_emb = common.wechat_category_columns(dims, sequence_feed=True)
    return common.dense_columns(), cols, [feedid_emb[0]], [feedid_emb[1]], common.label_columns()


total_feature_columns: list = []
label_feature_columns: list = []
example_parser = common.make_example_parser(lambda: (total_feature_columns, label_feature_columns))


def din_model_fn(features, labels, mode, params):
    """din.py:186-290."""
    training = mode == ModeKeys.TRAIN
    parts = []
    with variable_scope("dense_input"):
        dense_cols = params.get("dense_feature_columns") or []
        if dense_cols:
            parts.append(fc.input_layer(features, dense_cols))
    # the three lookups are issued together: their `prepare` work (bucket counts, catch-up of lagging rows, sweep share) is one
    # launch per arena instead of one per lookup (sparse.batch_lookups); nothing in the block reads a lookup's output
    from recalgorithm_amd import sparse as _sparse
    with _sparse.batch_lookups():
        with variable_scope("category_input"):
            category_input = fc.input_layer(features, params["category_feature_columns"])
        with variable_scope("target_input"):
            target_input, _ = fc.sequence_input_layer(features, params["target_feedid_feature_columns"], max_length=1)
            target_input = target_input.squeeze(1)                                   # (B, H)
        with variable_scope("his_seq_input"):
            seq_input, seq_length = fc.sequence_input_layer(features, params["sequence_feature_columns"],
                                                            max_length=params.get("sequence_max_length"))
    # the target embedding feeds the attention (as the query) and the fcn input: the gradient block of the second is added to
    # d(query) inside the attention's backward kernel (nn.GradJoin) instead of by an accumulation launch
    target_join = nn.GradJoin()
    with variable_scope("attention_part"):
        attention_output = din_attention(target_input, seq_input, seq_length,
                                         is_softmax=params["use_softmax"], query_join=target_join)   # (B, H)
    # Mini-batch-aware regularisation (din.py:249-254): l2_lambda / 2 / B * sum(ev^2) over ev = [category,
    # target, attention output].  Without dense features ev IS concat_all, the input of the first fcn layer:
    # inside a seeded training step the term is then added to the loss as a value — the by-product of the kernel that
    # writes the concat (ops.concat_sumsq) — and its gradient (seed * l2_lambda / B * ev) is folded into that layer's
    # input-gradient GEMM, instead of a second concat, a square, a reduction and five gradient-accumulation launches.
    from recalgorithm_amd import ops as _ops
    use_mba = bool(params["mini_batch_aware_regularization"] and params["l2_lambda"] > 0)
    seed = _ops._loss_seed
    ev_parts = [category_input, target_input, attention_output]
    fused_mba = (use_mba and not parts and training and seed is not None and category_input.is_cuda
                 and all(t.dim() == 2 and t.dtype == torch.float32 for t in ev_parts) and not current_store().building)
    mba_coeff = params["l2_lambda"] / category_input.shape[0] if use_mba else 0.0
    mba_value = None
    if fused_mba:
        concat_all, mba_value = _ops.concat_sumsq(ev_parts, mba_coeff / 2, joins={1: target_join})
    else:
        concat_all = torch.cat(parts + ev_parts, dim=-1)

    with variable_scope("fcn"):
        net = concat_all
        for i, unit in enumerate(params["hidden_units"]):
            layer_index = i + 1
            # dense -> dice | prelu -> batch_normalization (din.py:262-266): one autograd node in a training step on the GPU
            net = nn.dense_activation_bn(net, unit, "dice" if params["activation"] == "dice" else "prelu", layer_index,
                                         bool(params["batch_norm"]), training,
                                         input_l2=(seed * mba_coeff if fused_mba and i == 0 else 0.0),
                                         # ... -> dropout (din.py:235-236): in the BatchNorm's store / its backward's loads
                                         dropout_rate=params["dropout_rate"] if "dropout_rate" in params else None)
        logit = nn.dense(net, 1)

    def mba_reg():
        if fused_mba:
            return mba_value.reshape(())
        if use_mba:
            ev = torch.cat([category_input, target_input, attention_output], dim=-1)
            return params["l2_lambda"] * (ev * ev).sum() / 2 / ev.shape[0]
        return None

    return finish_model_fn(mode, logit, labels, params, extra_loss=mba_reg)


def main(unused_argv):
    global total_feature_columns, label_feature_columns
    dense_cols, category_cols, target_cols, seq_cols, label_feature_columns = create_feature_columns()
    total_feature_columns = dense_cols + category_cols + target_cols + seq_cols
    params = {
        "dense_feature_columns": dense_cols,
        "category_feature_columns": category_cols,
        "sequence_feature_columns": seq_cols,
        "target_feedid_feature_columns": target_cols,
        "hidden_units": FLAGS.hidden_units.split(","),
        "dropout_rate": FLAGS.dropout_rate,
        "batch_norm": FLAGS.batch_norm,
        "learning_rate": FLAGS.learning_rate,
        "activation": FLAGS.activation,
        "mini_batch_aware_regularization": FLAGS.mini_batch_aware_regularization,
        "l2_lambda": FLAGS.l2_lambda,
        "use_softmax": FLAGS.use_softmax,
    }
    common.run_estimator(din_model_fn, params, example_parser)


if __name__ == "__main__":
    flags.run(main)
