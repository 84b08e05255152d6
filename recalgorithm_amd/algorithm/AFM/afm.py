"""AFM entry point — MI355X drop-in for /root/reference algorithm/AFM/afm.py (Xiao et al., IJCAI 2017): same flags,
`create_feature_columns`, `example_parser`, `afm_model_fn(features, labels, mode, params)`, `main`, same variable names
(`attention_part/attention_{w,b,h}`, `prediction_score_part/p`) and prediction keys (`logit`, `probabilities`).

SURVEY.md §8f-3 sibling, built from the hot-path kernels plus one small one:
  * per-field embeddings: gather / bag-mean kernels (one input_layer per column, afm.py:156-159);
  * pair Hadamard products e_i * e_j, i < j over ALL F fields (afm.py:163-167) = the FiBiNET bilinear kernel with W = I
    on the fields plus one zero field (that kernel pairs `combinations(range(F' - 1), 2)`, quirk B-3);
  * attention MLP relu(pairs @ w + b) @ h (afm.py:181-183): the fp32-MFMA dense kernel on [B * P, K] and the one-unit head;
  * softmax over the pairs + weighted sum (afm.py:184-188): `recalgo_attention_pool_*`;
  * afm_logit = weighted_sum @ p and the loss tail: the fused logit / loss kernel.
Quirk kept: `category_input = fc.input_layer(...)` (:150-151) creates a second, unused set of tables.
"""
from __future__ import annotations

from typing import Tuple

import torch

from ... import feature_column as fc
from ... import flags, nn, ops
from ...model_tail import finish_model_fn
from ...variables import Variable, current_store, variable_scope
from .. import _common as common
from ..NFM.nfm import sibling_category_columns

# flags: /root/reference algorithm/AFM/afm.py:17-41
common.define_common_flags(batch_size=1024, learning_rate=0.005)
flags.DEFINE_integer("embedding_dim", 8, "Embedding dimension")
flags.DEFINE_integer("attention_factor", 128, "Hidden layer size of the attention network")
FLAGS = flags.FLAGS


def create_feature_columns() -> Tuple[list, list, list]:
    """-> (dense_feature_columns, category_feature_columns, label_feature_columns); afm.py:44-110."""
    return common.dense_columns(), sibling_category_columns(FLAGS.embedding_dim), common.label_columns()


total_feature_columns: list = []
label_feature_columns: list = []
example_parser = common.make_example_parser(lambda: (total_feature_columns, label_feature_columns))

_identity = {}


def _identity_weight(K: int, device) -> Variable:
    """W = I for the bilinear kernel (a constant: its 'gradient' buffer is scratch)."""
    key = (K, str(device))
    v = _identity.get(key)
    if v is None:
        v = _identity[key] = Variable("afm/identity", torch.eye(K, device=device))
    return v


def afm_model_fn(features, labels, mode, params):
    """afm.py:130-237."""
    store = current_store()
    with variable_scope("dense_input"):
        dense_input = fc.input_layer(features, params["dense_feature_columns"])
        dense_logit = nn.dense(dense_input, 1, name="dense_logit")
    with variable_scope("category_input"):
        fc.input_layer(features, params["category_feature_columns"])            # unused by the model (afm.py:150-151)

    cols = params["category_feature_columns"]
    F, K, t = len(cols), int(params["embedding_dim"]), int(params["attention_factor"])
    P = F * (F - 1) // 2
    with variable_scope("pair_interaction_part"):
        fields = fc.input_layers_concat(features, cols)                           # (batch, F*K), list order
    with variable_scope("attention_part"):
        w = store.get_variable("attention_w", (K, t))
        b = store.get_variable("attention_b", (t,))
        h = store.get_variable("attention_h", (t, 1))
    with variable_scope("prediction_score_part"):
        p = store.get_variable("p", (K, 1))
    B = fields.shape[0]
    if store.building:
        return finish_model_fn(mode, fields.new_zeros(B, 1), labels, params,
                               predictions=lambda prob: {"logit": fields.new_zeros(B, 1), "probabilities": prob})
    # pair Hadamard products over all F fields: the bilinear kernel pairs the first F' - 1 of F' fields -> one zero field
    padded = torch.cat([fields, fields.new_zeros(B, K)], dim=1).reshape(B, F + 1, K)
    pairs = ops.bilinear_interaction(store, "all", padded.contiguous(), _identity_weight(K, fields.device))   # (B, P, K)
    att = nn.dense_with(pairs.reshape(B * P, K), w, b, relu=True)                # relu(pairs @ w + b)   (B*P, t)
    att = nn.dense_with(att, h, None, relu=False)                                 # @ h                   (B*P, 1)
    weighted = ops.attention_pool(pairs, att.reshape(B, P))                       # softmax over pairs, weighted sum (B, K)
    afm_logit = nn.dense_with(weighted, p, None, relu=False)                      # (B, 1)
    total_logit = dense_logit + afm_logit
    return finish_model_fn(mode, total_logit, labels, params,
                           predictions=lambda prob: {"logit": total_logit, "probabilities": prob})


def main(unused_argv):
    global total_feature_columns, label_feature_columns
    dense, cat, label_feature_columns = create_feature_columns()
    total_feature_columns = dense + cat
    params = {"dense_feature_columns": dense, "category_feature_columns": cat, "embedding_dim": FLAGS.embedding_dim,
              "attention_factor": FLAGS.attention_factor, "learning_rate": FLAGS.learning_rate}
    common.run_estimator(afm_model_fn, params, example_parser)


if __name__ == "__main__":
    flags.run(main)
