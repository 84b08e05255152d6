"""FwFM entry point — MI355X drop-in for /root/reference algorithm/FwFM/fwfm.py (Pan et al., WWW 2018):
same flags, `create_feature_columns`, `example_parser`, `fwfm_model_fn(features, labels, mode, params)`,
`main`, same prediction keys (`probabilities`, `total_logit`) and TF variable names.

SURVEY.md §8f-3: a sibling of the six north_star models that needs no new kernel —
  * first order (indicator -> dense(1), fwfm.py:135-137) and the per-field embedding lookups
    (:140-143) are the fused DeepFM sparse kernel (`recalgo_deepfm_sparse_*`; its FM second-order
    output is simply not used);
  * second order  sum_{i<j} r_ij <e_i, e_j>  (:146-158, a Python double loop of F(F-1)/2 batch_dots in the
    reference) = the IPNN Gram features (`recalgo_pnn_features_*`) followed by the one-unit head
    (`recalgo_dense1_*`) over the pair strengths.
"""
from __future__ import annotations

import os
from typing import Tuple

from ... import feature_column as fc
from ... import flags, ops
from ...model_tail import finish_model_fn
from ...variables import current_store, glorot_uniform, variable_scope
from .. import _common as common

# flags: /root/reference algorithm/FwFM/fwfm.py:16-40
common.define_common_flags(batch_size=1024, learning_rate=0.005)
flags.DEFINE_integer("embedding_dim", 8, "Embedding dimension")
FLAGS = flags.FLAGS

CATEGORICAL = ["userid", "feedid", "device", "authorid", "bgm_song_id", "bgm_singer_id"]


def create_feature_columns() -> Tuple[list, list, list]:
    """-> (first_order_feature_columns, second_order_feature_columns, label_feature_columns); fwfm.py:45-105."""
    cats = [fc.categorical_column_with_vocabulary_file(k, os.path.join(FLAGS.vocabulary_dir, k + ".txt"))
            for k in CATEGORICAL]
    first = [fc.indicator_column(c) for c in cats]
    second = [fc.embedding_column(c, FLAGS.embedding_dim) for c in cats]
    return first, second, common.label_columns()


total_feature_columns: list = []
label_feature_columns: list = []
example_parser = common.make_example_parser(lambda: (total_feature_columns, label_feature_columns))


def fwfm_model_fn(features, labels, mode, params):
    """fwfm.py:123-214."""
    fields_flat, fwfm_first_order_logit, _unused_fm2 = common.fused_fm_sparse_part(
        features, params, "fwfm_first_order", "fwfm_first_order_dense", "fwfm_model_fn")

    F = len(params["second_order_feature_columns"])
    K = params["second_order_feature_columns"][0].dimension
    n = F * (F - 1) // 2
    with variable_scope("fields_pair_strength"):
        fields_pair_strength_weight = current_store().get_variable("fields_pair_strength_weight", (n,), glorot_uniform)

    fwfm_second_order_logit = ops.field_pair_logit(current_store(), fields_flat, fields_pair_strength_weight, F, K)
    total_logit = fwfm_first_order_logit + fwfm_second_order_logit
    return finish_model_fn(mode, total_logit, labels, params,
                           predictions=lambda prob: {"probabilities": prob, "total_logit": total_logit})


def main(unused_argv):
    global total_feature_columns, label_feature_columns
    first, second, label_feature_columns = create_feature_columns()
    total_feature_columns = first + second
    params = {
        "first_order_feature_columns": first,
        "second_order_feature_columns": second,
        "embedding_dim": FLAGS.embedding_dim,
        "learning_rate": FLAGS.learning_rate,
    }
    common.run_estimator(fwfm_model_fn, params, example_parser)


if __name__ == "__main__":
    flags.run(main)
