"""FFM entry point — MI355X drop-in for /root/reference algorithm/FFM/ffm.py (Juan et al., RecSys 2016): same flags,
`create_feature_columns`, `example_parser`, `ffm_model_fn(features, labels, mode, params)`, `main`, same variable names
(`ffm_first_order/fm_first_order_dense/{kernel,bias}`, `embedding_variables/<field>_embedding` of shape (F-1, V, K))
and prediction keys (`logit`, `probabilities`).

SURVEY.md §8f-3 sibling on the hot-path kernels:
  * first order (ffm.py:122-124): multi-hot indicator rows @ (sum V, 1) kernel = a width-1 gather (bag fields: the bag
    mean times the number of valid ids — an id that occurs twice counts twice, A-5);
  * field-aware lookups (ffm.py:146-157): field i owns F-1 sub-tables; ALL (field, sub-table) rows of a batch come from
    ONE launch of the gather kernel over F * (F-1) "virtual fields" (id + s * V_i inside the field's (F-1) * V_i-row
    table); a multi-valued field goes through `to_sparse_tensor` semantics (utils.py:49-64: the DISTINCT ids, mean);
  * sum over pairs i < j of <v_i[j-1], v_j[i]> (ffm.py:146-160): `recalgo_ffm_pairs_*`.
"""
from __future__ import annotations

import math
import os
from typing import Tuple

import torch

from ... import feature_column as fc
from ... import flags, ops
from ...feature_column import Ragged
from ...model_tail import finish_model_fn
from ...variables import EmbeddingArena, current_store, glorot_uniform, variable_scope, zeros
from .. import _common as common

# flags: /root/reference algorithm/FFM/ffm.py:17-40
common.define_common_flags(batch_size=1024, learning_rate=0.005)
flags.DEFINE_integer("embedding_dim", 8, "Embedding dimension")
FLAGS = flags.FLAGS

CATEGORICAL = ["userid", "feedid", "device", "authorid", "bgm_song_id", "bgm_singer_id"]


def create_feature_columns() -> Tuple[list, list]:
    """-> (one_hot_category_feature_columns, label_feature_columns); ffm.py:45-88."""
    cats = [fc.categorical_column_with_vocabulary_file(k, os.path.join(FLAGS.vocabulary_dir, k + ".txt")) for k in CATEGORICAL]
    cats.append(fc.categorical_column_with_vocabulary_file("manual_tag_list", os.path.join(FLAGS.vocabulary_dir, "manual_tag_id.txt")))
    return [fc.indicator_column(c) for c in cats], common.label_columns()


total_feature_columns: list = []
label_feature_columns: list = []
example_parser = common.make_example_parser(lambda: (total_feature_columns, label_feature_columns))


def _distinct_bags(x: Ragged) -> Ragged:
    """utils.py:49-64 `to_sparse_tensor` of a multi-hot row: its DISTINCT valid ids, ascending."""
    vals, offs = x.values, x.offsets
    B = offs.numel() - 1
    bag = torch.repeat_interleave(torch.arange(B, device=vals.device), offs[1:] - offs[:-1])
    ok = vals >= 0
    key = bag[ok] * (int(vals.max().item()) + 2 if vals.numel() else 1) + vals[ok]
    uniq = torch.unique(key)                                     # sorted
    mult = int(vals.max().item()) + 2 if vals.numel() else 1
    ub, uv = torch.div(uniq, mult, rounding_mode="floor"), uniq % mult
    counts = torch.zeros(B, dtype=torch.int64, device=vals.device).index_add_(0, ub, torch.ones_like(ub))
    new_offs = torch.cat([torch.zeros(1, dtype=torch.int64, device=vals.device), counts.cumsum(0)])
    return Ragged(uv.contiguous(), new_offs)


def ffm_model_fn(features, labels, mode, params):
    """ffm.py:106-211."""
    store = current_store()
    cols = params["one_hot_category_feature_columns"]
    F, K = len(cols), int(params["embedding_dim"])
    dev = store.device
    # ---- variables --------------------------------------------------------------------------------------------
    w1_name, kprefix = "ffm_first_order_w1", "ffm_first_order/fm_first_order_dense/kernel/"
    w1 = store.arenas.get(w1_name)
    if w1 is None:
        w1 = store.arenas[w1_name] = EmbeddingArena(w1_name, 1, dev, seed=store.seed + 77)
    if w1.weight is None and not w1.tables:
        total_v = sum(c.categorical_column.num_buckets for c in cols)
        limit = math.sqrt(6.0 / (total_v + 1))                 # glorot-uniform of the (sum V, 1) kernel
        for c in sorted(cols, key=lambda c: c.name):           # input_layer lays the indicator columns out by name
            v = c.categorical_column.num_buckets
            w1.add_table(kprefix + c.key, v, (torch.rand(v, 1, generator=w1._gen) * 2 - 1) * limit)
    with variable_scope("ffm_first_order"):
        with variable_scope("fm_first_order_dense"):
            bias = store.get_variable("bias", (1,), zeros)
    aname = f"emb{K}"
    arena = store.arenas.get(aname)
    if arena is None:
        arena = store.arenas[aname] = EmbeddingArena(aname, K, dev, seed=store.seed + 1)
    tnames = []
    for name, vocab in params["fields_vocabulary_size_tuple"]:
        tn = f"embedding_variables/{name}_embedding"
        tnames.append(tn)
        if arena.weight is None and tn not in arena.tables:     # tf.get_variable default: glorot-uniform over (F-1, V, K)
            arena.add_table(tn, (F - 1) * int(vocab), glorot_uniform((F - 1, int(vocab), K), arena._gen).reshape(-1, K),
                            view_shape=(F - 1, int(vocab), K))
    B = fc._batch_size(features, cols[0])
    if store.building:
        z = torch.zeros(B, 1, device=dev)
        return finish_model_fn(mode, z, labels, params, predictions=lambda prob: {"logit": z, "probabilities": prob})

    ids = [c.categorical_column.ids(features, dev) for c in cols]
    vocab = [int(v) for _, v in params["fields_vocabulary_size_tuple"]]
    single = [i for i, x in enumerate(ids) if not isinstance(x, Ragged)]
    # ---- first order ------------------------------------------------------------------------------------------
    first = None
    if single:                                                   # every single-valued column in ONE gather of (B, n) weights
        rb1 = store.row_base_tensor(w1, [kprefix + cols[i].key for i in single])
        id1 = torch.stack([ids[i] for i in single], 1).contiguous()                                  # (B, n): both lookups' ids
        first = ops.embedding_gather(store, id1, w1, rb1).sum(dim=1, keepdim=True)                   # (B, 1)
    for c, x in zip(cols, ids):
        if not isinstance(x, Ragged):
            continue
        tn = kprefix + c.key
        mean = ops.embedding_bag_mean(store, x.values, x.offsets, w1, tn)                            # (B, 1)
        lens = x.offsets[1:] - x.offsets[:-1]
        bag = torch.repeat_interleave(torch.arange(B, device=dev), lens)
        cnt = torch.zeros(B, device=dev).index_add_(0, bag, (x.values >= 0).float())
        term = mean * cnt.unsqueeze(1)                           # counts: a repeated id counts twice (A-5)
        first = term if first is None else first + term
    first = first + bias.data                                    # (the bias gradient: sum of d logit, below)
    # ---- field-aware lookups: X [B, F, F-1, K] ------------------------------------------------------------------
    if single:
        # view s of field i's (F-1, V, K) variable starts at arena row base_i + s * V_i: the F - 1 lookups of a field are the
        # SAME id against F - 1 row bases (OOV stays -1), so the id matrix is one expanding copy, not 5 launches per field
        key = (arena.name, "ffm_views", tuple(single))
        rbv = store._rb_cache.get(key)
        if rbv is None:
            rbv = store._rb_cache[key] = torch.tensor([arena.tables[tnames[i]][0] + s * vocab[i] for i in single for s in range(F - 1)],
                                                      dtype=torch.int64, device=dev)
        idv = id1.unsqueeze(2).expand(B, len(single), F - 1).reshape(B, -1)
        got = ops.embedding_gather(store, idv, arena, rbv)                                            # (B, n_single*(F-1)*K)
    if len(single) == F:
        X = got                                                  # all fields single-valued: the gather output IS X
    else:
        # runs of consecutive single-valued fields are one slice of the gather output each; a multi-valued field is F-1
        # bag-mean lookups over its distinct ids
        blocks, i, W = [], 0, (F - 1) * K
        while i < F:
            if isinstance(ids[i], Ragged):
                d = _distinct_bags(ids[i])
                blocks += [ops.embedding_bag_mean(store, torch.where(d.values >= 0, d.values + s * vocab[i], d.values),
                                                  d.offsets, arena, tnames[i]) for s in range(F - 1)]
                i += 1
            else:
                j = i
                while j < F and not isinstance(ids[j], Ragged):
                    j += 1
                a = single.index(i)
                blocks.append(got[:, a * W:(a + j - i) * W])
                i = j
        X = torch.cat(blocks, dim=1).contiguous()                # (B, F*(F-1)*K)
    second = ops.ffm_pairs(X, F, K)                              # (B, 1)
    total_logit = _BiasGrad.apply(first, bias) + second
    return finish_model_fn(mode, total_logit, labels, params,
                           predictions=lambda prob: {"logit": total_logit, "probabilities": prob})


class _BiasGrad(torch.autograd.Function):
    """identity on the logit that deposits d bias = sum_b d logit into the bias variable's gradient slot"""

    @staticmethod
    def forward(ctx, logit, bias):
        ctx.bias = bias
        return logit.view_as(logit)

    @staticmethod
    def backward(ctx, g):
        torch.sum(g, dim=0, out=ctx.bias.grad.view(1))
        return g, None


def main(unused_argv):
    global total_feature_columns, label_feature_columns
    cols, label_feature_columns = create_feature_columns()
    total_feature_columns = cols
    params = {"one_hot_category_feature_columns": cols, "embedding_dim": FLAGS.embedding_dim,
              "learning_rate": FLAGS.learning_rate,
              "fields_vocabulary_size_tuple": [(c.categorical_column.key, c.categorical_column.num_buckets) for c in cols]}
    common.run_estimator(ffm_model_fn, params, example_parser)


if __name__ == "__main__":
    flags.run(main)
