"""oracle/ref_models.py — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Model-level restatement of the six north_star model_fns (+ the FwFM sibling) (forward to logits / loss), composed
from oracle/ref_ops.py, op-for-op in the reference's order.  Parameters come in as a dict
`P` name -> tensor using the reference's TF variable names (scope/.../kernel etc.); gradients
are obtained by torch.autograd on the returned loss.  Columns are duck-typed: objects with
`.name`, `.key`, `.dimension` (embedding), `.shared_name` (shared tables).

Feature batch format: feats[key] is a LongTensor [B] (single-valued, -1 = OOV), a tuple
(values, offsets) (multi-valued), or a float tensor [B,1] (numeric).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

from . import ref_ops as R


def _sorted(cols):
    return sorted(cols, key=lambda c: c.name)     # [TF-ext A-1] input_layer sorts by name


def table_name(col, scope_layer: str) -> str:
    owner = col.shared_name if getattr(col, "shared_name", None) else col.name
    return f"{scope_layer}/{owner}/embedding_weights"


def _lookup(P, feats, col, scope_layer, shared_registry):
    if getattr(col, "shared_name", None):
        # [TF-ext A-4] the shared table is created once, by the first layer that uses it
        tn = shared_registry.setdefault(col.shared_name, table_name(col, scope_layer))
    else:
        tn = table_name(col, scope_layer)
    table = P[tn]
    ids = feats[col.key]
    if isinstance(ids, tuple):
        return R.embedding_lookup_mean(ids[0], ids[1], table)
    return R.embedding_lookup_single(ids, table)


def input_layer(P, feats, cols, scope_layer: str, shared_registry=None) -> torch.Tensor:
    shared_registry = {} if shared_registry is None else shared_registry
    parts = []
    for c in _sorted(cols):
        if hasattr(c, "dimension"):
            parts.append(_lookup(P, feats, c, scope_layer, shared_registry))
        else:                                      # numeric column
            parts.append(feats[c.key].reshape(feats[c.key].shape[0], -1))
    return torch.cat(parts, dim=1)


def _mlp(P, x, scope, names, relu=True):
    for n in names:
        x = R.dense(x, P[f"{scope}/{n}/kernel"], P.get(f"{scope}/{n}/bias"), relu=relu)
    return x


def _dropout(net, params, training, masks):
    """tf.layers.dropout(net, params["dropout_rate"], training=...) as the model scripts guard it (deepfm.py:208-209,
    din.py:235-236, fibinet.py:193-194, pnn.py:188-189): keep mask * 1 / (1 - rate) in TRAIN mode, identity otherwise.
    TF's random stream cannot be reproduced: `masks` is the list of keep masks in call order (the golden's
    aux/dropout_mask_<i>), consumed from the front."""
    rate = float(params.get("dropout_rate") or 0.0)
    if not (training and 0.0 < rate < 1.0):
        return net
    if not masks:
        raise ValueError("training-mode dropout needs its keep masks (dropout_masks=[...], call order)")
    return net * masks.pop(0).to(net.dtype) / (1.0 - rate)


def _tail(logit, labels, extra=None):
    out = {"logit": logit, "prob": torch.sigmoid(logit)}
    if labels is not None:
        loss = R.ce_loss(labels, logit)
        if extra is not None:
            loss = loss + extra
        out["loss"] = loss
    return out


# --------------------------------------------------------------------------------------------
def dcn(P, feats, labels, params, training=False):
    """algorithm/DCN/dcn.py:134-191."""
    dense_cols = params.get("dense_feature_columns") or []
    reg = {}
    cat = input_layer(P, feats, params["category_feature_columns"], "category_input/input_layer", reg)
    if dense_cols:
        dense_in = input_layer(P, feats, dense_cols, "dense_input/input_layer")
        concat_all = torch.cat([dense_in, cat], dim=-1)                           # :155
    else:
        concat_all = cat
    L = int(params["num_cross_layer"])
    cross = R.cross_stack(concat_all, [P[f"cross_part/wl_{i}"] for i in range(L)],
                          [P[f"cross_part/bl_{i}"] for i in range(L)])            # :157-160
    dnn = _mlp(P, concat_all, "dnn_part", [f"dnn_dense_{i}" for i in range(len(params["hidden_units"]))])
    output = torch.cat([cross, dnn], dim=-1)                                      # :168
    logit = R.dense(output, P["output_part/dense/kernel"], P["output_part/dense/bias"])
    return _tail(logit, None if labels is None else labels["read_comment"])


def deepfm(P, feats, labels, params, training=False, bn_state=None, dropout_masks=None):
    """algorithm/DeepFM/deepfm.py:165-235.  MLP order dense(relu) -> dropout -> BN (:207-211); batch norm uses batch
    statistics when `training`; training-mode dropout takes its keep masks from `dropout_masks` (call order)."""
    masks = list(dropout_masks or [])
    first_cols = _sorted(params["first_order_feature_columns"])
    # fm_first_order: (B, sum V) multi-hot @ kernel + bias == sum of per-column weight lookups
    w1 = [P[f"fm_first_order/fm_first_order_dense/kernel/{c.key}"] for c in first_cols]
    fm1 = R.indicator_first_order([feats[c.key] for c in first_cols], w1,
                                  P["fm_first_order/fm_first_order_dense/bias"][0])
    fields = []
    for i, c in enumerate(params["second_order_feature_columns"]):                 # :187-190 list order
        layer = "input_layer" if i == 0 else f"input_layer_{i}"
        fields.append(_lookup(P, feats, c, layer, {}))
    fm2 = R.fm_second_order(fields)                                                # :192-200
    net = torch.cat(fields, dim=1)                                                 # :204
    for i, _ in enumerate(params["hidden_units"]):
        dn = "dense" if i == 0 else f"dense_{i}"
        net = R.dense(net, P[f"fm_deep/{dn}/kernel"], P[f"fm_deep/{dn}/bias"], relu=True)
        net = _dropout(net, params, training, masks)                               # :208-209
        if params.get("batch_norm"):
            bn = "batch_normalization" if i == 0 else f"batch_normalization_{i}"
            net = R.batch_norm(net, P[f"fm_deep/{bn}/gamma"], P[f"fm_deep/{bn}/beta"],
                               P[f"fm_deep/{bn}/moving_mean"], P[f"fm_deep/{bn}/moving_variance"], training)
    n = len(params["hidden_units"])
    dn = "dense" if n == 0 else f"dense_{n}"
    deep = R.dense(net, P[f"fm_deep/{dn}/kernel"], P[f"fm_deep/{dn}/bias"])
    logit = fm1 + fm2 + deep                                                       # :214
    out = _tail(logit, None if labels is None else labels["read_comment"])
    out.update(fm_first_order_logit=fm1, fm_second_order_logit=fm2, deep_logit=deep)
    return out


def fwfm(P, feats, labels, params, training=False):
    """algorithm/FwFM/fwfm.py:123-161 (SURVEY.md §8f-3 sibling).  First order as in DeepFM; second order
    the reference's double loop: sum over i < j of r[index_from_upper_triangular(i, j, F)] * <e_i, e_j>
    (utils.py:67-82: row-major strict upper triangle), accumulated in loop order."""
    first_cols = _sorted(params["first_order_feature_columns"])
    w1 = [P[f"fwfm_first_order/fwfm_first_order_dense/kernel/{c.key}"] for c in first_cols]
    first = R.indicator_first_order([feats[c.key] for c in first_cols], w1,
                                    P["fwfm_first_order/fwfm_first_order_dense/bias"][0])         # :135-137
    fields = []
    for i, c in enumerate(params["second_order_feature_columns"]):                               # :140-143 list order
        layer = "input_layer" if i == 0 else f"input_layer_{i}"
        fields.append(_lookup(P, feats, c, layer, {}))
    F = len(fields)
    r = P["fields_pair_strength/fields_pair_strength_weight"]                                     # :146-149
    second = torch.zeros(fields[0].shape[0], 1, dtype=fields[0].dtype)
    index = 0
    for i in range(F - 1):                                                                        # :152-158
        for j in range(i + 1, F):
            second = second + r[index] * (fields[i] * fields[j]).sum(dim=1, keepdim=True)
            index += 1
    logit = first + second                                                                        # :160
    out = _tail(logit, None if labels is None else labels["read_comment"])
    out.update(fwfm_first_order_logit=first, fwfm_second_order_logit=second)
    return out


def afm(P, feats, labels, params, training=False):
    """algorithm/AFM/afm.py:143-192 (SURVEY.md §8f-3 sibling).  Quirk: the model
    also builds `category_input = fc.input_layer(...)` (:150-151) — its tables exist as variables and get zero
    gradients — but the pair interactions use a second set of tables, one `input_layer` call per column under
    `pair_interaction_part` (:156-159)."""
    dense_in = input_layer(P, feats, params["dense_feature_columns"], "dense_input/input_layer")
    dense_logit = R.dense(dense_in, P["dense_input/dense_logit/kernel"], P["dense_input/dense_logit/bias"])  # :145-147
    fields = []
    for i, c in enumerate(params["category_feature_columns"]):                                         # :156-159
        layer = "pair_interaction_part/input_layer" + ("" if i == 0 else f"_{i}")
        fields.append(_lookup(P, feats, c, layer, {}))
    F = len(fields)
    pairs = torch.stack([fields[i] * fields[j] for i in range(F) for j in range(i + 1, F)], dim=1)      # :163-167 (B, P, K)
    w, b, h = P["attention_part/attention_w"], P["attention_part/attention_b"], P["attention_part/attention_h"]
    att = torch.relu(pairs @ w + b) @ h                                                                 # :181-183 (B, P, 1)
    score = torch.softmax(att, dim=1)                                                                   # :184
    weighted = (pairs * score).sum(dim=1)                                                               # :187-188 (B, K)
    afm_logit = weighted @ P["prediction_score_part/p"]                                                 # :189-190
    logit = dense_logit + afm_logit                                                                     # :192
    return _tail(logit, None if labels is None else labels["read_comment"])


def _bags(ids):
    """feature batch entry -> list (len B) of lists of valid ids, in input order."""
    if isinstance(ids, tuple):
        vals, offs = ids
        return [[int(v) for v in vals[int(offs[b]):int(offs[b + 1])] if int(v) >= 0] for b in range(len(offs) - 1)]
    return [[int(v)] if int(v) >= 0 else [] for v in ids]


def ffm(P, feats, labels, params, training=False):
    """algorithm/FFM/ffm.py:118-163 (SURVEY.md §8f-3 sibling).
    First order: multi-hot indicator rows @ (sum V, 1) kernel + bias — an id that occurs twice in a bag counts
    twice (A-5).  Second order: field i owns F-1 tables `<name>_embedding[(F-1), V_i, K]`; for a pair i < j the
    reference looks field i up in its sub-table j-1 and field j in its sub-table i (:150-157) through
    `to_sparse_tensor` (utils.py:49-64: the coordinates of the non-zero multi-hot entries, i.e. the DISTINCT
    ids in ascending order) + safe_embedding_lookup_sparse (mean), and sums <v_i, v_j> over the pairs."""
    cols = params["one_hot_category_feature_columns"]
    bags = {c.key: _bags(feats[c.key]) for c in cols}
    B = len(next(iter(bags.values())))
    kk = "ffm_first_order/fm_first_order_dense/kernel"      # whole (sum V, 1) kernel, or one slice per column (A-1)
    kern = (P[kk] if kk in P else torch.cat([P[f"{kk}/{c.key}"] for c in _sorted(cols)])).reshape(-1)
    first = torch.zeros(B, dtype=kern.dtype)
    row0 = 0
    for c in _sorted(cols):                                            # :118-119 input_layer: sorted by column name
        V = c.categorical_column.num_buckets
        for b in range(B):
            for v in bags[c.key][b]:
                first[b] = first[b] + kern[row0 + v]
        row0 += V
    first = (first + P["ffm_first_order/fm_first_order_dense/bias"][0]).unsqueeze(-1)              # :120
    F = len(cols)
    tables = [P[f"embedding_variables/{name}_embedding"] for name, _ in params["fields_vocabulary_size_tuple"]]  # :125-130

    def lookup(table2d, key):
        rows = []
        for b in range(B):
            ids = sorted(set(bags[key][b]))
            if ids:
                acc = table2d[ids[0]]
                for v in ids[1:]:
                    acc = acc + table2d[v]
                rows.append(acc / len(ids))
            else:
                rows.append(torch.zeros(table2d.shape[1], dtype=table2d.dtype))
        return torch.stack(rows, 0)

    second = torch.zeros(B, 1, dtype=kern.dtype)
    for i in range(F - 1):                                             # :146-160
        for j in range(i + 1, F):
            vi = lookup(tables[i][j - 1], cols[i].key)
            vj = lookup(tables[j][i], cols[j].key)
            second = second + (vi * vj).sum(dim=-1, keepdim=True)
    logit = first + second                                            # :163
    return _tail(logit, None if labels is None else labels["read_comment"])


def nfm(P, feats, labels, params, training=False, dropout_masks=None):
    """algorithm/NFM/nfm.py:143-184 (SURVEY.md §8f-3 sibling).  Bi-interaction pooling
    0.5 * ((sum_f e_f)^2 - sum_f e_f^2) -> BatchNorm `bi_interaction_bn` -> dropout with the HARD-CODED rate 0.1
    (:170, independent of the dropout_rate flag) -> MLP (dense(relu) -> BN -> dropout(rate flag)) -> dense(1).
    In training mode `dropout_masks[0]` is the keep mask of that dropout (TF's random stream cannot be
    reproduced; the golden records the mask it used).  Like AFM, the model also creates an unused
    `category_input` set of tables (:150-151)."""
    dense_in = input_layer(P, feats, params["dense_feature_columns"], "dense_input/input_layer")
    dense_logit = R.dense(dense_in, P["dense_input/dense_logit/kernel"], P["dense_input/dense_logit/bias"])      # :145-147
    fields = []
    for i, c in enumerate(params["category_feature_columns"]):                                                  # :158-161
        layer = "bi_interaction_part/input_layer" + ("" if i == 0 else f"_{i}")
        fields.append(_lookup(P, feats, c, layer, {}))
    s_, q_ = fields[0], fields[0] ** 2
    for e in fields[1:]:                                                                                        # add_n: in list order
        s_, q_ = s_ + e, q_ + e ** 2
    x = 0.5 * (s_ ** 2 - q_)                                                                                    # :163-167
    b = "bi_interaction_part/bi_interaction_bn"
    x = R.batch_norm(x, P[f"{b}/gamma"], P[f"{b}/beta"], P[f"{b}/moving_mean"], P[f"{b}/moving_variance"], training)  # :168
    masks = list(dropout_masks or [])
    if training:                                                                                                # :170
        if not masks:
            raise ValueError("nfm(training=True) needs the keep mask of the hard-coded dropout (rate 0.1)")
        x = x * masks.pop(0).to(x.dtype) / 0.9
    net = x
    for i, _ in enumerate(params["hidden_units"]):                                                              # :174-180
        dn = "dense" if i == 0 else f"dense_{i}"
        net = R.dense(net, P[f"dnn_part/{dn}/kernel"], P[f"dnn_part/{dn}/bias"], relu=True)
        if params.get("batch_norm"):
            bn = "batch_normalization" if i == 0 else f"batch_normalization_{i}"
            net = R.batch_norm(net, P[f"dnn_part/{bn}/gamma"], P[f"dnn_part/{bn}/beta"],
                               P[f"dnn_part/{bn}/moving_mean"], P[f"dnn_part/{bn}/moving_variance"], training)
        net = _dropout(net, params, training, masks)                                                            # :178-179 (behind the BN)
    n = len(params["hidden_units"])
    dn = "dense" if n == 0 else f"dense_{n}"
    nfm_logit = R.dense(net, P[f"dnn_part/{dn}/kernel"], P[f"dnn_part/{dn}/bias"])                               # :181
    logit = dense_logit + nfm_logit                                                                             # :183
    return _tail(logit, None if labels is None else labels["read_comment"])


def xdeepfm(P, feats, labels, params, training=False):
    """algorithm/xDeepFM/xdeepfm.py:139-207."""
    dense_cols = params.get("dense_feature_columns") or []
    cat = input_layer(P, feats, params["category_feature_columns"], "category_input/input_layer", {})
    if dense_cols:
        linear_vec = torch.cat([input_layer(P, feats, dense_cols, "dense_input/input_layer"), cat], dim=-1)  # :162
    else:
        linear_vec = cat
    linear_logit = R.dense(linear_vec, P["linear_part/dense/kernel"], P["linear_part/dense/bias"])            # :163
    m, D = len(params["category_feature_columns"]), int(params["embedding_dim"])
    x0 = cat.reshape(-1, m, D)                                                                               # :167
    filters = [P[f"cin_part/cin_layer_{i + 1}_filter"] for i in range(len(params["cin_layer_feature_maps"]))]
    _, p_plus = R.cin_stack(x0, filters)                                                                     # :170-174
    cin_logit = R.dense(p_plus, P["cin_part/dense/kernel"])                                                  # :175 no bias
    dnn = _mlp(P, linear_vec, "dnn_part", [f"dense_{i}" for i in range(len(params["hidden_units"]))])
    n = len(params["hidden_units"])
    dnn_logit = R.dense(dnn, P["dnn_part/dense/kernel"])                                                     # :182 no bias
    return _tail(linear_logit + cin_logit + dnn_logit, None if labels is None else labels["read_comment"])


def din(P, feats, labels, params, training=False, dropout_masks=None):
    """algorithm/DIN/din.py:186-257; fcn order dense -> dice | prelu -> BN -> dropout (:227-236)."""
    masks = list(dropout_masks or [])
    reg = {}
    parts = []
    dense_cols = params.get("dense_feature_columns") or []
    if dense_cols:
        parts.append(input_layer(P, feats, dense_cols, "dense_input/input_layer"))                 # :200-201
    category = input_layer(P, feats, params["category_feature_columns"], "category_input/input_layer", reg)
    tcol = params["target_feedid_feature_columns"][0]
    scol = params["sequence_feature_columns"][0]
    tname = reg.setdefault(tcol.shared_name, table_name(tcol, "target_input/sequence_input_layer"))
    table = P[tname]
    target = R.embedding_lookup_single(feats[tcol.key], table)                                     # :208-210
    vals, offs = feats[scol.key]
    seq, seq_len = R.sequence_lookup(vals, offs, table, params.get("sequence_max_length"))         # :213-214
    a = "attention_part"
    att = R.din_attention(target, seq, seq_len,
                          P[f"{a}/f1_att/kernel"], P[f"{a}/f1_att/bias"], P[f"{a}/f2_att/kernel"],
                          P[f"{a}/f2_att/bias"], P[f"{a}/f3_att/kernel"], P[f"{a}/f3_att/bias"],
                          is_softmax=params["use_softmax"])                                        # :217-218
    net = torch.cat(parts + [category, target, att], dim=-1)                                       # :221
    for i, _ in enumerate(params["hidden_units"]):
        dn = "dense" if i == 0 else f"dense_{i}"
        net = R.dense(net, P[f"fcn/{dn}/kernel"], P[f"fcn/{dn}/bias"])                              # :227
        if params["activation"] == "dice":
            net = R.dice(net, P[f"fcn/dice_alpha_{i + 1}"])
        else:
            net = R.prelu(net, P[f"fcn/prelu_alpha_{i + 1}"])
        if params["batch_norm"]:
            bn = "batch_normalization" if i == 0 else f"batch_normalization_{i}"
            net = R.batch_norm(net, P[f"fcn/{bn}/gamma"], P[f"fcn/{bn}/beta"], P[f"fcn/{bn}/moving_mean"],
                               P[f"fcn/{bn}/moving_variance"], training)
        net = _dropout(net, params, training, masks)                                               # :235-236
    n = len(params["hidden_units"])
    dn = "dense" if n == 0 else f"dense_{n}"
    logit = R.dense(net, P[f"fcn/{dn}/kernel"], P[f"fcn/{dn}/bias"])
    extra = None
    if labels is not None and params["mini_batch_aware_regularization"] and params["l2_lambda"] > 0:
        extra = R.din_mba_reg(category, target, att, params["l2_lambda"])                          # :254-257
    return _tail(logit, None if labels is None else labels["read_comment"], extra)


def _dnn_bn(P, net, scope, hidden_units, batch_norm, training, params=None, masks=None):
    """dense(relu) -> dropout -> BN, the DeepFM/PNN/FiBiNET MLP order (quirk B-7)."""
    for i, _ in enumerate(hidden_units):
        dn = "dense" if i == 0 else f"dense_{i}"
        net = R.dense(net, P[f"{scope}/{dn}/kernel"], P[f"{scope}/{dn}/bias"], relu=True)
        net = _dropout(net, params or {}, training, masks)
        if batch_norm:
            bn = "batch_normalization" if i == 0 else f"batch_normalization_{i}"
            net = R.batch_norm(net, P[f"{scope}/{bn}/gamma"], P[f"{scope}/{bn}/beta"],
                               P[f"{scope}/{bn}/moving_mean"], P[f"{scope}/{bn}/moving_variance"], training)
    n = len(hidden_units)
    dn = "dense" if n == 0 else f"dense_{n}"
    return R.dense(net, P[f"{scope}/{dn}/kernel"], P[f"{scope}/{dn}/bias"])


def fibinet(P, feats, labels, params, training=False, dropout_masks=None):
    """algorithm/FiBiNET/fibinet.py:143-221."""
    dense_cols = params.get("dense_feature_columns") or []
    cat = input_layer(P, feats, params["category_feature_columns"], "category_input/input_layer", {})
    F, K = len(params["category_feature_columns"]), int(params["embedding_dim"])
    cat = cat.reshape(-1, F, K)                                                                    # :163
    t = params["bilinear_interaction_type"]
    bi = R.fibinet_interaction(cat, P["senet_part/senet_w1"], P["senet_part/senet_w2"],
                               P[f"bilinear_interaction_part/orginal_w_{t}"],
                               P[f"bilinear_interaction_part/senet_w_{t}"], t)                     # :171-187
    logit = _dnn_bn(P, bi, "dnn_part", params["hidden_units"], params.get("batch_norm"), training, params,
                    list(dropout_masks or []))                                                     # :189-197
    if dense_cols:
        dense_in = input_layer(P, feats, dense_cols, "dense_input/input_layer")
        logit = R.dense(dense_in, P["linear_part/dense/kernel"], P["linear_part/dense/bias"]) + logit   # :168,199
    return _tail(logit, None if labels is None else labels["read_comment"])


def pnn(P, feats, labels, params, training=False, dropout_masks=None):
    """algorithm/PNN/pnn.py:112-214."""
    fields, reg = [], {}
    for i, c in enumerate(params["category_feature_columns"]):                                     # :126-129 list order
        layer = "input_layer" if i == 0 else f"input_layer_{i}"
        fields.append(_lookup(P, feats, c, layer, reg))      # shared table: made by the first layer using it
    emb = torch.cat(fields, dim=-1)                                                                # :130
    F, K = len(fields), int(params["embedding_dim"])
    method = params["product_method"]
    pw = P["product_part/inner_product_w"] if method == "IPNN" else P["product_part/outer_product_w"]
    _, _, product_final = R.pnn_product_fast(emb, P["linear_part/linear_w"], pw, P["bias"], F, K, method)   # :133-181
    logit = _dnn_bn(P, product_final, "fcn", params["hidden_units"], params.get("batch_norm"), training, params,
                    list(dropout_masks or []))                                                              # :184-193
    extra = None
    wr = float(params.get("weight_regularizer") or 0.0)
    if labels is not None and wr > 0:            # tf.contrib.layers.l2_regularizer(scale): scale * sum(w^2) / 2
        extra = wr * 0.5 * ((P["linear_part/linear_w"] ** 2).sum() + (pw ** 2).sum())               # :138,151,164,209-211
    return _tail(logit, None if labels is None else labels["read_comment"], extra)
