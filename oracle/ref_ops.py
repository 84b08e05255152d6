"""oracle/ref_ops.py — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Unfused, op-for-op torch-CPU restatement of the reference's hot-path sub-graphs
(SURVEY.md §8a rows a2..a15).  Every function cites the reference lines it
follows (paths relative to /root/reference).  All functions are dtype-generic:
run them in float32 to mimic the TF1 CPU graph, in float64 to bound the error of
both implementations.  Gradients come from torch.autograd through this same
code (the reference has no backward code: TF autodiff).

TF-1.14 primitive semantics that are not visible in the reference source are
restated from SURVEY.md Appendix A and flagged [TF-ext].

Pinning: primitive-level parity unpinned (TF not installable); composition
pinned by tests/golden/*.npz generated from the reference's own sources run on
oracle/tf1_shim (see oracle/gen_golden.py).
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------- #
# a1/a2/a3/a10: feature columns
# --------------------------------------------------------------------------- #
def embedding_lookup_single(ids: Tensor, table: Tensor) -> Tensor:
    """Single-valued `embedding_column` through `fc.input_layer`
    (algorithm/DeepFM/deepfm.py:83-93,187-190).  [TF-ext A-3]
    safe_embedding_lookup_sparse: id < 0 (OOV / '') is dropped, the then-empty
    row is overwritten with zeros; a single valid id yields the exact table row
    (mean over one element)."""
    valid = ids >= 0
    rows = table[ids.clamp(min=0)]
    return torch.where(valid.unsqueeze(-1), rows, torch.zeros_like(rows))


def embedding_lookup_mean(values: Tensor, offsets: Tensor, table: Tensor) -> Tensor:
    """Multi-valued `embedding_column(..., combiner='mean')`
    (algorithm/DCN/dcn.py:97-103 `manual_tag_id_emb`, shared `feedid_emb[1]`).
    [TF-ext A-3] ids < 0 dropped; SparseSegmentMean = sequential fp32 sum in id
    order, then divide by the count of valid ids; no valid id -> zero row.
    `values` (nnz,) int64, `offsets` (B+1,) int64 CSR."""
    B = offsets.numel() - 1
    K = table.shape[1]
    lens = (offsets[1:] - offsets[:-1])
    maxlen = int(lens.max().item()) if B > 0 and values.numel() > 0 else 0
    acc = torch.zeros(B, K, dtype=table.dtype)
    cnt = torch.zeros(B, dtype=table.dtype)
    for j in range(maxlen):  # j-th element of every bag, in order
        has = lens > j
        pos = (offsets[:-1] + j).clamp(max=max(values.numel() - 1, 0))
        ids = torch.where(has, values[pos], torch.full_like(pos, -1))
        ok = ids >= 0
        rows = table[ids.clamp(min=0)]
        acc = acc + torch.where(ok.unsqueeze(-1), rows, torch.zeros_like(rows))
        cnt = cnt + ok.to(table.dtype)
    out = acc / cnt.clamp(min=1.0).unsqueeze(-1)
    return torch.where((cnt > 0).unsqueeze(-1), out, torch.zeros_like(out))


def sequence_lookup(values: Tensor, offsets: Tensor, table: Tensor,
                    T: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """`tf.contrib.feature_column.sequence_input_layer`
    (algorithm/DIN/din.py:207-214).  [TF-ext A-6] output (B, T, H) zero padded,
    T = max length in the batch; sequence_length counts entries including OOV
    ones, whose rows are zero."""
    B = offsets.numel() - 1
    lens = offsets[1:] - offsets[:-1]
    if T is None:
        T = int(lens.max().item()) if B > 0 else 0
    H = table.shape[1]
    out = torch.zeros(B, T, H, dtype=table.dtype)
    for j in range(T):
        has = lens > j
        pos = (offsets[:-1] + j).clamp(max=max(values.numel() - 1, 0))
        ids = torch.where(has, values[pos], torch.full_like(pos, -1))
        ok = ids >= 0
        rows = table[ids.clamp(min=0)]
        out[:, j, :] = torch.where(ok.unsqueeze(-1), rows, torch.zeros_like(rows))
    return out, lens


def indicator_first_order(ids_per_col: Sequence[Tensor], w_per_col: Sequence[Tensor],
                          bias: Tensor) -> Tensor:
    """FM first order: `indicator_column` -> `input_layer` -> `dense(1)`
    (algorithm/DeepFM/deepfm.py:72-80,179-181).  [TF-ext A-5] the multi-hot row
    times the (sum V, 1) kernel equals bias + sum_f w_f[id]; OOV contributes
    nothing.  Columns must be passed already sorted by column name (A-1)."""
    B = ids_per_col[0].shape[0]
    acc = torch.zeros(B, dtype=w_per_col[0].dtype)
    for ids, w in zip(ids_per_col, w_per_col):
        w = w.reshape(-1)               # the column's slice of the (sum V, 1) kernel
        ok = ids >= 0
        acc = acc + torch.where(ok, w[ids.clamp(min=0)], torch.zeros_like(acc))
    return (acc + bias).unsqueeze(-1)


# --------------------------------------------------------------------------- #
# a4: FM second order
# --------------------------------------------------------------------------- #
def fm_second_order(fields: Sequence[Tensor]) -> Tensor:
    """algorithm/DeepFM/deepfm.py:184-200.  `fields` = F tensors (B, K)."""
    squared = [e * e for e in fields]                     # :190 tf.square per field
    s = fields[0]
    for e in fields[1:]:                                   # :194 tf.add_n, input order
        s = s + e
    sum_then_square = s * s                                # :194 tf.square
    q = squared[0]
    for e in squared[1:]:                                  # :196 tf.add_n
        q = q + e
    return (0.5 * (sum_then_square - q)).sum(dim=1, keepdim=True)   # :198-200


# --------------------------------------------------------------------------- #
# a5: DCN cross layer
# --------------------------------------------------------------------------- #
def cross_layer(x0: Tensor, xl: Tensor, wl: Tensor, bl: Tensor) -> Tensor:
    """algorithm/DCN/cross_layer.py:16-24.  wl, bl: (d, 1)."""
    xl_wl = xl @ wl                                        # :21 (B,1)
    x0_xl_wl = x0 * xl_wl                                  # :22
    out = x0_xl_wl + bl.t()                                # :23
    return out + xl                                        # :24


def cross_stack(x0: Tensor, ws: Sequence[Tensor], bs: Sequence[Tensor]) -> Tensor:
    """algorithm/DCN/dcn.py:157-160."""
    xl = x0
    for w, b in zip(ws, bs):
        xl = cross_layer(x0, xl, w, b)
    return xl


# --------------------------------------------------------------------------- #
# a6: xDeepFM CIN
# --------------------------------------------------------------------------- #
def cin_layer(x0: Tensor, xk: Tensor, filters: Tensor) -> Tensor:
    """algorithm/xDeepFM/cin_layer.py:17-28.
    x0 (B, m, D), xk (B, hk, D), filters (1, hk*m, hk_1) -> (B, hk_1, D)."""
    B, m, D = x0.shape
    hk = xk.shape[1]
    outer = torch.einsum('bik,bjk->bkij', xk, x0)          # :21 (B, D, hk, m)
    outer = outer.reshape(B, D, hk * m)                    # :22
    # :26 conv1d, width-1 filter, stride 1, VALID == per-position matmul [TF-ext A-11]
    xk_1 = outer @ filters[0]                              # (B, D, hk_1)
    return xk_1.permute(0, 2, 1)                           # :28


def cin_stack(x0: Tensor, filters: Sequence[Tensor]) -> Tuple[List[Tensor], Tensor]:
    """algorithm/xDeepFM/xdeepfm.py:166-174: returns the layer outputs and
    p_plus = concat of sum-pooling over D."""
    xk = x0
    xs = []
    for f in filters:
        xk = cin_layer(x0, xk, f)
        xs.append(xk)
    p_plus = torch.cat([x.sum(dim=-1) for x in xs], dim=-1)
    return xs, p_plus


# --------------------------------------------------------------------------- #
# a11/a12: DIN attention, activations
# --------------------------------------------------------------------------- #
def din_attention(query: Tensor, keys: Tensor, keys_length: Tensor,
                  f1_w: Tensor, f1_b: Tensor, f2_w: Tensor, f2_b: Tensor,
                  f3_w: Tensor, f3_b: Tensor, is_softmax: bool = False) -> Tensor:
    """algorithm/DIN/din_attention.py:17-41.
    query (B,H), keys (B,T,H), keys_length (B,), f1_w (4H,64), f2_w (64,32),
    f3_w (32,1)."""
    B, T, H = keys.shape
    q = query.repeat(1, T).reshape(B, T, H)                              # :18-19
    cross_all = torch.cat([q, keys, q - keys, q * keys], dim=-1)         # :20
    d1 = torch.relu(cross_all @ f1_w + f1_b)                             # :21
    d2 = torch.relu(d1 @ f2_w + f2_b)                                    # :22
    d3 = d2 @ f3_w + f3_b                                                # :23 (B,T,1)
    w = d3
    mask = (torch.arange(T).unsqueeze(0) < keys_length.unsqueeze(1)).unsqueeze(-1)   # :27-28
    if is_softmax:
        paddings = torch.ones_like(w) * float(-2 ** 32 + 1)              # :31
        w = torch.where(mask, w, paddings)                               # :32
        w = w / (H ** 0.5)                                               # :34
        w = torch.softmax(w, dim=1)                                      # :35
    else:
        w = w * mask.to(w.dtype)                                         # :37-38
    out = w.transpose(1, 2) @ keys                                       # :40 (B,1,H)
    return out.squeeze(1)                                                # :41


def prelu(x: Tensor, alpha: Tensor) -> Tensor:
    """algorithm/DIN/activations.py:13-17."""
    zero = torch.zeros((), dtype=x.dtype)
    return torch.maximum(zero, x) + alpha * torch.minimum(zero, x)


def dice(x: Tensor, alpha: Tensor) -> Tensor:
    """algorithm/DIN/activations.py:29-37.  The BN call has no `training=` so it
    is always inference-mode with the never-updated moving stats (0, 1),
    center=False, scale=False, eps=1e-3 [TF-ext A-8, quirk B-5]."""
    x_norm = x / math.sqrt(1.0 + 1e-3)
    px = torch.sigmoid(x_norm)
    return x * px + alpha * x * (1 - px)


# --------------------------------------------------------------------------- #
# a8/a9: FiBiNET
# --------------------------------------------------------------------------- #
def senet(inp: Tensor, w1: Tensor, w2: Tensor) -> Tensor:
    """algorithm/FiBiNET/senet.py:26-34.  inp (B,F,K), w1 (F,r), w2 (r,F)."""
    z = inp.mean(dim=-1)              # :26
    a = torch.relu(z @ w1)            # :27-28
    a = torch.relu(a @ w2)            # :29-30
    return inp * a.unsqueeze(-1)      # :31-34


def bilinear_interaction(inp: Tensor, w: Tensor, type: str) -> Tensor:
    """algorithm/FiBiNET/bilinear_interaction_layer.py:20-40.  Pairs are
    combinations(range(F-1), 2): the last field never participates (quirk B-3)."""
    F = inp.shape[1]
    pairs = list(itertools.combinations(range(F - 1), 2))
    if type == "all":                 # :20-24  w (K,K)
        v_w = inp @ w
        p = [v_w[:, i, :] * inp[:, j, :] for i, j in pairs]
    elif type == "each":              # :26-29  w (F-1,K,K)
        v_w = [inp[:, i, :] @ w[i] for i in range(F - 1)]
        p = [v_w[i] * inp[:, j, :] for i, j in pairs]
    elif type == "interaction":       # :31-34  w (F(F-1)/2,K,K); zip truncation
        p = [(inp[:, i, :] @ w[k]) * inp[:, j, :]
             for (i, j), k in zip(pairs, range(F * (F - 1) // 2))]
    else:
        raise ValueError(
            f"Bilinear Interaction type must be in ['all','each','interaction'], got '{type}'")
    return torch.stack(p, dim=1)      # :40


def fibinet_interaction(cat: Tensor, senet_w1: Tensor, senet_w2: Tensor,
                        w_orig: Tensor, w_senet: Tensor, type: str) -> Tensor:
    """algorithm/FiBiNET/fibinet.py:171-187 -> (B, P*2K) flattened."""
    v = senet(cat, senet_w1, senet_w2)
    p0 = bilinear_interaction(cat, w_orig, type)
    p1 = bilinear_interaction(v, w_senet, type)
    tot = torch.cat([p0, p1], dim=-1)
    return tot.reshape(tot.shape[0], -1)


# --------------------------------------------------------------------------- #
# a7: PNN product layer
# --------------------------------------------------------------------------- #
def pnn_product(emb_flat: Tensor, linear_w: Tensor, product_w: Tensor, bias: Tensor,
                F: int, K: int, method: str) -> Tuple[Tensor, Tensor, Tensor]:
    """algorithm/PNN/pnn.py:133-181.  Returns (lz, lp, relu(lz+lp+bias)).
    IPNN product_w (D,F); OPNN product_w (D,K,K)."""
    lz = emb_flat @ linear_w                                   # :139
    E = emb_flat.reshape(-1, F, K)                             # :143
    D = linear_w.shape[1]
    outs = []
    if method == "IPNN":
        for i in range(D):                                     # :152
            theta = product_w[i].unsqueeze(1)                  # :153 (F,1)
            delta = (E * theta).sum(dim=1)                     # :155-156 (B,K)
            outs.append((delta * delta).sum(dim=1, keepdim=True))   # :157
    else:
        s = E.sum(dim=1)                                       # :165
        p = s.unsqueeze(2) @ s.unsqueeze(1)                    # :166 (B,K,K)
        for i in range(D):                                     # :167
            wi = product_w[i]
            upper = torch.triu(wi)                             # :169 band_part(0,-1)
            wi = upper + upper.t() - torch.diag(torch.diag(wi))     # :170
            outs.append((p * wi).sum(dim=(1, 2)).unsqueeze(1))      # :171-172
    lp = torch.cat(outs, dim=1)                                # :175
    return lz, lp, torch.relu(lz + lp + bias)                  # :181


def pnn_product_fast(emb_flat: Tensor, linear_w: Tensor, product_w: Tensor, bias: Tensor,
                     F: int, K: int, method: str) -> Tuple[Tensor, Tensor, Tensor]:
    """Same math as pnn_product without the D-iteration Python loop (used only
    to keep large-D oracle runs tractable; checked against pnn_product)."""
    lz = emb_flat @ linear_w
    E = emb_flat.reshape(-1, F, K)
    if method == "IPNN":
        delta = torch.einsum('df,bfk->bdk', product_w, E)
        lp = (delta * delta).sum(dim=-1)
    else:
        s = E.sum(dim=1)
        up = torch.triu(product_w)
        sym = up + up.transpose(1, 2) - torch.diag_embed(torch.diagonal(product_w, dim1=1, dim2=2))
        lp = torch.einsum('ba,dac,bc->bd', s, sym, s)
    return lz, lp, torch.relu(lz + lp + bias)


# --------------------------------------------------------------------------- #
# a13/a14: loss tail
# --------------------------------------------------------------------------- #
def sigmoid_cross_entropy_with_logits(labels: Tensor, logits: Tensor) -> Tensor:
    """[TF-ext A-9] max(x,0) - x*z + log(1+exp(-|x|))."""
    zero = torch.zeros((), dtype=logits.dtype)
    return torch.maximum(logits, zero) - logits * labels + torch.log1p(torch.exp(-logits.abs()))


def ce_loss(labels: Tensor, logits: Tensor) -> Tensor:
    """algorithm/DeepFM/deepfm.py:235."""
    return sigmoid_cross_entropy_with_logits(labels, logits).mean()


def din_mba_reg(category: Tensor, target: Tensor, att: Tensor, l2_lambda: float) -> Tensor:
    """algorithm/DIN/din.py:254-257: lambda * l2_loss(concat) / B,
    tf.nn.l2_loss(t) = sum(t**2)/2."""
    ev = torch.cat([category, target, att], dim=-1)
    return l2_lambda * (ev * ev).sum() / 2 / ev.shape[0]


def tf_metrics_auc(labels: Tensor, preds: Tensor, num_thresholds: int = 200) -> float:
    """[TF-ext A-9] tf.metrics.auc default: 200 thresholds, trapezoidal ROC."""
    eps = 1e-7
    th = [(i + 1) * 1.0 / (num_thresholds - 1) for i in range(num_thresholds - 2)]
    th = torch.tensor([0.0 - eps] + th + [1.0 + eps], dtype=torch.float64)
    p = preds.reshape(-1).double().unsqueeze(0)
    y = labels.reshape(-1).double().unsqueeze(0) > 0.5
    pred_pos = p > th.unsqueeze(1)
    tp = (pred_pos & y).sum(1).double()
    fp = (pred_pos & ~y).sum(1).double()
    fn = (~pred_pos & y).sum(1).double()
    tn = (~pred_pos & ~y).sum(1).double()
    tpr = (tp + eps) / (tp + fn + eps)
    fpr = fp / (fp + tn + eps)
    return float(((fpr[:-1] - fpr[1:]) * (tpr[:-1] + tpr[1:]) / 2).sum())


# --------------------------------------------------------------------------- #
# tf.layers pieces used by the model_fns (context for end-to-end parity)
# --------------------------------------------------------------------------- #
def dense(x: Tensor, kernel: Tensor, bias: Optional[Tensor] = None, relu: bool = False) -> Tensor:
    y = x @ kernel
    if bias is not None:
        y = y + bias
    return torch.relu(y) if relu else y


def batch_norm(x: Tensor, gamma: Tensor, beta: Tensor, moving_mean: Tensor, moving_var: Tensor,
               training: bool, eps: float = 1e-3) -> Tensor:
    """[TF-ext A-8] tf.layers.batch_normalization on (B, C): training uses the
    biased batch variance; inference uses moving stats.  (Moving-stat update,
    momentum 0.99, is done by the caller.)"""
    if training:
        mean = x.mean(dim=0)
        var = ((x - mean) ** 2).mean(dim=0)
    else:
        mean, var = moving_mean, moving_var
    inv = torch.rsqrt(var + eps) * gamma
    return x * inv + (beta - mean * inv)


# --------------------------------------------------------------------------- #
# a15: TF1 Adam
# --------------------------------------------------------------------------- #
def adam_tf1_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
                  beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8) -> None:
    """[TF-ext A-10] tf.train.AdamOptimizer (algorithm/DeepFM/deepfm.py:246-250),
    in place.  `step` is 1-based.  eps is added outside the bias correction.
    Sparse (IndexedSlices) gradients are de-duplicated by summation and then
    applied with *dense* decay of m and v, i.e. exactly this dense update with
    g = scatter_add of the slices."""
    # TF holds lr/beta1/beta2/eps as float32 tensors (cast to the variable dtype), so the decay
    # constants are the fp32-rounded values and (1 - beta) is formed from those (exact by
    # Sterbenz): 1 - float32(0.999) = 0.00100004673, not 0.001.
    f32 = lambda x: float(torch.tensor(x, dtype=torch.float32))
    lr, beta1, beta2, eps = f32(lr), f32(beta1), f32(beta2), f32(eps)
    lr_t = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).add_(g * g, alpha=1.0 - beta2)
    p.sub_(lr_t * m / (v.sqrt() + eps))


def lazy_adam_step(p: Tensor, indices: Tensor, values: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
                   beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8) -> None:
    """[TF-ext] tf.contrib.opt.LazyAdamOptimizer — the optimizer of algorithm/DIEN/dien.py:328 — applied to ONE
    embedding variable's IndexedSlices gradient (indices [n] int64, values [n, K]), in place.  `step` is 1-based.

    TF 1.14 (documented behaviour, restated):
      * Optimizer._apply_sparse_duplicate_indices first de-duplicates the slices: unique indices, values of
        duplicates summed (tf.unsorted_segment_sum) — so a row that appears several times in a batch takes ONE update
        with the summed gradient;
      * LazyAdamOptimizer._apply_sparse then touches only those rows:
            m[i] = beta1 * m[i] + (1 - beta1) * g_i
            v[i] = beta2 * v[i] + (1 - beta2) * g_i^2
            var[i] -= lr_t * m[i] / (sqrt(v[i]) + eps),   lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
        A row that is IN the slices with a summed gradient of exactly zero is still updated (its m, v decay and the
        variable moves); every row NOT in the slices keeps var, m and v unchanged — the difference to
        tf.train.AdamOptimizer (adam_tf1_step above), whose m, v decay for all rows."""
    f32 = lambda x: float(torch.tensor(x, dtype=torch.float32))
    lr, beta1, beta2, eps = f32(lr), f32(beta1), f32(beta2), f32(eps)
    lr_t = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    uniq, inv = torch.unique(indices, return_inverse=True)
    g = torch.zeros(uniq.numel(), values.shape[1], dtype=values.dtype).index_add_(0, inv, values)
    m_t = beta1 * m[uniq] + (1.0 - beta1) * g
    v_t = beta2 * v[uniq] + (1.0 - beta2) * g * g
    m[uniq] = m_t
    v[uniq] = v_t
    p[uniq] = p[uniq] - lr_t * m_t / (v_t.sqrt() + eps)


# --------------------------------------------------------------------------- #
# utils.py
# --------------------------------------------------------------------------- #
def index_from_upper_triangular(i: int, j: int, n: int) -> int:
    """algorithm/utils.py:67-82 (closed form of the loop)."""
    return i * (2 * n - i - 1) // 2 + (j - i - 1)
