"""Seeded synthetic "WeChat-shaped" CTR data.

26 single-valued categorical fields (the six reference names + c06..c25), keys
"<name>_<int>" like /root/reference dataset/wechat_algo_data1/DataGenerator.py:147-159, Zipf
ids, 1 % OOV (''), 16 dense floats log1p(Poisson(3)) (DataGenerator.py:374-380), an optional
<=50-long `his_read_comment_7d_seq` history over the feedid vocabulary and a `manual_tag_list`
bag, label `read_comment` ~ Bernoulli(0.0356) (EDA.ipynb cell 30).

Two forms: (a) vocabulary files + TFRecord of tf.train.Example (the plumbing config), and
(b) device-resident already-encoded id tensors (the throughput configs).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from . import tfrecord

REAL_FIELDS = ["userid", "feedid", "device", "authorid", "bgm_song_id", "bgm_singer_id"]
REAL_VOCABS = [20000, 106444, 2, 18789, 25159, 17500]
DENSE_FEATURES = [
    "videoplayseconds", "u_read_comment_7d_sum", "u_like_7d_sum", "u_click_avatar_7d_sum",
    "u_forward_7d_sum", "u_comment_7d_sum", "u_follow_7d_sum", "u_favorite_7d_sum",
    "i_read_comment_7d_sum", "i_like_7d_sum", "i_click_avatar_7d_sum", "i_forward_7d_sum",
    "i_comment_7d_sum", "i_follow_7d_sum", "i_favorite_7d_sum", "c_user_author_read_comment_7d_sum",
]
LABELS = ["read_comment", "comment", "like", "click_avatar", "forward", "follow", "favorite"]


def field_names(n_fields: int = 26) -> List[str]:
    return REAL_FIELDS[:n_fields] + [f"c{i:02d}" for i in range(len(REAL_FIELDS), n_fields)]


def vocab_sizes(n_fields: int = 26, max_vocab: int = 1_000_000) -> List[int]:
    """Six real-field sizes + log-spaced sizes in [1e2, max_vocab] (seed independent)."""
    extra = n_fields - len(REAL_VOCABS)
    if extra <= 0:
        return REAL_VOCABS[:n_fields]
    logs = np.linspace(2.0, np.log10(max_vocab), extra)
    return REAL_VOCABS + [int(round(10 ** x)) for x in logs]


def zipf_ids(rng: np.random.Generator, n: int, vocab: int, s: float = 1.05) -> np.ndarray:
    """Truncated Zipf(s) over [0, vocab) by inverse-CDF on the exact pmf (vocabularies above 2^24:
    rejection from numpy's unbounded Zipf sampler — the same distribution without a vocab-sized table)."""
    if vocab > (1 << 24):
        out = np.empty(n, dtype=np.int64)
        filled = 0
        while filled < n:
            x = rng.zipf(s, size=2 * (n - filled) + 16)
            x = x[x <= vocab][:n - filled]
            out[filled:filled + x.size] = x - 1
            filled += x.size
        return out
    cdf = _ZIPF_CDF.get((vocab, s))
    if cdf is None:                       # the table depends on (vocab, s) only: built once, not per batch
        ranks = np.arange(1, vocab + 1, dtype=np.float64)
        cdf = np.cumsum(ranks ** (-s))
        cdf /= cdf[-1]
        if len(_ZIPF_CDF) < 256:
            _ZIPF_CDF[(vocab, s)] = cdf
    return np.searchsorted(cdf, rng.random(n), side="left").astype(np.int64)


_ZIPF_CDF: Dict[tuple, np.ndarray] = {}


@dataclass
class SynthSpec:
    n_fields: int = 26
    max_vocab: int = 1_000_000
    oov_frac: float = 0.01
    with_dense: bool = False
    with_history: bool = False
    with_tags: bool = False
    history_len: Optional[int] = None        # fixed length (throughput) or None = U{0..50}
    max_history: int = 50
    tag_vocab: int = 350
    seed: int = 1234
    names: List[str] = field(default_factory=list)
    vocabs: List[int] = field(default_factory=list)

    def __post_init__(self):
        if not self.names:
            self.names = field_names(self.n_fields)
        if not self.vocabs:
            self.vocabs = vocab_sizes(self.n_fields, self.max_vocab)


def make_id_batch(spec: SynthSpec, B: int, batch_index: int = 0, names: Optional[List[str]] = None):
    """-> (ids int64 [B, F] in the order of `names` (default spec.names), labels float32 [B,1],
    dense float32 [B,16] or None, history (values, offsets) or None, tags (values, offsets) or None).
    ids == -1 marks OOV."""
    rng = np.random.default_rng([spec.seed, batch_index])
    names = names or spec.names
    vmap = dict(zip(spec.names, spec.vocabs))
    ids = np.empty((B, len(names)), dtype=np.int64)
    for j, nm in enumerate(names):
        col = zipf_ids(rng, B, vmap[nm])
        if spec.oov_frac > 0:
            col[rng.random(B) < spec.oov_frac] = -1
        ids[:, j] = col
    labels = (rng.random((B, 1)) < 0.0356).astype(np.float32)
    dense = np.log1p(rng.poisson(3.0, size=(B, 16))).astype(np.float32) if spec.with_dense else None
    hist = tags = None
    if spec.with_history:
        lens = (np.full(B, spec.history_len) if spec.history_len is not None
                else rng.integers(0, spec.max_history + 1, size=B))
        off = np.zeros(B + 1, dtype=np.int64)
        off[1:] = np.cumsum(lens)
        vals = zipf_ids(rng, int(off[-1]), vmap["feedid"])
        hist = (vals, off)
    if spec.with_tags:
        lens = rng.integers(0, 6, size=B)
        off = np.zeros(B + 1, dtype=np.int64)
        off[1:] = np.cumsum(lens)
        tags = (zipf_ids(rng, int(off[-1]), spec.tag_vocab), off)
    return ids, labels, dense, hist, tags


def device_features(spec: SynthSpec, B: int, device, batch_index: int = 0, sorted_layout: bool = True):
    """Already-encoded, device-resident (features, labels) for the throughput configs.  The id
    matrix is laid out in sorted(column-name) order so that fc.input_layer can hand it to the
    gather kernel without a copy; features[name] are column views of it.  With a history (the DIN layout) the target
    `feedid` is a sequence column looked up on its own: it gets its own contiguous [B] vector after the matrix of the other
    fields, so that both lookups read the batch in place."""
    from ..feature_column import Ragged
    names = sorted(spec.names) if sorted_layout else list(spec.names)
    ids, labels, dense, hist, tags = make_id_batch(spec, B, batch_index, names)
    # ONE allocation for the whole batch (int64 [B, F] id matrix, float32 [B, 1] labels, then the ragged history / tag
    # lists, each piece 16-byte aligned): a training loop that copies a batch into the static input buffers of a captured
    # step then moves it with a single copy
    own = [nm for nm in names if hist is not None and sorted_layout and nm == "feedid"]
    own_cols = [np.ascontiguousarray(ids[:, names.index(nm)]) for nm in own]
    if own:
        keep = [j for j, nm in enumerate(names) if nm not in own]
        ids, names = ids[:, keep], [names[j] for j in keep]
    pieces = [np.ascontiguousarray(ids), np.ascontiguousarray(labels)]
    for rag in (hist, tags):
        if rag is not None:
            pieces += [np.ascontiguousarray(rag[0]), np.ascontiguousarray(rag[1])]
    pieces[2:2] = own_cols
    offs, total = [], 0
    for a in pieces:
        offs.append(total)
        total += (a.nbytes + 15) // 16 * 16
    host = torch.zeros(total, dtype=torch.uint8)
    for a, o in zip(pieces, offs):
        host[o:o + a.nbytes] = torch.from_numpy(a).view(-1).view(torch.uint8)
    buf = host.to(device)

    def piece(k):
        a = pieces[k]
        return buf[offs[k]:offs[k] + a.nbytes].view(getattr(torch, a.dtype.name)).view(a.shape)
    nid = ids.size * 8
    mat = piece(0)
    feats: Dict[str, object] = {nm: mat[:, j] for j, nm in enumerate(names)}
    feats_meta = {"__ids_matrix__": mat, "__ids_names__": names}
    if dense is not None:
        d = torch.from_numpy(dense).to(device)
        for j, nm in enumerate(DENSE_FEATURES):
            feats[nm] = d[:, j:j + 1]
    k = 2
    for nm in own:
        feats[nm] = piece(k)
        k += 1
    if hist is not None:
        feats["his_read_comment_7d_seq"] = Ragged(piece(k), piece(k + 1))
        k += 2
    if tags is not None:
        feats["manual_tag_list"] = Ragged(piece(k), piece(k + 1))
    lab = {"read_comment": piece(1).view(B, 1)}
    return feats, lab, feats_meta


_ZIPF_CDF_DEV: Dict[tuple, torch.Tensor] = {}


def device_fresh_batches(spec: SynthSpec, B: int, device, n: int, seed: int):
    """n batches of the same distribution as make_id_batch (truncated Zipf(1.05) ids per field, oov_frac, 3.56 %
    positives) drawn ON the device — a long optimizer-state sweep needs thousands of never-repeated batches and the
    host generator makes ~15 per second.  Single-valued id fields only (no history / tags / dense features), vocabularies
    up to 2^24.  -> list of (features, labels) laid out exactly like device_features (one allocation per batch: id
    matrix followed by the labels), from its own random stream (not the host generator's)."""
    if spec.with_history or spec.with_tags or spec.with_dense or max(spec.vocabs) > (1 << 24):
        raise ValueError("device_fresh_batches: single-valued id fields with vocabularies <= 2^24 only")
    names = sorted(spec.names)
    vmap = dict(zip(spec.names, spec.vocabs))
    F = len(names)
    gen = torch.Generator(device=device).manual_seed(int(seed))
    ids = torch.empty(n, B, F, dtype=torch.int64, device=device)
    for j, nm in enumerate(names):
        V = vmap[nm]
        cdf = _ZIPF_CDF_DEV.get((V, str(device)))
        if cdf is None:
            ranks = torch.arange(1, V + 1, dtype=torch.float64, device=device)
            cdf = torch.cumsum(ranks.pow(-1.05), 0)
            cdf = _ZIPF_CDF_DEV[(V, str(device))] = cdf / cdf[-1]
        u = torch.rand(n * B, dtype=torch.float64, device=device, generator=gen)
        col = torch.searchsorted(cdf, u).clamp_(max=V - 1)
        if spec.oov_frac > 0:
            col = torch.where(torch.rand(n * B, device=device, generator=gen) < spec.oov_frac, torch.full_like(col, -1), col)
        ids[:, :, j] = col.view(n, B)
    labels = (torch.rand(n, B, device=device, generator=gen) < 0.0356).to(torch.float32)
    nid = B * F * 8
    out = []
    for i in range(n):
        buf = torch.empty(nid + B * 4, dtype=torch.uint8, device=device)
        buf[:nid] = ids[i].reshape(-1).view(torch.uint8)
        buf[nid:] = labels[i].view(torch.uint8)
        mat = buf[:nid].view(torch.int64).view(B, F)
        out.append(({nm: mat[:, j] for j, nm in enumerate(names)}, {"read_comment": buf[nid:].view(torch.float32).view(B, 1)}))
    return out


# ---- on-disk form: vocabulary files + TFRecord ------------------------------------------------
def key_of(name: str, i: int) -> bytes:
    return f"{name}_{i}".encode()


def write_vocabularies(spec: SynthSpec, vocab_dir: str) -> None:
    os.makedirs(vocab_dir, exist_ok=True)
    for nm, v in zip(spec.names, spec.vocabs):
        with open(os.path.join(vocab_dir, nm + ".txt"), "wb") as f:
            f.write(b"\n".join(key_of(nm, i) for i in range(v)) + b"\n")
    with open(os.path.join(vocab_dir, "manual_tag_id.txt"), "wb") as f:
        f.write(b"\n".join(key_of("manual_tag_id", i) for i in range(spec.tag_vocab)) + b"\n")


def write_tfrecord(spec: SynthSpec, path: str, n_examples: int, chunk: int = 4096,
                   as_sequence_example: bool = False) -> int:
    """tf.train.Example per row: 16 float dense, bytes categorical ('' for OOV), 7 float labels
    (DataGenerator.py:409-425).  List features go into multi-valued bytes_list (the layout the
    models' parse spec expects) or, with as_sequence_example, into feature_lists exactly like the
    checked-in writer (quirk B-9: they then parse as empty)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)

    def gen():
        done = 0
        bi = 0
        while done < n_examples:
            B = min(chunk, n_examples - done)
            ids, labels, dense, hist, tags = make_id_batch(spec, B, bi)
            rng = np.random.default_rng([spec.seed, bi, 7])
            for r in range(B):
                ctx = {}
                if dense is not None:
                    for j, nm in enumerate(DENSE_FEATURES):
                        ctx[nm] = ("float", [float(dense[r, j])])
                for j, nm in enumerate(spec.names):
                    i = int(ids[r, j])
                    ctx[nm] = ("bytes", [key_of(nm, i) if i >= 0 else b""])
                ctx["read_comment"] = ("float", [float(labels[r, 0])])
                for lb in LABELS[1:]:
                    ctx[lb] = ("float", [float(rng.random() < 0.02)])
                lists = {}
                if hist is not None:
                    hv = hist[0][hist[1][r]:hist[1][r + 1]]
                    lists["his_read_comment_7d_seq"] = ("bytes", [key_of("feedid", int(i)) for i in hv])
                if tags is not None:
                    tv = tags[0][tags[1][r]:tags[1][r + 1]]
                    lists["manual_tag_list"] = ("bytes", [key_of("manual_tag_id", int(i)) for i in tv])
                if as_sequence_example:
                    yield tfrecord.encode_sequence_example(ctx, lists)
                else:
                    ctx.update(lists)
                    yield tfrecord.encode_example(ctx)
            done += B
            bi += 1

    return tfrecord.write_records(path, gen())
