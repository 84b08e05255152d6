"""tf.feature_column look-alikes (the subset the six hot-path models use).

Reference call sites: algorithm/DeepFM/deepfm.py:44-99,179-190; algorithm/DCN/dcn.py:45-113,
148-153; algorithm/DIN/din.py:50-120,200-214.  TF-1.14 semantics restated from SURVEY.md
Appendix A (A-1 column order, A-2 vocabulary ids, A-3 safe mean lookup, A-4 shared tables,
A-5 indicator, A-6 sequence layer).

`features[key]` for a categorical key may be
  * an int64 tensor [B] / [B,1] of already-encoded ids (id < 0 == OOV), single-valued;
  * a `Ragged(values, offsets)` of already-encoded ids, multi-valued / sequence;
  * raw keys: a list/ndarray of bytes|str (single-valued) or a list of lists (multi-valued);
    they are encoded on the host with the column's vocabulary (line number, OOV -> -1).
"""
from __future__ import annotations

import os
from collections import namedtuple
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops, sparse
from .variables import EmbeddingArena, current_store, truncated_normal

Ragged = namedtuple("Ragged", ["values", "offsets"])   # int64 [nnz], int64 [B+1]

_VOCAB_CACHE: Dict[str, Dict[bytes, int]] = {}


def _load_vocab(path: str) -> Dict[bytes, int]:
    v = _VOCAB_CACHE.get(path)
    if v is None:
        with open(path, "rb") as f:
            keys = f.read().split(b"\n")
        if keys and keys[-1] == b"":
            keys.pop()
        v = {k: i for i, k in enumerate(keys)}          # id = 0-based line index (A-2)
        _VOCAB_CACHE[path] = v
    return v


# --------------------------------------------------------------------------------------------
# columns
# --------------------------------------------------------------------------------------------
class NumericColumn:
    def __init__(self, key, shape=(1,), default_value=None):
        self.key, self.shape, self.default_value = key, tuple(shape), default_value
        self.name = key

    def parse_spec(self):
        return {self.key: ("fixed", np.float32, self.shape, self.default_value)}


class CategoricalColumn:
    """categorical_column_with_vocabulary_file / _with_identity (+ sequence_ variants)."""

    def __init__(self, key, vocabulary_file=None, num_buckets=None, is_sequence=False,
                 default_value=-1):
        self.key, self.vocabulary_file, self.is_sequence = key, vocabulary_file, is_sequence
        self.default_value = default_value
        self._num_buckets = num_buckets
        self.name = key

    @property
    def num_buckets(self) -> int:
        if self._num_buckets is None:
            self._num_buckets = len(_load_vocab(self.vocabulary_file))
        return self._num_buckets

    def parse_spec(self):
        return {self.key: ("varlen", np.bytes_ if self.vocabulary_file else np.int64)}

    # -- host side string -> id (a1; bit-exact integer work) -----------------------------
    def encode(self, raw) -> "torch.Tensor | Ragged":
        vocab = _load_vocab(self.vocabulary_file) if self.vocabulary_file else None

        def one(k):
            if vocab is None:
                k = int(k)
                return k if 0 <= k < self.num_buckets else self.default_value
            if isinstance(k, str):
                k = k.encode()
            return vocab.get(bytes(k), self.default_value)

        if len(raw) and isinstance(raw[0], (list, tuple, np.ndarray)):
            offsets = np.zeros(len(raw) + 1, dtype=np.int64)
            vals: List[int] = []
            for i, row in enumerate(raw):
                vals.extend(one(k) for k in row)
                offsets[i + 1] = len(vals)
            return Ragged(torch.tensor(vals, dtype=torch.int64), torch.from_numpy(offsets))
        return torch.tensor([one(k) for k in raw], dtype=torch.int64)

    def ids(self, features, device):
        x = features[self.key]
        if not isinstance(x, (torch.Tensor, Ragged)):
            x = self.encode(x)
            if isinstance(x, Ragged) and not self.is_sequence:
                lens = x.offsets[1:] - x.offsets[:-1]
                if x.offsets.numel() > 1 and int(lens.max()) <= 1:
                    # a VarLen feature that is single-valued in this batch: dense [B] ids
                    # (-1 where the list is empty); mean over one row == the row itself
                    dense = torch.full((lens.numel(),), -1, dtype=torch.int64)
                    dense[lens == 1] = x.values
                    x = dense
        x = self._in_range(x)
        if isinstance(x, Ragged):
            return Ragged(x.values.to(device), x.offsets.to(device))
        if x.dim() == 2 and x.shape[1] == 1:
            x = x[:, 0]
        return x.to(device)

    def _in_range(self, x):
        """Pre-encoded ids are not checked by the kernels (include/recalgo.h: an id >= vocab would read and
        atomic-add outside its table).  Host-resident ids are clamped to OOV (default_value) here; device-resident
        ids are trusted, and verified only under RECALGO_CHECK_IDS=1 (one host sync per lookup)."""
        import os
        vals = x.values if isinstance(x, Ragged) else x
        if not isinstance(vals, torch.Tensor) or vals.dtype != torch.int64:
            return x
        nb = self.num_buckets
        if vals.device.type == "cpu":
            bad = vals >= nb
            if bool(bad.any()):
                vals = torch.where(bad, torch.full_like(vals, self.default_value), vals)
                return Ragged(vals, x.offsets) if isinstance(x, Ragged) else vals
        elif os.environ.get("RECALGO_CHECK_IDS") == "1" and vals.numel() and int(vals.max()) >= nb:
            raise ValueError(f"feature {self.key}: id {int(vals.max())} >= vocabulary size {nb}")
        return x


class EmbeddingColumn:
    def __init__(self, categorical_column, dimension, combiner="mean", shared_name=None):
        self.categorical_column, self.dimension, self.combiner = categorical_column, int(dimension), combiner
        self.shared_name = shared_name
        suffix = "_shared_embedding" if shared_name else "_embedding"
        self.name = categorical_column.key + suffix
        if combiner != "mean":
            raise ValueError("only combiner='mean' is used by the reference models")

    @property
    def key(self):
        return self.categorical_column.key

    def parse_spec(self):
        return self.categorical_column.parse_spec()


class IndicatorColumn:
    def __init__(self, categorical_column):
        self.categorical_column = categorical_column
        self.name = categorical_column.key + "_indicator"

    @property
    def key(self):
        return self.categorical_column.key

    def parse_spec(self):
        return self.categorical_column.parse_spec()


def numeric_column(key, shape=(1,), default_value=None, dtype=None):
    return NumericColumn(key, shape, default_value)


def categorical_column_with_vocabulary_file(key, vocabulary_file, vocabulary_size=None,
                                            num_oov_buckets=0, default_value=None):
    if num_oov_buckets:
        raise NotImplementedError("num_oov_buckets is not used by the reference")
    return CategoricalColumn(key, vocabulary_file, vocabulary_size,
                             default_value=-1 if default_value is None else default_value)


def sequence_categorical_column_with_vocabulary_file(key, vocabulary_file, vocabulary_size=None,
                                                     num_oov_buckets=0, default_value=None):
    return CategoricalColumn(key, vocabulary_file, vocabulary_size, is_sequence=True,
                             default_value=-1 if default_value is None else default_value)


def categorical_column_with_identity(key, num_buckets, default_value=None):
    return CategoricalColumn(key, None, int(num_buckets))


def embedding_column(categorical_column, dimension, combiner="mean"):
    return EmbeddingColumn(categorical_column, dimension, combiner)


def shared_embedding_columns(categorical_columns, dimension, combiner="mean"):
    """One table named after the sorted keys; columns returned in INPUT order (A-4)."""
    shared = "_".join(sorted(c.key for c in categorical_columns)) + "_shared_embedding"
    return [EmbeddingColumn(c, dimension, combiner, shared_name=shared) for c in categorical_columns]


def indicator_column(categorical_column):
    return IndicatorColumn(categorical_column)


def make_parse_example_spec(feature_columns) -> dict:
    spec = {}
    for c in feature_columns:
        spec.update(c.parse_spec())
    return spec


# --------------------------------------------------------------------------------------------
# tables
# --------------------------------------------------------------------------------------------
def _arena_for(store, K: int) -> EmbeddingArena:
    name = f"emb{K}"
    ar = store.arenas.get(name)
    if ar is None:
        ar = EmbeddingArena(name, K, store.device, seed=store.seed + 1000 + K)
        store.arenas[name] = ar
    return ar


def _table_for(store, col: EmbeddingColumn, layer_scope: str) -> Tuple[EmbeddingArena, str]:
    """TF names: <scope>/<input_layer>/<col.name>/embedding_weights, shared tables
    <scope>/<input_layer>/<shared_name>/embedding_weights (created once, reused)."""
    ar = _arena_for(store, col.dimension)
    if col.shared_name:
        key = ("shared", col.shared_name)
        tname = store.shared_tables.get(key)
        if tname is None:
            tname = f"{layer_scope}/{col.shared_name}/embedding_weights"
            store.shared_tables[key] = tname
    else:
        tname = f"{layer_scope}/{col.name}/embedding_weights"
    if tname not in ar.tables:
        ar.add_table(tname, col.categorical_column.num_buckets)
    return ar, tname


def _as_matrix(cols: List[torch.Tensor]) -> torch.Tensor:
    """[B] id vectors -> [B, F].  Zero-copy when they are adjacent columns of one row-major
    matrix in this order (what the synthetic / TFRecord batcher produces)."""
    t0 = cols[0]
    F = len(cols)
    if F == 1:
        return t0.reshape(-1, 1).contiguous()
    if all(t.dim() == 1 and t.stride(0) == F and t.numel() == t0.numel()
           and t.untyped_storage().data_ptr() == t0.untyped_storage().data_ptr()
           and t.storage_offset() == t0.storage_offset() + i for i, t in enumerate(cols)):
        return torch.as_strided(t0, (t0.numel(), F), (F, 1))
    return torch.stack(cols, dim=1)


# --------------------------------------------------------------------------------------------
# input layers
# --------------------------------------------------------------------------------------------
def input_layer(features, feature_columns, _layer_name: Optional[str] = None) -> torch.Tensor:
    """fc.input_layer: per-column outputs concatenated in sorted(column.name) order (A-1)."""
    store = current_store()
    layer = _layer_name or store.auto_name("input_layer")
    scope = store.full_name(layer)
    cols = sorted(feature_columns, key=lambda c: c.name)
    dev = store.device
    widths = []
    for c in cols:
        if isinstance(c, NumericColumn):
            widths.append(int(np.prod(c.shape)))
        elif isinstance(c, EmbeddingColumn):
            widths.append(c.dimension)
        elif isinstance(c, IndicatorColumn):
            widths.append(c.categorical_column.num_buckets)
        else:
            raise TypeError(f"unsupported column {c}")
    D = sum(widths)
    B = _batch_size(features, cols[0])

    # register tables first (so that a building pass sees every table)
    tables = {}
    for c in cols:
        if isinstance(c, EmbeddingColumn):
            tables[c.name] = _table_for(store, c, scope)
    if store.building:
        return torch.zeros(B, D, device=dev)

    if all(isinstance(c, EmbeddingColumn) for c in cols) and len({c.dimension for c in cols}) == 1:
        idl = [c.categorical_column.ids(features, dev) for c in cols]
        if all(isinstance(i, torch.Tensor) for i in idl):
            # hot path: one fused multi-field gather straight into the [B, F*K] output
            ar = tables[cols[0].name][0]
            rb = store.row_base_tensor(ar, [tables[c.name][1] for c in cols])
            return ops.embedding_gather(store, _as_matrix(idl), ar, rb)

    parts = []
    for c in cols:
        if isinstance(c, NumericColumn):
            x = features[c.key]
            x = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x, dtype=np.float32))
            parts.append(x.to(dev, torch.float32).reshape(B, -1))
        elif isinstance(c, EmbeddingColumn):
            ar, tname = tables[c.name]
            ids = c.categorical_column.ids(features, dev)
            if isinstance(ids, Ragged):
                parts.append(ops.embedding_bag_mean(store, ids.values, ids.offsets, ar, tname))
            else:
                rb = store.row_base_tensor(ar, [tname])
                parts.append(ops.embedding_gather(store, ids.reshape(-1, 1).contiguous(), ar, rb))
        else:  # IndicatorColumn: multi-hot counts (A-5); only materialised on request
            ids = c.categorical_column.ids(features, dev)
            V = c.categorical_column.num_buckets
            mh = torch.zeros(B, V, device=dev)
            if isinstance(ids, Ragged):
                lens = ids.offsets[1:] - ids.offsets[:-1]
                rows = torch.repeat_interleave(torch.arange(B, device=dev), lens)
                ok = ids.values >= 0
                mh.index_put_((rows[ok], ids.values[ok]), torch.ones(int(ok.sum()), device=dev), accumulate=True)
            else:
                ok = ids >= 0
                mh[torch.arange(B, device=dev)[ok], ids[ok]] = 1.0
            parts.append(mh)
    if len(parts) > 1:
        ops.flush_lazy_gathers()             # (a gather left to its consumer, ops.gather_feeds_cross, must have run before the concat reads it)
        sparse.flush_batch()                 # (the same for lookups collected by sparse.batch_lookups)
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)


def input_layers_concat(features, feature_columns) -> torch.Tensor:
    """tf.concat([fc.input_layer(features, [c]) for c in feature_columns], axis=-1): one
    input_layer call PER COLUMN in LIST order (pnn.py:126-130, deepfm.py:187-190,204), so the
    TF layer names are input_layer, input_layer_1, ... and the field order is the list order,
    not the alphabetical one.  Single-valued columns of one width go through ONE fused gather."""
    store = current_store()
    dev = store.device
    cols = list(feature_columns)
    tables = []
    for c in cols:
        layer = store.auto_name("input_layer")
        if not isinstance(c, EmbeddingColumn):
            raise TypeError("input_layers_concat: embedding columns only")
        tables.append(_table_for(store, c, store.full_name(layer)))
    B = _batch_size(features, cols[0])
    if store.building:
        return torch.zeros(B, sum(c.dimension for c in cols), device=dev)
    idl = [c.categorical_column.ids(features, dev) for c in cols]
    if len({c.dimension for c in cols}) == 1 and all(isinstance(i, torch.Tensor) for i in idl) \
            and len({id(t[0]) for t in tables}) == 1:
        ar = tables[0][0]
        rb = store.row_base_tensor(ar, [t for _, t in tables])
        return ops.embedding_gather(store, _as_matrix(idl), ar, rb)
    parts = []
    for c, ids, (ar, tname) in zip(cols, idl, tables):
        if isinstance(ids, Ragged):
            parts.append(ops.embedding_bag_mean(store, ids.values, ids.offsets, ar, tname))
        else:
            rb = store.row_base_tensor(ar, [tname])
            parts.append(ops.embedding_gather(store, ids.reshape(-1, 1).contiguous(), ar, rb))
    if len(parts) > 1:
        ops.flush_lazy_gathers()
        sparse.flush_batch()
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)


_UNIT_OFFSETS = {}


def _unit_offsets(B: int, dev) -> torch.Tensor:
    """offsets [0, 1, ..., B] of B length-1 sequences (made once per batch size: not a launch per step)."""
    key = (int(B), str(dev))
    t = _UNIT_OFFSETS.get(key)
    if t is None:
        if len(_UNIT_OFFSETS) > 8:
            _UNIT_OFFSETS.clear()
        t = _UNIT_OFFSETS[key] = torch.arange(B + 1, device=dev, dtype=torch.int64)
    return t


def sequence_input_layer(features, feature_columns, max_length: Optional[int] = None):
    """tf.contrib.feature_column.sequence_input_layer (A-6): (B, T, H) zero padded with T the
    longest sequence of the batch (or `max_length` when given, for static shapes), and
    sequence_length (B,).  A single-valued [B] id tensor is a length-1 sequence."""
    store = current_store()
    layer = store.auto_name("sequence_input_layer")
    scope = store.full_name(layer)
    dev = store.device
    outs, lens = [], None
    for c in sorted(feature_columns, key=lambda c: c.name):
        ar, tname = _table_for(store, c, scope)
        if store.building:
            B = _batch_size(features, c)
            T = max_length or 1
            outs.append(torch.zeros(B, T, c.dimension, device=dev))
            lens = torch.zeros(B, dtype=torch.int32, device=dev)
            continue
        ids = c.categorical_column.ids(features, dev)
        if isinstance(ids, torch.Tensor):
            B = ids.numel()
            ids = Ragged(ids.contiguous(), _unit_offsets(B, dev))
            T = 1
        else:
            T = max_length or max(int((ids.offsets[1:] - ids.offsets[:-1]).max().item()), 1)
        o, l = ops.sequence_gather(store, ids.values, ids.offsets, ar, tname, T)
        outs.append(o)
        lens = l
    if len(outs) > 1:
        sparse.flush_batch()                 # (inside sparse.batch_lookups(): the outputs are read by the cat)
    return (outs[0] if len(outs) == 1 else torch.cat(outs, dim=-1)), lens


def _batch_size(features, col) -> int:
    x = features[col.key]
    if isinstance(x, Ragged):
        return int(x.offsets.numel() - 1)
    if isinstance(x, torch.Tensor):
        return int(x.shape[0])
    return len(x)
