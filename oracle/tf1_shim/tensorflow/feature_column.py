"""oracle/tf1_shim/tensorflow/feature_column.py — TEST INFRASTRUCTURE ONLY.

tf.feature_column / tf.contrib.feature_column restated for the reference's use
([TF-ext]).  Feature batch format of the shim:
  categorical key -> list (len B) of lists of str/bytes vocabulary keys ('' / unknown = OOV)
  numeric key     -> T of shape (B, 1)
"""
from __future__ import annotations

import math

import torch

import tensorflow as tf


def _key_str(k):
    return k.decode() if isinstance(k, (bytes, bytearray)) else str(k)


class _Numeric:
    def __init__(self, key, shape=(1,), default_value=None, dtype=None):
        self.key, self.name, self.shape, self.default_value = key, key, tuple(shape), default_value


class _VocabFile:
    """[TF-ext A-2] id = 0-based line index, num_oov_buckets=0, default_value=-1."""

    def __init__(self, key, vocabulary_file, sequence=False):
        self.key, self.name, self.is_sequence = key, key, sequence
        with open(vocabulary_file) as f:
            lines = [ln.rstrip("\n") for ln in f]
        while lines and lines[-1] == "":
            lines.pop()
        self.vocab = {w: i for i, w in enumerate(lines)}
        self.num_buckets = len(lines)

    def ids(self, features):
        """-> list (len B) of int lists (OOV = -1)."""
        out = []
        for row in features[self.key]:
            if isinstance(row, (str, bytes)):
                row = [row]
            out.append([self.vocab.get(_key_str(w), -1) for w in row])
        return out


class _Embedding:
    def __init__(self, cat, dimension, combiner="mean", shared_name=None):
        self.categorical_column, self.dimension, self.combiner = cat, int(dimension), combiner
        self.shared_name = shared_name
        self.key = cat.key
        self.name = f"{cat.key}_shared_embedding" if shared_name else f"{cat.key}_embedding"


class _Indicator:
    def __init__(self, cat):
        self.categorical_column, self.key, self.name = cat, cat.key, f"{cat.key}_indicator"

    @property
    def variable_shape(self):                 # TensorShape([num_buckets]) (ffm.py:234)
        return (self.categorical_column.num_buckets,)


def numeric_column(key, shape=(1,), default_value=None, dtype=None, normalizer_fn=None):
    return _Numeric(key, shape, default_value, dtype)


def categorical_column_with_vocabulary_file(key, vocabulary_file, vocabulary_size=None, **_kw):
    return _VocabFile(key, vocabulary_file)


def sequence_categorical_column_with_vocabulary_file(key, vocabulary_file, vocabulary_size=None, **_kw):
    return _VocabFile(key, vocabulary_file, sequence=True)


def embedding_column(categorical_column, dimension, combiner="mean", **_kw):
    return _Embedding(categorical_column, dimension, combiner)


def shared_embedding_columns(categorical_columns, dimension, combiner="mean", **_kw):
    """[TF-ext A-4] one table named after the sorted keys; columns returned in INPUT order."""
    shared = "_".join(sorted(c.key for c in categorical_columns)) + "_shared_embedding"
    return [_Embedding(c, dimension, combiner, shared_name=shared) for c in categorical_columns]


def indicator_column(categorical_column):
    return _Indicator(categorical_column)


def make_parse_example_spec(feature_columns):
    spec = {}
    for c in feature_columns:
        if isinstance(c, _Numeric):
            spec[c.key] = ("FixedLenFeature", c.shape, "float32", c.default_value)
        else:
            spec[c.key] = ("VarLenFeature", "string")
    return spec


def _table(col: _Embedding, layer_scope: str):
    """[TF-ext A-3/A-7] truncated_normal(stddev=1/sqrt(dim)) table `<scope>/<layer>/<col>/embedding_weights`;
    a shared table is created once, by the first layer that touches it."""
    g = tf.get_default_graph()
    init = tf.truncated_normal_initializer(stddev=1.0 / math.sqrt(col.dimension))
    V = col.categorical_column.num_buckets
    if col.shared_name:
        v = g.shared_tables.get(col.shared_name)
        if v is None:
            with tf.variable_scope(layer_scope), tf.variable_scope(col.shared_name):
                v = tf.get_variable("embedding_weights", (V, col.dimension), initializer=init)
            g.shared_tables[col.shared_name] = v
        return v
    with tf.variable_scope(layer_scope), tf.variable_scope(col.name):
        return tf.get_variable("embedding_weights", (V, col.dimension), initializer=init)


def _combine_mean(table, id_lists):
    """[TF-ext A-3] safe_embedding_lookup_sparse(combiner='mean'): ids < 0 dropped, sequential sum
    in id order divided by the count, zeros when nothing is left."""
    rows = []
    for ids in id_lists:
        valid = [i for i in ids if i >= 0]
        if not valid:
            rows.append(torch.zeros(table.t.shape[1], dtype=table.t.dtype))
            continue
        acc = table.t[valid[0]]
        for i in valid[1:]:
            acc = acc + table.t[i]
        rows.append(acc / len(valid) if len(valid) > 1 else acc)
    return torch.stack(rows, 0)


def input_layer(features, feature_columns, **_kw):
    """[TF-ext A-1] outputs concatenated in sorted(column.name) order, NOT list order."""
    layer = tf._unique("input_layer")
    parts = []
    for c in sorted(feature_columns, key=lambda c: c.name):
        if isinstance(c, _Numeric):
            x = tf._raw(features[c.key])
            parts.append(x.reshape(x.shape[0], -1).to(tf.FLOAT))
        elif isinstance(c, _Embedding):
            table = _table(c, layer)
            parts.append(_combine_mean(table, c.categorical_column.ids(features)))
        elif isinstance(c, _Indicator):
            # [TF-ext A-5] multi-hot counts; OOV contributes nothing
            ids = c.categorical_column.ids(features)
            V = c.categorical_column.num_buckets
            mh = torch.zeros(len(ids), V, dtype=tf.FLOAT)
            for b, row in enumerate(ids):
                for i in row:
                    if i >= 0:
                        mh[b, i] += 1.0
            parts.append(mh)
        else:
            raise TypeError(c)
    return tf.T(torch.cat(parts, dim=1))


def sequence_input_layer(features, feature_columns, **_kw):
    """[TF-ext A-6] (B, T_max_in_batch, H) zero padded + sequence_length (B,), which counts OOV
    entries (their rows are zero)."""
    layer = tf._unique("sequence_input_layer")
    outs, lens = [], None
    for c in sorted(feature_columns, key=lambda c: c.name):
        table = _table(c, layer)
        ids = c.categorical_column.ids(features)
        L = [len(r) for r in ids]
        Tm = max(L) if L else 0
        out = torch.zeros(len(ids), Tm, c.dimension, dtype=tf.FLOAT)
        rows = []
        for b, r in enumerate(ids):
            steps = [table.t[i] if i >= 0 else torch.zeros(c.dimension, dtype=tf.FLOAT) for i in r]
            steps += [torch.zeros(c.dimension, dtype=tf.FLOAT)] * (Tm - len(r))
            rows.append(torch.stack(steps, 0) if steps else torch.zeros(0, c.dimension, dtype=tf.FLOAT))
        out = torch.stack(rows, 0) if rows else out
        outs.append(out)
        lens = torch.tensor(L, dtype=torch.int64)
    return tf.T(outs[0] if len(outs) == 1 else torch.cat(outs, dim=-1)), tf.T(lens)
