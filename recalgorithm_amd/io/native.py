"""ctypes binding of librecalgo_host.so (include/recalgo_host.h): native TFRecord reading,
tf.train.Example decoding and vocabulary lookup — the host-side step right before the hot path
(SURVEY.md §8f-2).  `NativeDataset` is the drop-in producer behind `train_input_fn` /
`eval_input_fn` (algorithm/utils.py) when the parser was built from feature columns: it yields the
same (features, labels) batches as the pure-Python path, with categorical features already
encoded as ids (int64 [B], or Ragged for multi-valued / sequence columns)."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint32, c_uint64, c_void_p
from typing import Dict, List, Optional

import numpy as np
import torch

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_HERE, "librecalgo_host.so")

SIGNATURES = {
    "recalgo_host_abi_version": (c_int, []),
    "recalgo_crc32c": (c_uint32, [c_void_p, c_uint64]),
    "recalgo_vocab_open": (c_void_p, [c_char_p]),
    "recalgo_vocab_size": (c_int64, [c_void_p]),
    "recalgo_vocab_lookup": (c_int64, [c_void_p, c_char_p, c_uint64]),
    "recalgo_vocab_close": (None, [c_void_p]),
    "recalgo_reader_open": (c_void_p, [c_char_p, c_int]),
    "recalgo_reader_close": (None, [c_void_p]),
    "recalgo_reader_rewind": (c_int, [c_void_p]),
    "recalgo_reader_configure": (None, [c_void_p, c_int64, c_int64, c_uint64]),
    "recalgo_reader_error": (c_char_p, [c_void_p]),
    "recalgo_reader_next_batch": (c_int64, [c_void_p, c_int64]),
    "recalgo_reader_float_feature": (c_int, [c_void_p, c_char_p, c_int, c_float, c_int, c_void_p]),
    "recalgo_reader_id_feature": (c_int64, [c_void_p, c_char_p, c_void_p, c_void_p, c_void_p, c_int64]),
    "recalgo_reader_id_matrix": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "recalgo_pipeline_open": (c_void_p, [c_char_p, c_int, c_int64, c_int64, c_uint64, c_int64, c_int, c_void_p, c_void_p, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int]),
    "recalgo_pipeline_next": (c_int64, [c_void_p, c_void_p]),
    "recalgo_pipeline_release": (None, [c_void_p, c_int]),
    "recalgo_pipeline_ids": (c_void_p, [c_void_p, c_int]),
    "recalgo_pipeline_floats": (c_void_p, [c_void_p, c_int]),
    "recalgo_pipeline_error": (c_char_p, [c_void_p]),
    "recalgo_pipeline_threads": (c_int, [c_void_p]),
    "recalgo_pipeline_close": (None, [c_void_p]),
}

_lib = None


def load(path: str = LIB_PATH) -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build it with `python -m recalgorithm_amd.build`")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.recalgo_host_abi_version() != 1:
        raise RuntimeError("librecalgo_host.so ABI version mismatch")
    _lib = lib
    return lib


def available() -> bool:
    return os.path.exists(LIB_PATH)


def crc32c(data: bytes) -> int:
    return int(load().recalgo_crc32c(data, len(data)))


class Vocabulary:
    _cache: Dict[str, "Vocabulary"] = {}

    def __init__(self, path: str):
        self.path = path
        self.h = load().recalgo_vocab_open(path.encode())
        if not self.h:
            raise IOError(f"cannot read vocabulary file {path}")

    @classmethod
    def get(cls, path: str) -> "Vocabulary":
        v = cls._cache.get(path)
        if v is None:
            v = cls._cache[path] = Vocabulary(path)
        return v

    def __len__(self):
        return int(load().recalgo_vocab_size(self.h))

    def lookup(self, key) -> int:
        if isinstance(key, str):
            key = key.encode()
        return int(load().recalgo_vocab_lookup(self.h, key, len(key)))


class PackedBatch(dict):
    """A decoded feature batch whose single-valued id features are column views of ONE contiguous [B, F] int64 matrix,
    columns in SORTED key order (the order feature_column.input_layer consumes them in): `packed_ids = (matrix, keys)`.
    Estimator._pack_host_columns then moves that matrix with one contiguous copy instead of re-stacking F strided
    column views (26 passes over the matrix per batch on the training loop's thread)."""
    packed_ids = None


class _NotSingleValued(ValueError):
    """a column declared single-valued holds several values in some record"""


_MULTI_VALUED_KEYS: Dict[tuple, set] = {}    # (absolute path, mtime, size) of a file -> feature keys found to be multi-valued (process-wide)


class NativeDataset:
    """TFRecordDataset(filepath)[.shuffle(buf)].repeat(epochs).batch(bs).map(parse) with the record
    framing, Example decoding and vocabulary lookup done in C++."""

    def __init__(self, filepath: str, feature_columns, label_keys, batch_size: int, num_epochs: Optional[int] = 1,
                 shuffle_buffer_size: int = 0, seed: int = 0, verify_crc: bool = False):
        from ..feature_column import CategoricalColumn, NumericColumn
        self.filepath, self.bs = filepath, int(batch_size)
        self.epochs = -1 if num_epochs is None else int(num_epochs)
        self.shuffle, self.seed, self.verify = int(shuffle_buffer_size or 0), int(seed), verify_crc
        # keys seen holding several values per record: decoded as ragged features.  Remembered PER FILE, not per object: an input_fn
        # that builds a new dataset per call (the usual Estimator pattern) must not rediscover the column mid-stream every time
        try:
            st = os.stat(filepath)
            ident = (os.path.abspath(filepath), st.st_mtime_ns, st.st_size)
        except OSError:
            ident = (os.path.abspath(filepath), 0, 0)
        self._multi = _MULTI_VALUED_KEYS.setdefault(ident, set())
        self.label_keys = list(label_keys)
        self.numeric: List[NumericColumn] = []
        self.categorical: List[CategoricalColumn] = []
        seen = set()
        for c in feature_columns:
            base = getattr(c, "categorical_column", c)
            if base.key in seen:
                continue
            seen.add(base.key)
            if isinstance(base, NumericColumn):
                self.numeric.append(base)
            elif isinstance(base, CategoricalColumn):
                if not base.vocabulary_file:
                    raise ValueError(f"native decoding needs a vocabulary file for {base.key}")
                self.categorical.append(base)
            else:
                raise TypeError(base)

    # -- the asynchronous pipeline (recalgo_pipeline_*): every categorical column single-valued, numeric columns of fixed length --
    PIPELINE_DEPTH = 4

    def _pipeline_columns(self):
        """-> (id columns in sorted key order, numeric columns) when the asynchronous pipeline can serve this dataset, else None"""
        if os.environ.get("RECALGO_READER_PIPELINE", "1") == "0":
            return None
        if any(c.is_sequence or c.key in self._multi for c in self.categorical):
            return None
        if any(c.default_value is not None and not np.isscalar(c.default_value) for c in self.numeric):
            return None
        return sorted(self.categorical, key=lambda c: c.key), list(self.numeric)

    def _iter_pipeline(self, flat, numeric):
        """Batches from the ring of the C++ pipeline: decoded ahead of the consumer by its own threads; what is handed out is a
        copy of the slot (the slot goes back to the producer at once), so a batch stays valid for as long as it is referenced."""
        lib = load()
        vocabs = [Vocabulary.get(c.vocabulary_file) for c in flat]
        F, NF = len(flat), sum(int(np.prod(c.shape)) for c in numeric)
        keys = (ctypes.c_char_p * max(F, 1))(*[c.key.encode() for c in flat])
        vh = (c_void_p * max(F, 1))(*[v.h for v in vocabs])
        fkeys = (ctypes.c_char_p * max(len(numeric), 1))(*[c.key.encode() for c in numeric])
        fn = (ctypes.c_int32 * max(len(numeric), 1))(*[int(np.prod(c.shape)) for c in numeric])
        fdef = (ctypes.c_float * max(len(numeric), 1))(*[float(c.default_value or 0.0) for c in numeric])
        fhas = (ctypes.c_int32 * max(len(numeric), 1))(*[int(c.default_value is not None) for c in numeric])
        h = lib.recalgo_pipeline_open(self.filepath.encode(), int(self.verify), self.epochs, self.shuffle, self.seed, self.bs, F,
                                      keys, vh, len(numeric), fkeys, fn, fdef, fhas, self.PIPELINE_DEPTH, None, None, 0)
        if not h:
            raise IOError(f"cannot open {self.filepath}")
        try:
            slot = ctypes.c_int(0)
            id_keys = [c.key for c in flat]
            while True:
                B = int(lib.recalgo_pipeline_next(h, ctypes.byref(slot)))
                if B == 0:
                    return
                if B == -2:
                    raise _NotSingleValued(lib.recalgo_pipeline_error(h).decode())
                if B == -3:
                    raise ValueError(lib.recalgo_pipeline_error(h).decode())
                if B < 0:
                    raise IOError(f"{self.filepath}: {lib.recalgo_pipeline_error(h).decode()}")
                feats: Dict[str, object] = PackedBatch()
                if F:
                    src = (ctypes.c_int64 * (B * F)).from_address(lib.recalgo_pipeline_ids(h, slot.value))
                    tmat = torch.from_numpy(np.frombuffer(src, dtype=np.int64).reshape(B, F).copy())
                    feats.update(zip(id_keys, tmat.unbind(1)))      # (the F column views in one call: 26 slicing calls were 25 us per batch)
                    feats.packed_ids = (tmat, id_keys)
                if NF:
                    srcf = (ctypes.c_float * (B * NF)).from_address(lib.recalgo_pipeline_floats(h, slot.value))
                    fmat = torch.from_numpy(np.frombuffer(srcf, dtype=np.float32).reshape(B, NF).copy())
                    off = 0
                    for c in numeric:
                        n = int(np.prod(c.shape))
                        feats[c.key] = fmat[:, off:off + n].reshape((B,) + tuple(c.shape))
                        off += n
                lib.recalgo_pipeline_release(h, slot.value)
                labels = {k: feats.pop(k) for k in self.label_keys}
                if feats.packed_ids is not None and any(k in self.label_keys for k in feats.packed_ids[1]):
                    feats.packed_ids = None
                yield feats, labels
        finally:
            lib.recalgo_pipeline_close(h)

    def __iter__(self):
        cols = self._pipeline_columns()
        if cols is not None:
            # (a file whose "single-valued" columns turn out to hold several values per record is found out by the first batch
            # that has one — before anything of it was handed out when it is the first batch, which is where a dataset's layout
            # shows; later on the error is raised: the batches already consumed cannot be replayed in a shuffled stream)
            it = self._iter_pipeline(*cols)
            try:
                first = next(it)
            except StopIteration:
                return
            except _NotSingleValued as e:
                self._note_multi(str(e))           # (the next iteration of this dataset goes straight to the ragged accessors)
                it = None
            if it is not None:
                yield first
                try:
                    yield from it
                except _NotSingleValued as e:
                    self._note_multi(str(e))
                    raise ValueError(f"{self.filepath}: {e} — found after batches of the shuffled stream were already consumed, "
                                     "which the asynchronous pipeline cannot replay; read the file again (the column is now known "
                                     "as multi-valued for every dataset over this file in this process and is read as a ragged "
                                     "feature) or set RECALGO_READER_PIPELINE=0") from None
                return
        yield from self._iter_sync()

    def _note_multi(self, message: str) -> None:
        """`feature <key> holds more than one value in a record` (recalgo_pipeline_error): remember the column (a key may
        itself contain spaces: everything between the fixed prefix and suffix is the key)."""
        pre, suf = "feature ", " holds more than one value in a record"
        if message.startswith(pre) and message.endswith(suf):
            self._multi.add(message[len(pre):len(message) - len(suf)])

    def _iter_sync(self):
        from ..feature_column import Ragged
        lib = load()
        h = lib.recalgo_reader_open(self.filepath.encode(), int(self.verify))
        if not h:
            raise IOError(f"cannot open {self.filepath}")
        try:
            lib.recalgo_reader_configure(h, self.epochs, self.shuffle, self.seed)
            vocabs = {c.key: Vocabulary.get(c.vocabulary_file) for c in self.categorical}
            while True:
                B = int(lib.recalgo_reader_next_batch(h, self.bs))
                if B < 0:
                    raise IOError(f"{self.filepath}: {lib.recalgo_reader_error(h).decode()}")
                if B == 0:
                    return
                feats: Dict[str, object] = PackedBatch()
                for c in self.numeric:
                    n = int(np.prod(c.shape))
                    out = np.empty((B, n), dtype=np.float32)
                    has_def = c.default_value is not None
                    rc = lib.recalgo_reader_float_feature(h, c.key.encode(), n, float(c.default_value or 0.0), int(has_def),
                                                          out.ctypes.data_as(c_void_p))
                    if rc != 0:
                        raise ValueError(lib.recalgo_reader_error(h).decode())
                    feats[c.key] = torch.from_numpy(out.reshape((B,) + tuple(c.shape)))
                # single-valued id features: one [B, F] matrix from one parallel pass; a key that turns out to
                # hold several values per record (or is declared a sequence) goes through the ragged call
                flat = sorted((c for c in self.categorical if not c.is_sequence and c.key not in self._multi),
                              key=lambda c: c.key)
                if flat:
                    keys = (ctypes.c_char_p * len(flat))(*[c.key.encode() for c in flat])
                    vh = (c_void_p * len(flat))(*[vocabs[c.key].h for c in flat])
                    mat = np.empty((B, len(flat)), dtype=np.int64)
                    multi = np.zeros(len(flat), dtype=np.int32)
                    if lib.recalgo_reader_id_matrix(h, len(flat), keys, vh, mat.ctypes.data_as(c_void_p),
                                                    multi.ctypes.data_as(c_void_p)) != 0:
                        raise IOError(f"{self.filepath}: {lib.recalgo_reader_error(h).decode()}")
                    tmat = torch.from_numpy(mat)
                    for j, c in enumerate(flat):
                        if multi[j]:
                            self._multi.add(c.key)
                        else:
                            feats[c.key] = tmat[:, j]
                    if not multi.any():
                        feats.packed_ids = (tmat, [c.key for c in flat])
                for c in self.categorical:
                    if c.key in feats:
                        continue
                    offs = np.empty(B + 1, dtype=np.int64)
                    vals = np.empty(max(B, 1) * 4, dtype=np.int64)
                    nnz = int(lib.recalgo_reader_id_feature(h, c.key.encode(), vocabs[c.key].h, offs.ctypes.data_as(c_void_p),
                                                            vals.ctypes.data_as(c_void_p), vals.size))
                    if nnz > vals.size:
                        vals = np.empty(nnz, dtype=np.int64)
                        lib.recalgo_reader_id_feature(h, c.key.encode(), vocabs[c.key].h, offs.ctypes.data_as(c_void_p),
                                                      vals.ctypes.data_as(c_void_p), vals.size)
                    vals = vals[:nnz]
                    lens = np.diff(offs)
                    if not c.is_sequence and (lens <= 1).all():
                        dense = np.full(B, -1, dtype=np.int64)       # single-valued in this batch: [B] ids, -1 = missing
                        dense[lens == 1] = vals
                        feats[c.key] = torch.from_numpy(dense)
                    else:
                        feats[c.key] = Ragged(torch.from_numpy(vals.copy()), torch.from_numpy(offs))
                labels = {k: feats.pop(k) for k in self.label_keys}
                if feats.packed_ids is not None and any(k in self.label_keys for k in feats.packed_ids[1]):
                    feats.packed_ids = None        # (a label that is an id column: the matrix no longer mirrors the features)
                yield feats, labels
        finally:
            lib.recalgo_reader_close(h)
