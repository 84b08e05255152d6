"""Shared helpers of the model scripts — drop-in for /root/reference algorithm/utils.py:
`train_input_fn`, `eval_input_fn` (same signatures, :4-46), `to_sparse_tensor` (:49-64),
`index_from_upper_triangular` (:67-82), plus `parse_example` (tf.parse_example stand-in).

The input functions return a re-iterable of (features, labels) host batches; the last partial
batch is kept (no drop_remainder), like the reference's `dataset.batch(batch_size)`.
"""
from __future__ import annotations

import random
from typing import Callable, Dict, Iterator, List

import numpy as np
import torch

from ..io import tfrecord


def parse_example(serialized: List[bytes], spec: Dict[str, tuple]) -> Dict[str, object]:
    """tf.parse_example(serialized, features=make_parse_example_spec(columns)).
    FixedLen float features -> float32 tensor (B, *shape) with default_value when absent;
    VarLen features -> list (len B) of lists of raw values (bytes / int)."""
    decoded = [tfrecord.decode_example(s) for s in serialized]
    out: Dict[str, object] = {}
    for key, sp in spec.items():
        if sp[0] == "fixed":
            _, dtype, shape, default = sp
            n = int(np.prod(shape))
            arr = np.empty((len(decoded), n), dtype=np.float32)
            for i, ex in enumerate(decoded):
                v = ex.get(key)
                if v is None or len(v) == 0:
                    if default is None:
                        raise ValueError(f"feature {key} is required but missing")
                    arr[i] = default
                else:
                    arr[i] = v[:n]
            out[key] = torch.from_numpy(arr.reshape((len(decoded),) + tuple(shape)))
        else:
            out[key] = [ex.get(key, []) for ex in decoded]
    return out


class _Dataset:
    """TFRecordDataset(filepath)[.shuffle(buf)].repeat(epochs).batch(bs).map(parser)."""

    def __init__(self, filepath, example_parser, batch_size, num_epochs, shuffle_buffer_size, seed=0):
        self.filepath, self.parser, self.bs = filepath, example_parser, int(batch_size)
        self.epochs, self.buf, self.seed = num_epochs, int(shuffle_buffer_size or 0), seed

    def _records(self) -> Iterator[bytes]:
        paths = [self.filepath] if isinstance(self.filepath, str) else list(self.filepath)
        rng = random.Random(self.seed)
        ep = 0
        while self.epochs is None or ep < self.epochs:
            ep += 1
            src = (r for p in paths for r in tfrecord.read_records(p))
            if self.buf > 0:       # tf.data shuffle-buffer semantics
                buf: List[bytes] = []
                for r in src:
                    if len(buf) < self.buf:
                        buf.append(r)
                        continue
                    j = rng.randrange(len(buf))
                    yield buf[j]
                    buf[j] = r
                rng.shuffle(buf)
                yield from buf
            else:
                yield from src

    def __iter__(self):
        batch: List[bytes] = []
        for r in self._records():
            batch.append(r)
            if len(batch) == self.bs:
                yield self.parser(batch)
                batch = []
        if batch:
            yield self.parser(batch)


class _Prefetch:
    """dataset.prefetch(n) (utils.py:24,43): a background thread runs the upstream pipeline up to n batches ahead of the
    consumer, so decoding batch i+1 overlaps the host side of step i (packing, the host-to-device copy, the launches; the
    native decoder releases the GIL, and the device runs asynchronously anyway).
    Still an iterable of batches: type checks on the dataset itself look at `.upstream`."""
    _END = object()

    def __init__(self, upstream, n: int = 1):
        self.upstream, self.n = upstream, max(int(n), 1)

    def __iter__(self):
        import queue
        import threading
        q: "queue.Queue" = queue.Queue(maxsize=self.n)
        stop = threading.Event()

        def put(item) -> bool:
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    continue
            return False

        def work():
            try:
                for item in self.upstream:
                    if not put(item):
                        return
                put(self._END)
            except BaseException as e:  # noqa: BLE001   (delivered to the consumer, which re-raises it)
                put(e)
        t = threading.Thread(target=work, name="recalgo-prefetch", daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is self._END:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()                   # a consumer that stops early (steps=..., an error) releases the producer


def _dataset(filepath, example_parser, batch_size, num_epochs, shuffle_buffer_size):
    return _Prefetch(_dataset_inline(filepath, example_parser, batch_size, num_epochs, shuffle_buffer_size), 1)


def _dataset_inline(filepath, example_parser, batch_size, num_epochs, shuffle_buffer_size):
    """The native (C++) reader/decoder serves parsers that declare their feature columns
    (`example_parser.columns_getter`, set by the model scripts) over a single file; anything else —
    an arbitrary parser callable, a list of files — takes the pure-Python path.  Both yield the same
    batches (tests/test_native_reader.py); only the shuffle order differs (different seeded generators)."""
    import os
    from ..io import native
    seed = _shuffle_seed()
    getter = getattr(example_parser, "columns_getter", None)
    if getter is not None and isinstance(filepath, str) and native.available() \
            and os.environ.get("RECALGO_PYTHON_READER", "0") != "1":
        total, label = getter()
        try:
            return native.NativeDataset(filepath, list(total) + list(label), [c.key for c in label], batch_size,
                                        num_epochs, shuffle_buffer_size, seed=seed)
        except (ValueError, TypeError):
            pass                     # e.g. identity columns without a vocabulary file
    return _Dataset(filepath, example_parser, batch_size, num_epochs, shuffle_buffer_size, seed=seed)


def _shuffle_seed() -> int:
    """dataset.shuffle(buffer) without a seed (utils.py:19) draws a fresh order every run; RECALGO_SHUFFLE_SEED pins it."""
    import os
    e = os.environ.get("RECALGO_SHUFFLE_SEED")
    return int(e) if e is not None else int.from_bytes(os.urandom(8), "little") >> 1


def train_input_fn(filepath, example_parser: Callable, batch_size, num_epochs, shuffle_buffer_size):
    return _dataset(filepath, example_parser, batch_size, num_epochs, shuffle_buffer_size)


def eval_input_fn(filepath, example_parser: Callable, batch_size):
    return _dataset(filepath, example_parser, batch_size, 1, 0)


def to_sparse_tensor(one_hot_tensor: torch.Tensor):
    """one-hot / multi-hot (B, V) -> (indices (nnz, 2), values (nnz,), dense_shape); values are
    the column indices, as the reference builds for safe_embedding_lookup_sparse."""
    idx = torch.nonzero(one_hot_tensor != 0)
    return idx, idx[:, 1], tuple(one_hot_tensor.shape)


def index_from_upper_triangular(i: int, j: int, n: int) -> int:
    """Flattened index of entry (i, j), i < j, of an n x n strict upper triangle."""
    return i * (2 * n - i - 1) // 2 + (j - i - 1)
