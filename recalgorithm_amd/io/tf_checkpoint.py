"""TensorFlow checkpoint files (the V2 "tensor bundle" tf.train.Saver / tf.estimator write: `<prefix>.index` +
`<prefix>.data-00000-of-0000N`, plus the text `checkpoint` state file) read and written WITHOUT TensorFlow, so that a
model_dir trained by a reference script (/root/reference algorithm/DeepFM/deepfm.py:289-296 RunConfig(model_dir=...,
save_checkpoints_steps=...)) loads straight into `Estimator.load_variables`, and weights trained
here can be handed back in the reference's own format.

PARITY UNPINNED: no TensorFlow and no checkpoint file exist in this environment, so this reader has never seen a file
written by TF.  It restates two published formats —
  * the LevelDB table format TF's `.index` file uses verbatim (tensorflow/core/lib/io/{table_builder,format,block}.cc,
    a copy of LevelDB's table/): blocks of prefix-compressed (key, value) entries with a restart array, a 5-byte block
    trailer (compression type, masked crc32c), an index block of block handles, a 48-byte footer ending in the magic
    0xdb4775248b80fb57; TF's BundleWriter writes it uncompressed;
  * tensorflow/core/protobuf/tensor_bundle.proto: key "" -> BundleHeaderProto {num_shards = 1, endianness = 2,
    version = 3}; key <variable name> -> BundleEntryProto {dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5,
    crc32c = 6 (fixed32, masked), slices = 7}; the data files hold the raw little-endian tensor bytes —
and is tested against the writer in this file plus one index file assembled byte by byte in the test from the format
description (tests/test_tf_checkpoint.py).  `scripts/tf_ckpt_to_npz.py` (which needs TensorFlow, on the reference
side) remains the pinned route.  Not supported: snappy-compressed blocks, partitioned variables (slices), string tensors.
"""
from __future__ import annotations

import os
import re
import struct
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

from .tfrecord import _fields, _ld, _read_varint, _varint, crc32c

_MAGIC = 0xdb4775248b80fb57
_FOOTER = 48
_BLOCK_SIZE = 4096                 # leveldb's default block_size
_RESTART_INTERVAL = 16

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


def _mask(c: int) -> int:
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _signed64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


# ---- LevelDB table: reading ---------------------------------------------------------------------------------------
def _read_block(buf: bytes, offset: int, size: int, verify: bool) -> bytes:
    if offset + size + 5 > len(buf):
        raise ValueError("tf checkpoint index: block handle points past the end of the file")
    contents = buf[offset:offset + size]
    ctype = buf[offset + size]
    if verify:
        (stored,) = struct.unpack_from("<I", buf, offset + size + 1)
        if stored != _mask(crc32c(contents + bytes([ctype]))):
            raise ValueError("tf checkpoint index: block checksum mismatch")
    if ctype != 0:
        raise NotImplementedError("tf checkpoint index: compressed blocks (type %d) are not supported" % ctype)
    return contents


def _block_entries(block: bytes) -> Iterator[Tuple[bytes, bytes]]:
    if len(block) < 4:
        raise ValueError("tf checkpoint index: block too short")
    (n_restarts,) = struct.unpack_from("<I", block, len(block) - 4)
    end = len(block) - 4 * (n_restarts + 1)
    if end < 0:
        raise ValueError("tf checkpoint index: bad restart array")
    pos, key = 0, b""
    while pos < end:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > end:
            raise ValueError("tf checkpoint index: corrupt block entry")
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _handle(buf: bytes, pos: int) -> Tuple[int, int, int]:
    off, pos = _read_varint(buf, pos)
    size, pos = _read_varint(buf, pos)
    return off, size, pos


def read_table(path: str, verify_crc: bool = True) -> Dict[bytes, bytes]:
    """Every (key, value) of a LevelDB-format table file, in key order."""
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < _FOOTER:
        raise ValueError(f"{path}: too short for a table file")
    footer = buf[-_FOOTER:]
    if struct.unpack_from("<Q", footer, 40)[0] != _MAGIC:
        raise ValueError(f"{path}: not a TensorFlow checkpoint index (bad table magic)")
    _, _, p = _handle(footer, 0)                              # metaindex handle (unused)
    ioff, isize, _ = _handle(footer, p)
    out: Dict[bytes, bytes] = {}
    for _, hv in _block_entries(_read_block(buf, ioff, isize, verify_crc)):
        boff, bsize, _ = _handle(hv, 0)
        for k, v in _block_entries(_read_block(buf, boff, bsize, verify_crc)):
            out[k] = v
    return out


# ---- tensor bundle --------------------------------------------------------------------------------------------------
def _parse_shape(buf: bytes) -> Tuple[int, ...]:
    dims = []
    for field, wt, v in _fields(buf):
        if field == 2 and wt == 2:                            # Dim {size = 1, name = 2}
            size = 0
            for f2, wt2, v2 in _fields(v):
                if f2 == 1 and wt2 == 0:
                    size = _signed64(v2)
            dims.append(size)
        elif field == 3 and wt == 0 and v:
            raise ValueError("tf checkpoint: tensor of unknown rank")
    return tuple(dims)


def _parse_entry(buf: bytes) -> dict:
    e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "slices": 0}
    for field, wt, v in _fields(buf):
        if field == 1 and wt == 0:
            e["dtype"] = v
        elif field == 2 and wt == 2:
            e["shape"] = _parse_shape(v)
        elif field == 3 and wt == 0:
            e["shard_id"] = v
        elif field == 4 and wt == 0:
            e["offset"] = v
        elif field == 5 and wt == 0:
            e["size"] = v
        elif field == 6 and wt == 5:
            e["crc32c"] = struct.unpack("<I", v)[0]
        elif field == 7:
            e["slices"] += 1
    return e


def _parse_header(buf: bytes) -> dict:
    h = {"num_shards": 1, "endianness": 0}
    for field, wt, v in _fields(buf):
        if field == 1 and wt == 0:
            h["num_shards"] = v
        elif field == 2 and wt == 0:
            h["endianness"] = v
    return h


def list_variables(prefix: str, verify_crc: bool = True) -> Dict[str, Tuple[tuple, type]]:
    """tf.train.list_variables: {name: (shape, numpy dtype)} of the checkpoint `<prefix>.index` describes."""
    out = {}
    for k, v in read_table(prefix + ".index", verify_crc).items():
        if k == b"":
            continue
        e = _parse_entry(v)
        out[k.decode()] = (e["shape"], _DTYPES.get(e["dtype"]))
    return out


def read_checkpoint(prefix: str, names: Optional[List[str]] = None, verify_crc: bool = True,
                    verify_data_crc: bool = False) -> Dict[str, np.ndarray]:
    """tf.train.load_checkpoint(prefix).get_tensor(name) for every (or the named) variable -> {name: array}.
    verify_crc: checksums of the index blocks; verify_data_crc: also of every tensor's bytes (pure-Python crc32c: slow
    for embedding tables)."""
    table = read_table(prefix + ".index", verify_crc)
    if b"" not in table:
        raise ValueError(f"{prefix}.index: no bundle header")
    header = _parse_header(table[b""])
    if header["endianness"] != 0:
        raise NotImplementedError("tf checkpoint: big-endian bundle")
    n_shards = header["num_shards"]
    shards: Dict[int, np.memmap] = {}
    out: Dict[str, np.ndarray] = {}
    want = None if names is None else set(names)
    for k, v in table.items():
        if k == b"":
            continue
        name = k.decode()
        if want is not None and name not in want:
            continue
        e = _parse_entry(v)
        if e["slices"]:
            raise NotImplementedError(f"tf checkpoint: {name} is a partitioned variable (slices)")
        dt = _DTYPES.get(e["dtype"])
        if dt is None:
            raise NotImplementedError(f"tf checkpoint: {name} has unsupported dtype enum {e['dtype']}")
        count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if count * np.dtype(dt).itemsize != e["size"]:
            raise ValueError(f"tf checkpoint: {name}: {e['size']} bytes do not match shape {e['shape']} of {np.dtype(dt).name}")
        sid = e["shard_id"]
        if sid not in shards:
            path = "%s.data-%05d-of-%05d" % (prefix, sid, n_shards)
            shards[sid] = np.memmap(path, dtype=np.uint8, mode="r") if os.path.getsize(path) else np.zeros(0, np.uint8)
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        if raw.shape[0] != e["size"]:
            raise ValueError(f"tf checkpoint: {name}: data file is shorter than the index says")
        if verify_data_crc and e["crc32c"] is not None and _mask(crc32c(raw.tobytes())) != e["crc32c"]:
            raise ValueError(f"tf checkpoint: {name}: tensor checksum mismatch")
        out[name] = np.array(raw.view(dt).reshape(e["shape"]))          # one copy, straight off the memmap slice
    if want is not None and want - set(out):
        raise KeyError(f"tf checkpoint: not in {prefix}: {sorted(want - set(out))[:5]}")
    return out


def latest_checkpoint(model_dir: str) -> Optional[str]:
    """tf.train.latest_checkpoint: the prefix named by `model_checkpoint_path` in <model_dir>/checkpoint."""
    state = os.path.join(model_dir, "checkpoint")
    if not os.path.isfile(state):
        return None
    with open(state) as f:
        m = re.search(r'^\s*model_checkpoint_path:\s*"([^"]*)"', f.read(), re.M)
    if not m:
        return None
    p = m.group(1)
    return p if os.path.isabs(p) else os.path.join(model_dir, p)


# ---- writing (hand weights back in the reference's format; the reader's test partner) --------------------------------
class _BlockBuilder:
    def __init__(self):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b""

    def add(self, key: bytes, value: bytes):
        shared = 0
        if self.count < _RESTART_INTERVAL:
            n = min(len(key), len(self.last))
            while shared < n and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last, self.count = key, self.count + 1

    def size(self) -> int:
        return len(self.buf) + 4 * (len(self.restarts) + 1)

    def finish(self) -> bytes:
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _emit_block(out: bytearray, contents: bytes) -> bytes:
    """append contents + trailer; -> its BlockHandle encoding"""
    handle = _varint(len(out)) + _varint(len(contents))
    out += contents + b"\x00" + struct.pack("<I", _mask(crc32c(contents + b"\x00")))
    return handle


def write_table(path: str, items: List[Tuple[bytes, bytes]]) -> None:
    """A LevelDB-format table of the (key, value) pairs (sorted by key here), uncompressed, 4 KB blocks."""
    items = sorted(items)
    out = bytearray()
    index = _BlockBuilder()
    blk = _BlockBuilder()
    for k, v in items:
        blk.add(k, v)
        if blk.size() >= _BLOCK_SIZE:
            index.add(blk.last, _emit_block(out, blk.finish()))          # separator: the block's last key
            blk = _BlockBuilder()
    if blk.count or not items:
        index.add(blk.last, _emit_block(out, blk.finish()))
    meta_handle = _emit_block(out, _BlockBuilder().finish())
    index_handle = _emit_block(out, index.finish())
    footer = meta_handle + index_handle
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC)
    with open(path, "wb") as f:
        f.write(out)


def _shape_proto(shape) -> bytes:
    return b"".join(_ld(2, b"\x08" + _varint(int(d))) for d in shape)


def write_checkpoint(prefix: str, arrays: Dict[str, np.ndarray], update_state_file: bool = True) -> None:
    """tf.train.Saver().save equivalent for plain arrays: `<prefix>.index`, `<prefix>.data-00000-of-00001` and (optionally)
    the `checkpoint` state file next to them."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    header = b"\x08\x01" + b"\x10\x00" + _ld(3, b"\x08\x01")              # num_shards 1, LITTLE, version {producer 1}
    items = [(b"", header)]
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as data:
        for name in sorted(arrays):
            a = np.asarray(arrays[name])
            if a.ndim and not a.flags.c_contiguous:               # (ascontiguousarray would turn a scalar into shape (1,))
                a = np.ascontiguousarray(a)
            if a.dtype not in _DTYPE_IDS:
                raise NotImplementedError(f"write_checkpoint: dtype {a.dtype} of {name}")
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            entry = b"\x08" + _varint(_DTYPE_IDS[a.dtype]) + _ld(2, _shape_proto(a.shape)) + b"\x18\x00"
            entry += b"\x20" + _varint(offset) + b"\x28" + _varint(len(raw)) + b"\x35" + struct.pack("<I", _mask(crc32c(raw)))
            items.append((name.encode(), entry))
            data.write(raw)
            offset += len(raw)
    write_table(prefix + ".index", items)
    if update_state_file:
        base = os.path.basename(prefix)
        with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"), "w") as f:
            f.write(f'model_checkpoint_path: "{base}"\nall_model_checkpoint_paths: "{base}"\n')
