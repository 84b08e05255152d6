"""TFRecord framing + tf.train.Example wire codec, host side, dependency-free.

Format spec taken from the reference's writer, /root/reference
dataset/wechat_algo_data1/DataGenerator.py:390-447 (`tf.io.TFRecordWriter`, `tf.train.Example`
/ `SequenceExample`) and SURVEY.md Appendix B-9:
    record = uint64 len | uint32 masked_crc32c(len) | bytes | uint32 masked_crc32c(bytes)
    mask(c) = ((c >> 15 | c << 17) + 0xa282ead8) mod 2^32
    Example{1: Features{1: map<string, Feature{1: BytesList | 2: FloatList | 3: Int64List}>}}
`parse_example` on SequenceExample bytes reads field 1 (context) and skips field 2
(feature_lists) exactly like tf.parse_example does (Appendix A-13).
"""
from __future__ import annotations

import struct
from typing import Dict, Iterable, Iterator, List, Sequence, Union

import numpy as np

# ---- crc32c (Castagnoli), table driven -------------------------------------------------------
_POLY = 0x82F63B78
_TABLE = np.zeros(256, dtype=np.uint32)
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (_POLY if _c & 1 else 0)
    _TABLE[_i] = _c
_TABLE_L = _TABLE.tolist()


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    t = _TABLE_L
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def write_records(path: str, records: Iterable[bytes]) -> int:
    n = 0
    with open(path, "wb") as f:
        for r in records:
            hdr = struct.pack("<Q", len(r))
            f.write(hdr)
            f.write(struct.pack("<I", masked_crc32c(hdr)))
            f.write(r)
            f.write(struct.pack("<I", masked_crc32c(r)))
            n += 1
    return n


def read_records(path: str, verify_crc: bool = False) -> Iterator[bytes]:
    with open(path, "rb") as f:
        while True:
            hdr = f.read(8)
            if not hdr:
                return
            if len(hdr) < 8:
                raise IOError(f"{path}: truncated record header")
            (n,) = struct.unpack("<Q", hdr)
            (hcrc,) = struct.unpack("<I", f.read(4))
            data = f.read(n)
            (dcrc,) = struct.unpack("<I", f.read(4))
            if len(data) < n:
                raise IOError(f"{path}: truncated record")
            if verify_crc and (hcrc != masked_crc32c(hdr) or dcrc != masked_crc32c(data)):
                raise IOError(f"{path}: crc mismatch")
            yield data


# ---- protobuf wire helpers ---------------------------------------------------------------------
def _varint(n: int) -> bytes:
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field: int, payload: bytes) -> bytes:
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


FeatureValue = Union[Sequence[bytes], Sequence[float], Sequence[int]]


def encode_feature(values, kind: str) -> bytes:
    if kind == "bytes":
        body = b"".join(_ld(1, v if isinstance(v, bytes) else str(v).encode()) for v in values)
        return _ld(1, body)
    if kind == "float":
        body = _ld(1, struct.pack(f"<{len(values)}f", *values)) if len(values) else b""
        return _ld(2, body)
    if kind == "int64":
        body = _ld(1, b"".join(_varint(int(v)) for v in values)) if len(values) else b""
        return _ld(3, body)
    raise ValueError(kind)


def encode_features(feats: Dict[str, tuple]) -> bytes:
    """feats: name -> (kind, values)."""
    out = bytearray()
    for name in feats:
        kind, values = feats[name]
        entry = _ld(1, name.encode()) + _ld(2, encode_feature(values, kind))
        out += _ld(1, entry)
    return bytes(out)


def encode_example(feats: Dict[str, tuple]) -> bytes:
    return _ld(1, encode_features(feats))


def encode_sequence_example(context: Dict[str, tuple], feature_lists: Dict[str, tuple]) -> bytes:
    """SequenceExample{1: context Features, 2: FeatureLists{1: map<string, FeatureList{1: Feature*}>}}
    as written by DataGenerator.py:429-443 (each list element is its own Feature)."""
    fl = bytearray()
    for name, (kind, values) in feature_lists.items():
        flist = b"".join(_ld(1, encode_feature([v], kind)) for v in values)
        fl += _ld(1, _ld(1, name.encode()) + _ld(2, flist))
    return _ld(1, encode_features(context)) + _ld(2, bytes(fl))


def _read_varint(buf: bytes, pos: int):
    n = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        n |= (b & 0x7F) << shift
        if not b & 0x80:
            return n, pos
        shift += 7


def _fields(buf: bytes):
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 2:
            n, pos = _read_varint(buf, pos)
            yield field, wt, buf[pos:pos + n]
            pos += n
        elif wt == 0:
            v, pos = _read_varint(buf, pos)
            yield field, wt, v
        elif wt == 5:
            yield field, wt, buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            yield field, wt, buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError(f"unsupported wire type {wt}")


def decode_feature(buf: bytes):
    for field, wt, payload in _fields(buf):
        if field == 1:   # BytesList
            return [p for f, _, p in _fields(payload) if f == 1]
        if field == 2:   # FloatList (packed or not)
            vals: List[float] = []
            for f, w, p in _fields(payload):
                if f == 1 and w == 2:
                    vals.extend(struct.unpack(f"<{len(p) // 4}f", p))
                elif f == 1 and w == 5:
                    vals.append(struct.unpack("<f", p)[0])
            return vals
        if field == 3:   # Int64List
            ivals: List[int] = []
            for f, w, p in _fields(payload):
                if f == 1 and w == 2:
                    pos = 0
                    while pos < len(p):
                        v, pos = _read_varint(p, pos)
                        ivals.append(v - (1 << 64) if v >> 63 else v)
                elif f == 1 and w == 0:
                    ivals.append(p - (1 << 64) if p >> 63 else p)
            return ivals
    return []


def decode_example(buf: bytes) -> Dict[str, list]:
    """Example or SequenceExample bytes -> {name: list}; only field 1 is read (A-13)."""
    out: Dict[str, list] = {}
    for field, wt, payload in _fields(buf):
        if field != 1 or wt != 2:
            continue            # SequenceExample.feature_lists (field 2) is skipped
        for f, _, entry in _fields(payload):
            if f != 1:
                continue
            name, feat = None, b""
            for ef, _, ep in _fields(entry):
                if ef == 1:
                    name = ep.decode()
                elif ef == 2:
                    feat = ep
            if name is not None:
                out[name] = decode_feature(feat)
    return out
