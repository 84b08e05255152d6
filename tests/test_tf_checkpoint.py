"""TensorFlow V2 checkpoint files without TensorFlow (recalgorithm_amd/io/tf_checkpoint.py).
PARITY UNPINNED against real TF output (none exists here): the reader is checked against (1) an index file assembled
byte by byte below from the LevelDB table format and tensor_bundle.proto, independent of the module's writer, and
(2) the module's own writer over many variables (multi-block tables, prefix compression, restarts)."""
import os
import struct

import numpy as np
import pytest

from recalgorithm_amd.io import tf_checkpoint as C
from recalgorithm_amd.io.tfrecord import crc32c


def _mask(c):
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _trailer(contents: bytes) -> bytes:
    return b"\x00" + struct.pack("<I", _mask(crc32c(contents + b"\x00")))


def test_reader_on_a_hand_assembled_bundle(tmp_path):
    """One data block with two entries — "" -> BundleHeaderProto, "w/kernel" -> BundleEntryProto — an empty metaindex
    block, an index block with one handle, the 48-byte footer; every byte written out from the format description."""
    w = np.array([[1.5, -2.0, 0.25], [4.0, 8.0, -16.0]], dtype="<f4")
    raw = w.tobytes()
    prefix = str(tmp_path / "model.ckpt-7")
    open(prefix + ".data-00000-of-00001", "wb").write(b"\xAA" * 8 + raw)           # the tensor starts at offset 8
    header = bytes([0x08, 0x01,                  # num_shards = 1
                    0x1A, 0x02, 0x08, 0x01])     # version { producer: 1 }   (endianness LITTLE = 0: omitted, proto3)
    entry = bytes([0x08, 0x01,                                       # dtype = DT_FLOAT
                   0x12, 0x08, 0x12, 0x02, 0x08, 0x02, 0x12, 0x02, 0x08, 0x03,   # shape { dim {size: 2} dim {size: 3} }
                   0x20, 0x08,                                       # offset = 8      (shard_id 0: omitted)
                   0x28, 0x18,                                       # size = 24
                   0x35]) + struct.pack("<I", _mask(crc32c(raw)))    # crc32c (fixed32, masked)
    # data block: entries (shared, non_shared, value_len, key delta, value); restart array [0]; num_restarts = 1
    data = bytes([0, 0, len(header)]) + header
    data += bytes([0, 8, len(entry)]) + b"w/kernel" + entry
    data += struct.pack("<I", 0) + struct.pack("<I", 1)
    meta = struct.pack("<I", 0) + struct.pack("<I", 1)                              # empty block
    f = bytearray()
    data_off = len(f); f += data + _trailer(data)
    meta_off = len(f); f += meta + _trailer(meta)
    handle = bytes([data_off, len(data)])                                           # varints < 128
    index = bytes([0, 8, len(handle)]) + b"w/kernel" + handle + struct.pack("<I", 0) + struct.pack("<I", 1)
    index_off = len(f); f += index + _trailer(index)
    assert max(meta_off, len(meta), index_off, len(index)) < 128
    footer = bytes([meta_off, len(meta), index_off, len(index)])
    f += footer + b"\x00" * (40 - len(footer)) + bytes([0x57, 0xFB, 0x80, 0x8B, 0x24, 0x75, 0x47, 0xDB])
    open(prefix + ".index", "wb").write(f)
    open(tmp_path / "checkpoint", "w").write('model_checkpoint_path: "model.ckpt-7"\nall_model_checkpoint_paths: "model.ckpt-7"\n')

    assert C.latest_checkpoint(str(tmp_path)) == prefix
    assert C.list_variables(prefix) == {"w/kernel": ((2, 3), np.float32)}
    got = C.read_checkpoint(prefix, verify_data_crc=True)
    assert list(got) == ["w/kernel"] and got["w/kernel"].dtype == np.float32 and np.array_equal(got["w/kernel"], w)
    # corruption is noticed: a flipped byte in the data block, a wrong tensor checksum, a wrong magic
    bad = bytearray(f); bad[data_off + 5] ^= 1
    open(prefix + ".index", "wb").write(bad)
    with pytest.raises(ValueError, match="checksum"):
        C.read_checkpoint(prefix)
    open(prefix + ".index", "wb").write(f)
    open(prefix + ".data-00000-of-00001", "wb").write(b"\xAA" * 8 + raw[:-1] + b"\x00")
    with pytest.raises(ValueError, match="tensor checksum"):
        C.read_checkpoint(prefix, verify_data_crc=True)
    bad = bytearray(f); bad[-1] ^= 0xFF
    open(prefix + ".index", "wb").write(bad)
    with pytest.raises(ValueError, match="magic"):
        C.read_checkpoint(prefix)


def test_writer_reader_round_trip_many_variables(tmp_path):
    rng = np.random.default_rng(3)
    arrays = {"global_step": np.array(12345, dtype=np.int64), "beta1_power": np.array(0.9, dtype=np.float32)}
    for i in range(300):                         # > 4 KB of index entries: several data blocks, shared key prefixes, restarts
        shape = [(), (7,), (3, 5), (2, 3, 4)][i % 4]
        arrays[f"dnn_part/dense_{i}/kernel"] = rng.standard_normal(shape).astype(np.float32)
        arrays[f"dnn_part/dense_{i}/kernel/Adam"] = rng.standard_normal(shape).astype(np.float32)
    arrays["ids"] = rng.integers(-5, 5, size=(4, 2)).astype(np.int32)
    arrays["empty"] = np.zeros((0, 16), dtype=np.float32)
    prefix = str(tmp_path / "m" / "model.ckpt-12345")
    C.write_checkpoint(prefix, arrays)
    assert os.path.getsize(prefix + ".index") > 3 * 4096
    assert C.latest_checkpoint(str(tmp_path / "m")) == prefix
    got = C.read_checkpoint(prefix, verify_data_crc=True)
    assert set(got) == set(arrays)
    for k, a in arrays.items():
        assert got[k].dtype == a.dtype and got[k].shape == a.shape and np.array_equal(got[k], a), k
    some = C.read_checkpoint(prefix, names=["ids", "global_step"])
    assert set(some) == {"ids", "global_step"} and int(some["global_step"]) == 12345
    with pytest.raises(KeyError):
        C.read_checkpoint(prefix, names=["nope"])
    lv = C.list_variables(prefix)
    assert lv["dnn_part/dense_2/kernel"] == ((3, 5), np.float32) and lv["global_step"] == ((), np.int64)
