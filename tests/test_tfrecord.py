"""CPU: the data plumbing in front of the hot path — TFRecord framing (crc32c), the
tf.train.Example wire codec, tf.parse_example semantics and the input functions
(/root/reference algorithm/utils.py:4-46; dataset/wechat_algo_data1/DataGenerator.py:390-447;
SURVEY.md Appendix A-2, A-13, B-9).  Known answers are the published CRC-32C check values
(RFC 3720 B.4) and hand-assembled protobuf wire bytes."""
import struct

import numpy as np
import pytest
import torch

from recalgorithm_amd import feature_column as fc
from recalgorithm_amd.algorithm.utils import eval_input_fn, parse_example, train_input_fn
from recalgorithm_amd.io import synth, tfrecord as T


def test_crc32c_known_answers():
    assert T.crc32c(b"") == 0x00000000
    assert T.crc32c(b"123456789") == 0xE3069283                     # the CRC-32C "check" value
    assert T.crc32c(bytes(32)) == 0x8A9136AA                         # RFC 3720 B.4: 32 bytes of zeros
    assert T.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43                # 32 bytes of ones
    assert T.crc32c(bytes(range(32))) == 0x46DD794E                  # 32 incrementing bytes
    c = T.crc32c(b"123456789")
    assert T.masked_crc32c(b"123456789") == (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def test_record_framing_roundtrip_and_crc_check(tmp_path):
    recs = [b"", b"a", bytes(range(256)) * 3, b"\x00" * 17]
    path = str(tmp_path / "x.tfrecord")
    assert T.write_records(path, recs) == len(recs)
    raw = open(path, "rb").read()
    # layout of the first (empty) record: uint64 len | uint32 masked crc(len) | data | uint32 masked crc(data)
    assert raw[:8] == struct.pack("<Q", 0)
    assert raw[8:12] == struct.pack("<I", T.masked_crc32c(struct.pack("<Q", 0)))
    assert raw[12:16] == struct.pack("<I", T.masked_crc32c(b""))
    assert list(T.read_records(path, verify_crc=True)) == recs
    bad = bytearray(raw)
    bad[-6] ^= 0x01                                                  # flip one payload bit of the last record
    open(path, "wb").write(bytes(bad))
    assert len(list(T.read_records(path))) == len(recs)              # unchecked read still frames
    with pytest.raises(IOError):
        list(T.read_records(path, verify_crc=True))
    open(path, "wb").write(raw[:-3])                                 # truncated file
    with pytest.raises((IOError, struct.error)):
        list(T.read_records(path, verify_crc=True))


def test_example_wire_bytes_known_answer():
    # Example{features{feature{"a": Feature{int64_list{value:[1]}}}}} assembled by hand:
    #   Int64List{1: packed [1]}            0a 01 01
    #   Feature{3: Int64List}               1a 03 0a 01 01
    #   map entry {1: "a", 2: Feature}      0a 01 61 12 05 1a 03 0a 01 01
    #   Features{1: entry}                  0a 0a <entry>
    #   Example{1: Features}                0a 0c <features>
    want = bytes.fromhex("0a0c" "0a0a" "0a0161" "1205" "1a03" "0a0101")
    assert T.encode_example({"a": ("int64", [1])}) == want
    # Feature{1: BytesList{1: "xy"}} and Feature{2: FloatList{1: packed [1.0]}}
    assert T.encode_feature([b"xy"], "bytes") == bytes.fromhex("0a04" "0a027879")
    assert T.encode_feature([1.0], "float") == bytes.fromhex("1206" "0a04" "0000803f")
    assert T.decode_example(want) == {"a": [1]}


def test_example_codec_roundtrip_edge_cases():
    feats = {
        "ids": ("int64", [0, 1, -1, 2 ** 62, -(2 ** 63)]),          # negative varints are 10 bytes
        "f": ("float", [0.0, -1.5, 3.4028234663852886e38]),
        "s": ("bytes", [b"", b"userid_17", "feedid_3"]),
        "empty_b": ("bytes", []), "empty_f": ("float", []), "empty_i": ("int64", []),
    }
    dec = T.decode_example(T.encode_example(feats))
    assert dec["ids"] == feats["ids"][1]
    assert np.array_equal(np.float32(dec["f"]), np.float32(feats["f"][1]))
    assert dec["s"] == [b"", b"userid_17", b"feedid_3"]
    assert dec["empty_b"] == [] and dec["empty_f"] == [] and dec["empty_i"] == []
    # unpacked repeated scalars (what some writers emit) decode too
    unpacked_float = bytes.fromhex("12" "0a") + bytes.fromhex("0d0000803f" "0d00000040")
    assert T.decode_feature(unpacked_float) == [1.0, 2.0]


def test_parse_example_semantics_and_sequence_example_quirk():
    """tf.parse_example: FixedLen float -> (B, 1) with the default when absent; VarLen -> ragged lists.
    Quirk B-9: the checked-in ETL writes SequenceExample with the list features under feature_lists,
    which tf.parse_example does not read -> they parse as EMPTY."""
    cols = [fc.numeric_column("d0", default_value=0.0), fc.numeric_column("read_comment", default_value=0.0)]
    cat = fc.categorical_column_with_identity("userid", 10)
    cat.vocabulary_file = None
    spec = fc.make_parse_example_spec(cols)
    spec["userid"] = ("varlen", np.bytes_)
    spec["his_read_comment_7d_seq"] = ("varlen", np.bytes_)
    ex0 = T.encode_example({"d0": ("float", [2.5]), "userid": ("bytes", [b"userid_3"]),
                            "his_read_comment_7d_seq": ("bytes", [b"feedid_1", b"feedid_2"]),
                            "read_comment": ("float", [1.0])})
    ex1 = T.encode_example({"userid": ("bytes", [b""])})                       # d0 / label absent -> defaults
    ex2 = T.encode_sequence_example({"d0": ("float", [7.0]), "userid": ("bytes", [b"userid_9"])},
                                    {"his_read_comment_7d_seq": ("bytes", [b"feedid_1", b"feedid_2"])})
    out = parse_example([ex0, ex1, ex2], spec)
    assert out["d0"].shape == (3, 1) and out["d0"].dtype == torch.float32
    assert out["d0"].flatten().tolist() == [2.5, 0.0, 7.0]
    assert out["read_comment"].flatten().tolist() == [1.0, 0.0, 0.0]
    assert out["userid"] == [[b"userid_3"], [b""], [b"userid_9"]]
    assert out["his_read_comment_7d_seq"] == [[b"feedid_1", b"feedid_2"], [], []]   # [2]: feature_lists skipped
    with pytest.raises(ValueError):
        parse_example([ex1], {"d0": ("fixed", np.float32, (1,), None)})            # required, missing


def _write_dataset(tmp_path, n=50, fields=5):
    spec = synth.SynthSpec(n_fields=fields, max_vocab=200, seed=3, oov_frac=0.1, with_dense=True,
                           with_history=True, with_tags=True)
    vocab_dir = str(tmp_path / "vocabulary") + "/"
    synth.write_vocabularies(spec, vocab_dir)
    path = str(tmp_path / "train.tfrecord")
    assert synth.write_tfrecord(spec, path, n, chunk=16) == n
    return spec, vocab_dir, path


def test_input_fns_batching_epochs_and_shuffle(tmp_path, monkeypatch):
    monkeypatch.setenv("RECALGO_SHUFFLE_SEED", "11")       # (unset: a fresh order per run, like tf's unseeded shuffle)
    spec, vocab_dir, path = _write_dataset(tmp_path)
    cols = [fc.numeric_column("read_comment", default_value=0.0)] + \
           [fc.categorical_column_with_vocabulary_file(n, vocab_dir + n + ".txt") for n in spec.names]

    def parser(serialized):
        f = parse_example(serialized, fc.make_parse_example_spec(cols))
        y = f.pop("read_comment")
        return f, {"read_comment": y}
    batches = list(eval_input_fn(path, parser, 16))
    assert [b[1]["read_comment"].shape[0] for b in batches] == [16, 16, 16, 2]     # last partial batch kept
    assert sum(b[1]["read_comment"].shape[0] for b in train_input_fn(path, parser, 16, 3, 0)) == 150   # repeat(3)
    a = [f["userid"] for f, _ in train_input_fn(path, parser, 50, 1, 10)]
    b = [f["userid"] for f, _ in train_input_fn(path, parser, 50, 1, 10)]
    c = [f["userid"] for f, _ in eval_input_fn(path, parser, 50)]
    assert a == b and a != c and sorted(map(tuple, a[0])) == sorted(map(tuple, c[0]))   # seeded shuffle, same multiset
    # the written keys decode back to the ids the generator drew ('' = OOV -> -1, A-2)
    ids, labels, *_ = synth.make_id_batch(spec, 16, 0)
    f0, l0 = batches[0]
    for j, nm in enumerate(spec.names):
        col = cols[1 + j]
        got = col.ids({nm: f0[nm]}, torch.device("cpu"))
        assert got.tolist() == ids[:, j].tolist()
    assert l0["read_comment"].flatten().tolist() == labels[:, 0].tolist()


def test_prefetch_stage_order_errors_and_early_stop(monkeypatch):
    """dataset.prefetch(1) (utils.py:24): same batches in the same order, upstream errors reach the consumer, a consumer
    that stops early does not leave the producer blocked."""
    import threading
    import time
    from recalgorithm_amd.algorithm.utils import _Prefetch
    assert list(_Prefetch(range(50), 1)) == list(range(50))
    assert list(_Prefetch([], 1)) == []

    def broken():
        yield 1
        raise ValueError("decode failed")
    it = iter(_Prefetch(broken(), 1))
    assert next(it) == 1
    with pytest.raises(ValueError, match="decode failed"):
        next(it)
    produced = []

    def slow():
        for i in range(1000):
            produced.append(i)
            yield i
    g = iter(_Prefetch(slow(), 1))
    assert [next(g), next(g)] == [0, 1]
    g.close()                                         # consumer gives up
    time.sleep(0.3)
    n = len(produced)
    time.sleep(0.2)
    assert len(produced) == n and n <= 5              # the producer stopped, at most the prefetch depth ahead
    assert not [t for t in threading.enumerate() if t.name == "recalgo-prefetch" and t.is_alive()]
