"""CPU: librecalgo_host.so (include/recalgo_host.h) — the native TFRecord / tf.train.Example /
vocabulary plumbing — against the pure-Python codec (recalgorithm_amd/io/tfrecord.py) and the
known answers it is itself tested with; plus a decode-throughput figure."""
import ctypes
import os
import re
import time

import numpy as np
import pytest
import torch

from recalgorithm_amd import build as B
from recalgorithm_amd import feature_column as fc
from recalgorithm_amd.algorithm.utils import _Dataset, eval_input_fn, parse_example, train_input_fn
from recalgorithm_amd.io import native, synth, tfrecord as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def host_lib():
    B.build_host(verbose=False)
    return native.load()


def test_header_and_binding_agree():
    hdr = open(os.path.join(ROOT, "include", "recalgo_host.h")).read()
    declared = set(re.findall(r"\b(recalgo_\w+)\s*\(", hdr))
    assert declared == set(native.SIGNATURES), declared ^ set(native.SIGNATURES)
    lib = ctypes.CDLL(native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_crc32c_known_answers_native():
    assert native.crc32c(b"") == 0 and native.crc32c(b"123456789") == 0xE3069283
    assert native.crc32c(bytes(32)) == 0x8A9136AA and native.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    rng = np.random.default_rng(0)
    for n in (1, 7, 8, 9, 63, 1000, 4097):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert native.crc32c(b) == T.crc32c(b)


def test_vocabulary_lookup(tmp_path):
    p = tmp_path / "v.txt"
    p.write_bytes(b"userid_0\nuserid_1\n\nuserid_1\nlast_without_newline")
    v = native.Vocabulary(str(p))
    assert v.lookup(b"userid_0") == 0 and v.lookup("userid_1") == 1          # first occurrence wins
    assert v.lookup(b"") == 2 and v.lookup(b"last_without_newline") == 4 and v.lookup(b"nope") == -1
    with pytest.raises(IOError):
        native.Vocabulary(str(tmp_path / "missing.txt"))


def _dataset(tmp_path, n=300, seq_example=False):
    spec = synth.SynthSpec(n_fields=6, max_vocab=300, seed=5, oov_frac=0.1, with_dense=True, with_history=True,
                           with_tags=True)
    vocab_dir = str(tmp_path / "vocabulary") + "/"
    synth.write_vocabularies(spec, vocab_dir)
    path = str(tmp_path / ("seq.tfrecord" if seq_example else "ex.tfrecord"))
    synth.write_tfrecord(spec, path, n, chunk=64, as_sequence_example=seq_example)
    from recalgorithm_amd.algorithm._common import DENSE_FEATURES
    cols = [fc.numeric_column(k, default_value=0.0) for k in DENSE_FEATURES]
    cols += [fc.embedding_column(fc.categorical_column_with_vocabulary_file(nm, vocab_dir + nm + ".txt"), 8) for nm in spec.names
             if nm != "feedid"]
    feed = fc.sequence_categorical_column_with_vocabulary_file("feedid", vocab_dir + "feedid.txt")
    his = fc.sequence_categorical_column_with_vocabulary_file("his_read_comment_7d_seq", vocab_dir + "feedid.txt")
    cols += fc.shared_embedding_columns([feed, his], 8)
    cols += [fc.embedding_column(fc.categorical_column_with_vocabulary_file("manual_tag_list", vocab_dir + "manual_tag_id.txt"), 8)]
    labels = [fc.numeric_column("read_comment", default_value=0.0)]

    def parser(serialized):
        f = parse_example(serialized, fc.make_parse_example_spec(cols + labels))
        y = f.pop("read_comment")
        return f, {"read_comment": y}
    parser.columns_getter = lambda: (cols, labels)
    return spec, path, cols, labels, parser


def _encoded(cols, feats):
    """Python-path features (raw keys) -> ids, the way the model's input layers would encode them."""
    out = {}
    for c in cols:
        base = getattr(c, "categorical_column", c)
        x = feats[base.key]
        out[base.key] = x if isinstance(x, torch.Tensor) else base.ids({base.key: x}, torch.device("cpu"))
    return out


@pytest.mark.parametrize("seq_example", [False, True])
def test_native_batches_equal_python_batches(tmp_path, seq_example):
    spec, path, cols, labels, parser = _dataset(tmp_path, seq_example=seq_example)
    nat = eval_input_fn(path, parser, 64)
    assert isinstance(nat.upstream, native.NativeDataset)          # (behind the prefetch stage)
    py = _Dataset(path, parser, 64, 1, 0)
    nb, pb = list(nat), list(py)
    assert [b[1]["read_comment"].shape[0] for b in nb] == [64, 64, 64, 64, 44] == [b[1]["read_comment"].shape[0] for b in pb]
    for (nf, nl), (pf, pl) in zip(nb, pb):
        assert torch.equal(nl["read_comment"], pl["read_comment"])
        enc = _encoded(cols, pf)
        assert set(nf) == set(enc)
        for k, v in enc.items():
            if isinstance(v, torch.Tensor):
                assert isinstance(nf[k], torch.Tensor) and torch.equal(nf[k], v), k
            else:
                assert torch.equal(nf[k].values, v.values) and torch.equal(nf[k].offsets, v.offsets), k
    if seq_example:      # quirk B-9: list features written under feature_lists parse as empty
        assert all(int(f["his_read_comment_7d_seq"].values.numel()) == 0 for f, _ in nb)
    else:
        assert any(int(f["his_read_comment_7d_seq"].values.numel()) > 0 for f, _ in nb)


def test_native_repeat_shuffle_and_crc(tmp_path, monkeypatch):
    monkeypatch.setenv("RECALGO_SHUFFLE_SEED", "7")            # pinned: two runs give the same order
    spec, path, cols, labels, parser = _dataset(tmp_path, n=100)
    n3 = sum(l["read_comment"].shape[0] for _, l in train_input_fn(path, parser, 32, 3, 0))
    assert n3 == 300                                                          # repeat(3) then batch
    sizes = [l["read_comment"].shape[0] for _, l in train_input_fn(path, parser, 32, 3, 0)]
    assert sizes == [32] * 9 + [12]
    a = torch.cat([f["userid"] for f, _ in train_input_fn(path, parser, 100, 1, 16)])
    b = torch.cat([f["userid"] for f, _ in train_input_fn(path, parser, 100, 1, 16)])
    c = torch.cat([f["userid"] for f, _ in eval_input_fn(path, parser, 100)])
    assert torch.equal(a, b) and not torch.equal(a, c) and torch.equal(a.sort().values, c.sort().values)
    # shuffle(buffer).repeat(epochs), the reference's order (utils.py:19-21): every pass is a permutation of the file —
    # records of different epochs never mix (native reader and Python reader alike)
    monkeypatch.setenv("RECALGO_PYTHON_READER", "0")
    for reader in ("native", "python"):
        if reader == "python":
            monkeypatch.setenv("RECALGO_PYTHON_READER", "1")
        two = torch.cat([f["userid"] for f, _ in train_input_fn(path, parser, 50, 2, 16)]) if reader == "native" else \
            torch.cat([torch.as_tensor(_encoded(cols, f)["userid"]) for f, _ in train_input_fn(path, parser, 50, 2, 16)])
        assert two.numel() == 200
        assert torch.equal(two[:100].sort().values, c.sort().values) and torch.equal(two[100:].sort().values, c.sort().values)
        assert not torch.equal(two[:100], c)
    monkeypatch.delenv("RECALGO_PYTHON_READER")
    # without a pinned seed every run draws its own order (dataset.shuffle without a seed)
    monkeypatch.delenv("RECALGO_SHUFFLE_SEED")
    d = torch.cat([f["userid"] for f, _ in train_input_fn(path, parser, 100, 1, 16)])
    e = torch.cat([f["userid"] for f, _ in train_input_fn(path, parser, 100, 1, 16)])
    assert not torch.equal(d, e) and torch.equal(d.sort().values, e.sort().values)
    raw = bytearray(open(path, "rb").read())
    raw[40] ^= 0x10
    bad = str(tmp_path / "bad.tfrecord")
    open(bad, "wb").write(bytes(raw))
    ds = native.NativeDataset(bad, cols + labels, ["read_comment"], 32, verify_crc=True)
    with pytest.raises(IOError):
        list(ds)
    # a file cut in the middle of its last record, and an empty file
    cut = str(tmp_path / "cut.tfrecord")
    open(cut, "wb").write(open(path, "rb").read()[:-3])
    with pytest.raises(IOError):
        list(native.NativeDataset(cut, cols + labels, ["read_comment"], 32))
    empty = str(tmp_path / "empty.tfrecord")
    open(empty, "wb").close()
    assert list(native.NativeDataset(empty, cols + labels, ["read_comment"], 32, num_epochs=3)) == []
    # a required feature without default
    req = [fc.numeric_column("not_there")]
    with pytest.raises(ValueError):
        list(native.NativeDataset(path, req, [], 32))


def test_native_decode_throughput(tmp_path, capsys):
    spec = synth.SynthSpec(n_fields=26, max_vocab=100000, seed=9)
    vocab_dir = str(tmp_path / "vocabulary") + "/"
    synth.write_vocabularies(spec, vocab_dir)
    path = str(tmp_path / "big.tfrecord")
    n = 20000
    synth.write_tfrecord(spec, path, n)
    cols = [fc.embedding_column(fc.categorical_column_with_vocabulary_file(nm, vocab_dir + nm + ".txt"), 16) for nm in spec.names]
    labels = [fc.numeric_column("read_comment", default_value=0.0)]

    def parser(serialized):
        f = parse_example(serialized, fc.make_parse_example_spec(cols + labels))
        y = f.pop("read_comment")
        return f, {"read_comment": y}
    parser.columns_getter = lambda: (cols, labels)
    list(eval_input_fn(path, parser, 4096))                                   # warm: vocabularies loaded
    t0 = time.perf_counter()
    got = sum(l["read_comment"].shape[0] for _, l in eval_input_fn(path, parser, 4096))
    t_nat = time.perf_counter() - t0
    t0 = time.perf_counter()
    m = 4096
    f, _ = next(iter(_Dataset(path, parser, m, 1, 0)))
    for c in cols:
        c.categorical_column.ids({c.key: f[c.key]}, torch.device("cpu"))
    t_py = (time.perf_counter() - t0) * n / m
    assert got == n
    with capsys.disabled():
        print(f"\n[native reader] 26 string fields/example: native {n / t_nat:,.0f} ex/s, python {n / t_py:,.0f} ex/s "
              f"({t_py / t_nat:.0f}x)")
    assert t_nat < t_py


def test_native_reader_survives_fork(tmp_path):
    """The decode worker pool does not exist in a forked child (e.g. a fork-based data-loader worker): the
    child must fall back to inline decoding instead of waiting for the parent's threads."""
    import multiprocessing as mp
    spec, path, cols, labels, parser = _dataset(tmp_path, n=200)
    parent = [l["read_comment"].shape[0] for _, l in eval_input_fn(path, parser, 64)]      # pool is up in the parent
    q = mp.get_context("fork").Queue()

    def child():
        q.put([l["read_comment"].shape[0] for _, l in eval_input_fn(path, parser, 64)])
    p = mp.get_context("fork").Process(target=child)
    p.start()
    p.join(60)
    alive = p.is_alive()
    if alive:
        p.terminate()
    assert not alive, "forked child hung in the native reader"
    assert p.exitcode == 0 and q.get(timeout=5) == parent == [64, 64, 64, 8]


def test_native_decoder_rejects_or_survives_corrupt_examples(tmp_path):
    """Bit-flipped and random record payloads (valid TFRecord framing, crc check off): the decoder must either
    report a malformed batch or decode something — never read out of bounds or crash."""
    spec, path, cols, labels, parser = _dataset(tmp_path, n=40)
    recs = list(T.read_records(path)) if hasattr(T, "read_records") else None
    if recs is None:
        pytest.skip("python codec has no record iterator")
    rng = np.random.default_rng(3)
    outcomes = {"ok": 0, "error": 0}
    for case in range(60):
        muts = []
        for r in recs[:20]:
            b = bytearray(r)
            kind = case % 3
            if kind == 0 and b:                                   # flip a few bytes
                for _ in range(3):
                    b[int(rng.integers(0, len(b)))] ^= int(rng.integers(1, 256))
            elif kind == 1:                                       # truncate
                b = b[:int(rng.integers(0, len(b) + 1))]
            else:                                                 # random bytes
                b = bytearray(rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8).tobytes())
            muts.append(bytes(b))
        p = str(tmp_path / f"fuzz_{case}.tfrecord")
        T.write_records(p, muts)
        try:
            for _ in native.NativeDataset(p, cols + labels, ["read_comment"], 8):
                pass
            outcomes["ok"] += 1
        except (IOError, ValueError):
            outcomes["error"] += 1
    assert outcomes["ok"] + outcomes["error"] == 60 and outcomes["error"] > 0


def test_host_header_arg_types_match_binding():
    """include/recalgo_host.h parameter / return types against io/native.py's ctypes signatures."""
    from ctypes import c_char_p, c_float, c_int, c_int64, c_uint32, c_uint64, c_void_p
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "recalgo_host.h")).read(), flags=re.S)

    def ctype_of(decl):
        decl = decl.strip()
        if "char" in decl and decl.count("*") == 1:
            return c_char_p
        if "*" in decl:
            return c_void_p
        base = decl.rsplit(" ", 1)[0].replace("const ", "").strip()
        return {"int": c_int, "int64_t": c_int64, "uint64_t": c_uint64, "uint32_t": c_uint32, "float": c_float}[base]

    for name, (res, args) in native.SIGNATURES.items():
        m = re.search(r"([A-Za-z_0-9 \*]+?)\b" + name + r"\s*\(([^)]*)\)", src)
        assert m, name
        ret = m.group(1).strip()
        want = None if ret == "void" else (c_char_p if "char" in ret else (c_void_p if "*" in ret else
               {"int": c_int, "int64_t": c_int64, "uint32_t": c_uint32}[ret]))
        assert res is want, f"{name}: returns `{ret}`, binding {res}"
        params = [p for p in m.group(2).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(args), name
        for i, (decl, bound) in enumerate(zip(params, args)):
            got = ctype_of(decl)
            # byte buffers are passed as c_void_p / c_char_p interchangeably
            assert got is bound or {got, bound} == {c_char_p, c_void_p}, f"{name} arg {i} `{decl.strip()}`: binding {bound.__name__}"


def test_id_matrix_map_semantics_on_hand_encoded_records(tmp_path):
    """The fused one-pass decoder (recalgo_reader_id_matrix) against records written entry by entry: a repeated map
    key keeps its LAST entry (protobuf map semantics), an empty BytesList and a missing key give -1, features nobody
    asked for are skipped, the entries may come in any order (the position hint must not be trusted blindly), an
    unknown vocabulary key gives -1, and the direct-scan float accessor agrees with the indexed one."""
    vd = tmp_path / "v"
    vd.mkdir()
    (vd / "a.txt").write_bytes(b"a_0\na_1\na_2\n")
    (vd / "b.txt").write_bytes(b"b_0\nb_1\nkey_longer_than_the_inline_slot_of_the_table_0\n")
    ld, ent = T._ld, lambda name, feat: T._ld(1, T._ld(1, name) + T._ld(2, feat))
    blist = lambda *vals: T._ld(1, b"".join(T._ld(1, v) for v in vals))
    flist = lambda *vals: T._ld(2, T._ld(1, b"".join(np.float32(v).tobytes() for v in vals)))       # packed
    recs = [
        ld(1, ent(b"a", blist(b"a_1")) + ent(b"b", blist(b"b_0")) + ent(b"y", flist(1.0))),
        ld(1, ent(b"b", blist(b"b_1")) + ent(b"zzz", blist(b"ignored")) + ent(b"a", blist(b"a_2")) + ent(b"y", flist(0.0))),   # reordered + extra
        ld(1, ent(b"a", blist(b"a_0")) + ent(b"a", blist(b"a_2")) + ent(b"b", blist(b"nope")) + ent(b"y", flist(1.0))),          # repeated key: last wins
        ld(1, ent(b"a", blist(b"a_1")) + ent(b"a", blist()) + ent(b"y", flist(0.0))),                                              # last entry empty -> -1; b missing
        ld(1, ent(b"b", blist(b"key_longer_than_the_inline_slot_of_the_table_0")) + ent(b"a", blist(b"a_0", b"a_1"))),           # long key; two values; y missing
    ]
    path = str(tmp_path / "hand.tfrecord")
    T.write_records(path, recs)
    lib = native.load()
    va, vb = native.Vocabulary(str(vd / "a.txt")), native.Vocabulary(str(vd / "b.txt"))
    rd = ctypes.c_void_p(lib.recalgo_reader_open(path.encode(), 1))
    assert lib.recalgo_reader_next_batch(rd, 16) == 5
    keys = (ctypes.c_char_p * 2)(b"a", b"b")
    vs = (ctypes.c_void_p * 2)(va.h, vb.h)
    out, multi = np.full((5, 2), 99, np.int64), np.zeros(2, np.int32)
    yv = np.zeros((5, 1), np.float32)
    # float accessor BEFORE the id matrix: served by the direct scan (no index yet) ...
    assert lib.recalgo_reader_float_feature(rd, b"y", 1, ctypes.c_float(-7.0), 1, yv.ctypes.data_as(ctypes.c_void_p)) == 0
    assert yv.reshape(-1).tolist() == [1.0, 0.0, 1.0, 0.0, -7.0]
    assert lib.recalgo_reader_id_matrix(rd, 2, keys, vs, out.ctypes.data_as(ctypes.c_void_p), multi.ctypes.data_as(ctypes.c_void_p)) == 0
    assert out.tolist() == [[1, 0], [2, 1], [2, -1], [-1, -1], [0, 2]]
    assert multi.tolist() == [1, 0]
    # ... and after two more accessor calls by the index: same answers
    for _ in range(3):
        y2 = np.zeros((5, 1), np.float32)
        assert lib.recalgo_reader_float_feature(rd, b"y", 1, ctypes.c_float(-7.0), 1, y2.ctypes.data_as(ctypes.c_void_p)) == 0
        assert np.array_equal(y2, yv)
    offs, vals = np.zeros(6, np.int64), np.zeros(8, np.int64)
    nnz = lib.recalgo_reader_id_feature(rd, b"a", va.h, offs.ctypes.data_as(ctypes.c_void_p), vals.ctypes.data_as(ctypes.c_void_p), 8)
    assert nnz == 5 and offs.tolist() == [0, 1, 2, 3, 3, 5] and vals[:5].tolist() == [1, 2, 2, 0, 1]
    # a required float feature missing in a record is an error, not a silent zero
    assert lib.recalgo_reader_float_feature(rd, b"y", 1, ctypes.c_float(0.0), 0, yv.ctypes.data_as(ctypes.c_void_p)) == -1
    lib.recalgo_reader_close(rd)


def test_pipeline_batches_equal_the_synchronous_reader(tmp_path, monkeypatch):
    """recalgo_pipeline_* (producer + worker threads decoding several batches ahead) hands out exactly the batches of the
    synchronous accessors: same shuffle / repeat order (same seeded stream), same ids, floats and defaults, the last partial
    batch, and the Python parser's values."""
    spec = synth.SynthSpec(n_fields=7, max_vocab=400, seed=11, oov_frac=0.1, with_dense=True)
    vocab_dir = str(tmp_path / "vocabulary") + "/"
    synth.write_vocabularies(spec, vocab_dir)
    path = str(tmp_path / "single.tfrecord")
    synth.write_tfrecord(spec, path, 333, chunk=64)
    from recalgorithm_amd.algorithm._common import DENSE_FEATURES
    cols = [fc.numeric_column(k, default_value=0.0) for k in DENSE_FEATURES]
    cols += [fc.embedding_column(fc.categorical_column_with_vocabulary_file(nm, vocab_dir + nm + ".txt"), 8) for nm in spec.names]
    cols += [fc.numeric_column("never_written", default_value=2.5)]
    labels = [fc.numeric_column("read_comment", default_value=0.0)]

    def batches(pipeline, **kw):
        monkeypatch.setenv("RECALGO_READER_PIPELINE", "1" if pipeline else "0")
        ds = native.NativeDataset(path, cols + labels, ["read_comment"], 50, seed=5, **kw)
        assert (ds._pipeline_columns() is not None) == pipeline
        return list(ds)

    for kw in (dict(num_epochs=1), dict(num_epochs=3, shuffle_buffer_size=40)):
        a, b = batches(True, **kw), batches(False, **kw)
        assert len(a) == len(b) and [l["read_comment"].shape[0] for _, l in a] == [l["read_comment"].shape[0] for _, l in b]
        for (fa, la), (fb, lb) in zip(a, b):
            assert set(fa) == set(fb)
            for k in fa:
                assert fa[k].dtype == fb[k].dtype and fa[k].shape == fb[k].shape and torch.equal(fa[k], fb[k]), k
            assert torch.equal(la["read_comment"], lb["read_comment"])
            assert fa.packed_ids is not None and fa.packed_ids[1] == fb.packed_ids[1] and torch.equal(fa.packed_ids[0], fb.packed_ids[0])
            assert float(fa["never_written"].min()) == 2.5 == float(fa["never_written"].max())
    # an iterator dropped half way joins its threads (no hang, no leak of the handle)
    it = iter(native.NativeDataset(path, cols + labels, ["read_comment"], 50, num_epochs=None))
    for _ in range(3):
        next(it)
    it.close()


def test_pipeline_hands_a_multi_valued_column_back_to_the_ragged_reader(tmp_path, monkeypatch):
    """A column declared like any other categorical column whose records hold SEVERAL values (`manual_tag_list`) is found out
    by the asynchronous pipeline only at run time: recalgo_pipeline_next answers -2 naming the column (not the error state)
    and the dataset re-reads — same batches as with the pipeline switched off, the bag column ragged."""
    spec = synth.SynthSpec(n_fields=6, max_vocab=300, seed=5, oov_frac=0.1, with_tags=True)
    vocab_dir = str(tmp_path / "vocabulary") + "/"
    synth.write_vocabularies(spec, vocab_dir)
    path = str(tmp_path / "tags.tfrecord")
    synth.write_tfrecord(spec, path, 200, chunk=64)
    cols = [fc.embedding_column(fc.categorical_column_with_vocabulary_file(nm, vocab_dir + nm + ".txt"), 8) for nm in spec.names]
    cols += [fc.embedding_column(fc.categorical_column_with_vocabulary_file("manual_tag_list", vocab_dir + "manual_tag_id.txt"), 8)]
    labels = [fc.numeric_column("read_comment", default_value=0.0)]

    def batches(pipeline):
        monkeypatch.setenv("RECALGO_READER_PIPELINE", "1" if pipeline else "0")
        return list(native.NativeDataset(path, cols + labels, ["read_comment"], 64, num_epochs=2, shuffle_buffer_size=7, seed=3))

    a, b = batches(True), batches(False)
    assert len(a) == len(b) == 7
    ragged = 0
    for (fa, la), (fb, lb) in zip(a, b):
        assert set(fa) == set(fb) and torch.equal(la["read_comment"], lb["read_comment"])
        for k in fa:
            if isinstance(fb[k], torch.Tensor):
                assert torch.equal(fa[k], fb[k]), k
            else:
                ragged += 1
                assert k == "manual_tag_list" and torch.equal(fa[k].values, fb[k].values) and torch.equal(fa[k].offsets, fb[k].offsets)
    assert ragged == 7
