"""CPU (no kernels launched): host logic of the mirrored reference interface.
  * every mirrored model_fn, built from the mirror's own create_feature_columns(), creates exactly
    the variables (TF names and shapes) that the reference's model_fn created when it was executed
    for the golden vectors (oracle/gen_golden.py) — the variable-registration pass is launch-free;
  * a1: vocabulary-file string -> id encoding is exact (line number, '' / unknown -> -1), for
    single-valued, multi-valued and sequence columns;
  * flags / string hyper-parameters / utils known answers."""
import re

import numpy as np
import pytest
import torch

from recalgorithm_amd import feature_column as fc
from recalgorithm_amd.estimator import Estimator, RunConfig
from tests import golden_util as GU


@pytest.mark.parametrize("name", GU.MODELS)
def test_mirror_creates_the_reference_variables(name, tmp_path):
    vocab_dir = GU.write_vocab_dir(str(tmp_path / "vocabulary"))
    model_fn, params, _ = GU.mirror_setup(name, vocab_dir)
    d = GU.load(name)
    sfeats, labels = GU.string_batch()
    feats = {k: (v.float() if isinstance(v, torch.Tensor) else v) for k, v in sfeats.items()}
    est = Estimator(model_fn, params, RunConfig(device="cpu", seed=3, use_hip_graph=False))
    est.build(feats, {"read_comment": labels.float()})          # registration pass only: no HIP call
    arrays = est.store.named_arrays()
    gv = GU.golden_to_oracle_vars(name, GU.section(d, "var/"), params)
    # Dice's BN statistics are never trained (quirk B-5): constants of the kernel, not variables here
    missing = [k for k in gv if k not in arrays and "dice_bn" not in k]
    # cross_part/wl, cross_part/bl: the contiguous blocks behind wl_i / bl_i (fused kernel operands)
    extra = [k for k in arrays if k not in gv and not re.search(r"/(wl|bl)$", k)]
    assert not missing, f"reference variables absent from the mirror: {missing}"
    assert not extra, f"mirror variables the reference does not have: {extra}"
    for k, v in gv.items():
        if k in arrays:
            assert tuple(arrays[k].shape) == tuple(v.shape), (k, arrays[k].shape, v.shape)


def test_vocabulary_encoding_is_exact(tmp_path):
    vocab_dir = GU.write_vocab_dir(str(tmp_path / "vocabulary"))
    sfeats, _ = GU.string_batch()
    b = GU.load("batch")

    def expect(word, stem):
        m = re.fullmatch(rf"{stem}_(\d+)", word)
        return int(m.group(1)) if m and int(m.group(1)) < int(b[f"vocab/{stem}"]) else -1
    for key, stem, seq in [("userid", "userid", False), ("device", "device", False),
                           ("manual_tag_list", "manual_tag_id", False), ("his_read_comment_7d_seq", "feedid", True)]:
        mk = fc.sequence_categorical_column_with_vocabulary_file if seq else fc.categorical_column_with_vocabulary_file
        col = mk(key, vocab_dir + stem + ".txt")
        assert col.num_buckets == int(b[f"vocab/{stem}"])
        enc = col.encode(sfeats[key])
        flat = [expect(w, stem) for row in sfeats[key] for w in row]
        assert enc.values.tolist() == flat
        assert enc.offsets.tolist() == np.cumsum([0] + [len(r) for r in sfeats[key]]).tolist()
        assert -1 in flat          # the batch exercises the OOV path
    # single-valued VarLen feature -> dense [B] ids, -1 where missing
    col = fc.categorical_column_with_vocabulary_file("userid", vocab_dir + "userid.txt")
    ids = col.ids({"userid": sfeats["userid"]}, torch.device("cpu"))
    assert ids.shape == (len(sfeats["userid"]),) and ids.dtype == torch.int64
    assert ids.tolist() == [expect(r[0], "userid") if r else -1 for r in sfeats["userid"]]


def test_input_layer_sorts_columns_by_name():
    """fc.input_layer concatenates in sorted(column.name) order; shared columns are
    named <key>_shared_embedding and returned in input order (A-4)."""
    cats = {k: fc.categorical_column_with_identity(k, 10) for k in ("userid", "device", "authorid", "feedid", "his")}
    shared = fc.shared_embedding_columns([cats["feedid"], cats["his"]], 8)
    assert [c.key for c in shared] == ["feedid", "his"]
    assert shared[0].shared_name == shared[1].shared_name == "feedid_his_shared_embedding"
    cols = [fc.embedding_column(cats["userid"], 8), fc.embedding_column(cats["device"], 8),
            fc.embedding_column(cats["authorid"], 8)] + shared
    assert [c.name for c in sorted(cols, key=lambda c: c.name)] == [
        "authorid_embedding", "device_embedding", "feedid_shared_embedding", "his_shared_embedding", "userid_embedding"]


def test_utils_known_answers():
    from recalgorithm_amd.algorithm.utils import index_from_upper_triangular
    n = 7
    k = 0
    for i in range(n):
        for j in range(i + 1, n):
            assert index_from_upper_triangular(i, j, n) == k      # utils.py:67-82
            k += 1


def test_string_hyperparameters_and_enum_flag():
    from recalgorithm_amd import flags
    from recalgorithm_amd.algorithm.FiBiNET import fibinet  # noqa: F401  (defines the flags)
    flags.FLAGS._overrides.clear()       # programmatic overrides (other tests) shadow parsed values
    rest = flags.FLAGS._parse(["--bilinear_interaction_type=each", "--hidden_units=64,32", "--batch_norm=False"])
    assert rest == [] and flags.FLAGS.bilinear_interaction_type == "each"
    assert flags.FLAGS.hidden_units.split(",") == ["64", "32"] and flags.FLAGS.batch_norm is False
    with pytest.raises(SystemExit):
        flags.FLAGS._parse(["--bilinear_interaction_type=bogus"])
    flags.FLAGS._parse([])


@pytest.mark.parametrize("name", ["model_deepfm", "model_fwfm", "model_dcn", "model_din_dice"])
def test_load_and_export_variables_by_reference_names(name, tmp_path):
    """Estimator.load_variables takes the reference's own variable dictionary (TF names and shapes — here the
    golden `var/` section, produced by the reference sources) incl. the (sum V, 1) first-order kernel, and
    export_variables gives it back unchanged."""
    import numpy as np
    from recalgorithm_amd.estimator import Estimator, RunConfig
    vocab_dir = GU.write_vocab_dir(str(tmp_path / "vocabulary"))
    model_fn, params, _ = GU.mirror_setup(name, vocab_dir)
    d = GU.load(name)
    sfeats, labels = GU.string_batch()
    feats = {k: (v.float() if isinstance(v, torch.Tensor) else v) for k, v in sfeats.items()}
    est = Estimator(model_fn, params, RunConfig(device="cpu", seed=5, use_hip_graph=False))
    est.build(feats, {"read_comment": labels.float()})
    ref = {k: v for k, v in GU.section(d, "var/").items() if "dice_bn" not in k}
    # optimizer slots and counters of a real checkpoint are ignored
    noisy = dict(ref, global_step=np.asarray(7), beta1_power=np.asarray(0.9))
    noisy[next(iter(ref)) + "/Adam"] = np.zeros(3)
    assigned = est.load_variables(noisy)
    assert len(assigned) >= len(ref)
    back = est.export_variables()
    for k, v in ref.items():
        assert k in back, k
        np.testing.assert_array_equal(back[k], v.astype(np.float32).reshape(back[k].shape))
    with pytest.raises(ValueError):
        est.load_variables({k: (v[:-1] if v.ndim and v.shape[0] > 1 else v) for k, v in ref.items()})
    with pytest.raises(KeyError):
        est.load_variables(dict(ref, **{"no/such/variable": np.zeros(2)}))
    # the same through TensorFlow checkpoint FILES (io/tf_checkpoint.py): what a reference model_dir holds — variables,
    # Adam slots, counters — is written in TF's V2 bundle format, and a fresh estimator loads the model_dir natively
    from recalgorithm_amd.io import tf_checkpoint
    ckpt = {k: v.astype(np.float32) for k, v in ref.items()}
    ckpt.update({next(iter(ref)) + "/Adam": np.zeros(3, np.float32), "global_step": np.array(4321, np.int64),
                 "beta1_power": np.array(0.5, np.float32), "beta2_power": np.array(0.9, np.float32)})
    model_dir = str(tmp_path / "reference_model_dir")
    tf_checkpoint.write_checkpoint(model_dir + "/model.ckpt-4321", ckpt)
    est2 = Estimator(model_fn, params, RunConfig(device="cpu", seed=6, use_hip_graph=False))
    est2.build(feats, {"read_comment": labels.float()})
    assert est2.load_tf_checkpoint(model_dir) == 4321
    back2 = est2.export_variables()
    for k, v in ref.items():
        np.testing.assert_array_equal(back2[k], v.astype(np.float32).reshape(back2[k].shape))
    # and back out: save_tf_checkpoint -> read_checkpoint gives the reference-named arrays
    est2.global_step = 99
    out = est2.save_tf_checkpoint(str(tmp_path / "handback" / "model.ckpt-99"))
    again = tf_checkpoint.read_checkpoint(out)
    assert int(again["global_step"]) == 99
    for k, v in ref.items():
        np.testing.assert_array_equal(again[k], v.astype(np.float32).reshape(again[k].shape))


def test_host_columns_are_packed_into_one_matrix_per_kind():
    """Estimator._pack_host_columns: [B] int64 id vectors -> column views of one [B, F] matrix in sorted key
    order (the layout feature_column._as_matrix reads in place), [B, 1] float columns -> views of one [B, n]
    matrix; values unchanged, other entries untouched."""
    from recalgorithm_amd import feature_column as fc
    from recalgorithm_amd.estimator import Estimator, RunConfig
    g = torch.Generator().manual_seed(1)
    B = 33
    feats = {f"f{j:02d}": torch.randint(-1, 50, (B,), generator=g) for j in (3, 0, 7, 1)}
    feats.update({f"d{j}": torch.randn(B, 1, generator=g) for j in range(3)})
    feats["odd_len"] = torch.zeros(B + 1, dtype=torch.int64)
    feats["strings"] = [["a"]] * B
    feats["ragged"] = fc.Ragged(torch.arange(5), torch.tensor([0, 5] + [5] * (B - 1)))
    est = Estimator(lambda *a: None, {}, RunConfig(device="cpu"))
    assert est._pack_host_columns(feats) is feats                      # nothing to do for a CPU estimator
    out = est._pack_host_columns(feats, force=True)
    ids = [out[k] for k in ("f00", "f01", "f03", "f07")]
    for k in feats:
        if isinstance(feats[k], torch.Tensor):
            assert torch.equal(out[k], feats[k]), k
    assert out["strings"] is feats["strings"] and out["ragged"] is feats["ragged"] and out["odd_len"] is feats["odd_len"]
    m = fc._as_matrix(ids)
    assert m.data_ptr() == ids[0].data_ptr() and m.shape == (B, 4) and m.is_contiguous()      # zero-copy
    assert torch.equal(m, torch.stack([feats[k] for k in ("f00", "f01", "f03", "f07")], 1))
    d = [out[f"d{j}"] for j in range(3)]
    assert d[1].data_ptr() == d[0].data_ptr() + 4 and d[0].stride() == (3, 1)


def test_checkpoint_round_trip_restores_variables_moments_and_step(tmp_path):
    """Estimator.save_checkpoint / restore-on-build (the reference gets this from tf.estimator's model_dir,
    deepfm.py:290-293): variables, both Adam moments (dense and per arena), the optimizer step and the global
    step survive; the arena's live-row bookkeeping is rebuilt from the restored moments."""
    from recalgorithm_amd.estimator import Estimator, RunConfig
    vocab_dir = GU.write_vocab_dir(str(tmp_path / "vocabulary"))
    model_fn, params, _ = GU.mirror_setup("model_dcn", vocab_dir)
    sfeats, labels = GU.string_batch()
    feats = {k: (v.float() if isinstance(v, torch.Tensor) else v) for k, v in sfeats.items()}
    lab = {"read_comment": labels.float()}
    md = str(tmp_path / "model_dir")
    a = Estimator(model_fn, params, RunConfig(device="cpu", seed=5, use_hip_graph=False, model_dir=md))
    a.build(feats, lab)
    g = torch.Generator().manual_seed(0)
    for v in a.store.named_arrays().values():
        v.copy_(torch.randn(v.shape, generator=g))
    a.store.flat_m.copy_(torch.randn(a.store.flat_m.shape, generator=g))
    a.store.flat_v.copy_(torch.rand(a.store.flat_v.shape, generator=g))
    for ar in a.store.arenas.values():
        ar.m[::3] = torch.randn(ar.m[::3].shape, generator=g)
        ar.v[::3] = torch.rand(ar.v[::3].shape, generator=g)
    a.store.opt_state = {"step": torch.tensor([17]), "lr_t": torch.zeros(1)}
    a.global_step = 17
    a.save_checkpoint()
    b = Estimator(model_fn, params, RunConfig(device="cpu", seed=99, use_hip_graph=False, model_dir=md))
    b.build(feats, lab)                                     # different seed: everything must come from the file
    assert b.global_step == 17 and int(b.store.opt_state["step"]) == 17
    for k, v in a.store.named_arrays().items():
        assert torch.equal(b.store.named_arrays()[k], v), k
    assert torch.equal(b.store.flat_m, a.store.flat_m) and torch.equal(b.store.flat_v, a.store.flat_v)
    for n, ar in a.store.arenas.items():
        assert torch.equal(b.store.arenas[n].m, ar.m) and torch.equal(b.store.arenas[n].v, ar.v)
        assert b.store.arenas[n].live is None               # rebuilt lazily from the moments (variables.live_state)


def test_checkpoint_restore_refuses_a_different_model(tmp_path):
    """A checkpoint whose variables do not match the model (another vocabulary, other hidden_units) must not be
    half-applied: restoring the step counter and Adam's bias correction on a partly re-initialised model is silent
    corruption.  Also: a matching state round-trips, moments included."""
    import torch
    from recalgorithm_amd.estimator import collect_checkpoint_state, restore_checkpoint_state
    from recalgorithm_amd.variables import EmbeddingArena, VariableStore

    def store(vocab=11, hidden=4):
        st = VariableStore("cpu", seed=3)
        st.get_variable("dense/kernel", (6, hidden))
        ar = st.arenas["emb"] = EmbeddingArena("emb", 4, "cpu", seed=5)
        ar.add_table("t0", vocab)
        st.finalize()
        return st
    a = store()
    a.flat_m.uniform_(0, 1); a.flat_v.uniform_(0, 1)
    a.arenas["emb"].m.uniform_(0, 1); a.arenas["emb"].v.uniform_(0, 1)
    a.opt_state = {"step": torch.tensor([17]), "lr_t": torch.zeros(1)}
    state, writer = collect_checkpoint_state(a, global_step=17)
    assert writer
    path = tmp_path / "ck.pt"
    torch.save(state, path)
    state = torch.load(path, weights_only=True)              # the file is plain tensors / ints
    b = store()
    b.vars["dense/kernel"].data.zero_()
    assert restore_checkpoint_state(b, state, torch.device("cpu")) == 17
    assert torch.equal(b.vars["dense/kernel"].data, a.vars["dense/kernel"].data)
    assert torch.equal(b.arenas["emb"].m, a.arenas["emb"].m) and torch.equal(b.flat_v, a.flat_v)
    assert int(b.opt_state["step"]) == 17
    for other in (store(vocab=12), store(hidden=5)):
        before = {k: v.clone() for k, v in other.named_arrays().items()}
        with pytest.raises(RuntimeError, match="does not match the model"):
            restore_checkpoint_state(other, state, torch.device("cpu"))
        assert all(torch.equal(v, before[k]) for k, v in other.named_arrays().items())     # nothing was applied
        assert other.opt_state is None


def test_preencoded_ids_out_of_range_become_oov():
    """The gather / scatter kernels do not bounds-check ids against the vocabulary: host-resident pre-encoded ids
    that are out of range are mapped to OOV before they reach the device."""
    import torch
    from recalgorithm_amd import feature_column as fc
    from recalgorithm_amd.feature_column import Ragged
    c = fc.categorical_column_with_identity("x", 10)
    ids = c.ids({"x": torch.tensor([0, 9, 10, -1, 12345])}, torch.device("cpu"))
    assert ids.tolist() == [0, 9, -1, -1, -1]
    r = c.ids({"x": Ragged(torch.tensor([3, 10, 2]), torch.tensor([0, 2, 3]))}, torch.device("cpu"))
    assert r.values.tolist() == [3, -1, 2]


def test_train_loop_raises_on_row_exchange_overflow():
    """The static row exchange drops requests that do not fit a bucket and only raises a device flag; Estimator.train
    polls it (every OVERFLOW_POLL_EVERY steps, at every checkpoint and at the end) and refuses to go on."""
    import types
    import torch
    from recalgorithm_amd.estimator import Estimator
    sharding = types.SimpleNamespace(overflow=torch.zeros(1, dtype=torch.bool))
    arena = types.SimpleNamespace(sharding=sharding)
    stub = types.SimpleNamespace(shard_spec=object(), store=types.SimpleNamespace(arenas={"a": arena}))
    Estimator._check_exchange_overflow(stub)                      # flag clear: fine
    sharding.overflow[0] = True
    with pytest.raises(RuntimeError, match="bucket overflow"):
        Estimator._check_exchange_overflow(stub)
    Estimator._check_exchange_overflow(types.SimpleNamespace())   # not data parallel: nothing to poll


def test_packed_batches_of_the_native_reader_move_as_one_matrix(tmp_path):
    """io.native.NativeDataset yields PackedBatch dicts: the single-valued id features are column views of ONE [B, F]
    matrix in sorted key order, and Estimator._pack_host_columns takes that matrix whole (same result as the generic
    re-stacking path: values, order, zero-copy _as_matrix)."""
    from recalgorithm_amd import feature_column as fc
    from recalgorithm_amd.estimator import Estimator, RunConfig
    from recalgorithm_amd.io import native, synth
    if not native.available():
        pytest.skip("librecalgo_host.so not built")
    spec = synth.SynthSpec(n_fields=5, max_vocab=40, seed=3)
    vd, path = str(tmp_path / "vocabulary") + "/", str(tmp_path / "t.tfrecord")
    synth.write_vocabularies(spec, vd)
    synth.write_tfrecord(spec, path, 70)
    # columns deliberately NOT in sorted order
    names = list(reversed(spec.names))
    cats = [fc.categorical_column_with_vocabulary_file(n, vd + n + ".txt") for n in names]
    cols = [fc.embedding_column(c, 4) for c in cats]
    label = fc.numeric_column("read_comment", default_value=0.0)
    ds = native.NativeDataset(path, cols + [label], ["read_comment"], 32)
    est = Estimator(lambda *a: None, {}, RunConfig(device="cpu"))
    n_batches = 0
    for feats, labels in ds:
        n_batches += 1
        assert isinstance(feats, native.PackedBatch) and feats.packed_ids is not None
        mat, keys = feats.packed_ids
        assert keys == sorted(names) and mat.is_contiguous() and mat.shape == (labels["read_comment"].shape[0], len(names))
        fast = est._pack_host_columns(feats, force=True)
        slow = est._pack_host_columns(dict(feats), force=True)             # a plain dict: the generic path
        for k in names:
            assert torch.equal(fast[k], slow[k]) and torch.equal(fast[k], feats[k])
        m = fc._as_matrix([fast[k] for k in sorted(names)])
        assert m.data_ptr() == mat.data_ptr() and m.is_contiguous()         # the reader's matrix itself, no copy on a CPU device
    assert n_batches == 3


def test_graphed_step_input_load_copies_views_of_one_allocation_at_once():
    """GraphedTrainStep.load (the per-step host path of a graph-replayed training loop): inputs that are views of one
    allocation with the same relative layout on both sides move as ONE copy of the byte span they cover; anything else
    is copied tensor by tensor; a changed structure is refused.  (CPU tensors: the logic is device independent.)"""
    from recalgorithm_amd.estimator import GraphedTrainStep, _tree_tensors
    from recalgorithm_amd import feature_column as fc
    B, F = 37, 5

    def batch(seed, pad=0):
        g = torch.Generator().manual_seed(seed)
        base = torch.randint(0, 1000, (pad + B * F + B,), generator=g)
        m = base[pad:pad + B * F].view(B, F)
        feats = {f"f{i}": m[:, i] for i in range(F)}
        feats["dense"] = torch.randn(B, 3, generator=g)
        feats["seq"] = fc.Ragged(torch.arange(7) + seed, torch.tensor([0, 7] + [7] * (B - 1)))
        return feats, {"y": base[pad + B * F:]}

    f0, l0 = batch(0)
    g = GraphedTrainStep.__new__(GraphedTrainStep)
    g._static = list(_tree_tensors(f0, "f")) + list(_tree_tensors(l0, "l"))
    static_ptrs = [t.data_ptr() for _, t in g._static]
    for seed, pad in ((1, 0), (2, 11)):                      # the source may sit anywhere in its own allocation
        f1, l1 = batch(seed, pad)
        g.load(f1, l1)
        for (k, dst), (_, src) in zip(g._static, list(_tree_tensors(f1, "f")) + list(_tree_tensors(l1, "l"))):
            assert torch.equal(dst, src), k
    assert [t.data_ptr() for _, t in g._static] == static_ptrs            # the static buffers themselves never move
    # separately allocated columns on the source side: copied one by one, same result
    f2, l2 = batch(3)
    f2 = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in f2.items()}
    g.load(f2, l2)
    assert all(torch.equal(f0[k], f2[k]) for k in f2 if isinstance(f2[k], torch.Tensor))
    # a different dtype is converted by the per-tensor path
    f3, l3 = batch(4)
    f3["dense"] = f3["dense"].double()
    g.load(f3, l3)
    assert torch.equal(f0["dense"], f3["dense"].float())
    f4, l4 = batch(5)
    f4.pop("f3")
    with pytest.raises(ValueError):
        g.load(f4, l4)
    f5, l5 = batch(6)
    f5["dense"] = torch.zeros(B, 4)
    with pytest.raises(ValueError):
        g.load(f5, l5)


def test_graphed_step_input_load_never_clobbers_a_static_input_inside_a_span():
    """A group's one-span copy overwrites every byte between its first and last view — so it is only taken when no OTHER static
    input lives in that range of the static allocation (an input whose new source sits in another storage would otherwise be
    clobbered whenever it had been copied earlier); and per-tensor copies run after all span copies."""
    from recalgorithm_amd.estimator import GraphedTrainStep, _tree_tensors
    from recalgorithm_amd.nn import BNLink
    B = 16
    static = torch.zeros(4 * B, dtype=torch.int64)
    sf = {"a": static[0:B], "b": static[B:2 * B], "intruder": static[2 * B:3 * B]}
    sl = {"y": static[3 * B:4 * B]}
    g = GraphedTrainStep.__new__(GraphedTrainStep)
    g._static = list(_tree_tensors(sf, "f")) + list(_tree_tensors(sl, "l"))
    src = torch.arange(4 * B, dtype=torch.int64) + 100            # a, b, (gap), y in ONE storage with the static layout ...
    other = torch.arange(B, dtype=torch.int64) + 7000             # ... the intruder's new value in ANOTHER
    nf = {"a": src[0:B], "b": src[B:2 * B], "intruder": other}
    nl = {"y": src[3 * B:4 * B]}
    g.load(nf, nl)
    assert torch.equal(sf["a"], src[0:B]) and torch.equal(sf["b"], src[B:2 * B]) and torch.equal(sl["y"], src[3 * B:])
    assert torch.equal(sf["intruder"], other)                     # (a span copy a .. y would have left src's gap bytes here)
    # BNLink (nn.py): the sums a dense layer's backward left are handed out only for the gradient tensor they belong to
    x = torch.zeros(130, 8)
    link = BNLink(x, torch.zeros(8), torch.ones(8))
    link.sums, gr = torch.ones(3, 16), torch.zeros(130, 8)
    link.grad_ptr = gr.data_ptr()
    assert link.take(gr) is not None and link.take(gr) is None    # consumed once
    link.sums = torch.ones(3, 16)
    assert link.take(torch.zeros(130, 8)) is None and link.sums is None      # another tensor (autograd summed two consumers)
    link.sums = torch.ones(3, 16)
    assert link.take(gr.t().contiguous().t()) is None             # not the contiguous tensor the sums were computed from


@pytest.mark.parametrize("F", [2, 3, 6, 26, 40])
def test_pair_strength_gradient_jobs_cover_the_strict_triangle(F):
    """ops._PairKernel.colsum_jobs (FwFM head inside the fused loss tail): the F - 1 column-sum jobs together deliver column
    t(i, j) of the Gram triangle (diagonal included) to r.grad[index_from_upper_triangular(i, j)] (reference utils.py:67-82)
    for every i < j, each entry exactly once, and never touch a diagonal column."""
    import torch
    from recalgorithm_amd import ops
    from recalgorithm_amd.variables import Variable
    n, T = F * (F - 1) // 2, F * (F + 1) // 2
    r = Variable("fields_pair_strength/fields_pair_strength_weight", torch.zeros(n))
    rows, stride, col0 = 3, T + 2 + 5, 5                      # the head's columns start at column 5 of the partial rows
    partials = torch.arange(rows * stride, dtype=torch.float32).reshape(rows, stride)
    jobs = ops._PairKernel(r, torch.zeros(T, 1), F, None).colsum_jobs(partials, col0, rows, stride)
    assert len(jobs) == F - 1
    hit = torch.zeros(n, dtype=torch.int64)
    for part, off, nrows, st, cnt, out in jobs:               # what the deferred-sum launch does with a job
        assert part is partials and nrows == rows and st == stride and out.numel() == cnt
        out.copy_(partials[:, off:off + cnt].sum(0))
        first = (out.data_ptr() - r.grad.data_ptr()) // 4
        hit[first:first + cnt] += 1
    assert bool((hit == 1).all())
    index = 0
    for i in range(F - 1):
        for j in range(i + 1, F):
            t = i * F - i * (i - 1) // 2 + (j - i)
            assert float(r.grad[index]) == float(partials[:, col0 + t].sum()), (i, j)
            index += 1
    assert ops._pair_index(F, "cpu").tolist() == [i * F - i * (i - 1) // 2 + (j - i) for i in range(F - 1) for j in range(i + 1, F)]
