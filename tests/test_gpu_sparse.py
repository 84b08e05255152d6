"""Owner-computes row-gradient scatter + fused sparse optimizer (csrc/sparse.hip, recalgorithm_amd/sparse.py).

  * GRAD mode: the per-row sums equal the fp32 sum in the kernels' documented order bit for bit (tiles of 256 examples of
    a field in example order, then the tiles in order; rows up to 24 requests per tile / 48 tiles), equal the fp64 sum
    within fp32 rounding everywhere (hot rows, large buckets), and are bit-reproducible from run to run;
  * ADAM mode (deferred-exact TF1 Adam): after the flush, weights and both moments are BIT-IDENTICAL to the dense
    TF1 Adam pass over the whole arena (tf.train.AdamOptimizer semantics, /root/reference algorithm/DeepFM/deepfm.py:246-250),
    and every forward reads the same weights the dense pass would have produced;
  * LAZY_ADAM mode == oracle.ref_ops.lazy_adam_step (tf.contrib.opt.LazyAdamOptimizer, algorithm/DIEN/dien.py:328).
"""
import numpy as np
import pytest
import torch

from tests.util import assert_bit_exact, assert_close

pytestmark = pytest.mark.gpu


class _Store:
    """The two things sparse.py needs of a VariableStore."""

    def __init__(self, dev):
        self.opt_state = {"step": torch.zeros(1, dtype=torch.int64, device=dev), "lr_t": torch.zeros(1, device=dev)}
        self.arenas = {}


def _arena(dev, rows, K, seed=1, name="t"):
    from recalgorithm_amd.variables import EmbeddingArena
    ar = EmbeddingArena(name, K, dev, seed=seed)
    ar.add_table("t0", rows)
    ar.materialize()
    return ar


def _skewed_ids(gen, n_ex, F, rows, hot=0.3, oov=0.05):
    ids = torch.randint(0, rows, (n_ex, F), generator=gen)
    ids[torch.rand(n_ex, F, generator=gen) < hot] = 3                     # one hot row
    ids[torch.rand(n_ex, F, generator=gen) < oov] = -1
    return ids


def _request_order_sum(ids, g, rows, K):
    """The fp32 sum in the kernels' (fixed) order, with numpy float32 adds: an id matrix is walked FIELD-MAJOR in tiles of
    256 examples; the requests of a row inside a tile are added in example order (the tile's partial sum, csrc/sparse.hip
    `place`), the tiles' partial sums of a row in tile order (`apply`)."""
    out = np.zeros((rows, K), dtype=np.float32)
    idn = ids.numpy()
    gv = g.reshape(ids.shape[0], ids.shape[1], K).numpy().astype(np.float32)
    n_ex, F = idn.shape
    for f in range(F):
        for e0 in range(0, n_ex, 256):
            part = {}
            for e in range(e0, min(n_ex, e0 + 256)):
                r = int(idn[e, f])
                if r >= 0:
                    part[r] = gv[e, f].copy() if r not in part else part[r] + gv[e, f]
            for r, p_ in part.items():
                out[r] = out[r] + p_
    return torch.from_numpy(out)


@pytest.mark.parametrize("rows,K,n_ex,F", [(5003, 16, 300, 5), (700, 8, 257, 1), (64, 64, 100, 3), (1500, 1, 999, 2),
                                           (900, 6, 50, 7), (4000, 2, 1000, 1), (100000, 16, 4096, 26)])
def test_scatter_grad_is_the_request_order_sum(dev, rows, K, n_ex, F):
    from recalgorithm_amd import sparse
    gen = torch.Generator().manual_seed(rows + K)
    ar = _arena(dev, rows, K)
    store = _Store(dev)
    ids = torch.randint(0, rows, (n_ex, F), generator=gen)
    # duplicates, but no row above 40 requests (longer rows are summed by the whole workgroup in a different fixed order)
    dup = torch.randint(0, min(rows, 50), (n_ex, F), generator=gen)
    ids = torch.where(torch.rand(n_ex, F, generator=gen) < min(0.3, 1000.0 / (n_ex * F)), dup, ids)
    ids[torch.rand(n_ex, F, generator=gen) < 0.05] = -1
    g = torch.randn(n_ex, F * K, generator=gen)
    with torch.enable_grad():
        src = sparse.begin_lookup(ar, store, ids.to(dev), None, None, 0, n_ex, F)
    assert src is not None
    src.set_grad(g.to(dev))
    store.arenas["t"] = ar
    sparse.materialize_grads(store)
    counts = torch.bincount(ids[ids >= 0].reshape(-1), minlength=rows)
    assert int(counts.max()) <= 48
    assert_bit_exact(ar.grad.cpu(), _request_order_sum(ids, g, rows, K), "scatter GRAD vs request-order fp32 sum")
    # a second pass over the same plan adds the same sums again (+=), and the plan came back clean
    first = ar.grad.clone()
    sparse.plan_of(ar).grad_materialized = False
    sparse.materialize_grads(store)
    assert_close(ar.grad, 2 * first.double(), what="second GRAD pass accumulates")


@pytest.mark.parametrize("hot_requests", [500, 3000, 9000])
def test_scatter_hot_rows_and_oversize_buckets(dev, hot_requests):
    """A row with thousands of requests (a two-valued field like the reference's `device`): the bucket exceeds the LDS
    sort and takes the global merge path; the whole workgroup sums the row.  Deterministic, and right."""
    from recalgorithm_amd import sparse
    rows, K, F = 3000, 16, 2
    n_ex = hot_requests
    gen = torch.Generator().manual_seed(hot_requests)
    ids = torch.randint(0, rows, (n_ex, F), generator=gen)
    ids[:, 0] = torch.where(torch.rand(n_ex, generator=gen) < 0.7, torch.full((n_ex,), 17), torch.full((n_ex,), 18))
    g = torch.randn(n_ex, F * K, generator=gen)
    ref = torch.zeros(rows, K, dtype=torch.float64).index_add_(0, ids.reshape(-1), g.reshape(-1, K).double())
    outs = []
    for rep in range(2):
        ar = _arena(dev, rows, K)
        store = _Store(dev)
        store.arenas["t"] = ar
        with torch.enable_grad():
            src = sparse.begin_lookup(ar, store, ids.to(dev), None, None, 0, n_ex, F)
        src.set_grad(g.to(dev))
        sparse.materialize_grads(store)
        outs.append(ar.grad.clone())
    assert_bit_exact(outs[0], outs[1], "scatter is bit-reproducible")
    assert_close(outs[0], ref, what=f"hot row with {hot_requests} requests", reduced=True)


def test_scatter_sources_ragged_and_broadcast(dev):
    """The three request shapes of the models in one plan: an id matrix with per-field row bases, a ragged sequence
    (DIN's history, din.py:207-214) and a shared per-example gradient row (DeepFM's first-order weights)."""
    from recalgorithm_amd import sparse
    from recalgorithm_amd.variables import EmbeddingArena
    gen = torch.Generator().manual_seed(5)
    K, B, F, T = 8, 37, 3, 6
    ar = EmbeddingArena("t", K, dev, seed=2)
    vocabs = [50, 70, 30, 90]
    for i, v in enumerate(vocabs):
        ar.add_table(f"t{i}", v)
    ar.materialize()
    rows = sum(vocabs)
    rb = torch.tensor([ar.tables[f"t{i}"][0] for i in range(3)], dtype=torch.int64)
    ids = torch.stack([torch.randint(-1, vocabs[i], (B,), generator=gen) for i in range(3)], 1)
    lens = torch.randint(0, T + 3, (B,), generator=gen)          # some longer than T: truncated
    offs = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)])
    vals = torch.randint(-1, vocabs[3], (int(lens.sum()),), generator=gen)
    g_ids = torch.randn(B, F * K, generator=gen)
    g_seq = torch.randn(B, T, K, generator=gen)
    store = _Store(dev)
    store.arenas["t"] = ar
    with torch.enable_grad():
        s0 = sparse.begin_lookup(ar, store, ids.to(dev), None, rb.to(dev), 0, B, F)
        s1 = sparse.begin_lookup(ar, store, vals.to(dev), offs.to(dev), None, ar.tables["t3"][0], B, T)
    s0.set_grad(g_ids.to(dev))
    s1.set_grad(g_seq.to(dev))
    sparse.materialize_grads(store)
    ref = torch.zeros(rows, K, dtype=torch.float64)
    for b in range(B):
        for f in range(F):
            if ids[b, f] >= 0:
                ref[rb[f] + ids[b, f]] += g_ids[b, f * K:(f + 1) * K].double()
        for t in range(min(int(lens[b]), T)):
            v = int(vals[offs[b] + t])
            if v >= 0:
                ref[ar.tables["t3"][0] + v] += g_seq[b, t].double()
    assert_close(ar.grad, ref, what="id matrix + ragged sequence in one plan", reduced=True)
    # broadcast rows: K = 1 arena, every field of example b receives g1[b]
    w1 = EmbeddingArena("w1", 1, dev, seed=3)
    for i, v in enumerate(vocabs[:3]):
        w1.add_table(f"t{i}", v)
    w1.materialize()
    store.arenas = {"w1": w1}
    g1 = torch.randn(B, 1, generator=gen)
    with torch.enable_grad():
        s = sparse.begin_lookup(w1, store, ids.to(dev), None, rb.to(dev), 0, B, F)
    s.set_grad(g1.to(dev), fmul=0)
    sparse.materialize_grads(store)
    ref1 = torch.zeros(sum(vocabs[:3]), 1, dtype=torch.float64)
    for b in range(B):
        for f in range(F):
            if ids[b, f] >= 0:
                ref1[rb[f] + ids[b, f]] += g1[b].double()
    assert_close(w1.grad, ref1, what="broadcast gradient rows (first-order weights)", reduced=True)


@pytest.mark.parametrize("rows,K,F,period", [(3001, 16, 3, 4), (777, 8, 1, 3), (2000, 2, 2, 5), (1500, 1, 2, 4), (300, 64, 1, 2),
                                             (40000, 16, 4, 32), (3_000_000, 4, 1, 2)])     # (the last: sweep groups walk 16 rows each)
def test_deferred_adam_is_bit_identical_to_dense_tf1_adam(dev, rows, K, F, period, monkeypatch):
    """N steps of (lookup -> row gradients -> optimizer) on two copies of an arena: copy A takes the summed gradients
    (GRAD mode) and the DENSE TF1 Adam pass over every row; copy B the fused deferred-exact path.  Every step the rows the
    lookup reads are bit-identical, and after the flush so are all of w, m, v."""
    import ctypes
    from recalgorithm_amd import _lib, ops, sparse
    lib = _lib.load()
    rb0 = torch.zeros(F, dtype=torch.int64, device=dev)
    monkeypatch.setenv("RECALGO_ADAM_SWEEP_PERIOD", str(period))
    gen = torch.Generator().manual_seed(rows * 7 + K)
    A, Bn = _arena(dev, rows, K, seed=9, name="a"), _arena(dev, rows, K, seed=9, name="b")
    assert_bit_exact(A.weight, Bn.weight)
    stA, stB = _Store(dev), _Store(dev)
    stA.arenas["a"], stB.arenas["b"] = A, Bn
    lr = 0.01
    n_steps = 3 * period + 5
    for step in range(1, n_steps + 1):
        n_ex = 150
        # a sliding window of "recent" rows + a few uniform ones: rows return after gaps of every length up to > period
        lo = (step * 37) % max(rows - 200, 1)
        ids = lo + torch.randint(0, min(200, rows), (n_ex, F), generator=gen)
        far = torch.randint(0, rows, (n_ex, F), generator=gen)
        ids = torch.where(torch.rand(n_ex, F, generator=gen) < 0.2, far, ids).clamp_(max=rows - 1)
        ids[torch.rand(n_ex, F, generator=gen) < 0.05] = -1
        ids[0, 0] = 5                                                      # a row touched every step
        g = torch.randn(n_ex, F * K, generator=gen)
        g[0, :K] = 0.0 if step % 2 else g[0, :K]                           # ... sometimes with a zero gradient
        ids_d, g_d = ids.to(dev), g.to(dev)
        with torch.enable_grad():
            sB = sparse.begin_lookup(Bn, stB, ids_d, None, None, 0, n_ex, F)
            sA = sparse.begin_lookup(A, stA, ids_d, None, None, 0, n_ex, F)
        # the forward lookup on the deferred arena reads every row as of this step (lagging rows replayed in registers,
        # nothing written back): bit-identical to the lookup on the densely updated copy
        outA, outB = torch.empty(n_ex, F * K, device=dev), torch.empty(n_ex, F * K, device=dev)
        dvB, stpB = sparse.deferred_view(Bn, stB)
        pp = lambda t_: ctypes.c_void_p(t_.data_ptr())
        st_ = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.recalgo_embedding_gather_fwd_deferred(pp(ids_d), pp(Bn.weight), pp(rb0), n_ex, F, K, pp(outB), F * K, 0,
                                                             dvB, stpB, 0, st_), "gather deferred")
        _lib.check(lib.recalgo_embedding_gather_fwd(pp(ids_d), pp(A.weight), pp(rb0), n_ex, F, K, pp(outA), F * K, 0, st_), "gather")
        assert_bit_exact(outB, outA, f"step {step}: what the forward reads")
        sA.set_grad(g_d)
        sB.set_grad(g_d)
        sparse.materialize_grads(stA)                                      # A.grad = summed rows (same order as B's)
        sparse.new_forward(stA)                                            # (A's sources are done)
        ops.adam_tf1_advance_(stA.opt_state["step"], stA.opt_state["lr_t"], lr)
        ops.adam_tf1_(A.weight.view(-1), A.grad.view(-1), A.m.view(-1), A.v.view(-1), step=-1, lr=lr,
                      lr_t_dev=stA.opt_state["lr_t"])                      # dense pass over ALL rows, zeroes A.grad
        stB.opt_state["step"] += 1
        sparse.apply(Bn, False, stB.opt_state["step"], lr, 0.9, 0.999, 1e-8)
        if step == n_steps // 2:                                            # a flush in the middle (EVAL, checkpoint)
            sparse.sync_store(stB)
            for a, b, nm in ((A.weight, Bn.weight, "w"), (A.m, Bn.m, "m"), (A.v, Bn.v, "v")):
                assert_bit_exact(b, a, f"mid-run flush: {nm}")
    lag = int(stB.opt_state["step"]) - sparse.plan_of(Bn).last_step
    touched = sparse.plan_of(Bn).last_step > 0
    assert int(lag[touched].max()) <= period + 1, "the sweep bounds every row's lag"
    assert bool((lag[touched] > 0).any()), "the test must actually defer something"
    sparse.sync_store(stB)
    for a, b, nm in ((A.weight, Bn.weight, "w"), (A.m, Bn.m, "m"), (A.v, Bn.v, "v")):
        assert_bit_exact(b, a, f"deferred vs dense TF1 Adam: {nm}")
    assert float(A.grad.abs().sum()) == 0.0


@pytest.mark.parametrize("rows,K", [(500, 16), (300, 2), (200, 12), (150, 1)])
def test_lazy_adam_matches_tf_lazy_adam(dev, rows, K):
    """LazyAdamOptimizer (dien.py:328): exactly the rows of the step's slices move — whole rows, also a row whose
    summed gradient is exactly zero; all others keep w, m, v bit for bit (widths that are not 4 * 2^n included)."""
    from oracle import ref_ops as R
    from recalgorithm_amd import sparse
    gen = torch.Generator().manual_seed(rows + K)
    ar = _arena(dev, rows, K, seed=4)
    store = _Store(dev)
    store.arenas["t"] = ar
    p = ar.weight.cpu().double()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    lr = 0.02
    for step in range(1, 5):
        n_ex, F = 60, 2
        ids = _skewed_ids(gen, n_ex, F, rows // 2)                          # the upper half of the table is never touched
        g = torch.randn(n_ex, F * K, generator=gen)
        ids[ids == 7] = 8
        ids[1, 0], ids[2, 0] = 7, 7                                        # row 7: twice in every batch ...
        g[1, :K] = 1.5
        g[2, :K] = 1.5 if step == 1 else -1.5                              # ... from step 2 on with a summed gradient == 0
        before = (ar.weight.clone(), ar.m.clone(), ar.v.clone())
        with torch.enable_grad():
            s = sparse.begin_lookup(ar, store, ids.to(dev), None, None, 0, n_ex, F)
        s.set_grad(g.to(dev))
        store.opt_state["step"] += 1
        sparse.apply(ar, True, store.opt_state["step"], lr, 0.9, 0.999, 1e-8)
        ok = ids.reshape(-1) >= 0
        R.lazy_adam_step(p, ids.reshape(-1)[ok], g.reshape(-1, K)[ok].double(), m, v, step, lr)
        assert_close(ar.weight, p, what=f"lazy adam w, step {step}", rtol=2e-5)
        assert_close(ar.m, m, what=f"lazy adam m, step {step}", rtol=2e-5)
        assert_close(ar.v, v, what=f"lazy adam v, step {step}", rtol=2e-5)
        untouched = torch.ones(rows, dtype=torch.bool)
        untouched[ids.reshape(-1)[ok]] = False
        for a, b, nm in zip(before, (ar.weight, ar.m, ar.v), ("w", "m", "v")):
            assert_bit_exact(b[untouched.to(dev)], a[untouched.to(dev)], f"rows outside the batch keep {nm}")
        if step >= 2:
            assert float((ar.weight[7] - before[0][7]).abs().max()) > 0, "a zero-gradient row of the batch still moves"
    assert sparse.plan_of(ar).last_step is None


@pytest.mark.parametrize("model", ["dcn", "deepfm"])
def test_models_train_identically_on_the_owner_and_the_atomic_paths(dev, model, monkeypatch):
    """DCN / DeepFM, three steps: the owner-computes path (deferred Adam; DeepFM's first-order arena as the companion of its
    embedding arena) and the round-1 path that arenas outside the plan's domain still take (LDS-aggregated float atomics +
    live-list dense Adam; sparse.SCATTER_MODE, a test hook) agree — the same arithmetic, summed in a different order."""
    from recalgorithm_amd import feature_column as fc
    from recalgorithm_amd import sparse
    from recalgorithm_amd.algorithm.DCN.dcn import dcn_model_fn
    from recalgorithm_amd.algorithm.DeepFM.deepfm import deepfm_model_fn
    from recalgorithm_amd.estimator import Estimator, RunConfig
    from recalgorithm_amd.io import synth
    spec = synth.SynthSpec(n_fields=6, max_vocab=300, seed=11)
    results = {}
    for mode in ("owner", "atomic"):
        monkeypatch.setattr(sparse, "SCATTER_MODE", mode)
        cats = [fc.categorical_column_with_identity(n, v) for n, v in zip(spec.names, spec.vocabs)]
        if model == "dcn":
            fn = dcn_model_fn
            params = {"category_feature_columns": [fc.embedding_column(c, 16) for c in cats], "dense_feature_columns": [],
                      "hidden_units": ["32", "16"], "num_cross_layer": 2, "learning_rate": 0.01}
        else:
            fn = deepfm_model_fn
            params = {"first_order_feature_columns": [fc.indicator_column(c) for c in cats],
                      "second_order_feature_columns": [fc.embedding_column(c, 16) for c in cats],
                      "hidden_units": ["32", "16"], "dropout_rate": 0.0, "batch_norm": True, "learning_rate": 0.01}
        est = Estimator(fn, params, RunConfig(device=dev, seed=5))
        losses = []
        for i in range(3):
            feats, labels, _ = synth.device_features(spec, 128, dev, batch_index=i)
            est.build(feats, labels)
            losses.append(float(est.train_step(feats, labels)))
        plans = [sparse.plan_of(a) for a in est.store.arenas.values()]
        if mode == "owner":                    # the path under test really ran: deferred state exists, no live list was built
            assert all(pl is not None and pl.last_step is not None for pl in plans)
            assert all(a.live is None for a in est.store.arenas.values())
        else:
            assert all(pl is None for pl in plans)
        results[mode] = (losses, {k: v.detach().cpu().double() for k, v in est.store.named_arrays().items()})
    for a, b in zip(*[results[m][0] for m in ("owner", "atomic")]):
        assert abs(a - b) <= 1e-5 * abs(b)
    for k, ref in results["atomic"][1].items():
        assert_close(results["owner"][1][k], ref, what=f"owner vs atomic path: {k}", rtol=1e-4, reduced=True)


@pytest.mark.parametrize("lazy", [False, True])
def test_companion_arena_equals_its_own_plan_bit_for_bit(dev, lazy, monkeypatch):
    """DeepFM's first-order arena (one float per row, looked up with the embedding arena's requests, deepfm.py:125-141)
    rides on the embedding arena's plan: `place` also sums the scalar gradients of a tile's duplicates and
    recalgo_scatter_apply_companion walks the same placed entries.  Same entries, same order, same arithmetic as a plan
    of its own: weights, both moments and the materialised gradient are bit-identical — deferred Adam and LazyAdam."""
    from recalgorithm_amd import sparse
    monkeypatch.setenv("RECALGO_ADAM_SWEEP_PERIOD", "3")
    rows, K, F, n_ex = 4000, 16, 3, 700
    out = {}
    for companion in ("1", "0"):
        monkeypatch.setattr(sparse, "COMPANION", companion == "1")
        gen = torch.Generator().manual_seed(77)
        E, W = _arena(dev, rows, K, seed=3, name="e"), _arena(dev, rows, 1, seed=4, name="w")
        store = _Store(dev)
        store.arenas["e"], store.arenas["w"] = E, W
        grads = None
        for step in range(1, 9):
            lo = (step * 301) % (rows - 600)
            ids = lo + torch.randint(0, 600, (n_ex, F), generator=gen)
            ids[torch.rand(n_ex, F, generator=gen) < 0.3] = 3             # a hot row: duplicates inside every tile
            ids[torch.rand(n_ex, F, generator=gen) < 0.05] = -1
            g = torch.randn(n_ex, F * K, generator=gen).to(dev)
            g1 = torch.randn(n_ex, 1, generator=gen).to(dev)
            ids_d = ids.to(dev)
            sparse.new_forward(store)
            with torch.enable_grad():
                s, s1 = sparse.begin_lookup_pair(E, W, store, ids_d, None, n_ex, F)
            assert isinstance(s1, sparse.CompanionSource) == (companion == "1")
            s.set_grad(g)
            s1.set_grad(g1, fmul=0)
            if step == 2:                                                  # the gradient arenas on request (named_grads)
                grads = (E.grad.clone(), W.grad.clone())
                ref1 = torch.zeros(rows, 1, dtype=torch.float64)
                ok = ids >= 0
                ref1.index_add_(0, ids[ok], g1.cpu().double().expand(n_ex, F)[ok].reshape(-1, 1))
                assert_close(grads[1], ref1, what="companion GRAD vs fp64 sum", reduced=True)
            store.opt_state["step"] += 1
            for ar in sorted((E, W), key=sparse.has_companions):
                sparse.apply(ar, lazy, store.opt_state["step"], 0.01, 0.9, 0.999, 1e-8)
            assert float(W.grad.abs().sum()) == 0.0 and float(E.grad.abs().sum()) == 0.0
        if not lazy:
            lag = int(store.opt_state["step"]) - sparse.plan_of(W).last_step
            assert int(lag[sparse.plan_of(W).last_step > 0].max()) <= 3 + 1, "the companion arena is swept too"
        sparse.sync_store(store)
        out[companion] = (grads, [t.clone() for t in (E.weight, E.m, E.v, W.weight, W.m, W.v)])
    for a, b, nm in zip(out["1"][0], out["0"][0], ("E.grad", "W.grad")):
        assert_bit_exact(a, b, f"companion vs own plan: {nm}")
    for a, b, nm in zip(out["1"][1], out["0"][1], ("E.w", "E.m", "E.v", "W.w", "W.m", "W.v")):
        assert_bit_exact(a, b, f"companion vs own plan: {nm}")
    assert float((out["1"][1][3] - _arena(dev, rows, 1, seed=4).weight).abs().max()) > 0


def test_companion_dissolves_when_the_arena_is_also_looked_up_alone(dev):
    """An arena that rides on another one's plan AND is looked up on its own in the same step falls back to a plan of its
    own (every request counted once)."""
    from recalgorithm_amd import sparse
    rows, K, F, n_ex = 900, 8, 2, 300
    gen = torch.Generator().manual_seed(5)
    E, W = _arena(dev, rows, K, seed=3, name="e"), _arena(dev, rows, 1, seed=4, name="w")
    store = _Store(dev)
    store.arenas["e"], store.arenas["w"] = E, W
    ids = torch.randint(0, rows, (n_ex, F), generator=gen)
    ids2 = torch.randint(0, rows, (50, 1), generator=gen)
    g1, g2 = torch.randn(n_ex, 1, generator=gen), torch.randn(50, 1, generator=gen)
    with torch.enable_grad():
        s, s1 = sparse.begin_lookup_pair(E, W, store, ids.to(dev), None, n_ex, F)
        s2 = sparse.begin_lookup(W, store, ids2.to(dev), None, None, 0, 50, 1)
    assert s.companion is None and s1.regular is not None
    s.set_grad(torch.randn(n_ex, F * K, generator=gen).to(dev))
    s1.set_grad(g1.to(dev), fmul=0)
    s2.set_grad(g2.to(dev))
    ref = torch.zeros(rows, 1, dtype=torch.float64)
    ref.index_add_(0, ids.reshape(-1), g1.double().expand(n_ex, F).reshape(-1, 1))
    ref.index_add_(0, ids2.reshape(-1), g2.double())
    assert_close(W.grad, ref, what="dissolved companion + own lookup", reduced=True)


@pytest.mark.parametrize("requests,nb_env", [((700, 9), None), ((4096, 26), None), ((3000, 40), "12"), ((5000, 60), "13")])
def test_plan_prefix_from_the_optimizer_launch_equals_place_scanning_itself(dev, requests, nb_env, monkeypatch):
    """RECALGO_SCATTER_PRESCANNED: the prefix of the plan's bucket totals (offs) and the bucket dispatch order (sched) written
    by the extra workgroup of recalgo_adam_tf1_step_plans (csrc/plan_scan.h; counters kept in registers up to 4096 buckets,
    re-read above) give the same placement as `place` scanning the counters itself: weights, moments and last_step of two
    identical arenas agree bit for bit after three steps, one arena on each path."""
    from recalgorithm_amd import ops, sparse as sp
    if nb_env is not None:
        monkeypatch.setattr(sp, "NB_LOG2", int(nb_env))
    n_ex, F = requests
    rows, K = 20000, 16
    gen = torch.Generator().manual_seed(n_ex + F)
    arenas = [_arena(dev, rows, K, seed=5, name=f"a{i}") for i in range(2)]
    stores = [_Store(dev) for _ in range(2)]
    flat = [torch.zeros(8, device=dev) for _ in range(4)]
    for step in range(1, 4):
        ids = _skewed_ids(gen, n_ex, F, rows).to(dev)
        g = torch.randn(n_ex, F * K, generator=gen).to(dev)
        for i, (ar, st) in enumerate(zip(arenas, stores)):
            st.arenas = {"a": ar}
            src = sp.begin_lookup(ar, st, ids, None, None, 0, n_ex, F, True)
            src.set_grad(g)
            st.opt_state["step"] += 1
            scans = []
            if i == 0:
                rec = sp.plan_scan_record(ar, False)
                assert rec is not None                        # (one lookup: the workspace's counts are this step's)
                scans = [rec]
            ops.adam_tf1_step_(flat[0], flat[1], flat[2], flat[3], [], st.opt_state["step"], None, 0.01, plan_scans=scans)
            sp.apply(ar, False, st.opt_state["step"], 0.01, 0.9, 0.999, 1e-8)
    a, b = arenas
    assert_bit_exact(a.weight, b.weight, "weights: prescanned vs self-scanned placement")
    assert_bit_exact(a.m, b.m, "first moments")
    assert_bit_exact(a.v, b.v, "second moments")
    assert torch.equal(sp.plan_of(a).last_step, sp.plan_of(b).last_step)
