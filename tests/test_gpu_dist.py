"""-m gpu: the row-sharded embedding path (recalgorithm_amd/parallel.py) end to end on the real HIP
kernels and the RCCL backend, on the one GPU a test box has: a 1-rank process group.  With N = 1
every row is "remote-local" (owner 0, local row = global row), so the whole machinery — bucketing,
the all_to_alls, owner-side HIP gather / scatter-add, staged arena, identity ids, the fused DeepFM
kernel on staged rows, gradient push, dense all-reduce hook — runs, and its result must equal the
plain single-GPU step.  (N = 2 is covered on CPU/gloo by tests/test_dist_gloo.py.)"""
import os
import re
import socket

import pytest
import torch

from recalgorithm_amd import feature_column as fc
from recalgorithm_amd.estimator import Estimator, RunConfig
from recalgorithm_amd.io import synth
from tests.util import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg(request):
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    # captured graphs that hold RCCL kernel nodes, and work still queued on the push stream, must be gone before the
    # communicator is
    import gc
    import threading
    # a FAILED test's traceback can keep a captured graph alive, and destroying the communicator under it has been seen
    # to block for minutes: bound the teardown (the run has failed anyway)
    guard = None
    if request.session.testsfailed:
        guard = threading.Timer(60.0, lambda: os._exit(1))
        guard.daemon = True
        guard.start()
    gc.collect()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    if guard is not None:
        guard.cancel()


def _make(model, dev, batch_norm=True, before_build=None):
    spec = synth.SynthSpec(n_fields=8, max_vocab=300, seed=31, oov_frac=0.05,
                           with_history=(model == "din"), with_dense=(model == "din"))
    cats = [fc.categorical_column_with_identity(n, v) for n, v in zip(spec.names, spec.vocabs)]
    if model == "dcn":
        from recalgorithm_amd.algorithm.DCN.dcn import dcn_model_fn as fn
        params = {"category_feature_columns": [fc.embedding_column(c, 16) for c in cats], "dense_feature_columns": [],
                  "hidden_units": ["32", "16"], "num_cross_layer": 2, "learning_rate": 0.005}
    elif model == "deepfm":
        from recalgorithm_amd.algorithm.DeepFM.deepfm import deepfm_model_fn as fn
        params = {"first_order_feature_columns": [fc.indicator_column(c) for c in cats],
                  "second_order_feature_columns": [fc.embedding_column(c, 16) for c in cats],
                  "hidden_units": ["32", "16"], "dropout_rate": 0.0, "batch_norm": batch_norm, "learning_rate": 0.005}
    else:
        from recalgorithm_amd.algorithm.DIN.din import din_model_fn as fn
        cmap = dict(zip(spec.names, cats))
        his = fc.categorical_column_with_identity("his_read_comment_7d_seq", cmap["feedid"].num_buckets)
        his.is_sequence = True
        feed = cmap.pop("feedid")
        feed.is_sequence = True
        shared = fc.shared_embedding_columns([feed, his], 16, combiner="mean")
        params = {"dense_feature_columns": [], "category_feature_columns": [fc.embedding_column(c, 16) for c in cmap.values()],
                  "target_feedid_feature_columns": [shared[0]], "sequence_feature_columns": [shared[1]],
                  "hidden_units": ["32", "16"], "dropout_rate": 0.0, "batch_norm": True, "learning_rate": 0.005,
                  "activation": "dice", "mini_batch_aware_regularization": True, "l2_lambda": 0.2,
                  "use_softmax": False}
    est = Estimator(fn, params, RunConfig(device=dev, seed=9, use_hip_graph=False))
    feats, labels, _ = synth.device_features(spec, 192, dev)
    if before_build is not None:
        before_build(est)
    est.build(feats, labels)
    # DIN's alpha = 1 makes dice/prelu the identity and the fcn stack affine: whole families of
    # gradients (biases and BN offsets ahead of the next BatchNorm) then vanish analytically and their
    # Adam steps are rounding noise — move alpha away so that every parameter has a real gradient
    g = torch.Generator().manual_seed(99)
    for name, v in est.store.vars.items():
        if "alpha" in name:
            v.data.copy_((0.25 + 0.5 * torch.rand(v.data.shape, generator=g)).to(dev))
    return est, feats, labels


@pytest.mark.parametrize("model", ["dcn", "deepfm", "din"])
def test_sharded_single_rank_equals_plain(dev, pg, model):
    from recalgorithm_amd.parallel import attach_data_parallel, unshard_arena
    ref, feats, labels = _make(model, dev)
    shd, _, _ = _make(model, dev)
    attach_data_parallel(shd, pg)
    assert all(getattr(a, "sharding", None) is not None for a in shd.store.arenas.values())
    for step in range(3):
        l0 = ref.train_step(feats, labels)
        l1 = shd.train_step(feats, labels)
        assert_close(l1, l0, what=f"{model} loss step {step}")
    # the OWNER side of the exchange ran on the owner-computes path: the shard has deferred-Adam state, no live-row list
    from recalgorithm_amd import sparse
    for a in shd.store.arenas.values():
        assert sparse.plan_of(a) is not None and sparse.plan_of(a).last_step is not None and a.live is None
    a0, a1 = ref.store.named_arrays(), shd.store.named_arrays(gather=True)       # (tables of a sharded arena: collective gather)
    for k in a0:
        if "embedding_weights" in k or "kernel/" in k:
            continue
        if re.search(r"/dense(_\d+)?/bias$", k) and any(n.startswith(k.rsplit("/", 2)[0] + "/batch_normalization") for n in a0):
            continue    # a bias ahead of a training-mode BatchNorm: zero gradient analytically, its Adam step is rounding noise
        assert_close(a1[k], a0[k], rtol=3e-4, what=f"{model} {k} after 3 steps", reduced=True)
    for name, ar in ref.store.arenas.items():
        full = unshard_arena(shd.store.arenas[name], "weight")
        assert_close(full, ar.weight, rtol=3e-4, what=f"{model} arena {name} after 3 steps", reduced=True)


def test_attach_before_build_creates_the_arenas_sharded(dev, pg):
    """The production order: attach_data_parallel on a not yet built Estimator -> the arenas are materialised as
    this rank's rows only (here: all of them, world 1) with the single-process initial values, and training matches."""
    from recalgorithm_amd.parallel import attach_data_parallel, unshard_arena
    ref, feats, labels = _make("deepfm", dev)
    shd, _, _ = _make("deepfm", dev, before_build=lambda est: attach_data_parallel(est, pg))
    for name, ar in ref.store.arenas.items():
        assert shd.store.arenas[name].sharding is not None
        assert torch.equal(unshard_arena(shd.store.arenas[name], "weight"), ar.weight)
    for step in range(2):
        assert_close(shd.train_step(feats, labels), ref.train_step(feats, labels), what=f"loss step {step}")


@pytest.mark.parametrize("M,vocab,oov", [(106496, 5000, 0.05), (4097, 50, 0.3), (1000, 1 << 40, 0.0), (64, 3, 0.5), (1, 1, 0.0),
                                         (0, 1, 0.0)])
def test_dedup_rows_kernel_against_torch(dev, M, vocab, oov):
    """recalgo_dedup_rows (hash table, atomicCAS + atomicMin) vs the sort-based restatement: the representative of a
    row is its SMALLEST request index, so the result does not depend on the arrival order — bit-identical."""
    from recalgorithm_amd.parallel import hip_dedup_rows
    from tests.dist_doubles import torch_dedup_rows
    g = torch.Generator().manual_seed(M + 1)
    rows = torch.randint(0, vocab, (M,), generator=g, dtype=torch.int64)
    rows = torch.where(torch.rand(M, generator=g) < oov, torch.full_like(rows, -1), rows).to(dev)
    for _ in range(2):                       # second call: fresh workspace contents must not matter
        u, rep = hip_dedup_rows(rows)
        u_ref, rep_ref = torch_dedup_rows(rows)
        assert torch.equal(rep, rep_ref) and torch.equal(u, u_ref)
    if M:
        assert int((u >= 0).sum()) == int(torch.unique(rows[rows >= 0]).numel())


def test_static_exchange_step_is_graph_capturable(dev, pg):
    """The fixed-capacity exchange has static shapes and no host sync: the sharded training step
    (RCCL all_to_alls included) is captured into one hipGraph and replays to the eager result."""
    from recalgorithm_amd.estimator import GraphedTrainStep
    from recalgorithm_amd.parallel import attach_data_parallel, exchange_overflowed
    eager, feats, labels = _make("dcn", dev)
    graphd, _, _ = _make("dcn", dev)
    attach_data_parallel(eager, pg)
    attach_data_parallel(graphd, pg)
    g = GraphedTrainStep(graphd.train_step, feats, labels, warmup=2)      # 2 eager steps + 1 captured (not run)
    for _ in range(2):
        eager.train_step(feats, labels)
    for _ in range(3):
        l0 = eager.train_step(feats, labels)
        l1 = g()
        assert_close(l1, l0, what="graph replay loss vs eager", rtol=1e-5)
    assert not exchange_overflowed(graphd)


@pytest.mark.timeout(240, method="thread")
def test_captured_rccl_step_with_side_stream_replays_120_times(dev, pg):
    """the sharded DeepFM step — RCCL all_to_alls, the gradient push on the second stream (fork / join by
    events inside the capture) — as ONE hipGraph, replayed 120 times on four rotating batches, against the same 120 steps
    launched eagerly: the loss trajectory tracks the eager one (the sharded scatter's float atomics make the two runs
    differ by rounding, which 120 Adam steps amplify: tight over the first 20 steps, loose over all), the tables agree at
    the end, no bucket overflow, and the replay loop stays healthy (no replay slower than 20 x the median).
    Everything is copied to the host and the graph is destroyed BEFORE anything is asserted: a failing assertion must not
    keep a captured graph with RCCL nodes alive past the communicator (the module fixture's teardown)."""
    import gc
    import time
    from recalgorithm_amd import parallel
    from recalgorithm_amd.estimator import GraphedTrainStep
    from recalgorithm_amd.parallel import attach_data_parallel, exchange_overflowed, unshard_arena
    spec = synth.SynthSpec(n_fields=8, max_vocab=300, seed=31, oov_frac=0.05)
    batches = [synth.device_features(spec, 192, dev, batch_index=i)[:2] for i in range(4)]
    eager, feats, labels = _make("deepfm", dev)
    graphd, _, _ = _make("deepfm", dev)
    attach_data_parallel(eager, pg)
    attach_data_parallel(graphd, pg)
    g = GraphedTrainStep(graphd.train_step, feats, labels, warmup=2)
    for _ in range(2):
        eager.train_step(feats, labels)
    le, lg, dts = [], [], []
    for i in range(120):
        f, l = batches[i % 4]
        le.append(eager.train_step(f, l).detach().reshape(()).clone())
        t0 = time.perf_counter()
        lg.append(g(f, l).detach().reshape(()).clone())          # (the graph returns its static loss buffer)
        if i % 10 == 9:
            torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    le, lg = torch.stack(le).cpu().double(), torch.stack(lg).cpu().double()
    overflow = exchange_overflowed(graphd)
    used_side_stream = bool(parallel._push_streams)
    tables = {}
    for name, ar in eager.store.arenas.items():
        tables[name] = (unshard_arena(graphd.store.arenas[name], "weight").cpu().double(), unshard_arena(ar, "weight").cpu().double())
    del g, graphd, eager, ar
    gc.collect()
    torch.cuda.synchronize()
    # ---- nothing below holds a graph or an estimator -------------------------------------------------------------
    assert bool(torch.isfinite(lg).all())
    rel = ((lg - le).abs() / le.abs().clamp(min=1e-6))
    assert float(rel[:20].max()) < 1e-3, f"replayed losses leave the eager ones within 20 steps: {float(rel[:20].max()):.3g}"
    assert float(rel.max()) < 5e-2, f"replayed loss trajectory diverges from the eager one: {float(rel.max()):.3g}"
    assert float(lg[-4:].mean()) < float(lg[:4].mean()), "the replayed run does not train"
    assert not overflow
    assert used_side_stream, "the gradient push never used its side stream"
    for name, (a, b) in tables.items():
        err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-9)
        assert err < 5e-2, f"arena {name} after 120 steps: max deviation {err:.3g} of the table's scale"
    med = sorted(dts)[len(dts) // 2]
    assert max(dts) < max(20 * med, 0.5), f"a replay stalled: max {max(dts):.3f}s, median {med * 1e3:.2f} ms"


@pytest.mark.parametrize("world,M,capf", [(1, 1000, 1.0), (2, 4097, 2.0), (3, 999, 1.5), (8, 106496, 2.0), (8, 106496, 0.9),
                                          (64, 20000, 2.0), (100, 5000, 3.0), (4, 63, 4.0), (8, 0, 2.0)])
def test_exchange_plan_kernel_against_torch_bucketing(dev, world, M, capf):
    """recalgo_exchange_plan vs the stable torch bucketing (tests/dist_doubles.py): same overflow flag,
    same bucket contents as multisets (slot order inside a bucket is unspecified), and send_local /
    send_pos / req_slot mutually consistent."""
    import ctypes
    from recalgorithm_amd import _lib
    from tests.dist_doubles import torch_exchange_plan
    lib = _lib.load()
    g = torch.Generator().manual_seed(world * 7919 + M)
    rows = torch.randint(0, 50000, (M,), generator=g)
    rows = torch.where(torch.rand(M, generator=g) < 0.3, (rows % 7) * world, rows)        # hot rows on owner 0
    rows = torch.where(torch.rand(M, generator=g) < 0.1, torch.full_like(rows, -1), rows).to(dev)
    cap = max(8, int(M * capf / world + 7) // 8 * 8)
    ovf_ref = torch.zeros(1, dtype=torch.bool, device=dev)
    sl_ref, rs_ref = torch_exchange_plan(rows, world, cap, ovf_ref)
    send_local = torch.empty(world * cap, dtype=torch.int64, device=dev)
    send_pos = torch.empty(world * cap, dtype=torch.int64, device=dev)
    req_slot = torch.empty(M, dtype=torch.int64, device=dev)
    counters = torch.empty(world, dtype=torch.int32, device=dev)
    ovf = torch.zeros(1, dtype=torch.bool, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.recalgo_exchange_plan(p(rows), M, world, cap, p(send_local), p(send_pos), p(req_slot), p(counters), p(ovf), st),
               "recalgo_exchange_plan")
    torch.cuda.synchronize()
    assert bool(ovf.item()) == bool(ovf_ref.item())
    valid = rows >= 0
    want = torch.bincount((rows[valid] % world), minlength=world)
    assert torch.equal(counters.long(), want)                                   # requests per owner
    filled = (send_local.reshape(world, cap) >= 0).sum(1)
    assert torch.equal(filled, want.clamp(max=cap))
    if M:
        ok = req_slot >= 0
        assert not bool((ok & ~valid).any())
        assert int(ok.sum()) == int(filled.sum())
        assert torch.equal(send_local[req_slot[ok]], rows[ok] // world)
        assert torch.equal(req_slot[ok] // cap, rows[ok] % world)
        assert torch.equal(send_pos[req_slot[ok]], torch.nonzero(ok).squeeze(1))
        assert int((send_pos >= 0).sum()) == int(ok.sum())
    if not bool(ovf.item()):
        assert bool((req_slot[valid] >= 0).all()) if M else True
        a = send_local.reshape(world, cap).sort(1).values
        b = sl_ref.reshape(world, cap).sort(1).values
        assert torch.equal(a, b)
    # send_pos is optional
    sl2 = torch.empty_like(send_local)
    _lib.check(lib.recalgo_exchange_plan(p(rows), M, world, cap, p(sl2), None, p(req_slot), p(counters), p(ovf), st), "plan")
    torch.cuda.synchronize()
    if not bool(ovf.item()):          # which requests an overflowing bucket drops is unspecified
        assert torch.equal(sl2.reshape(world, cap).sort(1).values, send_local.reshape(world, cap).sort(1).values)
    assert torch.equal((sl2.reshape(world, cap) >= 0).sum(1), filled)
