"""CPU: the host side of the owner-computes path (recalgorithm_amd/sparse.py) — which launches it asks the C-ABI for, in
which order and with which plan layout — against a recording stand-in for librecalgo_hip.so.  (What the launches compute
is tests/test_gpu_sparse.py's business; nothing here touches a GPU or oracle/.)"""
import ctypes

import pytest
import torch

from recalgorithm_amd import _lib, sparse
from recalgorithm_amd.variables import EmbeddingArena


class _FakeLib:
    """Records the scatter entry points' calls; sizes like the real library."""

    def __init__(self):
        self.calls = []

    def recalgo_scatter_plan_buckets_log2(self, n):
        return 10

    def recalgo_scatter_plan_workspace_bytes(self, n, l, K):
        return 4096 + 64 * int(n)

    def recalgo_scatter_plan_header_bytes(self, l):
        return 64 + 8 * (1 << int(l))

    def recalgo_scatter_prepare(self, src, K, ws, cap, nb, first, flags, deferred, comp_deferred, rows, comp_rows, period, step,
                                off, stream):
        n_ex, F = (src.n_ex, src.F) if src is not None else (0, 0)
        self.calls.append(("prepare", {"K": K, "first": int(first), "count": bool(flags & sparse.PREPARE_COUNT),
                                       "sweep": bool(flags & sparse.PREPARE_SWEEP), "deferred": deferred is not None,
                                       "companion_deferred": comp_deferred is not None, "n_ex": n_ex, "F": F,
                                       "rows": int(rows), "step_offset": int(off)}))
        return 0

    def recalgo_scatter_apply(self, arr, n, comp, K, ws, cap, nb, mode, w, m, v, grad, deferred, rows, live, step, off,
                              lr, b1, b2, eps, stream):
        self.calls.append(("apply", {"n_sources": int(n), "companion": comp is not None, "K": K, "mode": int(mode),
                                     "deferred": deferred is not None, "grad": grad is not None, "rows": int(rows),
                                     "sources": [(arr[i].n_ex, arr[i].F, bool(arr[i].g)) for i in range(n)]}))
        return 0

    def recalgo_scatter_plan_scan(self, ws, cap, nb, out):
        self.calls.append(("plan_scan", {"cap": int(cap), "nb": int(nb)}))
        return 0

    def recalgo_adam_deferred_sweep(self, d, K, r0, r1, step, off, stream):
        self.calls.append(("sweep", {"K": K, "rows": (int(r0), int(r1))}))
        return 0


class _Store:
    def __init__(self):
        self.opt_state = {"step": torch.zeros(1, dtype=torch.int64), "lr_t": torch.zeros(1)}
        self.arenas = {}


@pytest.fixture()
def lib(monkeypatch):
    fake = _FakeLib()
    monkeypatch.setattr(_lib, "load", lambda *a, **k: fake)
    monkeypatch.setattr(sparse, "_supported", lambda arena: True)
    monkeypatch.setattr(sparse, "_stream", lambda t: None)
    monkeypatch.setattr(sparse.ctypes, "byref", lambda x: x)          # (the stand-in reads the structs directly)
    monkeypatch.setattr(sparse, "SCATTER_MODE", "owner")
    monkeypatch.setattr(sparse, "COMPANION", True)
    return fake


def _arena(rows, K, name):
    ar = EmbeddingArena(name, K, "cpu", seed=1)
    ar.add_table("t0", rows)
    ar.materialize()
    return ar


def _ids(n_ex, F, rows=50, seed=0):
    return torch.randint(0, rows, (n_ex, F), generator=torch.Generator().manual_seed(seed))


def test_slot_space_of_a_plan():
    """Tiles of 256: an id matrix takes F x (examples rounded up to 256) slots (field-major), a ragged source its
    n_ex * F requests rounded up — the layout recalgo_scatter_source_slots documents."""
    assert sparse.Source(_ids(300, 5), None, None, 0, 300, 5).slots == 5 * 512
    assert sparse.Source(_ids(256, 1), None, None, 0, 256, 1).slots == 256
    assert sparse.Source(torch.arange(10), torch.arange(4), None, 0, 3, 50).slots == 256
    assert sparse.Source(torch.arange(10), torch.arange(4), None, 0, 30, 50).slots == 1536
    assert sparse.Source(_ids(0, 3), None, None, 0, 0, 3).slots == 0


def test_lookups_join_one_plan_and_the_optimizer_applies_them_once(lib):
    E, st = _arena(100, 16, "e"), _Store()
    st.arenas["e"] = E

    def step():
        s1 = sparse.begin_lookup(E, st, _ids(300, 5), None, None, 0, 300, 5, True)
        s2 = sparse.begin_lookup(E, st, _ids(40, 1), None, None, 0, 40, 1, True)
        s1.set_grad(torch.zeros(300, 5 * 16))
        s2.set_grad(torch.zeros(40, 16))
        st.opt_state["step"] += 1
        sparse.apply(E, False, st.opt_state["step"], 0.01, 0.9, 0.999, 1e-8)

    # first step: the workspace grows with the second lookup (grow-only), which invalidates the first lookup's counts — the
    # optimizer call takes them again; no lookup's launch could carry the sweep (no optimizer state yet): its own launch,
    # up to the step before this one
    step()
    kinds = [c[0] for c in lib.calls]
    assert kinds == ["prepare", "prepare", "prepare", "prepare", "prepare", "apply"]
    assert lib.calls[0][1]["first"] == 0 and lib.calls[1][1]["first"] == 5 * 512        # second source after the first's slots
    assert all(c[1]["count"] and not c[1]["sweep"] for c in lib.calls[:4])
    assert not lib.calls[0][1]["deferred"]                                              # no optimizer state before the first step
    sw = lib.calls[4][1]
    assert sw["sweep"] and not sw["count"] and sw["n_ex"] == 0 and sw["deferred"] and sw["rows"] == 100 and sw["step_offset"] == -1
    a = lib.calls[-1][1]
    assert a["n_sources"] == 2 and a["mode"] == sparse.MODE_ADAM and a["deferred"] and not a["companion"]
    plan = sparse.plan_of(E)
    assert plan.sources == [] and plan.last_step is not None and int(plan.last_step.max()) == 0     # fresh arena: no row has state
    # steady state: ONE launch per lookup (counts + catch-up of its lagging rows; the arena's first lookup of the step also
    # carries the sweep), then the optimizer's place + apply: three launches for a one-lookup model, no recount
    n0 = len(lib.calls)
    step()
    assert [c[0] for c in lib.calls[n0:]] == ["prepare", "prepare", "apply"]
    assert all(c[1]["deferred"] and c[1]["count"] for c in lib.calls[n0:n0 + 2])
    assert [c[1]["sweep"] for c in lib.calls[n0:n0 + 2]] == [True, False]


def test_lookups_outside_training_are_not_registered(lib):
    E, st = _arena(100, 16, "e"), _Store()
    assert sparse.begin_lookup(E, st, _ids(8, 2), None, None, 0, 8, 2, False) is None
    E.trainable = False
    assert sparse.begin_lookup(E, st, _ids(8, 2), None, None, 0, 8, 2, True) is None
    assert lib.calls == []


def test_a_forward_without_backward_is_recounted(lib):
    """A lookup whose output never received a gradient is dropped at the optimizer: the counts in the workspace then
    describe more requests than are applied, so they are taken again for exactly the sources with gradients."""
    E, st = _arena(100, 8, "e"), _Store()
    s1 = sparse.begin_lookup(E, st, _ids(64, 2), None, None, 0, 64, 2, True)
    sparse.begin_lookup(E, st, _ids(64, 1, seed=1), None, None, 0, 64, 1, True)          # no gradient
    s1.set_grad(torch.zeros(64, 16))
    st.opt_state["step"] += 1
    sparse.apply(E, True, st.opt_state["step"], 0.01, 0.9, 0.999, 1e-8)
    kinds = [c[0] for c in lib.calls]
    assert kinds == ["prepare", "prepare", "prepare", "apply"]
    assert lib.calls[2][1]["first"] == 0 and not lib.calls[2][1]["deferred"]
    assert lib.calls[3][1]["n_sources"] == 1 and lib.calls[3][1]["mode"] == sparse.MODE_LAZY_ADAM
    assert sparse.plan_of(E).last_step is None                                           # LazyAdam keeps no deferred state


def test_companion_arena_rides_on_the_main_plan(lib):
    E, W, st = _arena(100, 16, "e"), _arena(100, 1, "w"), _Store()
    st.arenas.update(e=E, w=W)
    ids = _ids(300, 3)
    s, s1 = sparse.begin_lookup_pair(E, W, st, ids, None, 300, 3, True)
    assert isinstance(s1, sparse.CompanionSource) and s.companion is s1
    assert [c[0] for c in lib.calls] == ["prepare"]                                      # ONE prepare for both arenas
    assert sparse.has_companions(W) and not sparse.has_companions(E)
    with pytest.raises(RuntimeError):
        sparse.apply(W, False, st.opt_state["step"], 0.01, 0.9, 0.999, 1e-8)             # the main arena must go first
    s.set_grad(torch.zeros(300, 48))
    s1.set_grad(torch.zeros(300, 1), fmul=0)
    st.opt_state["step"] += 1
    for ar in sorted((W, E), key=sparse.has_companions):                                 # (what the Estimator does)
        sparse.apply(ar, False, st.opt_state["step"], 0.01, 0.9, 0.999, 1e-8)
    applies = [c[1] for c in lib.calls if c[0] == "apply"]
    assert len(applies) == 1 and applies[0]["companion"] and applies[0]["K"] == 16       # W's step ran with E's launches
    assert sparse.plan_of(W).last_step is not None and not sparse.has_companions(W)
    # next step: E's prepare also catches the companion's rows up
    sparse.begin_lookup_pair(E, W, st, ids, None, 300, 3, True)
    assert lib.calls[-1][0] == "prepare" and lib.calls[-1][1]["companion_deferred"]


def test_companion_is_dissolved_when_the_arena_is_also_looked_up_alone(lib):
    E, W, st = _arena(100, 8, "e"), _arena(100, 1, "w"), _Store()
    s, s1 = sparse.begin_lookup_pair(E, W, st, _ids(64, 2), None, 64, 2, True)
    s2 = sparse.begin_lookup(W, st, _ids(10, 1), None, None, 0, 10, 1, True)
    assert s.companion is None and s1.regular is not None and not sparse.has_companions(W)
    s.set_grad(torch.zeros(64, 16)); s1.set_grad(torch.zeros(64, 1), fmul=0); s2.set_grad(torch.zeros(10, 1))
    st.opt_state["step"] += 1
    sparse.apply(E, False, st.opt_state["step"], 0.01, 0.9, 0.999, 1e-8)
    assert not lib.calls[-1][1]["companion"]
    n0 = len(lib.calls)
    sparse.apply(W, False, st.opt_state["step"], 0.01, 0.9, 0.999, 1e-8)
    kinds = [(c[0], c[1].get("count"), c[1].get("sweep")) for c in lib.calls[n0:]]
    assert kinds == [("prepare", True, False), ("prepare", True, False), ("prepare", False, True), ("apply", None, None)]
    assert lib.calls[-1][1]["n_sources"] == 2                                                  # both lookups of W, recounted
    assert sorted(x[:2] for x in lib.calls[-1][1]["sources"]) == [(10, 1), (64, 2)]


def test_lazy_adam_gives_up_a_partial_companion(lib):
    """LazyAdam must touch exactly the rows of the second arena's own lookups: when only some lookups of the main plan carry
    the companion, the arenas run separately."""
    E, W, st = _arena(100, 8, "e"), _arena(100, 1, "w"), _Store()
    s, s1 = sparse.begin_lookup_pair(E, W, st, _ids(64, 2), None, 64, 2, True)
    s3 = sparse.begin_lookup(E, st, _ids(32, 1), None, None, 0, 32, 1, True)              # E alone
    s.set_grad(torch.zeros(64, 16)); s1.set_grad(torch.zeros(64, 1), fmul=0); s3.set_grad(torch.zeros(32, 8))
    st.opt_state["step"] += 1
    for ar in sorted((W, E), key=sparse.has_companions):
        sparse.apply(ar, True, st.opt_state["step"], 0.01, 0.9, 0.999, 1e-8)
    applies = [c[1] for c in lib.calls if c[0] == "apply"]
    assert len(applies) == 2 and not applies[0]["companion"] and applies[0]["n_sources"] == 2
    assert applies[1]["K"] == 1 and applies[1]["n_sources"] == 1 and applies[1]["mode"] == sparse.MODE_LAZY_ADAM


def test_more_lookups_than_sources_are_merged(lib):
    E, st = _arena(100, 4, "e"), _Store()
    srcs = [sparse.begin_lookup(E, st, _ids(8, 1, seed=i), None, None, 0, 8, 1, True) for i in range(20)]
    for s in srcs:
        s.set_grad(torch.ones(8, 4))
    assert all(s_.caught_up for s_ in srcs)                                              # every lookup's launch catches its rows up
    st.opt_state["step"] += 1
    sparse.apply(E, False, st.opt_state["step"], 0.01, 0.9, 0.999, 1e-8)
    a = lib.calls[-1][1]
    assert a["n_sources"] == 1 and a["sources"][0][:2] == (160, 1)                       # 20 x 8 requests as one source of rows


def test_gradient_arena_on_request_then_optimizer(lib):
    """Reading arena.grad (tests, named_grads) runs the plan in GRAD mode; the optimizer call that follows takes the counts
    again and passes the gradient arena so that the touched rows are zeroed."""
    E, st = _arena(100, 8, "e"), _Store()
    st.arenas["e"] = E
    s = sparse.begin_lookup(E, st, _ids(64, 2), None, None, 0, 64, 2, True)
    s.set_grad(torch.zeros(64, 16))
    _ = E.grad
    assert lib.calls[-1][0] == "apply" and lib.calls[-1][1]["mode"] == sparse.MODE_GRAD and lib.calls[-1][1]["grad"]
    _ = E.grad                                                                          # (once per plan)
    assert sum(c[0] == "apply" for c in lib.calls) == 1
    st.opt_state["step"] += 1
    sparse.apply(E, False, st.opt_state["step"], 0.01, 0.9, 0.999, 1e-8)
    tail = [(c[0], c[1].get("count"), c[1].get("sweep")) for c in lib.calls[-3:]]
    assert tail == [("prepare", True, False), ("prepare", False, True), ("apply", None, None)] and lib.calls[-1][1]["grad"]


def test_whole_table_readers_flush_the_deferred_state(lib):
    E, st = _arena(100, 8, "e"), _Store()
    st.arenas["e"] = E
    sparse.sync_store(st)
    assert lib.calls == []                                                               # nothing deferred yet
    s = sparse.begin_lookup(E, st, _ids(8, 1), None, None, 0, 8, 1, True)
    s.set_grad(torch.zeros(8, 8))
    st.opt_state["step"] += 1
    sparse.apply(E, False, st.opt_state["step"], 0.01, 0.9, 0.999, 1e-8)
    sparse.sync_store(st)
    assert lib.calls[-1] == ("sweep", {"K": 8, "rows": (0, 100)})
    sparse.sync_arena(E)                                                                # (callers without a store: parallel.unshard_arena)
    assert lib.calls[-1][0] == "sweep" and len([c for c in lib.calls if c[0] == "sweep"]) == 2
    sparse.reset(E)
    assert sparse.plan_of(E).last_step is None


def test_scatter_mode_hook_and_sweep_period_knob(monkeypatch):
    assert sparse.scatter_mode() == "owner"
    monkeypatch.setattr(sparse, "SCATTER_MODE", "atomic")
    assert sparse.scatter_mode() == "atomic"
    monkeypatch.setattr(sparse, "SCATTER_MODE", "bogus")
    with pytest.raises(ValueError):
        sparse.scatter_mode()
    monkeypatch.setenv("RECALGO_ADAM_SWEEP_PERIOD", "0")
    with pytest.raises(ValueError):
        sparse.sweep_period()


def test_the_plan_prefix_scan_rides_on_the_optimizer_launch_only_when_the_counts_are_those_of_the_applied_sources(lib):
    """sparse.plan_scan_record hands the optimizer launch a record (and `apply` then runs RECALGO_SCATTER_PRESCANNED) exactly
    when the workspace's bucket totals are those of the sources `apply` will place; whenever `apply` has to count again (first
    step: the workspace grew; a lookup without a gradient) there is no record and `place` scans the counters itself."""
    E, st = _arena(100, 16, "e"), _Store()
    st.arenas["e"] = E

    def step(with_record=True, drop_second_grad=False):
        s1 = sparse.begin_lookup(E, st, _ids(300, 5), None, None, 0, 300, 5, True)
        s2 = sparse.begin_lookup(E, st, _ids(40, 1), None, None, 0, 40, 1, True)
        s1.set_grad(torch.zeros(300, 5 * 16))
        if not drop_second_grad:
            s2.set_grad(torch.zeros(40, 16))
        st.opt_state["step"] += 1
        rec = sparse.plan_scan_record(E, False) if with_record else None
        sparse.apply(E, False, st.opt_state["step"], 0.01, 0.9, 0.999, 1e-8)
        return rec

    assert step() is None                                   # first step: the second lookup re-sized the workspace -> recount
    assert lib.calls[-1][0] == "apply" and lib.calls[-1][1]["mode"] == sparse.MODE_ADAM
    n0 = len(lib.calls)
    assert step() is not None                               # steady state: counted == the applied sources
    assert [c[0] for c in lib.calls[n0:]] == ["prepare", "prepare", "plan_scan", "apply"]
    assert lib.calls[-1][1]["mode"] == sparse.MODE_ADAM | sparse.MODE_PRESCANNED
    step(with_record=False)                                 # an optimizer path without the fused launch: no flag
    assert lib.calls[-1][1]["mode"] == sparse.MODE_ADAM
    n0 = len(lib.calls)
    assert step(drop_second_grad=True) is None              # counted two lookups, one is applied: `apply` counts again
    assert lib.calls[-1][1]["mode"] == sparse.MODE_ADAM and lib.calls[-1][1]["n_sources"] == 1
    assert step() is not None                               # ... and the step after is prescanned again
    assert lib.calls[-1][1]["mode"] == sparse.MODE_ADAM | sparse.MODE_PRESCANNED
