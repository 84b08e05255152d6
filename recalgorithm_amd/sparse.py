"""Host side of csrc/sparse.hip: the per-arena request PLAN of a training step (row-gradient scatter without float
atomics, fused with the sparse optimizer) and the deferred-exact TF1 Adam state.

    forward of a lookup   begin_lookup(arena, ...)    registers the lookup's requests as a `Source`, launches
                                                      recalgo_scatter_prepare (ONE launch: bucket counts, catch-up of the
                                                      lookup's lagging rows, the step's share of the deferred-Adam sweep)
    backward of a lookup  Source.set_grad(g)          records where the per-request gradient rows are
    optimizer             apply(arena, mode, ...)     recalgo_scatter_apply (two launches: place, apply) over all sources of the
                                                      arena (the prefix of the bucket counts rides on the dense optimizer
                                                      launch: plan_scan_record): TF1 Adam with dense semantics
                                                      evaluated lazily but exactly (tf.train.AdamOptimizer,
                                                      /root/reference algorithm/DeepFM/deepfm.py:246-250) or
                                                      tf.contrib.opt.LazyAdamOptimizer (algorithm/DIEN/dien.py:328)
    anyone reading whole  sync(arena) / sync_store    recalgo_adam_deferred_sweep: every row brought to the current step
    tables                                            (named_arrays, checkpoints, export)
    tests / tools         materialize_grads(store)    the summed row gradients written to arena.grad (GRAD mode)

This is the path of every arena the plan supports — for local arenas, and for the OWNER side of a row-sharded arena's
exchange (parallel.StagedArena: the rows the peers request are a lookup of the shard's plan); `atomic` / `sorted` keep the
round-2 kernels (LDS-aggregated float atomics / torch.sort + ordered segment sums) with the live-row-list optimizer.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional

import torch

from . import _lib

MODE_GRAD, MODE_ADAM, MODE_LAZY_ADAM = 0, 1, 2
PREPARE_COUNT, PREPARE_SWEEP, PREPARE_CATCHUP = 1, 2, 4          # include/recalgo.h RECALGO_PREPARE_*
MAX_SOURCES = 16
LR_RING = 1024


# Test hooks (module attributes, not environment knobs): tests/test_gpu_sparse.py compares the owner-computes path with the
# round-1 float-atomic scatter + live-row-list Adam (SCATTER_MODE = "atomic": what an arena outside the plan's domain — rows
# wider than 256 floats — and the requester side of a row-sharded arena still run), a companion arena with a plan of its own
# (COMPANION = False), and plans with a forced bucket count (NB_LOG2).
SCATTER_MODE = "owner"
COMPANION = True
NB_LOG2 = None


def scatter_mode() -> str:
    if SCATTER_MODE not in ("owner", "atomic"):
        raise ValueError(f"sparse.SCATTER_MODE = {SCATTER_MODE!r}: expected owner | atomic")
    return SCATTER_MODE


def sweep_period() -> int:
    """Deferred Adam: 1/P of every arena is brought up to date per step (no row lags more than P + 1 steps)."""
    p = int(os.environ.get("RECALGO_ADAM_SWEEP_PERIOD", "32"))
    if not 1 <= p <= LR_RING - 8:
        raise ValueError(f"RECALGO_ADAM_SWEEP_PERIOD={p}: expected 1 .. {LR_RING - 8}")
    return p


class _CSource(ctypes.Structure):            # include/recalgo.h recalgo_scatter_source_t
    _fields_ = [("ids", ctypes.c_void_p), ("offsets", ctypes.c_void_p), ("row_base", ctypes.c_void_p), ("base", ctypes.c_int64),
                ("n_ex", ctypes.c_int), ("F", ctypes.c_int), ("g", ctypes.c_void_p), ("g_stride", ctypes.c_int64),
                ("g_col", ctypes.c_int), ("g_fmul", ctypes.c_int),
                ("fm_scale", ctypes.c_void_p), ("fm_sum", ctypes.c_void_p), ("fm_emb", ctypes.c_void_p)]


class _CCompanion(ctypes.Structure):         # include/recalgo.h recalgo_scatter_companion_t
    _fields_ = [("sources", ctypes.c_void_p), ("w", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("grad", ctypes.c_void_p), ("deferred", ctypes.c_void_p), ("rows", ctypes.c_int64)]


class _CPlanScan(ctypes.Structure):          # include/recalgo.h recalgo_plan_scan_t
    _fields_ = [("total", ctypes.c_void_p), ("offs", ctypes.c_void_p), ("sched", ctypes.c_void_p),
                ("counter_shift", ctypes.c_uint32), ("nb_log2", ctypes.c_uint32)]


class _CDeferred(ctypes.Structure):          # include/recalgo.h recalgo_deferred_adam_t
    _fields_ = [("w", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p), ("last_step", ctypes.c_void_p),
                ("lr_ring", ctypes.c_void_p), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float)]


class Source:
    """One lookup's requests: ids [n_ex, F] (offsets None) or ragged (values, offsets) with F steps per example."""

    def __init__(self, ids, offsets, row_base, base: int, n_ex: int, F: int):
        self.ids, self.offsets, self.row_base, self.base, self.n_ex, self.F = ids, offsets, row_base, int(base), int(n_ex), int(F)
        self.arena = None                      # set by begin_lookup
        self.g = None
        self.g_stride = self.g_col = self.g_fmul = 0
        self.fm = None                         # FM second-order epilogue: (g_fm2 [n_ex], field sum [n_ex, K], embeddings [n_ex, F*K])
        self.caught_up = True                  # the lookup's rows are current when its forward kernel runs
        self.companion: Optional["CompanionSource"] = None     # a one-float-per-row arena looked up with the same requests

    @property
    def n(self) -> int:
        return self.n_ex * self.F

    @property
    def slots(self) -> int:
        """Size of the source in the plan's slot space (include/recalgo.h recalgo_scatter_source_slots): whole tiles of 256;
        an id matrix takes F fields x (examples rounded up to 256), a ragged source its n_ex * F requests rounded up."""
        if self.n == 0:
            return 0
        pad = lambda x: (x + 255) // 256 * 256
        return pad(self.n) if self.offsets is not None else pad(self.n_ex) * self.F

    def set_grad(self, g: torch.Tensor, fmul: Optional[int] = None, fm=None):
        """g: [n_ex, F * K] / [n_ex, F, K] / [n_ex, K'] with the last dimension contiguous; request (e, f) reads its K
        floats at g[e].flat[f * fmul : f * fmul + K] (fmul defaults to K = row width; 0: all fields share the row).
        fm = (scale [n_ex], field_sum [n_ex, K], emb [n_ex, F * K]): the request's gradient row is
        g + scale[e] * (field_sum[e] - emb[e, f]), formed by the kernels on load (DeepFM's second-order term)."""
        self.fm = None
        if fm is not None:
            sc, fs, em = (t.contiguous() for t in fm)
            assert sc.numel() == self.n_ex and fs.numel() * self.F == em.numel()
            self.fm = (sc, fs, em)
        if g.dim() == 3:
            g = g.reshape(g.shape[0], -1) if g.is_contiguous() else g.contiguous().view(g.shape[0], -1)
        if g.dim() != 2 or (g.shape[1] > 1 and g.stride(1) != 1) or g.shape[0] != self.n_ex:
            g = g.contiguous().view(self.n_ex, -1)
        self.g = g
        self.g_stride = int(g.stride(0)) if g.shape[0] > 1 else int(g.shape[1])
        self.g_col = 0
        self.g_fmul = fmul

    def c_struct(self, K: int) -> _CSource:
        p = lambda t: None if t is None else t.data_ptr()
        fm = K if self.g_fmul is None else int(self.g_fmul)
        fs = self.fm if self.fm is not None else (None, None, None)
        return _CSource(p(self.ids), p(self.offsets), p(self.row_base), self.base, self.n_ex, self.F, p(self.g),
                        self.g_stride, self.g_col, fm, p(fs[0]), p(fs[1]), p(fs[2]))


class CompanionSource:
    """The lookup of a SECOND arena of one float per row made with exactly the requests of `main` (DeepFM's first-order
    weights beside its embeddings, /root/reference algorithm/DeepFM/deepfm.py:125-141).  It has no plan of its own: the
    main arena's `place` also sums the scalar gradients of a tile's duplicates, `apply` updates this arena's row beside
    the main arena's (the lane that owns the row's first piece), and its catch-up and share of the sweep ride in the main
    arena's `prepare` launch — no launch of its own."""

    def __init__(self, main: Source, arena):
        self.main, self.arena = main, arena
        self.caught_up = False                 # its forward reads lagging rows through the deferred view
        self.g = None
        self.g_stride = self.g_col = self.g_fmul = 0
        self.regular: Optional[Source] = None  # set when the pairing was dissolved (see _dissolve)

    def set_grad(self, g: torch.Tensor, fmul: Optional[int] = None):
        """g: [n_ex, 1] (fmul = 0: every field of an example adds the same scalar) or [n_ex, F] (fmul = 1)."""
        if self.regular is not None:
            return self.regular.set_grad(g, fmul)
        g = g.reshape(self.main.n_ex, -1)
        if g.shape[1] > 1 and g.stride(1) != 1:
            g = g.contiguous()
        self.g = g
        self.g_stride = int(g.stride(0)) if g.shape[0] > 1 else int(g.shape[1])
        self.g_col = 0
        self.g_fmul = 1 if fmul is None else int(fmul)

    def c_struct(self) -> _CSource:
        m = self.main
        p = lambda t: None if t is None else t.data_ptr()
        return _CSource(p(m.ids), p(m.offsets), p(m.row_base), m.base, m.n_ex, m.F, p(self.g), self.g_stride, self.g_col, self.g_fmul,
                        None, None, None)


class ArenaPlan:
    """Per-arena state of the owner-computes path (attached as `arena.sparse`)."""

    def __init__(self, arena):
        self.arena = arena
        self.sources: List[Source] = []
        self.ws: Optional[torch.Tensor] = None
        self.capacity = 0                      # requests the workspace was sized for
        self.nb_log2 = 10
        self.counted = None                    # signature of what the workspace's bucket totals currently hold
        self.swept = False                     # this step's share of the deferred-Adam sweep has been launched
        self.prescanned = None                 # signature the bucket-total prefix (offs / sched) was computed for, this step
        self.last_step: Optional[torch.Tensor] = None     # deferred-Adam: int32 [rows]
        self.lr_ring: Optional[torch.Tensor] = None
        self.betas = (0.9, 0.999, 1e-8)
        self.grad_materialized = False
        self.step_dev: Optional[torch.Tensor] = None      # the optimizer's device step counter (sync_arena)
        self.companions: List[CompanionSource] = []       # this arena is the SECOND arena of these lookups (this step)
        self.served = False                    # ... and the main arena's optimizer call has already applied them

    # -- workspace ------------------------------------------------------------------------------
    def _ensure_ws(self, n_requests: int):
        """Grow-only: the bucket count is fixed when the workspace is allocated (re-allocating invalidates the counts)."""
        if self.ws is not None and n_requests <= self.capacity:
            return
        lib = _lib.load()
        cap = max((n_requests + 255) // 256 * 256, 256)          # slots (whole tiles)
        self.nb_log2 = int(lib.recalgo_scatter_plan_buckets_log2(cap)) if NB_LOG2 is None else int(NB_LOG2)
        self.capacity = cap
        nbytes = int(lib.recalgo_scatter_plan_workspace_bytes(cap, self.nb_log2, self.arena.K))
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.arena.weight.device)
        self.header = int(lib.recalgo_scatter_plan_header_bytes(self.nb_log2))
        self.clear_counts()

    def clear_counts(self):
        """Bucket totals and cursors back to zero (`apply` leaves them that way; needed on a fresh workspace and whenever
        counted sources are dropped without having been applied)."""
        self.ws[:self.header].zero_()
        self.counted = (self.ws.data_ptr(), self.nb_log2, ())

    def _signature(self, sources):
        return (self.ws.data_ptr(), self.nb_log2, tuple(id(s) for s in sources))

    def _deferred_struct(self):
        if self.last_step is None:
            return None
        a = self.arena
        b1, b2, eps = self.betas
        return _CDeferred(a.weight.data_ptr(), a.m.data_ptr(), a.v.data_ptr(), self.last_step.data_ptr(),
                          self.lr_ring.data_ptr(), b1, b2, eps)


def _stream(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _supported(arena) -> bool:
    from .variables import EmbeddingArena
    if type(arena) is not EmbeddingArena or arena.weight is None or not arena.weight.is_cuda:
        return False
    K = arena.K
    return arena.weight.shape[0] < (1 << 31) and ((K % 4 == 0 and K // 4 <= 64) or K <= 64)


def plan_of(arena) -> Optional[ArenaPlan]:
    return getattr(arena, "sparse", None)


def view_for(src, arena, store):
    """The read view a lookup's forward kernel needs: none when the lookup was registered and its rows caught up."""
    if src is not None and src.caught_up:
        return None, None
    return deferred_view(arena, store)


def deferred_view(arena, store):
    """(recalgo_deferred_adam_t by reference, step counter pointer) of an arena with deferred-Adam state, for the
    `_deferred` forward lookups (rows whose state lags are read as of the current step, nothing is written back);
    (None, None) for an arena without such state: the plain lookup."""
    plan = plan_of(arena)
    if plan is None or plan.last_step is None or store is None or getattr(store, "opt_state", None) is None:
        return None, None
    d = plan._deferred_struct()
    plan._view = d                              # (keeps the struct alive for the duration of the call)
    return ctypes.byref(d), ctypes.c_void_p(store.opt_state["step"].data_ptr())


def begin_lookup(arena, store, ids: torch.Tensor, offsets: Optional[torch.Tensor], row_base: Optional[torch.Tensor],
                 base: int, n_ex: int, F: int, training: Optional[bool] = None, companion_arena=None,
                 can_defer: bool = False) -> Optional[Source]:
    """Called by a lookup's forward in TRAIN mode.  Returns the Source to attach the gradient to (owner mode), or None.
    Launches recalgo_scatter_prepare: the lookup's tiles are counted into the arena's plan and — deferred Adam — its
    lagging rows are brought up to date (and the arena's share of the sweep runs).  (Lookups that are NOT registered — EVAL / PREDICT — read lagging rows through
    deferred_view instead: replayed in registers, nothing written.)"""
    plan = plan_of(arena)
    if training is None:                       # (inside an autograd Function's forward grad mode is always off: callers
        training = torch.is_grad_enabled()     #  there pass the mode of the CALL)
    training = bool(training) and getattr(arena, "trainable", True)
    if not (training and scatter_mode() == "owner" and _supported(arena)):
        return None
    if plan is None:
        plan = arena.sparse = ArenaPlan(arena)
    if plan.companions:
        _dissolve(plan)                        # (the arena is also looked up on its own this step: it needs its own plan)
    src = Source(ids, offsets, row_base, base, n_ex, F)
    src.arena = arena
    if src.n == 0:
        plan.sources.append(src)
        return src
    lib = _lib.load()
    first = sum(s.slots for s in plan.sources)
    if plan.ws is None or first + src.slots > plan.capacity:
        # (a re-sized workspace starts with clean totals: the sources already counted are counted again by the optimizer's call)
        plan._ensure_ws(first + src.slots)
    if plan.counted is None or plan.counted[:2] != (plan.ws.data_ptr(), plan.nb_log2):
        plan.clear_counts()
    cs = src.c_struct(arena.K)
    # deferred Adam: the launch also brings the lookup's lagging rows up to date (claimed once per row, written back), so that
    # the forward kernel that follows — and `apply` at the end of the step — find them current; the first lookup of an arena
    # in a step carries the step's share of the sweep
    d = plan._deferred_struct()
    src.caught_up = True
    step = None if d is None else store.opt_state["step"]
    d1, c_rows = None, 0
    if companion_arena is not None and _pair(src, arena, companion_arena) and d is not None:
        cp = plan_of(companion_arena)
        d1 = cp._deferred_struct()             # (the second arena's rows are caught up, and swept, by the same launch)
        c_rows = companion_arena.weight.shape[0] if d1 is not None else 0
    # Only the catch-up has to precede the lookup's forward kernel; the bucket counts (needed by `place`, after the backward
    # pass) and the step's share of the sweep (needed by nobody before the next step) ride in the same launch.  (Tried in
    # round 4: those two on a side stream beside the backward pass — DCN 0.248 vs 0.242 ms, DIN 0.658 vs 0.608: the fork /
    # join of the second stream costs more than the overlap buys; removed.)
    flags = (0 if d is None else PREPARE_CATCHUP) | PREPARE_COUNT
    if d is not None and not plan.swept:
        flags |= PREPARE_SWEEP
        plan.swept = True
        if d1 is not None:
            plan_of(companion_arena).swept = True
    # (can_defer: the caller enqueues its forward kernel through defer_launch when the Source comes back `deferred` — a caller
    # that does not would read rows the pending launch has not caught up yet)
    if (_batch is not None and can_defer and d1 is None and companion_arena is None and getattr(arena, "sharding", None) is None
            and "__staged__" not in getattr(arena, "tables", {})):
        # the model issues several lookups together (batch_lookups): ONE launch for all of an arena's at the end of the block; the
        # forward kernels of these lookups are enqueued behind it (ops: defer_launch)
        src.deferred = True
        _batch["arenas"].setdefault(id(arena), (arena, plan, []))[2].append((src, cs, first, flags, d, step))
    elif flags:
        _lib.check(lib.recalgo_scatter_prepare(ctypes.byref(cs), arena.K, ctypes.c_void_p(plan.ws.data_ptr()), plan.capacity, plan.nb_log2,
                                               first, flags, None if d is None else ctypes.byref(d),
                                               None if d1 is None else ctypes.byref(d1), arena.weight.shape[0], c_rows, sweep_period(),
                                               None if step is None else ctypes.c_void_p(step.data_ptr()), 0, _stream(arena.weight)),
                   "recalgo_scatter_prepare")
    plan.sources.append(src)
    plan.counted = plan.counted[:2] + (plan.counted[2] + (id(src),),)
    return src


# ---- several lookups issued together: one `prepare` launch per arena -------------------------------------------------------------
_batch = None                # the active batch_lookups() block: {"arenas": {id(arena): (arena, plan, [entries])}, "launches": [fn]}
MAX_PREPARE_SOURCES = 4      # recalgo_scatter_prepare_multi
BATCH_LOOKUPS = True         # False: batch_lookups() blocks change nothing (tests compare the two)


class batch_lookups:
    """with sparse.batch_lookups(): a = fc.input_layer(..); b, n = fc.sequence_input_layer(..)
    The TRAIN lookups issued inside register with their arena's plan as always, but their `prepare` work (bucket counts, catch-up
    of lagging rows, the step's share of the sweep) is ONE launch per arena at the end of the block instead of one per lookup,
    and their forward kernels run behind it (DIN: three launches of 6 / 17 / 17 us -> one).  Nothing inside the block may READ a
    lookup's output (the tensors are returned unfilled); code that has to — a torch.cat of per-column outputs — calls flush_batch()
    first."""

    def __enter__(self):
        global _batch
        self.prev = _batch
        if BATCH_LOOKUPS:
            _batch = {"arenas": {}, "launches": []}
        return self

    def __exit__(self, et, ev, tb):
        global _batch
        if not BATCH_LOOKUPS and _batch is self.prev:
            return False
        b, _batch = _batch, self.prev
        if et is None and b is not None:
            _flush_batch(b)
        return False


def batching() -> bool:
    return _batch is not None


class _CLookupJob(ctypes.Structure):         # include/recalgo.h recalgo_lookup_job_t
    _fields_ = [("kind", ctypes.c_int), ("ids", ctypes.c_void_p), ("aux", ctypes.c_void_p), ("table", ctypes.c_void_p),
                ("B", ctypes.c_int), ("F_or_T", ctypes.c_int), ("K", ctypes.c_int), ("out", ctypes.c_void_p),
                ("out_stride", ctypes.c_int), ("out_col", ctypes.c_int), ("seq_len", ctypes.c_void_p)]


def defer_launch(fn, job=None) -> None:
    """the forward kernel of a lookup whose prepare work is pending: enqueued by flush_batch() / the end of the block.
    job = (kind, ids, aux, table, B, F_or_T, K, out, out_stride, out_col, seq_len | None): a plain lookup that may share ONE launch
    with the block's other plain lookups (recalgo_lookup_multi_fwd); fn launches it on its own."""
    _batch["launches"].append((fn, job))


def flush_batch() -> None:
    """inside a batch_lookups() block: launch what is pending now (the block goes on collecting)"""
    global _batch
    if _batch is not None and (_batch["arenas"] or _batch["launches"]):
        b, _batch = _batch, {"arenas": {}, "launches": []}
        _flush_batch(b)


def _flush_batch(b) -> None:
    lib = _lib.load()
    for arena, plan, entries in b["arenas"].values():
        for i in range(0, len(entries), MAX_PREPARE_SOURCES):
            chunk = entries[i:i + MAX_PREPARE_SOURCES]
            flags = 0
            for e in chunk:
                flags |= e[3]
            d = next((e[4] for e in chunk if e[4] is not None), None)
            step = next((e[5] for e in chunk if e[5] is not None), None)
            ws = ctypes.c_void_p(plan.ws.data_ptr())
            stp = None if step is None else ctypes.c_void_p(step.data_ptr())
            if len(chunk) == 1:
                _, cs, first, _, _, _ = chunk[0]
                _lib.check(lib.recalgo_scatter_prepare(ctypes.byref(cs), arena.K, ws, plan.capacity, plan.nb_log2, first, flags,
                                                       None if d is None else ctypes.byref(d), None, arena.weight.shape[0], 0,
                                                       sweep_period(), stp, 0, _stream(arena.weight)), "recalgo_scatter_prepare")
            else:
                arr = (_CSource * len(chunk))(*[e[1] for e in chunk])
                firsts = (ctypes.c_int64 * len(chunk))(*[e[2] for e in chunk])
                _lib.check(lib.recalgo_scatter_prepare_multi(arr, len(chunk), firsts, arena.K, ws, plan.capacity, plan.nb_log2, flags,
                                                             None if d is None else ctypes.byref(d), arena.weight.shape[0],
                                                             sweep_period(), stp, 0, _stream(arena.weight)),
                           "recalgo_scatter_prepare_multi")
                prepare_stats["merged"] += len(chunk) - 1
        # what the workspace holds now: EVERY entry of the block (also one registered before a later lookup re-sized — and
        # cleared — the workspace, whose id the book-keeping dropped then) beside the sources counted by launches of their own
        have = set(plan.counted[2]) | {id(e[0]) for e in entries}
        plan.counted = plan.counted[:2] + (tuple(id(s) for s in plan.sources if id(s) in have),)
    plain = [(fn, job) for fn, job in b["launches"] if job is not None]
    for fn, job in b["launches"]:
        if job is None:
            fn()
    for i in range(0, len(plain), 4):
        chunk = plain[i:i + 4]
        if len(chunk) == 1:
            chunk[0][0]()
            continue
        ptr = lambda t: None if t is None else t.data_ptr()
        arr = (_CLookupJob * len(chunk))(*[_CLookupJob(j[0], ptr(j[1]), ptr(j[2]), ptr(j[3]), j[4], j[5], j[6], ptr(j[7]), j[8], j[9],
                                                        ptr(j[10])) for _, j in chunk])
        _lib.check(lib.recalgo_lookup_multi_fwd(arr, len(chunk), _stream(chunk[0][1][7])), "recalgo_lookup_multi_fwd")
        prepare_stats["merged_lookups"] += len(chunk) - 1


prepare_stats = {"merged": 0, "merged_lookups": 0}        # `prepare` / forward launches saved by batch_lookups (tests read it)


def companion_enabled() -> bool:
    return bool(COMPANION)


def _pair(main: Source, main_arena, arena) -> bool:
    """Try to make `arena` (one float per row) the companion of the lookup `main` of `main_arena`."""
    plan = plan_of(arena)
    ok = (main.n > 0 and companion_enabled() and _supported(arena) and arena.K == 1 and
          main_arena.K * 256 * 4 <= 32 * 1024 and getattr(arena, "trainable", True) and
          (plan is None or not plan.sources))
    if ok:
        others = {id(s.companion.arena) for s in plan_of(main_arena).sources if s.companion is not None}
        ok = not others or others == {id(arena)}
    if not ok:
        return False
    if plan is None:
        plan = arena.sparse = ArenaPlan(arena)
    cs = CompanionSource(main, arena)
    main.companion = cs
    plan.companions.append(cs)
    return True


def begin_lookup_pair(main_arena, arena, store, ids, row_base, n_ex: int, F: int, training=None):
    """The two lookups of DeepFM's sparse part: `main_arena` [rows, K] and `arena` [rows, 1] with the SAME ids / row_base.
    -> (Source of the main lookup or None, CompanionSource / Source / None of the second)."""
    main = begin_lookup(main_arena, store, ids, None, row_base, 0, n_ex, F, training, companion_arena=arena)
    if main is not None and main.companion is not None:
        return main, main.companion
    return main, begin_lookup(arena, store, ids, None, row_base, 0, n_ex, F, training)


def _dissolve(plan: ArenaPlan) -> None:
    """The companions registered on this arena become lookups of its own plan (counted again by the optimizer's launch)."""
    for cs in plan.companions:
        m = cs.main
        r = Source(m.ids, m.offsets, m.row_base, m.base, m.n_ex, m.F)
        r.caught_up = False
        if cs.g is not None:
            r.g, r.g_stride, r.g_col, r.g_fmul = cs.g, cs.g_stride, cs.g_col, cs.g_fmul
        cs.regular = r
        m.companion = None
        plan.sources.append(r)
    plan.companions = []
    plan.counted = None


def new_forward(store) -> None:
    """A model_fn invocation starts: sources of an earlier forward that never reached the optimizer are stale."""
    for ar in store.arenas.values():
        plan = plan_of(ar)
        if plan is not None and plan.companions:
            plan.companions = []
            plan.grad_materialized = False
        if plan is not None and plan.sources:
            plan.sources = []
            plan.grad_materialized = False
            if plan.ws is not None and plan.counted is not None and plan.counted[2]:
                plan.clear_counts()            # (the abandoned forward's entries: normally consumed and cleared by `apply`)


def has_work(arena) -> bool:
    plan = plan_of(arena)
    return plan is not None and (bool(plan.sources) or bool(plan.companions) or plan.last_step is not None)


def has_companions(arena) -> bool:
    """The arena's lookups of this step ride on another arena's plan: that arena's optimizer call serves it (the
    Estimator applies arenas with companions LAST)."""
    plan = plan_of(arena)
    return plan is not None and bool(plan.companions)


def _merge_dense(sources: List[Source], K: int) -> List[Source]:
    """More lookups into one arena than recalgo_scatter_apply takes sources (a model whose columns are looked up one by
    one): the id-matrix sources are folded into ONE source of explicit arena rows with their gradient rows concatenated
    (two torch.cat per step — the fused id-matrix path of the benchmark models never gets here)."""
    dense = [s for s in sources if s.offsets is None and s.n]
    rest = [s for s in sources if s.offsets is not None and s.n]
    if len(rest) + 1 > MAX_SOURCES:
        raise NotImplementedError(f"more than {MAX_SOURCES - 1} ragged lookups into one arena in one step")
    rows, grads = [], []
    for s in dense:
        ids = s.ids.reshape(s.n_ex, s.F)
        r = ids + s.base
        if s.row_base is not None:
            r = r + s.row_base.reshape(1, -1)
        rows.append(torch.where(ids >= 0, r, torch.full_like(r, -1)).reshape(-1))
        fm = K if s.g_fmul is None else int(s.g_fmul)
        if fm == 0:
            g = s.g[:, :K].unsqueeze(1).expand(s.n_ex, s.F, K)
        else:
            g = torch.as_strided(s.g, (s.n_ex, s.F, K), (s.g.stride(0) if s.n_ex > 1 else s.F * fm, fm, 1))
        if s.fm is not None:
            # the FM second-order epilogue the kernels form on load (Source.set_grad): g + scale * (field_sum - emb)
            sc, fs, em = s.fm
            g = torch.addcmul(g, sc.reshape(-1, 1, 1), fs.reshape(s.n_ex, 1, K) - em.reshape(s.n_ex, s.F, K))
        grads.append(g.reshape(-1, K))
    allrows = torch.cat(rows).reshape(-1, 1).contiguous()
    merged = Source(allrows, None, None, 0, allrows.shape[0], 1)
    merged.set_grad(torch.cat(grads).contiguous())
    return rest + [merged]


MODE_PRESCANNED = 0x100        # include/recalgo.h RECALGO_SCATTER_PRESCANNED


def plan_scan_record(arena, lazy: bool):
    """The recalgo_plan_scan_t of the arena's plan, for the optimizer launch that runs between the step's last count and this
    arena's `apply` (ops.adam_tf1_step_(plan_scans=)) — or None when `apply` will not find the totals of exactly its sources in
    the workspace (it then counts again, and `place` scans them itself)."""
    plan = plan_of(arena)
    if plan is None or plan.ws is None or plan.companions or (plan.served and not plan.sources):
        return None
    srcs = [s for s in plan.sources if s.g is not None and s.n]
    if not srcs or len(srcs) > MAX_SOURCES or sum(s.slots for s in srcs) > plan.capacity or plan.counted != plan._signature(srcs):
        return None
    rec = _CPlanScan()
    _lib.check(_lib.load().recalgo_scatter_plan_scan(ctypes.c_void_p(plan.ws.data_ptr()), plan.capacity, plan.nb_log2,
                                                     ctypes.byref(rec)), "recalgo_scatter_plan_scan")
    plan.prescanned = plan._signature(srcs)
    return rec


def _run(plan: ArenaPlan, sources: List[Source], mode: int, step_dev, step_offset: int, lr: float, live=None):
    lib = _lib.load()
    prescanned, plan.prescanned = plan.prescanned, None
    a = plan.arena
    comp_arena = _companion_arena(sources, mode)
    if len([s for s in sources if s.n]) > MAX_SOURCES:
        if comp_arena is not None:
            _dissolve(plan_of(comp_arena))
            comp_arena = None
        sources = _merge_dense(sources, a.K)
    srcs = [s for s in sources if s.n]
    plan._ensure_ws(sum(s.slots for s in srcs))
    if not srcs or prescanned != plan._signature(srcs) or plan.counted != prescanned:
        prescanned = None
    if plan.counted != plan._signature(srcs):
        # the totals in the workspace are not those of exactly these sources (first step, a forward without a backward, a
        # GRAD pass before the optimizer, a re-sized workspace): count again
        if plan.counted is None or plan.counted[2] or plan.counted[:2] != (plan.ws.data_ptr(), plan.nb_log2):
            plan.clear_counts()
        first = 0
        for s in srcs:
            cs = s.c_struct(a.K)
            _lib.check(lib.recalgo_scatter_prepare(ctypes.byref(cs), a.K, ctypes.c_void_p(plan.ws.data_ptr()), plan.capacity,
                                                   plan.nb_log2, first, PREPARE_COUNT, None, None, 0, 0, 1, None, 0, _stream(a.weight)),
                       "recalgo_scatter_prepare")
            first += s.slots
    if not srcs:                               # (the sweep and the lr ring still need the launch)
        dummy = Source(a.weight, None, None, 0, 0, 1)
        dummy.g, dummy.g_fmul = a.weight, a.K
        srcs = [dummy]
    arr = (_CSource * len(srcs))(*[s.c_struct(a.K) for s in srcs])
    d = plan._deferred_struct() if mode == MODE_ADAM else None
    if d is not None and not plan.swept:
        # no lookup's launch carried this step's share of the sweep (a step without a TRAIN lookup into the arena, or the
        # arena's first deferred step): its own launch, up to the step before this one
        dc0 = None
        if comp_arena is not None and plan_of(comp_arena).last_step is not None and not plan_of(comp_arena).swept:
            dc0 = plan_of(comp_arena)._deferred_struct()
        _lib.check(lib.recalgo_scatter_prepare(None, a.K, ctypes.c_void_p(plan.ws.data_ptr()), plan.capacity, plan.nb_log2, 0,
                                               PREPARE_SWEEP, ctypes.byref(d), None if dc0 is None else ctypes.byref(dc0),
                                               a.weight.shape[0], comp_arena.weight.shape[0] if dc0 is not None else 0, sweep_period(),
                                               ctypes.c_void_p(step_dev.data_ptr()), step_offset - 1, _stream(a.weight)),
                   "recalgo_scatter_prepare (sweep)")
        if dc0 is not None:
            plan_of(comp_arena).swept = True
    if mode == MODE_ADAM and comp_arena is not None:
        # the companion's share of the sweep rides on the launch that carries the MAIN arena's (begin_lookup, or the launch
        # above).  When the main arena was swept by an earlier plain lookup of the step, the paired lookup's launch carries no
        # sweep at all: the companion then gets a launch of its own — unswept, its untouched rows would lag past the lr ring
        cpl = plan_of(comp_arena)
        if cpl.last_step is not None and not cpl.swept:
            dc1 = cpl._deferred_struct()
            _lib.check(lib.recalgo_scatter_prepare(None, 1, ctypes.c_void_p(plan.ws.data_ptr()), plan.capacity, plan.nb_log2, 0,
                                                   PREPARE_SWEEP, ctypes.byref(dc1), None, comp_arena.weight.shape[0], 0,
                                                   sweep_period(), ctypes.c_void_p(step_dev.data_ptr()), step_offset - 1,
                                                   _stream(a.weight)), "recalgo_scatter_prepare (companion sweep)")
            cpl.swept = True
    if mode != MODE_GRAD:
        plan.swept = False                     # (the next step's first lookup sweeps again)
    b1, b2, eps = plan.betas
    p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    comp = None
    if comp_arena is not None:
        # the second arena: same placed entries, its own rows / optimizer state.  Sources without a companion (or whose
        # companion never got a gradient) contribute nothing to it
        c, cp = comp_arena, plan_of(comp_arena)
        cp.betas = plan.betas
        if mode == MODE_ADAM and cp.last_step is None:
            _init_deferred(cp, step_dev)
        if mode == MODE_LAZY_ADAM and cp.last_step is not None:
            sync(c, step_dev, -1)
            cp.last_step = None
        if mode == MODE_GRAD:
            cp.grad_materialized = True
        cstructs = [s.companion.c_struct() if (s.companion is not None and s.companion.g is not None) else
                    _CSource(s.ids.data_ptr(), None if s.offsets is None else s.offsets.data_ptr(),
                             None if s.row_base is None else s.row_base.data_ptr(), s.base, s.n_ex, s.F, None, 0, 0, 0)
                    for s in srcs]
        carr = (_CSource * len(srcs))(*cstructs)
        dc = cp._deferred_struct() if mode == MODE_ADAM else None
        cgrad = c._grad if (mode == MODE_GRAD or cp.grad_materialized) else None
        ptr = lambda t: None if t is None else t.data_ptr()
        comp = _CCompanion(ctypes.addressof(carr), ptr(c.weight), ptr(c.m), ptr(c.v), ptr(cgrad),
                           None if dc is None else ctypes.addressof(dc), c.weight.shape[0])
    grad = a._grad if (mode == MODE_GRAD or plan.grad_materialized) else None
    _lib.check(lib.recalgo_scatter_apply(arr, len(srcs), None if comp is None else ctypes.byref(comp), a.K, p(plan.ws), plan.capacity,
                                         plan.nb_log2, mode | (MODE_PRESCANNED if prescanned is not None else 0), p(a.weight), p(a.m),
                                         p(a.v), p(grad),
                                         None if d is None else ctypes.byref(d), a.weight.shape[0],
                                         live, p(step_dev), step_offset, lr, b1, b2, eps, _stream(a.weight)),
               "recalgo_scatter_apply")
    if comp_arena is not None and mode != MODE_GRAD:
        cp.companions = []
        cp.grad_materialized = False
        cp.served = True
        cp.swept = False
    plan.counted = plan._signature([])         # (`apply` left the totals clean: the next step's `prepare` launches add to zero)


def _companion_arena(sources: List[Source], mode: int):
    """The second arena served by this plan's launches, or None.  LazyAdam must touch exactly the rows of the second arena's
    own lookups: a plan in which only SOME lookups carry a companion gives it up (the arenas then run separately)."""
    comps = [s.companion for s in sources if s.n and s.companion is not None]
    if not comps:
        return None
    arena = comps[0].arena
    if mode == MODE_LAZY_ADAM and len(comps) != len([s for s in sources if s.n]):
        _dissolve(plan_of(arena))
        return None
    return arena


def _init_deferred(plan: ArenaPlan, step_dev: torch.Tensor) -> None:
    # rows with state (m, v) are valid for the step before this one; untouched rows are marked 0
    arena = plan.arena
    rows = arena.weight.shape[0]
    plan.last_step = torch.zeros(rows, dtype=torch.int32, device=arena.weight.device)
    prev = (step_dev - 1).to(torch.int32)
    chunk = 1 << 22                            # (row chunks: the masks of a 100 M-row table are multi-GB transients otherwise)
    any_alive = torch.zeros((), dtype=torch.bool, device=arena.weight.device)
    for r0 in range(0, rows, chunk):
        r1 = min(rows, r0 + chunk)
        alive = (arena.m[r0:r1] != 0).any(dim=1) | (arena.v[r0:r1] != 0).any(dim=1)
        plan.last_step[r0:r1] = torch.where(alive, prev, torch.zeros_like(prev))
        any_alive |= alive.any()
    capturing = arena.weight.is_cuda and torch.cuda.is_current_stream_capturing()
    if not capturing and bool(any_alive) and int(step_dev) <= 1:
        # 0 doubles as "never touched": moments without the step they belong to (restored without opt_step, or set by hand)
        # cannot be given their pending g = 0 decay — refuse instead of silently deviating from the dense pass
        raise RuntimeError(f"arena {getattr(arena, 'name', '?')}: Adam moments are non-zero but the optimizer step is {int(step_dev)}: "
                           "restore the step counter together with the moments (estimator.restore_checkpoint_state does)")
    plan.lr_ring = torch.zeros(LR_RING, dtype=torch.float32, device=arena.weight.device)


def apply(arena, lazy: bool, step_dev: torch.Tensor, lr: float, beta1: float, beta2: float, eps: float) -> None:
    """The optimizer step of one arena (step_dev already advanced to this step)."""
    plan = plan_of(arena)
    plan.step_dev = step_dev
    if plan.companions:
        raise RuntimeError(f"arena {getattr(arena, 'name', '?')}: its lookups ride on another arena's plan, whose optimizer call "
                           "has not run yet (apply arenas with sparse.has_companions() last)")
    if plan.served and not plan.sources:       # (this step's update of the arena ran with the main arena's launches)
        plan.served = False
        return
    plan.served = False
    sources = [s for s in plan.sources if s.g is not None]
    plan.betas = (float(beta1), float(beta2), float(eps))
    if lazy:
        if plan.last_step is not None:
            sync(arena, step_dev, -1)          # (switching optimizers mid-run: finish the deferred updates first)
            plan.last_step = None
        if sources:
            _run(plan, sources, MODE_LAZY_ADAM, step_dev, 0, lr)
    else:
        if plan.last_step is None:
            _init_deferred(plan, step_dev)
        _run(plan, sources, MODE_ADAM, step_dev, 0, lr)
    for s in plan.sources:                     # companions whose main lookup never got a gradient: nothing to apply
        if s.companion is not None:
            cp = plan_of(s.companion.arena)
            cp.companions = [c for c in cp.companions if c is not s.companion]
    plan.sources = []
    plan.grad_materialized = False


def materialize_arena(arena) -> None:
    """arena.grad += the summed row gradients of the pending sources (what reading `arena.grad` triggers; the optimizer
    does not need it)."""
    plan = plan_of(arena)
    if plan is None or plan.grad_materialized:
        return
    if plan.companions:                        # the second arena of a pairing: the main arena's launch fills both
        for ma in {id(c.main.arena): c.main.arena for c in plan.companions}.values():
            materialize_arena(ma)
        return
    sources = [s for s in plan.sources if s.g is not None]
    if not sources:
        return
    plan.grad_materialized = True              # (set first: _run reads the raw arena._grad, never the property)
    _run(plan, sources, MODE_GRAD, None, 0, 0.0)


def materialize_grads(store) -> None:
    for ar in store.arenas.values():
        materialize_arena(ar)


def sync(arena, step_dev: Optional[torch.Tensor], step_offset: int = 0) -> None:
    """Deferred Adam: bring EVERY row of the arena to step_dev[0] + step_offset (no-op without deferred state)."""
    plan = plan_of(arena)
    if plan is None or plan.last_step is None or step_dev is None:
        return
    d = plan._deferred_struct()
    _lib.check(_lib.load().recalgo_adam_deferred_sweep(ctypes.byref(d), arena.K, 0, arena.weight.shape[0],
                                                      ctypes.c_void_p(step_dev.data_ptr()), step_offset, _stream(arena.weight)),
               "recalgo_adam_deferred_sweep")


def sync_arena(arena) -> None:
    """sync() with the step counter the arena's last optimizer call used (callers that have no store at hand)."""
    plan = plan_of(arena)
    if plan is not None and plan.last_step is not None and plan.step_dev is not None:
        sync(arena, plan.step_dev, 0)


def sync_store(store) -> None:
    """Every arena of the store reflects all completed optimizer steps (named_arrays, checkpoints, export)."""
    st = getattr(store, "opt_state", None)
    if st is None:
        return
    for ar in store.arenas.values():
        sync(ar, st["step"], 0)


def reset(arena) -> None:
    """Forget the deferred state (the arena's w / m / v were replaced: restore, load_variables, re-sharding)."""
    plan = plan_of(arena)
    if plan is not None:
        plan.last_step = None
        plan.sources = []
        plan.counted = None
        plan.swept = False
        plan.grad_materialized = False
