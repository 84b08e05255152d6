"""Torch-tensor level wrappers around the C-ABI (include/recalgo.h).

PyTorch is plumbing here: device memory, the current HIP stream, and autograd to chain the
activation gradients between the hand-written kernels.  All arithmetic of the hot path runs in
librecalgo_hip.so; there is no CPU or eager fallback — tensors must live on a HIP device.

Parameter gradients are written by the kernels directly into `Variable.grad` /
`EmbeddingArena.grad` (see variables.py); the autograd Functions return None for them.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch
from torch.autograd import Function

from . import _lib, sparse
from .variables import EmbeddingArena, Variable, VariableStore

_ACT = {"prelu": 0, "dice": 1}


def _lib_():
    return _lib.load()


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t: torch.Tensor):
    if not t.is_cuda:
        raise _lib.RecalgoError(
            "recalgo ops run only on a HIP device (no CPU fallback); got a CPU tensor")
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def copy_bytes(dst: torch.Tensor, src: torch.Tensor) -> None:
    """dst <- src for two contiguous device tensors of the same byte length on one device (recalgo_copy_bytes)."""
    n = dst.numel() * dst.element_size()
    if n != src.numel() * src.element_size() or not dst.is_contiguous() or not src.is_contiguous() or dst.device != src.device:
        raise ValueError("copy_bytes: contiguous tensors of equal byte length on one device")
    _lib.check(_lib_().recalgo_copy_bytes(_p(dst), _p(src), n, _stream(dst)), "recalgo_copy_bytes")


def _chk(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")


_ws_cache = {}


def arena_row_base(arena, table_name: str, device) -> torch.Tensor:
    """int64 [1] device tensor holding the first arena row of `table_name`.  Cached ON the arena object: a
    module-level cache keyed by id(arena) hands a dead arena's row bases to a new arena that re-uses its address."""
    cache = arena.__dict__.setdefault("_row_base_1", {})
    t = cache.get(table_name)
    if t is None or t.device != torch.device(device):
        t = cache[table_name] = torch.tensor([arena.tables[table_name][0]], dtype=torch.int64, device=device)
    return t


def anchor_store(anchor):
    """The VariableStore whose autograd anchor was passed to a lookup (its optimizer state holds the step counter the
    deferred-Adam catch-up needs)."""
    return getattr(anchor, "_recalgo_store", None)


def _flush(arena) -> None:
    """Row-sharded deployment (parallel.py): the kernel scattered into a staged gradient; send it
    to the rows' owners now."""
    f = getattr(arena, "flush_grad", None)
    if f is not None:
        f()


def mark_live_rows(arena, ids: torch.Tensor, row_base: Optional[torch.Tensor], F: int) -> None:
    """Record the rows a scatter has just touched in the arena's live-row list (first touch only)."""
    if not getattr(arena, "tracks_live_rows", False):
        return
    live, lst, cnt = arena.live_state()
    _lib.check(_lib_().recalgo_mark_live_rows(_p(ids), _p(row_base), ids.numel(), int(F), _p(live), _p(lst), _p(cnt),
                                              _stream(ids)), "recalgo_mark_live_rows")


class _Live(ctypes.Structure):          # include/recalgo.h recalgo_live_t
    _fields_ = [("row_live", ctypes.c_void_p), ("live_list", ctypes.c_void_p), ("live_count", ctypes.c_void_p),
                ("row_offset", ctypes.c_int64)]


def _live(arena, row_offset: int = 0):
    """recalgo_live_t* of the arena's live-row bookkeeping (the scatter kernels mark the rows they flush), or None
    for arenas without it (row-sharded staging buffers: their owners mark on receipt)."""
    if not getattr(arena, "tracks_live_rows", False):
        return None
    live, lst, cnt = arena.live_state()
    return ctypes.byref(_Live(live.data_ptr(), lst.data_ptr(), cnt.data_ptr(), int(row_offset)))


def scatter_rows_sorted(arena, rows: torch.Tensor, vals: torch.Tensor) -> None:
    """arena.grad[rows[i], :] += vals[i, :] (rows < 0 skipped), deterministically (stable sort + ordered segment sums); the
    rows join the live list.  Only for an arena that is NOT on the owner-computes path while its partner in a fused lookup is
    (DeepFM's two arenas, one of them frozen or of an unsupported width)."""
    rows = rows.reshape(-1).contiguous()
    vals = vals.reshape(rows.numel(), -1).contiguous()
    srt, perm = torch.sort(rows, stable=True)
    _lib.check(_lib_().recalgo_scatter_rows_sorted(_p(srt), _p(perm), _p(vals), rows.numel(), vals.shape[1], _p(arena.grad),
                                                   _stream(vals)), "recalgo_scatter_rows_sorted")
    mark_live_rows(arena, rows, None, 1)


def _staged(arena, rows: torch.Tensor, store=None):
    """(plan, staged arena, identity ids [M]) for a row-sharded arena, or None when the arena is local.  `store`: the
    owner side of the exchange joins the shard's owner-computes plan (sparse.py) when the call is a TRAIN forward."""
    sd = getattr(arena, "sharding", None)
    if sd is None:
        return None
    from . import parallel
    plan = sd.plan(rows)
    return plan, parallel.StagedArena(plan, arena, store, torch.is_grad_enabled()), plan.staged_ids(rows, rows.shape)


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Per-device grow-only scratch (allocated outside graph capture on first use)."""
    key = (device.type, device.index)
    w = _ws_cache.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = w
    return w


# =============================================================================================
# K1: embedding gather
# =============================================================================================
# A gather whose output the cross network consumes next (DCN, dcn.py:152-160) is not launched: CrossNet's forward kernel
# fetches the rows itself and writes x0 on the way (recalgo_gather_cross_fwd).  `gather_feeds_cross` is the model's promise;
# anything that is still pending when another consumer could read the tensor is launched by flush_lazy_gathers().
LAZY_GATHER = True                           # module hook (tests / A-B): False = every gather is its own launch
_lazy_gather_on = False
_lazy_hit = False
_lazy_gathers: list = []


class gather_feeds_cross:
    """with ops.gather_feeds_cross() as lz: x = fc.input_layer(...); lz.keep(x) — the single-launch gathers issued inside stay
    pending only if exactly one was issued and `x` IS its output (one width, no bags, no concat); else they run now."""

    def __enter__(self):
        global _lazy_gather_on
        flush_lazy_gathers()                 # (nothing is pending here unless an earlier model_fn call ended in an exception)
        self._prev, _lazy_gather_on = _lazy_gather_on, bool(LAZY_GATHER)
        self._kept = False
        return self

    def keep(self, x) -> None:
        self._kept = (len(_lazy_gathers) == 1 and isinstance(x, torch.Tensor) and _lazy_gathers[0][0] is x)

    def __exit__(self, *exc):
        global _lazy_gather_on
        _lazy_gather_on = self._prev
        if not self._kept:
            flush_lazy_gathers()
        return False


def flush_lazy_gathers() -> None:
    while _lazy_gathers:
        out, ids, arena, row_base = _lazy_gathers.pop()
        out._recalgo_lazy_gather = None
        B, F = ids.shape
        _lib.check(_lib_().recalgo_embedding_gather_fwd(_p(ids), _p(arena.weight), _p(row_base), B, F, arena.K, _p(out), F * arena.K, 0,
                                                        _stream(ids)), "recalgo_embedding_gather_fwd")


def _take_lazy_gather(x0: torch.Tensor):
    """The pending gather whose output is x0 (-> (ids, arena, row_base)), or None; any OTHER pending gather is launched."""
    lz = getattr(x0, "_recalgo_lazy_gather", None)
    if lz is not None:
        for i, ent in enumerate(_lazy_gathers):
            if ent[0] is x0:
                _lazy_gathers.pop(i)
                break
        x0._recalgo_lazy_gather = None
    flush_lazy_gathers()
    return lz


class _GatherFn(Function):
    @staticmethod
    def forward(ctx, anchor, ids, arena: EmbeddingArena, row_base, training=False, lazy_ok=False):
        B, F = ids.shape
        K = arena.K
        out = torch.empty(B, F * K, device=ids.device, dtype=torch.float32)
        # owner-computes scatter (sparse.py): the lookup joins the arena's plan; deferred-Adam rows are caught up first
        ctx.src = sparse.begin_lookup(arena, anchor_store(anchor), ids, None, row_base, 0, B, F, training, can_defer=True)
        # deferred Adam: a registered (TRAIN) lookup's rows were just caught up; any other lookup reads lagging rows as of now
        dv, stp = sparse.view_for(ctx.src, arena, anchor_store(anchor))
        ctx.arena, ctx.ids, ctx.row_base = arena, ids, row_base
        # (lazy_ok: only the plain lookup of embedding_gather() may be left to its consumer — not the staged rows of a sharded arena)
        batched = ctx.src is not None and getattr(ctx.src, "deferred", False)       # (sparse.batch_lookups: prepare work pending)
        ctx.lazy = (lazy_ok and _lazy_gather_on and not batched and dv is None and K % 4 == 0 and F * K <= 1024
                    and ids.is_contiguous())
        if ctx.lazy:
            global _lazy_hit
            _lazy_hit = True                 # (embedding_gather() hangs the pending launch on the tensor autograd hands out)
            return out

        def launch():
            _lib.check(_lib_().recalgo_embedding_gather_fwd_deferred(
                _p(ids), _p(arena.weight), _p(row_base), B, F, K, _p(out), F * K, 0, dv, stp, 0, _stream(ids)),
                "recalgo_embedding_gather_fwd")
        if batched:
            # behind the block's one `prepare` launch (the rows it catches up are read here); a plain lookup: it may share its
            # launch with the block's other plain lookups
            job = (0, ids, row_base, arena.weight, B, F, K, out, F * K, 0, None) if (dv is None and ids.is_contiguous()) else None
            sparse.defer_launch(launch, job)
        else:
            launch()
        return out

    @staticmethod
    def backward(ctx, g):
        ids, arena = ctx.ids, ctx.arena
        B, F = ids.shape
        if ctx.src is not None:
            ctx.src.set_grad(g)              # summed per row (and applied) by the optimizer's recalgo_scatter_apply
            return None, None, None, None, None, None
        g = g.contiguous()
        _lib.check(_lib_().recalgo_embedding_gather_bwd(
            _p(ids), _p(g), _p(ctx.row_base), B, F, arena.K, F * arena.K, 0, _p(arena.grad), _live(arena),
            _stream(ids)), "recalgo_embedding_gather_bwd")
        _flush(arena)
        return None, None, None, None, None, None


def embedding_gather(store: VariableStore, ids: torch.Tensor, arena: EmbeddingArena,
                     row_base: torch.Tensor) -> torch.Tensor:
    """ids [B,F] int64 (id<0 = OOV) -> [B, F*K]; gradient scattered into arena.grad."""
    _chk(ids, torch.int64, "ids")
    _chk(row_base, torch.int64, "row_base")
    if getattr(arena, "sharding", None) is not None:
        from . import parallel
        _, staged, ident = _staged(arena, parallel.global_rows(ids, row_base), store)
        return _GatherFn.apply(store.anchor, ident.reshape(ids.shape), staged, torch.zeros_like(row_base), False)
    global _lazy_hit, _lazy_gather_on
    _lazy_hit = False
    if _lazy_gather_on and _lazy_gathers:
        # a SECOND gather inside gather_feeds_cross(): the outputs are about to be combined by somebody else (input_layer's
        # per-column path + torch.cat), so neither can be x0 of the cross kernel — launch the pending one, go eager from here on
        flush_lazy_gathers()
        _lazy_gather_on = False
    out = _GatherFn.apply(store.anchor, ids, arena, row_base, torch.is_grad_enabled(), True)
    if _lazy_hit:
        _lazy_hit = False
        out._recalgo_lazy_gather = (ids, arena, row_base)
        _lazy_gathers.append((out, ids, arena, row_base))
    return out


class _BagMeanFn(Function):
    @staticmethod
    def forward(ctx, anchor, values, offsets, arena: EmbeddingArena, table_name, training=False):
        B = offsets.numel() - 1
        K = arena.K
        table = arena.table_view(table_name)
        out = torch.empty(B, K, device=values.device, dtype=torch.float32)
        ctx.src = None
        if table_name != "__staged__" and values.numel():
            # one request per bag entry; its gradient row (g[bag] / count) is expanded in the backward
            ctx.src = sparse.begin_lookup(arena, anchor_store(anchor), values, None, None, arena.tables[table_name][0],
                                          values.numel(), 1, training)
        dv, stp = (None, None) if table_name == "__staged__" else sparse.view_for(ctx.src, arena, anchor_store(anchor))
        rb0 = 0 if table_name == "__staged__" else arena.tables[table_name][0]
        _lib.check(_lib_().recalgo_embedding_bag_mean_fwd_deferred(
            _p(values), _p(offsets), _p(table), B, K, _p(out), K, 0, dv, rb0, stp, 0, _stream(offsets)),
            "recalgo_embedding_bag_mean_fwd")
        ctx.args = (values, offsets, arena, table_name)
        return out

    @staticmethod
    def backward(ctx, g):
        values, offsets, arena, table_name = ctx.args
        B = offsets.numel() - 1
        rb, vocab = arena.tables[table_name]
        g = g.contiguous()
        if ctx.src is not None:
            lens = offsets[1:] - offsets[:-1]
            bag = torch.repeat_interleave(torch.arange(B, device=values.device), lens)
            cnt = torch.zeros(B, device=values.device).index_add_(0, bag, (values >= 0).float()).clamp_(min=1.0)
            ctx.src.set_grad(g[bag] / cnt[bag].unsqueeze(1))
            return None, None, None, None, None, None
        gt = arena.grad[rb:rb + vocab]
        _lib.check(_lib_().recalgo_embedding_bag_mean_bwd(
            _p(values), _p(offsets), _p(g), B, arena.K, arena.K, 0, _p(gt), _live(arena, rb), _stream(offsets)),
            "recalgo_embedding_bag_mean_bwd")
        _flush(arena)
        return None, None, None, None, None, None


def embedding_bag_mean(store, values, offsets, arena, table_name) -> torch.Tensor:
    _chk(values, torch.int64, "values")
    _chk(offsets, torch.int64, "offsets")
    if getattr(arena, "sharding", None) is not None:
        rows = torch.where(values >= 0, values + arena.tables[table_name][0], torch.full_like(values, -1))
        _, staged, ident = _staged(arena, rows, store)
        return _BagMeanFn.apply(store.anchor, ident, offsets, staged, "__staged__", False)
    return _BagMeanFn.apply(store.anchor, values, offsets, arena, table_name, torch.is_grad_enabled())


class _SeqGatherFn(Function):
    @staticmethod
    def forward(ctx, anchor, values, offsets, arena: EmbeddingArena, table_name, T, training=False):
        B = offsets.numel() - 1
        K = arena.K
        table = arena.table_view(table_name)
        out = torch.empty(B, T, K, device=offsets.device, dtype=torch.float32)
        seq_len = torch.empty(B, device=offsets.device, dtype=torch.int32)
        ctx.src = None
        if table_name != "__staged__":
            ctx.src = sparse.begin_lookup(arena, anchor_store(anchor), values, offsets, None, arena.tables[table_name][0], B, T, training,
                                          can_defer=True)
        dv, stp = (None, None) if table_name == "__staged__" else sparse.view_for(ctx.src, arena, anchor_store(anchor))
        rb0 = 0 if table_name == "__staged__" else arena.tables[table_name][0]

        def launch():
            _lib.check(_lib_().recalgo_sequence_gather_fwd_deferred(
                _p(values), _p(offsets), _p(table), B, T, K, _p(out), _p(seq_len), dv, rb0, stp, 0, _stream(offsets)),
                "recalgo_sequence_gather_fwd")
        if ctx.src is not None and getattr(ctx.src, "deferred", False):
            # (sparse.batch_lookups: behind the block's one `prepare` launch, possibly sharing ONE forward launch)
            job = (1, values, offsets, table, B, T, K, out, 0, 0, seq_len) if (dv is None and values.is_contiguous()) else None
            sparse.defer_launch(launch, job)
        else:
            launch()
        ctx.args = (values, offsets, arena, table_name, T)
        ctx.mark_non_differentiable(seq_len)
        ctx.set_materialize_grads(False)      # (else autograd fills a zero "gradient" for seq_len: one launch per lookup and step)
        return out, seq_len

    @staticmethod
    def backward(ctx, g, _gl):
        values, offsets, arena, table_name, T = ctx.args
        B = offsets.numel() - 1
        rb, vocab = arena.tables[table_name]
        if g is None:                          # (the sequence output was not used)
            return None, None, None, None, None, None, None
        if ctx.src is not None:
            ctx.src.set_grad(g)              # (no `arena.grad` access here: reading it would sum the pending sources now)
            return None, None, None, None, None, None, None
        g = g.contiguous()
        gt = arena.grad[rb:rb + vocab]
        _lib.check(_lib_().recalgo_sequence_gather_bwd(
            _p(values), _p(offsets), _p(g), B, T, arena.K, _p(gt), _live(arena, rb), _stream(offsets)),
            "recalgo_sequence_gather_bwd")
        _flush(arena)
        return None, None, None, None, None, None, None


def sequence_gather(store, values, offsets, arena, table_name, T) -> Tuple[torch.Tensor, torch.Tensor]:
    _chk(values, torch.int64, "values")
    _chk(offsets, torch.int64, "offsets")
    if getattr(arena, "sharding", None) is not None:
        rows = torch.where(values >= 0, values + arena.tables[table_name][0], torch.full_like(values, -1))
        _, staged, ident = _staged(arena, rows, store)
        return _SeqGatherFn.apply(store.anchor, ident, offsets, staged, "__staged__", int(T), False)
    return _SeqGatherFn.apply(store.anchor, values, offsets, arena, table_name, int(T), torch.is_grad_enabled())


# =============================================================================================
# K1+K2+K3: DeepFM sparse path
# =============================================================================================
class _DeepFMSparseFn(Function):
    @staticmethod
    def forward(ctx, anchor, ids, arena: EmbeddingArena, w1: EmbeddingArena, bias: Variable, row_base, training=False):
        B, F = ids.shape
        K = arena.K
        emb = torch.empty(B, F * K, device=ids.device, dtype=torch.float32)
        fm1 = torch.empty(B, 1, device=ids.device, dtype=torch.float32)
        fm2 = torch.empty(B, 1, device=ids.device, dtype=torch.float32)
        fsum = torch.empty(B, K, device=ids.device, dtype=torch.float32)
        st = anchor_store(anchor)
        # the first-order arena is looked up with the same requests: it rides on the embedding arena's plan
        ctx.src, ctx.src1 = sparse.begin_lookup_pair(arena, w1, st, ids, row_base, B, F, training)
        dv, stp = sparse.view_for(ctx.src, arena, st)
        dv1, stp1 = sparse.view_for(ctx.src1, w1, st)
        _lib.check(_lib_().recalgo_deepfm_sparse_fwd_deferred(
            _p(ids), _p(arena.weight), _p(w1.weight), _p(bias.data), _p(row_base), B, F, K,
            _p(emb), _p(fm1), _p(fm2), _p(fsum), dv, dv1, stp if stp is not None else stp1, 0, _stream(ids)),
            "recalgo_deepfm_sparse_fwd")
        ctx.args = (ids, arena, w1, bias, row_base)
        ctx.save_for_backward(emb, fsum)
        ctx.set_materialize_grads(False)      # (FwFM never uses fm2: no zero "gradient" is filled for it, one launch less)
        return emb, fm1, fm2

    @staticmethod
    def backward(ctx, g_emb, g_fm1, g_fm2):
        ids, arena, w1, bias, row_base = ctx.args
        emb, fsum = ctx.saved_tensors
        B, F = ids.shape
        no_fm2 = g_fm2 is None
        g_emb = emb.new_zeros(B, F * arena.K) if g_emb is None else g_emb.contiguous()
        g_fm1 = emb.new_zeros(B, 1) if g_fm1 is None else g_fm1.contiguous()
        g_fm2 = (emb.new_zeros(B, 1) if ctx.src is None else None) if no_fm2 else g_fm2.contiguous()
        if ctx.src is not None or ctx.src1 is not None:
            # owner-computes path (sparse.py).  The second-order term's gradient g_emb + g_fm2 * (S - e) (Appendix D "FM2") is
            # the EPILOGUE of the embedding lookup's source: `place` / `apply` form it on load — no [B, F, K] tensor, no
            # elementwise launches.  An arena that is NOT on the owner path (frozen, or an unsupported width) while the
            # other one is keeps the deterministic sorted scatter for itself.
            K = arena.K
            rows = None
            if ctx.src is not None:
                ctx.src.set_grad(g_emb, fm=None if no_fm2 else (g_fm2.reshape(B), fsum, emb))
            elif getattr(arena, "trainable", True):
                rows = torch.where(ids >= 0, ids + row_base.unsqueeze(0), torch.full_like(ids, -1))
                e3, s3 = emb.reshape(B, F, K), fsum.reshape(B, 1, K)
                scatter_rows_sorted(arena, rows, torch.addcmul(g_emb.reshape(B, F, K), g_fm2.reshape(B, 1, 1), s3 - e3))
                _flush(arena)
            if ctx.src1 is not None:
                ctx.src1.set_grad(g_fm1.reshape(B, 1), fmul=0)      # every field of example b adds g_fm1[b] to its w1 row
            elif getattr(w1, "trainable", True):
                if rows is None:
                    rows = torch.where(ids >= 0, ids + row_base.unsqueeze(0), torch.full_like(ids, -1))
                scatter_rows_sorted(w1, rows, g_fm1.reshape(B, 1, 1).expand(B, F, 1))
                _flush(w1)
            if not colsum_of_dlogit(g_fm1, bias.grad.view(1)):      # (TRAIN step: a job of the step's deferred-sum launch)
                torch.sum(g_fm1, dim=0, out=bias.grad.view(1))
            return None, None, None, None, None, None, None
        _lib.check(_lib_().recalgo_deepfm_sparse_bwd(
            _p(ids), _p(emb), _p(fsum), _p(g_emb), _p(g_fm1), _p(g_fm2), _p(row_base), B, F, arena.K,
            _p(arena.grad), _p(w1.grad), _live(arena), _live(w1), _stream(ids)), "recalgo_deepfm_sparse_bwd")
        _flush(arena)
        _flush(w1)
        torch.sum(g_fm1, dim=0, out=bias.grad.view(1))
        return None, None, None, None, None, None, None


def deepfm_sparse(store, ids, arena, w1_arena, bias, row_base):
    """-> (deep_input [B,F*K], fm_first_order_logit [B,1], fm_second_order_logit [B,1])."""
    _chk(ids, torch.int64, "ids")
    if getattr(arena, "sharding", None) is not None:
        # the first-order arena mirrors the embedding arena's row layout: one exchange plan, two fetches
        from . import parallel
        plan, staged, ident = _staged(arena, parallel.global_rows(ids, row_base), store)
        staged_w1 = parallel.StagedArena(plan, w1_arena, store, torch.is_grad_enabled())
        return _DeepFMSparseFn.apply(store.anchor, ident.reshape(ids.shape), staged, staged_w1, bias,
                                     torch.zeros_like(row_base), False)
    return _DeepFMSparseFn.apply(store.anchor, ids, arena, w1_arena, bias, row_base, torch.is_grad_enabled())


# =============================================================================================
# K4: CrossNet
# =============================================================================================
def _pad4(t: torch.Tensor) -> torch.Tensor:
    """Zero-pad the last dimension to a multiple of 4 (the CrossNet kernels move float4s; the
    reference's default DCN input is d = 82).  Zero columns of x0 / w / b are inert: they add
    nothing to x_l . w_l and produce zero output columns."""
    pad = (-t.shape[-1]) % 4
    return t if pad == 0 else torch.nn.functional.pad(t, (0, pad))


class _CrossFn(Function):
    @staticmethod
    def forward(ctx, anchor, x0, w: Variable, b: Variable, grad_join):
        # w.data, b.data: [L, d].  grad_join: nn.GradJoin shared with the other consumer of x0 (or None).
        ctx.grad_join = grad_join
        B, d = x0.shape
        L = w.data.shape[0]
        lz = _take_lazy_gather(x0)           # x0 not gathered yet (ops.gather_feeds_cross): this kernel fetches the rows itself
        x0p, wp, bp = _pad4(x0), _pad4(w.data.reshape(L, d)), _pad4(b.data.reshape(L, d))
        dp = x0p.shape[1]
        out = torch.empty(B, dp, device=x0.device, dtype=torch.float32)
        if lz is not None and dp == d and x0.is_contiguous():
            ids, arena, row_base = lz
            _lib.check(_lib_().recalgo_gather_cross_fwd(_p(ids), _p(arena.weight), _p(row_base), B, ids.shape[1], arena.K, _p(wp), _p(bp), L,
                                                        _p(x0), d, _p(out), dp, _stream(x0)), "recalgo_gather_cross_fwd")
        else:
            if lz is not None:               # (cannot be fused after all: the plain gather first)
                ids, arena, row_base = lz
                _lib.check(_lib_().recalgo_embedding_gather_fwd(_p(ids), _p(arena.weight), _p(row_base), B, ids.shape[1], arena.K, _p(x0),
                                                                ids.shape[1] * arena.K, 0, _stream(ids)), "recalgo_embedding_gather_fwd")
                x0p = _pad4(x0)
            _lib.check(_lib_().recalgo_cross_fwd(
                _p(x0p), dp, _p(wp), _p(bp), B, dp, L, _p(out), dp, _stream(x0)),
                "recalgo_cross_fwd")
        ctx.vars = (w, b)
        ctx.d = d
        ctx.save_for_backward(x0p)
        return out if dp == d else out[:, :d]

    @staticmethod
    def backward(ctx, g):
        w, b = ctx.vars
        (x0p,) = ctx.saved_tensors
        B, dp = x0p.shape
        d = ctx.d
        L = w.data.shape[0]
        _cross_rider[:] = [e for e in _cross_rider if e[0] is not ctx]      # (nobody gave it a ride: computed here, as always)
        early, ctx.early = getattr(ctx, "early", None), None
        stale_dx = None
        if early is not None:
            # this backward already ran as a rider of a dense layer's launch (dense_bwd), on the gradient the fused tail produced
            eg, edx, jobs, used = early
            if eg.data_ptr() == g.data_ptr() and eg.shape == g.shape and eg.stride() == g.stride():
                if used():
                    return None, None, None, None, None          # ... and the MLP's first layer added dx0 to its own input gradient
                extra = ctx.grad_join.take() if ctx.grad_join is not None else None
                return None, (edx if extra is None else edx + extra.reshape(edx.shape)), None, None, None
            # the cross output had another consumer after all (autograd summed into a different tensor): the early result is
            # void — its deferred column sums are withdrawn, the backward runs here on the full gradient, and what the MLP's first
            # layer may already have added is taken back out
            for j in jobs:
                if any(j is e for e in _colsum_pending):
                    _colsum_pending[:] = [e for e in _colsum_pending if e is not j]
            stale_dx = edx if used() else None
        g = _pad4(g).contiguous()
        lib = _lib_()
        # unpadded width: the column sum of the partial rows joins the step's deferred-sum launch (its own scratch: the
        # rows are read after later kernels have run)
        defer = dp == d
        nbytes = int(lib.recalgo_cross_bwd_workspace_bytes(B, dp, L))
        if defer:
            key = ("cross", x0p.device.type, x0p.device.index, B, dp, L, w.grad.data_ptr())
            ws = _dense_ws.get(key)
            if ws is None:
                ws = _dense_ws[key] = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x0p.device)
        else:
            ws = _workspace(nbytes, x0p.device)
        dx0 = torch.empty_like(x0p)
        if dp == d:
            wp, bp, dw, db = w.data, b.data, w.grad, b.grad
        else:
            wp, bp = _pad4(w.data.reshape(L, d)), _pad4(b.data.reshape(L, d))
            dw, db = torch.empty_like(wp), torch.empty_like(bp)
        # the gradient the MLP branch parked for x0 is added in this kernel's epilogue (no separate add launch)
        extra = ctx.grad_join.take() if ctx.grad_join is not None else None
        fused_extra = extra is not None and dp == d and extra.is_contiguous() and tuple(extra.shape) == (B, d)
        _lib.check(lib.recalgo_cross_bwd(
            _p(x0p), dp, _p(wp), _p(bp), _p(g), dp, _p(extra) if fused_extra else None, B, dp, L, _p(dx0), _p(dw),
            _p(db), _p(ws), int(defer), _stream(x0p)), "recalgo_cross_bwd")
        if defer:
            rows, wsf = int(lib.recalgo_cross_bwd_partial_rows(B)), ws.view(torch.float32)
            _colsum_pending.append((wsf, 0, rows, 2 * L * dp, L * dp, dw))
            _colsum_pending.append((wsf, L * dp, rows, 2 * L * dp, L * dp, db))
        if dp != d:
            w.grad.copy_(dw[:, :d].reshape(w.grad.shape))
            b.grad.copy_(db[:, :d].reshape(b.grad.shape))
            dx0 = dx0[:, :d]
        if extra is not None and not fused_extra:
            dx0 = dx0 + extra.reshape(dx0.shape)
        if stale_dx is not None:
            dx0 = dx0 - stale_dx.reshape(dx0.shape)
        return None, dx0, None, None, None


def cross_stack(store, x0: torch.Tensor, w: Variable, b: Variable, grad_join=None) -> torch.Tensor:
    """Fused L-layer CrossNet: w, b are [L, d] block variables."""
    if store.building:
        return torch.zeros_like(x0)
    _chk(x0, torch.float32, "x0")
    out = _CrossFn.apply(store.anchor, x0, w, b, grad_join)
    if grad_join is not None and out.grad_fn is not None:
        out._recalgo_cross_node = out.grad_fn        # (the fused tail hands the branch's gradient over early: defer_cross_rider)
    return out


class _CrossLayerFn(Function):
    """The reference's un-fused signature cross_layer(x0, xl, index): xl distinct from x0."""

    @staticmethod
    def forward(ctx, anchor, x0, xl, w: Variable, b: Variable):
        B, d = x0.shape
        x0p, xlp = _pad4(x0), _pad4(xl)
        wp, bp = _pad4(w.data.reshape(1, d)), _pad4(b.data.reshape(1, d))
        dp = x0p.shape[1]
        out = torch.empty(B, dp, device=x0.device, dtype=torch.float32)
        _lib.check(_lib_().recalgo_cross_layer_fwd(
            _p(x0p), _p(xlp), dp, _p(wp), _p(bp), B, dp, _p(out), dp, _stream(x0)),
            "recalgo_cross_layer_fwd")
        ctx.vars = (w, b)
        ctx.d = d
        ctx.save_for_backward(x0p, xlp)
        return out if dp == d else out[:, :d]

    @staticmethod
    def backward(ctx, g):
        w, b = ctx.vars
        x0p, xlp = ctx.saved_tensors
        B, dp = x0p.shape
        d = ctx.d
        g = _pad4(g).contiguous()
        lib = _lib_()
        ws = _workspace(lib.recalgo_cross_bwd_workspace_bytes(B, dp, 1), x0p.device)
        dx0, dxl = torch.empty_like(x0p), torch.empty_like(x0p)
        if dp == d:
            wp, bp, dw, db = w.data, b.data, w.grad, b.grad
        else:
            wp, bp = _pad4(w.data.reshape(1, d)), _pad4(b.data.reshape(1, d))
            dw, db = torch.empty_like(wp), torch.empty_like(bp)
        _lib.check(lib.recalgo_cross_layer_bwd(
            _p(x0p), _p(xlp), dp, _p(wp), _p(bp), _p(g), dp, B, dp, _p(dx0), _p(dxl),
            _p(dw), _p(db), _p(ws), _stream(x0p)), "recalgo_cross_layer_bwd")
        if dp != d:
            w.grad.copy_(dw[:, :d].reshape(w.grad.shape))
            b.grad.copy_(db[:, :d].reshape(b.grad.shape))
            dx0, dxl = dx0[:, :d], dxl[:, :d]
        return None, dx0, dxl, None, None


def cross_layer(store, x0: torch.Tensor, xl: torch.Tensor, w: Variable, b: Variable) -> torch.Tensor:
    """One layer, w/b of shape (d, 1) or (d,): out = x0 * (xl . w) + b + xl."""
    if store.building:
        return torch.zeros_like(x0)
    _chk(x0, torch.float32, "x0")
    _chk(xl, torch.float32, "xl")
    return _CrossLayerFn.apply(store.anchor, x0, xl, w, b)


# =============================================================================================
# K5: CIN layer (fp32 MFMA implicit GEMM)
# =============================================================================================
class _CinFn(Function):
    """One launch-sized piece of a CIN layer: the feature maps [n0, n1) of the contribution of the previous layer's maps
    [h0, h1) (the whole layer when that is everything).  The kernels take <= 128 maps on either side per launch."""

    @staticmethod
    def forward(ctx, anchor, x0, xk, filt: Variable, h0, h1, n0, n1):
        B, m, D = x0.shape
        Hk_full = filt.data.shape[-2] // m
        N_full = filt.data.shape[-1]
        whole = (h0, h1, n0, n1) == (0, Hk_full, 0, N_full)
        if whole:
            w, xk_c = filt.data, xk
        else:
            w = filt.data.reshape(Hk_full, m, N_full)[h0:h1, :, n0:n1].reshape((h1 - h0) * m, n1 - n0).contiguous()
            xk_c = xk[:, h0:h1, :].contiguous()
        Hk, N = h1 - h0, n1 - n0
        out = torch.empty(B, N, D, device=x0.device, dtype=torch.float32)
        pool = torch.empty(B, N, device=x0.device, dtype=torch.float32)
        _lib.check(_lib_().recalgo_cin_layer_fwd(
            _p(x0), _p(xk_c), _p(w), B, m, Hk, N, D, _p(out), _p(pool), N, 0, _stream(x0)),
            "recalgo_cin_layer_fwd")
        ctx.filt, ctx.box, ctx.whole = filt, (h0, h1, n0, n1), whole
        ctx.xk_shape = xk.shape
        ctx.save_for_backward(x0, xk_c, w)
        ctx.set_materialize_grads(False)
        return out, pool

    @staticmethod
    def backward(ctx, g_out, g_pool):
        x0, xk, w = ctx.saved_tensors
        filt = ctx.filt
        h0, h1, n0, n1 = ctx.box
        B, m, D = x0.shape
        Hk, N = h1 - h0, n1 - n0
        dx0 = torch.empty_like(x0)
        dxk = torch.empty_like(xk)
        dw = filt.grad if ctx.whole else torch.empty_like(w)
        if g_out is None and g_pool is None:
            dw.zero_()
            dx0.zero_(), dxk.zero_()
        else:
            g_out = None if g_out is None else g_out.contiguous()
            g_pool = None if g_pool is None else g_pool.contiguous()
            lib = _lib_()
            ws = _workspace(lib.recalgo_cin_layer_bwd_workspace_bytes(B, m, Hk, N, D), x0.device)
            _lib.check(lib.recalgo_cin_layer_bwd(
                _p(x0), _p(xk), _p(w), _p(g_out), _p(g_pool), N, 0, B, m, Hk, N, D,
                _p(dx0), 0, _p(dxk), 0, _p(dw), _p(ws), _stream(x0)), "recalgo_cin_layer_bwd")
        if not ctx.whole:
            Hk_full = filt.data.shape[-2] // m
            filt.grad.reshape(Hk_full, m, filt.data.shape[-1])[h0:h1, :, n0:n1].copy_(dw.reshape(Hk, m, N))
            full = torch.zeros(ctx.xk_shape, device=x0.device, dtype=torch.float32)
            full[:, h0:h1, :] = dxk
            dxk = full
        return None, dx0, dxk, None, None, None, None, None


CIN_MAX_MAPS = 128          # feature maps per launch on either side (csrc/cin.hip)


class _CinStackFn(Function):
    """The whole CIN stack of xdeepfm.py:166-174 as ONE autograd node (every layer <= 128 maps: one launch each way per layer):
    layer i's sum-pooled maps are written by its kernel straight into their column block of p_plus (no torch.cat), the
    backward reads each layer's share of d(p_plus) in place (pool_stride / pool_col of recalgo_cin_layer_bwd: no slice
    copies), the gradient of a layer's output is the next layer's dX^k handed over as is, and dX^0 is accumulated by the
    kernels across the layers (dx0_accumulate) instead of by autograd (an elementwise launch per layer)."""

    @staticmethod
    def forward(ctx, anchor, x0, *filts):
        B, m, D = x0.shape
        lib = _lib_()
        Ns = [int(f.data.shape[-1]) for f in filts]
        p_plus = torch.empty(B, sum(Ns), device=x0.device, dtype=torch.float32)
        xs, xk, col = [], x0, 0
        for f, N in zip(filts, Ns):
            Hk = xk.shape[1]
            out = torch.empty(B, N, D, device=x0.device, dtype=torch.float32)
            _lib.check(lib.recalgo_cin_layer_fwd(_p(x0), _p(xk), _p(f.data), B, m, Hk, N, D, _p(out), _p(p_plus), p_plus.shape[1], col,
                                                 _stream(x0)), "recalgo_cin_layer_fwd")
            xs.append(out)
            xk = out
            col += N
        ctx.filts, ctx.Ns = filts, Ns
        ctx.save_for_backward(x0, *xs)
        ctx.set_materialize_grads(False)
        return (p_plus, *xs)

    @staticmethod
    def backward(ctx, g_pplus, *g_xs):
        x0, *xs = ctx.saved_tensors
        filts, Ns = ctx.filts, ctx.Ns
        B, m, D = x0.shape
        lib = _lib_()
        L = len(filts)
        if g_pplus is None and all(g is None for g in g_xs):
            for f in filts:
                f.grad.zero_()
            return (None, torch.zeros_like(x0)) + (None,) * L
        if g_pplus is not None and (g_pplus.stride(1) != 1 or g_pplus.shape[1] != sum(Ns)):
            g_pplus = g_pplus.contiguous()
        dx0 = torch.empty_like(x0)
        g_next = None                                # d(loss) / d(xs[i]) arriving from layer i + 1
        cols = [sum(Ns[:i]) for i in range(L)]
        first = True
        for i in range(L - 1, -1, -1):
            xk = x0 if i == 0 else xs[i - 1]
            Hk, N = xk.shape[1], Ns[i]
            g_out = g_next
            if g_xs[i] is not None:                  # a caller that also uses the layer's maps themselves (the reference does not)
                g_out = g_xs[i].contiguous() if g_out is None else g_out + g_xs[i]
            if g_out is None and g_pplus is None:
                filts[i].grad.zero_()
                g_next = None
                continue
            dxk = torch.empty_like(xk)
            ws = _workspace(lib.recalgo_cin_layer_bwd_workspace_bytes(B, m, Hk, N, D), x0.device)
            _lib.check(lib.recalgo_cin_layer_bwd(
                _p(x0), _p(xk), _p(filts[i].data), _p(g_out), _p(g_pplus), 0 if g_pplus is None else g_pplus.stride(0), cols[i],
                B, m, Hk, N, D, _p(dx0), 0 if first else 1, _p(dxk), 0, _p(filts[i].grad), _p(ws), _stream(x0)),
                "recalgo_cin_layer_bwd")
            first = False
            g_next = dxk
        if first:
            dx0.zero_()
        elif g_next is not None:
            dx0.add_(g_next)                         # layer 1: X^k IS X^0
        return (None, dx0) + (None,) * L


def cin_stack(store, x0: torch.Tensor, filts):
    """-> (p_plus [B, sum N_i], [X^1, ..., X^L]) of the CIN stack over x0 [B, m, D] with the filters `filts` (Variables of shape
    (1, H_{i-1} m, N_i)); None when a layer is wider than one launch takes (the caller then chains ops.cin_layer)."""
    if store.building or x0.dtype != torch.float32 or not x0.is_cuda:
        return None
    Hk = x0.shape[1]
    for f in filts:
        N = int(f.data.shape[-1])
        if Hk > CIN_MAX_MAPS or N > CIN_MAX_MAPS:
            return None
        Hk = N
    res = _CinStackFn.apply(store.anchor, x0.contiguous(), *filts)
    return res[0], list(res[1:])


def cin_layer(store, x0: torch.Tensor, xk: torch.Tensor, filt: Variable):
    """x0 [B,m,D], xk [B,Hk,D], filt (1, Hk*m, N) -> (xk_1 [B,N,D], sum-pooled [B,N]).  Layers wider than the kernels'
    128 x 128 maps per launch (e.g. --cin_layer_feature_maps=200,200) are tiled: output maps in column chunks
    (concatenated), previous-layer maps in row chunks (the layer is linear in X^k: the chunks' contributions add)."""
    if store.building:
        N = filt.data.shape[-1]
        return x0.new_zeros(x0.shape[0], N, x0.shape[2]), x0.new_zeros(x0.shape[0], N)
    _chk(x0, torch.float32, "x0")
    _chk(xk, torch.float32, "xk")
    Hk, N = xk.shape[1], filt.data.shape[-1]
    if Hk <= CIN_MAX_MAPS and N <= CIN_MAX_MAPS:
        return _CinFn.apply(store.anchor, x0, xk, filt, 0, Hk, 0, N)
    hs = [(h, min(h + CIN_MAX_MAPS, Hk)) for h in range(0, Hk, CIN_MAX_MAPS)]
    ns = [(n, min(n + CIN_MAX_MAPS, N)) for n in range(0, N, CIN_MAX_MAPS)]
    outs, pools = [], []
    for n0, n1 in ns:
        o = p = None
        for h0, h1 in hs:
            oc, pc = _CinFn.apply(store.anchor, x0, xk, filt, h0, h1, n0, n1)
            o, p = (oc, pc) if o is None else (o + oc, p + pc)
        outs.append(o)
        pools.append(p)
    return torch.cat(outs, dim=1), torch.cat(pools, dim=1)


# =============================================================================================
# K9: DIN attention
# =============================================================================================
class _DinAttentionFn(Function):
    @staticmethod
    def forward(ctx, anchor, query, keys, keys_length, vs, is_softmax: bool, query_join=None):
        # vs = (f1_w, f1_b, f2_w, f2_b, f3_w, f3_b) Variables
        B, T, H = keys.shape
        out = torch.empty(B, H, device=query.device, dtype=torch.float32)
        _lib.check(_lib_().recalgo_din_attention_fwd(
            _p(query), _p(keys), _p(keys_length), *[_p(v.data) for v in vs], B, T, H, int(is_softmax),
            _p(out), _stream(query)), "recalgo_din_attention_fwd")
        ctx.vs, ctx.is_softmax, ctx.kl = vs, is_softmax, keys_length
        ctx.query_join = query_join
        ctx.in_step = _loss_seed is not None      # built inside Estimator.train_step: its optimizer runs the deferred sums
        ctx.save_for_backward(query, keys)
        return out

    @staticmethod
    def backward(ctx, g):
        query, keys = ctx.saved_tensors
        B, T, H = keys.shape
        vs = ctx.vs
        lib = _lib_()
        # g may be a column block of a wider gradient matrix (the fcn input's): read in place when its rows are float4s
        if not (g.dim() == 2 and g.stride(1) == 1 and g.stride(0) % 4 == 0 and g.stride(0) >= H and g.data_ptr() % 16 == 0):
            g = g.contiguous()
        # the gradient the query's other consumer parked (nn.GradJoin): added in this kernel's epilogue
        extra = ctx.query_join.take() if ctx.query_join is not None else None
        fused_extra = extra is not None and extra.dim() == 2 and tuple(extra.shape) == (B, H) and extra.stride(1) == 1
        dq, dk = torch.empty_like(query), torch.empty_like(keys)
        nbytes = int(lib.recalgo_din_attention_bwd_workspace_bytes(B, T, H))
        if ctx.in_step:
            # inside a training step: the six parameter gradients are column sums of the kernel's partial rows — jobs of the
            # step's deferred-sum launch instead of a launch of their own (the partials need a buffer of their own until then)
            key = ("din_att", query.device.type, query.device.index, B, T, H, vs[0].grad.data_ptr())
            ws = _dense_ws.get(key)
            if ws is None:
                ws = _dense_ws[key] = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=query.device)
            grads = [None] * 6
        else:
            ws = _workspace(nbytes, query.device)
            grads = [_p(v.grad) for v in vs]
        _lib.check(lib.recalgo_din_attention_bwd_joined(
            _p(query), _p(keys), _p(ctx.kl), *[_p(v.data) for v in vs], _p(g), g.stride(0),
            _p(extra) if fused_extra else None, extra.stride(0) if fused_extra else 0, B, T, H,
            int(ctx.is_softmax), _p(dq), _p(dk), *grads, _p(ws), _stream(query)), "recalgo_din_attention_bwd")
        if ctx.in_step:
            rows, pf = int(lib.recalgo_din_attention_bwd_partial_rows(B)), int(lib.recalgo_din_attention_bwd_partial_floats(H))
            wsf, off = ws.view(torch.float32), 0
            flat = all(vs[i + 1].grad.data_ptr() == vs[i].grad.data_ptr() + 4 * vs[i].grad.numel()
                       and vs[i + 1].grad.untyped_storage().data_ptr() == vs[0].grad.untyped_storage().data_ptr()
                       for i in range(5))
            if flat:      # the six gradients lie in the flat gradient buffer in the partial row's order: ONE job
                g0 = vs[0].grad
                out = torch.empty(0, dtype=torch.float32, device=g0.device).set_(g0.untyped_storage(), g0.storage_offset(), (pf,), (1,))
                _colsum_pending.append((wsf, 0, rows, pf, pf, out))
            else:
                for v in vs:
                    n = v.grad.numel()
                    _colsum_pending.append((wsf, off, rows, pf, n, v.grad))
                    off += n
        if extra is not None and not fused_extra:
            dq = dq + extra.reshape(dq.shape)
        return None, dq, dk, None, None, None, None


def din_attention(store, query, keys, keys_length, vs, is_softmax=False, query_join=None) -> torch.Tensor:
    """query [B,H], keys [B,T,H], keys_length [B] int32 -> [B,H].  query_join (nn.GradJoin): the query's other consumer
    parks its input gradient there and this op's backward kernel adds it to d(query)."""
    if store.building:
        return torch.zeros_like(query)
    _chk(query, torch.float32, "query")
    _chk(keys, torch.float32, "keys")
    if keys_length.dtype != torch.int32:
        keys_length = keys_length.to(torch.int32)
    return _DinAttentionFn.apply(store.anchor, query, keys, keys_length.contiguous(), tuple(vs), bool(is_softmax), query_join)


# =============================================================================================
# K7/K8: FiBiNET SENET + bilinear interaction
# =============================================================================================
BILINEAR_TYPES = {"all": 0, "each": 1, "interaction": 2}


class _SenetFn(Function):
    @staticmethod
    def forward(ctx, anchor, emb, w1: Variable, w2: Variable):
        B, F, K = emb.shape
        Rd = w1.data.shape[1]
        v = torch.empty_like(emb)
        _lib.check(_lib_().recalgo_senet_fwd(_p(emb), _p(w1.data), _p(w2.data), B, F, K, Rd, _p(v), None,
                                             _stream(emb)), "recalgo_senet_fwd")
        ctx.vars = (w1, w2)
        ctx.save_for_backward(emb)
        return v

    @staticmethod
    def backward(ctx, g):
        w1, w2 = ctx.vars
        (emb,) = ctx.saved_tensors
        B, F, K = emb.shape
        Rd = w1.data.shape[1]
        lib = _lib_()
        g = g.contiguous()
        ws = _workspace(lib.recalgo_senet_bwd_workspace_bytes(B, F, K, Rd), emb.device)
        d = torch.empty_like(emb)
        _lib.check(lib.recalgo_senet_bwd(_p(emb), _p(w1.data), _p(w2.data), _p(g), B, F, K, Rd, _p(d), 0,
                                         _p(w1.grad), _p(w2.grad), _p(ws), _stream(emb)), "recalgo_senet_bwd")
        return None, d, None, None


def senet(store, emb: torch.Tensor, w1: Variable, w2: Variable) -> torch.Tensor:
    """emb [B,F,K], w1 (F, Rd), w2 (Rd, F) -> re-weighted embeddings [B,F,K]."""
    if store.building:
        return torch.zeros_like(emb)
    _chk(emb, torch.float32, "input")
    return _SenetFn.apply(store.anchor, emb, w1, w2)


class _BilinearFn(Function):
    @staticmethod
    def forward(ctx, anchor, btype: int, x0, w0: Variable, x1, w1: Optional[Variable]):
        B, F, K = x0.shape
        nv = 1 if x1 is None else 2
        P = (F - 1) * (F - 2) // 2
        out = torch.empty(B, P, nv * K, device=x0.device, dtype=torch.float32)
        _lib.check(_lib_().recalgo_bilinear_fwd(
            _p(x0), _p(w0.data), _p(x1), None if w1 is None else _p(w1.data), B, F, K, btype, _p(out), nv * K, 0,
            _stream(x0)), "recalgo_bilinear_fwd")
        ctx.vars, ctx.btype, ctx.nv = (w0, w1), btype, nv
        ctx.save_for_backward(x0, x1)
        return out

    @staticmethod
    def backward(ctx, g):
        w0, w1 = ctx.vars
        x0, x1 = ctx.saved_tensors
        B, F, K = x0.shape
        nv = ctx.nv
        lib = _lib_()
        g = g.contiguous()
        ws = _workspace(lib.recalgo_bilinear_bwd_workspace_bytes(B, F, K, nv, ctx.btype), x0.device)
        dx0 = torch.empty_like(x0)
        dx1 = None if x1 is None else torch.empty_like(x1)
        _lib.check(lib.recalgo_bilinear_bwd(
            _p(x0), _p(w0.data), _p(x1), None if w1 is None else _p(w1.data), _p(g), nv * K, 0, B, F, K, ctx.btype,
            _p(dx0), _p(w0.grad), _p(dx1), None if w1 is None else _p(w1.grad), _p(ws), _stream(x0)),
            "recalgo_bilinear_bwd")
        return None, None, dx0, None, dx1, None


def bilinear_interaction(store, btype: str, x0: torch.Tensor, w0: Variable,
                         x1: Optional[torch.Tensor] = None, w1: Optional[Variable] = None) -> torch.Tensor:
    """(x_s [B,F,K], W_s) for one or two sets -> [B, (F-1)(F-2)/2, n_sets*K] (sets concatenated on
    the last axis).  For type "interaction" only the first (F-1)(F-2)/2 slices of W_s are read and
    receive gradient (the reference's zip truncation); the rest keep a zero gradient."""
    if btype not in BILINEAR_TYPES:
        raise ValueError(f"Bilinear Interaction type must be in ['all','each','interaction'], got '{btype}'")
    if store.building:
        B, F, K = x0.shape
        return x0.new_zeros(B, (F - 1) * (F - 2) // 2, (1 if x1 is None else 2) * K)
    _chk(x0, torch.float32, "input")
    if x1 is not None:
        _chk(x1, torch.float32, "input")
    return _BilinearFn.apply(store.anchor, BILINEAR_TYPES[btype], x0, w0, x1, w1)


# =============================================================================================
# K6: PNN product layer
# =============================================================================================
PNN_METHODS = {"IPNN": 0, "OPNN": 1}


class _PnnProductFn(Function):
    """relu(emb @ linear_w + phi(emb) @ omega(product_w) + bias): feature / weight builders (csrc/pnn.hip) and
    the D-way contraction (csrc/dense.hip, fp32 MFMA: both operand pairs, the bias and the ReLU in ONE
    launch; pnn.py:139,146-181) are hand-written kernels, forward and backward."""

    @staticmethod
    def forward(ctx, anchor, emb_flat, linear_w: Variable, product_w: Variable, bias: Variable, F, K, method):
        B = emb_flat.shape[0]
        D = linear_w.data.shape[1]
        lib = _lib_()
        T = lib.recalgo_pnn_feature_count(F, K, method)
        T4 = (T + 3) // 4 * 4          # row stride of phi / row count of omega: float4-addressable GEMM operands
        st = _stream(emb_flat)
        phi = torch.empty(B, T4, device=emb_flat.device, dtype=torch.float32)      # padding columns zeroed by the kernel
        omega = _pnn_omega(product_w, T4, D)                                        # padding rows stay zero
        _lib.check(lib.recalgo_pnn_features_fwd(_p(emb_flat), B, F, K, method, _p(phi), T4, st), "recalgo_pnn_features_fwd")
        _lib.check(lib.recalgo_pnn_weights_fwd(_p(product_w.data), D, F, K, method, _p(omega), st),
                   "recalgo_pnn_weights_fwd")
        # lz + lp + bias, ReLU  (pnn.py:139,175,178,181)
        y = dense_fwd(emb_flat, linear_w.data.reshape(-1, D), bias.data.reshape(-1), True, x2=phi, w2=omega)
        ctx.vars = (linear_w, product_w, bias)
        ctx.dims = (F, K, method)
        ctx.save_for_backward(emb_flat, phi, omega, y)
        return y

    @staticmethod
    def backward(ctx, g):
        linear_w, product_w, bias = ctx.vars
        F, K, method = ctx.dims
        emb_flat, phi, omega, y = ctx.saved_tensors
        B, D = y.shape
        T4 = phi.shape[1]
        lib = _lib_()
        st = _stream(emb_flat)
        g = g.contiguous()
        # gz = g * [y > 0] is applied inside the kernels (never materialised); each operand pair's input and weight
        # gradients are one merged launch.  The omega gradient is needed right away (pnn_weights_bwd), so its split
        # sum is not deferred to the end of the step
        domega = torch.empty_like(omega)
        dphi = dense_bwd(phi, g, y, omega, domega, None)
        d_emb = dense_bwd(emb_flat, g, y, linear_w.data.reshape(-1, D), linear_w.grad.view(-1, D), bias.grad.view(-1),
                          defer=True)
        _lib.check(lib.recalgo_pnn_features_bwd(_p(emb_flat), _p(dphi), T4, B, F, K, method, _p(d_emb), 1, st),
                   "recalgo_pnn_features_bwd")
        _lib.check(lib.recalgo_pnn_weights_bwd(_p(product_w.data), _p(domega), D, F, K, method, _p(product_w.grad), st),
                   "recalgo_pnn_weights_bwd")
        return None, d_emb, None, None, None, None, None, None


_pnn_omega_cache = {}


def _pnn_omega(product_w: Variable, T4: int, D: int) -> torch.Tensor:
    """The [T4, D] omega buffer of one product layer, allocated once (zeros): recalgo_pnn_weights_fwd rewrites its first
    T rows every step, the padding rows stay zero."""
    key = (product_w.data.data_ptr(), T4, D)
    t = _pnn_omega_cache.get(key)
    if t is None:
        if len(_pnn_omega_cache) > 16:
            _pnn_omega_cache.clear()
        t = _pnn_omega_cache[key] = torch.zeros(T4, D, device=product_w.data.device, dtype=torch.float32)
    return t


def pnn_product_layer(store, emb_flat: torch.Tensor, linear_w: Variable, product_w: Variable, bias: Variable,
                      F: int, K: int, method: str) -> torch.Tensor:
    """emb_flat [B, F*K] -> relu(lz + lp + bias) [B, D] (pnn.py:133-181)."""
    if store.building:
        return emb_flat.new_zeros(emb_flat.shape[0], linear_w.data.shape[1])
    _chk(emb_flat, torch.float32, "fields_embeddings")
    return _PnnProductFn.apply(store.anchor, emb_flat, linear_w, product_w, bias, int(F), int(K),
                               PNN_METHODS["IPNN" if method == "IPNN" else "OPNN"])


class _IpnnFeaturesFn(Function):
    """phi[b, t(i,j)] = <e_i[b], e_j[b]>, i <= j: the Gram upper triangle of the fields, diagonal included
    (`recalgo_pnn_features_*`, method IPNN)."""

    @staticmethod
    def forward(ctx, emb_flat, F, K):
        B = emb_flat.shape[0]
        lib = _lib_()
        T = lib.recalgo_pnn_feature_count(F, K, 0)
        phi = torch.empty(B, T, device=emb_flat.device, dtype=torch.float32)
        _lib.check(lib.recalgo_pnn_features_fwd(_p(emb_flat), B, F, K, 0, _p(phi), T, _stream(emb_flat)),
                   "recalgo_pnn_features_fwd")
        ctx.dims = (F, K)
        ctx.save_for_backward(emb_flat)
        return phi

    @staticmethod
    def backward(ctx, dphi):
        (emb_flat,) = ctx.saved_tensors
        F, K = ctx.dims
        dphi = dphi.contiguous()
        d_emb = torch.empty_like(emb_flat)
        _lib.check(_lib_().recalgo_pnn_features_bwd(_p(emb_flat), _p(dphi), dphi.shape[1], emb_flat.shape[0], F, K, 0, _p(d_emb), 0,
                                                    _stream(emb_flat)), "recalgo_pnn_features_bwd")
        return d_emb, None, None


class _PairKernel:
    """The one-unit head over phi whose [T, 1] kernel is the pair strengths r scattered to the off-diagonal columns of the
    Gram triangle (0 on the diagonal).  Quacks like a Variable for the fused loss tail (`data`; its gradient = the column
    sums of the tail's partial rows, delivered straight into r.grad: row i of the strict triangle is one contiguous run
    of both index spaces, so the F - 1 runs are F - 1 jobs of the step's deferred-sum launch — no select launch)."""

    def __init__(self, r: Variable, w: torch.Tensor, F: int, anchor):
        self.r, self.data, self.F, self.anchor = r, w, int(F), anchor
        self.grad = None

    def colsum_jobs(self, partials, col, rows, stride):
        F, out, jobs, at = self.F, self.r.grad.view(-1), [], 0
        for i in range(F - 1):
            n = F - 1 - i
            jobs.append((partials, col + i * F - i * (i - 1) // 2 + 1, rows, stride, n, out[at:at + n]))
            at += n
        return jobs

    def apply_head(self, parts):
        return _PairHeadFn.apply(self.anchor, parts[0], self.r, self.F)


class _PairHeadFn(Function):
    """logit[b] = sum_{i<j} r[index(i,j)] * phi[b, t(i,j)]  (FwFM second order, fwfm.py:146-158) outside a TRAIN step's
    fused tail: the one-unit head kernels (`recalgo_dense1_*`) over phi."""

    @staticmethod
    def forward(ctx, anchor, phi, r: Variable, F):
        w = _pair_vector(r, F, phi.shape[1], phi.device)
        ctx.r, ctx.F = r, F
        ctx.save_for_backward(phi, w)
        return dense1_fwd([phi], w, None)

    @staticmethod
    def backward(ctx, g):
        phi, w = ctx.saved_tensors
        dphi = torch.empty_like(phi) if ctx.needs_input_grad[1] else None
        dw = torch.empty_like(w)
        dense1_bwd([phi], w, g.contiguous(), [dphi], dw, None)
        torch.index_select(dw.reshape(-1), 0, _pair_index(ctx.F, phi.device), out=ctx.r.grad.view(-1))
        return None, dphi, None, None


def _pair_vector(r: Variable, F: int, T: int, dev) -> torch.Tensor:
    """[T, 1]: r at the off-diagonal columns of the Gram triangle, 0 on its diagonal (allocated and cleared once)."""
    w = _pair_vectors.get((r.data.data_ptr(), T))
    if w is None:
        w = _pair_vectors[(r.data.data_ptr(), T)] = torch.zeros(T, 1, device=dev, dtype=torch.float32)
    w.index_copy_(0, _pair_index(F, dev), r.data.reshape(-1, 1))
    return w


_pair_index_cache = {}
_pair_vectors = {}


def _pair_index(F: int, device) -> torch.Tensor:
    """index_from_upper_triangular(i, j, F) (reference utils.py:67-82, row-major strict upper triangle) ->
    column t(i, j) = i*F - i(i-1)/2 + (j - i) of the Gram upper triangle with diagonal (include/recalgo.h)."""
    key = (F, device)
    t = _pair_index_cache.get(key)
    if t is None:
        cols = [i * F - i * (i - 1) // 2 + (j - i) for i in range(F - 1) for j in range(i + 1, F)]
        t = _pair_index_cache[key] = torch.tensor(cols, dtype=torch.int64, device=device)
    return t


def field_pair_logit(store, emb_flat: torch.Tensor, r: Variable, F: int, K: int):
    """emb_flat [B, F*K], r [F(F-1)/2] -> [B, 1]: sum_{i<j} r[index(i,j)] <e_i, e_j> (fwfm.py:146-158).  In a TRAIN step
    the sum over the pairs is left to the fused loss tail (nn.LazyLogit: one launch for this head, the loss and the backward
    of both)."""
    if store.building:
        return emb_flat.new_zeros(emb_flat.shape[0], 1)
    _chk(emb_flat, torch.float32, "fields_embeddings")
    F, K = int(F), int(K)
    phi = _IpnnFeaturesFn.apply(emb_flat.contiguous(), F, K)
    if logit_loss_supported([phi], []):
        from .nn import LazyLogit
        return LazyLogit([(_PairKernel(r, _pair_vector(r, F, phi.shape[1], phi.device), F, store.anchor), None, [phi])])
    return _PairHeadFn.apply(store.anchor, phi, r, F)


# =============================================================================================
# sibling models: NFM bi-interaction, AFM attention pooling, FFM pair dots (csrc/siblings.hip)
# =============================================================================================
class _BiInteractionFn(Function):
    @staticmethod
    def forward(ctx, emb, F, K):
        emb = emb.contiguous()
        B = emb.shape[0]
        out = torch.empty(B, K, device=emb.device, dtype=torch.float32)
        _lib.check(_lib_().recalgo_bi_interaction_fwd(_p(emb), B, F, K, _p(out), _stream(emb)), "recalgo_bi_interaction_fwd")
        ctx.dims = (F, K)
        ctx.save_for_backward(emb)
        return out

    @staticmethod
    def backward(ctx, g):
        (emb,) = ctx.saved_tensors
        F, K = ctx.dims
        d = torch.empty_like(emb)
        _lib.check(_lib_().recalgo_bi_interaction_bwd(_p(emb), _p(g.contiguous()), emb.shape[0], F, K, _p(d), _stream(emb)),
                   "recalgo_bi_interaction_bwd")
        return d, None, None


def bi_interaction(emb_flat: torch.Tensor, F: int, K: int) -> torch.Tensor:
    """[B, F*K] -> [B, K]: 0.5 * ((sum_f e_f)^2 - sum_f e_f^2) (nfm.py:155-167)."""
    _chk(emb_flat, torch.float32, "fields_embeddings")
    return _BiInteractionFn.apply(emb_flat, int(F), int(K))


class _AttentionPoolFn(Function):
    @staticmethod
    def forward(ctx, pairs, att):
        pairs, att = pairs.contiguous(), att.contiguous()
        B, P, K = pairs.shape
        out = torch.empty(B, K, device=pairs.device, dtype=torch.float32)
        score = torch.empty(B, P, device=pairs.device, dtype=torch.float32)
        _lib.check(_lib_().recalgo_attention_pool_fwd(_p(pairs), _p(att), B, P, K, _p(out), _p(score), _stream(pairs)),
                   "recalgo_attention_pool_fwd")
        ctx.save_for_backward(pairs, score)
        return out

    @staticmethod
    def backward(ctx, g):
        pairs, score = ctx.saved_tensors
        B, P, K = pairs.shape
        dp, da = torch.empty_like(pairs), torch.empty_like(score)
        _lib.check(_lib_().recalgo_attention_pool_bwd(_p(pairs), _p(score), _p(g.contiguous()), B, P, K, _p(dp), _p(da),
                                                      _stream(pairs)), "recalgo_attention_pool_bwd")
        return dp, da


def attention_pool(pairs: torch.Tensor, att: torch.Tensor) -> torch.Tensor:
    """pairs [B, P, K], att [B, P] -> sum_p softmax(att)[p] * pairs[:, p, :]  (afm.py:184-188)."""
    return _AttentionPoolFn.apply(pairs, att.reshape(pairs.shape[0], pairs.shape[1]))


class _FfmPairsFn(Function):
    @staticmethod
    def forward(ctx, x, F, K):
        x = x.contiguous()
        B = x.shape[0]
        out = torch.empty(B, 1, device=x.device, dtype=torch.float32)
        _lib.check(_lib_().recalgo_ffm_pairs_fwd(_p(x), B, F, K, _p(out), _stream(x)), "recalgo_ffm_pairs_fwd")
        ctx.dims = (F, K)
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        F, K = ctx.dims
        dx = torch.empty_like(x)
        _lib.check(_lib_().recalgo_ffm_pairs_bwd(_p(x), _p(g.contiguous()), x.shape[0], F, K, _p(dx), _stream(x)),
                   "recalgo_ffm_pairs_bwd")
        return dx, None, None


def ffm_pairs(x: torch.Tensor, F: int, K: int) -> torch.Tensor:
    """x [B, F*(F-1)*K] (field i in its sub-table s at [i, s]) -> [B, 1] = sum_{i<j} <x[i][j-1], x[j][i]> (ffm.py:146-160)."""
    _chk(x, torch.float32, "field-aware embeddings")
    return _FfmPairsFn.apply(x, int(F), int(K))


# =============================================================================================
# context-MLP glue (csrc/mlp.hip): dense backward epilogue, BatchNorm training
# =============================================================================================
def _mat(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 or not t.is_cuda:
        raise ValueError(f"{name}: expected a row-major fp32 device matrix, got {t.dtype} {tuple(t.shape)} strides {t.stride()}")
    return t


def bn_partial_rows(rows: int) -> int:
    return int(_lib_().recalgo_batchnorm_partial_rows(int(rows)))


def dense_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], relu: bool,
              x2: Optional[torch.Tensor] = None, w2: Optional[torch.Tensor] = None,
              bn_partials: Optional[torch.Tensor] = None, drop: Optional["DropSpec"] = None) -> torch.Tensor:
    """act(x @ w (+ x2 @ w2) + bias) on the fp32 matrix cores (include/recalgo.h recalgo_dense_fwd).  bn_partials
    [bn_partial_rows(M), 2 N]: the launch also leaves the per-tile batch moments of the result there (recalgo_dense_fwd_bn),
    for the BatchNorm layer that consumes it (batchnorm_train_fwd(..., partials=))."""
    x, w = _mat(x, "x"), _mat(w, "w")
    M, K = x.shape
    N = w.shape[1]
    if w.shape[0] != K or w.stride(0) != N:
        raise ValueError("dense_fwd: w must be a contiguous [K, N] matrix")
    if x2 is not None:
        x2, w2 = _mat(x2, "x2"), _mat(w2, "w2")
        if w2.shape != (x2.shape[1], N) or w2.stride(0) != N or x2.shape[0] != M:
            raise ValueError("dense_fwd: second operand pair does not match")
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    if bn_partials is not None and (tuple(bn_partials.shape) != (bn_partial_rows(M), 2 * N) or not bn_partials.is_contiguous()):
        raise ValueError("dense_fwd: bn_partials must be a contiguous [bn_partial_rows(M), 2 N] tensor")
    if drop is not None:
        # the dropout behind the layer rides in the epilogue (recalgo_dense_fwd_drop): y, and the moments, are those of the dropped tensor
        drop.check_mask((M, N))
        cd, _keep = _cdrop(drop)
        _lib.check(_lib_().recalgo_dense_fwd_drop(
            _p(x), x.stride(0), _p(w), K, _p(x2), 0 if x2 is None else x2.stride(0), _p(w2), 0 if x2 is None else x2.shape[1],
            _p(bias), M, N, int(relu), _p(y), N, _p(bn_partials), cd, _stream(x)), "recalgo_dense_fwd_drop")
        return y
    _lib.check(_lib_().recalgo_dense_fwd_bn(
        _p(x), x.stride(0), _p(w), K, _p(x2), 0 if x2 is None else x2.stride(0), _p(w2), 0 if x2 is None else x2.shape[1],
        _p(bias), M, N, int(relu), _p(y), N, _p(bn_partials), _stream(x)), "recalgo_dense_fwd")
    return y


def dense_fwd_act(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], kind: int, alpha: torch.Tensor,
                  bn_partials: torch.Tensor):
    """z = x @ w + bias, y = prelu | dice (z, alpha) and the per-tile batch moments of y in ONE launch
    (recalgo_dense_fwd_act_bn): the forward of DIN's dense -> activation -> batch_norm layers up to the BatchNorm merge.
    -> (z, y)"""
    x, w = _mat(x, "x"), _mat(w, "w")
    M, K = x.shape
    N = w.shape[1]
    if w.shape[0] != K or w.stride(0) != N:
        raise ValueError("dense_fwd_act: w must be a contiguous [K, N] matrix")
    if tuple(bn_partials.shape) != (bn_partial_rows(M), 2 * N) or not bn_partials.is_contiguous():
        raise ValueError("dense_fwd_act: bn_partials must be a contiguous [bn_partial_rows(M), 2 N] tensor")
    z = torch.empty(M, N, device=x.device, dtype=torch.float32)
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    _lib.check(_lib_().recalgo_dense_fwd_act_bn(
        _p(x), x.stride(0), _p(w), K, None, 0, None, 0, _p(bias), M, N, 0, int(kind), _p(alpha), _p(z), _p(y), N,
        _p(bn_partials), _stream(x)), "recalgo_dense_fwd_act_bn")
    return z, y


def dense_bwd_input(g: torch.Tensor, y_mask: Optional[torch.Tensor], w: torch.Tensor, c_in: Optional[torch.Tensor] = None,
                    beta: float = 0.0, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """dx = (g * [y_mask > 0]) @ w^T (+ beta * c_in); y_mask contiguous with g's layout (recalgo_dense_bwd_input)."""
    g, w = _mat(g, "g"), _mat(w, "w")
    M, N = g.shape
    K = w.shape[0]
    if w.shape[1] != N or w.stride(0) != N:
        raise ValueError("dense_bwd_input: w must be a contiguous [K, N] matrix")
    if y_mask is not None and (y_mask.shape != g.shape or y_mask.stride() != g.stride()):
        raise ValueError("dense_bwd_input: y_mask must have g's layout")
    dx = out if out is not None else torch.empty(M, K, device=g.device, dtype=torch.float32)
    _lib.check(_lib_().recalgo_dense_bwd_input(
        _p(g), g.stride(0), _p(y_mask), _p(w), M, N, K, _p(c_in), 0 if c_in is None else c_in.stride(0), float(beta),
        _p(dx), dx.stride(0), int(accumulate), _stream(g)), "recalgo_dense_bwd_input")
    return dx


_side_streams = {}
_side_keepalive = []         # tensors a side-stream kernel still reads (kept alive until the streams are joined)
_side_dirty = set()          # devices whose side stream has un-joined work


def side_stream(device) -> "torch.cuda.Stream":
    """The second HIP stream of the step (one per device): the weight-gradient GEMMs run on it, concurrently with the
    input-gradient chain on the main stream (they only meet again in the step's deferred-sum launch)."""
    key = (device.type, device.index)
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st


def join_side_streams() -> None:
    """Make the current stream wait for the side-stream work of this step."""
    for dev in list(_side_dirty):
        torch.cuda.current_stream(dev).wait_stream(side_stream(dev))
    _side_dirty.clear()
    _side_keepalive.clear()


_dense_ws = {}
_dense_pending = []          # deferred split reductions of this backward pass: (M, K, N, ws, dw, dbias)
_colsum_pending = []         # deferred plain column sums: (partials, element offset, rows, row_stride, n, out)


class _ColSum(ctypes.Structure):              # include/recalgo.h recalgo_colsum_t
    _fields_ = [("partials", ctypes.c_void_p), ("out", ctypes.c_void_p), ("rows", ctypes.c_int),
                ("row_stride", ctypes.c_int64), ("n", ctypes.c_int64)]


class _DenseSplit(ctypes.Structure):          # include/recalgo.h recalgo_dense_split_t
    _fields_ = [("M", ctypes.c_int), ("K", ctypes.c_int), ("N", ctypes.c_int), ("workspace", ctypes.c_void_p),
                ("dw", ctypes.c_void_p), ("dbias", ctypes.c_void_p)]


def dense_bwd_weights(x: torch.Tensor, g: torch.Tensor, y_mask: Optional[torch.Tensor], dw: torch.Tensor,
                      dbias: Optional[torch.Tensor], defer: bool = False) -> None:
    """dw = x^T (g * [y_mask > 0]), dbias = colsum(g * [y_mask > 0]) (recalgo_dense_bwd_weights; deterministic).
    `defer`: the fixed-order sum of the batch-split partials is left to `flush_dense_splits()` (one launch for all
    the layers of a backward pass; the optimizer and `named_grads` call it) — dw / dbias are valid only after it."""
    x, g = _mat(x, "x"), _mat(g, "g")
    M, K = x.shape
    N = g.shape[1]
    if g.shape[0] != M or tuple(dw.shape) != (K, N) or not dw.is_contiguous():
        raise ValueError("dense_bwd_weights: shape mismatch")
    if y_mask is not None and (y_mask.shape != g.shape or y_mask.stride() != g.stride()):
        raise ValueError("dense_bwd_weights: y_mask must have g's layout")
    lib = _lib_()
    if M == 0:
        dw.zero_()
        if dbias is not None:
            dbias.zero_()
        return
    # own scratch per weight tensor: a deferred reduction reads it after later layers have run
    nbytes = int(lib.recalgo_dense_bwd_weights_workspace_bytes(M, K, N))
    key = (x.device.type, x.device.index, M, K, N, dw.data_ptr() if defer else 0)
    ws = _dense_ws.get(key)
    if ws is None:
        ws = _dense_ws[key] = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    defer = bool(defer and nbytes > 0)
    _lib.check(lib.recalgo_dense_bwd_weights(_p(x), x.stride(0), _p(g), g.stride(0), _p(y_mask), M, K, N, _p(dw), _p(dbias),
                                             _p(ws), int(defer), _stream(x)), "recalgo_dense_bwd_weights")
    if defer:
        _dense_pending.append((M, K, N, ws, dw, dbias))


def dense_bwd(x: torch.Tensor, g: torch.Tensor, y_mask: Optional[torch.Tensor], w: torch.Tensor, dw: torch.Tensor,
              dbias: Optional[torch.Tensor], c_in: Optional[torch.Tensor] = None, beta: float = 0.0,
              defer: bool = False, bn=None, premask: Optional[torch.Tensor] = None, cross_rider: bool = True) -> torch.Tensor:
    """Both gradients of a dense layer in one launch (recalgo_dense_bwd): returns dx = (g * [y_mask > 0]) @ w^T
    (+ beta * c_in); dw / dbias as dense_bwd_weights (valid after flush_dense_splits() when `defer`).
    premask [M, K] (rows contiguous): dx is zeroed where premask <= 0 (before the beta * c_in term) — x itself when x is the
    ReLU output of the layer below, whose backward then needs no mask (nn.ReluSource).
    bn = (bn_x [M, K] contiguous, mean [K], rstd [K], partials [bn_partial_rows(M), 2 K]): x is the output of a training-mode
    BatchNorm over bn_x — the launch also leaves the sums that BatchNorm's backward starts with (recalgo_dense_bwd_bn)."""
    x, g, w = _mat(x, "x"), _mat(g, "g"), _mat(w, "w")
    M, K = x.shape
    N = g.shape[1]
    if M == 0:
        dense_bwd_weights(x, g, y_mask, dw, dbias)
        return torch.zeros(0, K, device=g.device, dtype=torch.float32)
    if g.shape[0] != M or tuple(dw.shape) != (K, N) or not dw.is_contiguous() or tuple(w.shape) != (K, N) or w.stride(0) != N:
        raise ValueError("dense_bwd: shape mismatch")
    if y_mask is not None and (y_mask.shape != g.shape or y_mask.stride() != g.stride()):
        raise ValueError("dense_bwd: y_mask must have g's layout")
    lib = _lib_()
    nbytes = int(lib.recalgo_dense_bwd_weights_workspace_bytes(M, K, N))
    key = (x.device.type, x.device.index, M, K, N, dw.data_ptr() if defer else 0)
    ws = _dense_ws.get(key)
    if ws is None:
        ws = _dense_ws[key] = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    defer = bool(defer and nbytes > 0)
    dx = torch.empty(M, K, device=g.device, dtype=torch.float32)
    bnp = (None, None, None, None)
    if bn is not None:
        bx, bmean, brstd, bpart = bn
        if (tuple(bx.shape) != (M, K) or not bx.is_contiguous() or bmean.numel() != K or brstd.numel() != K
                or tuple(bpart.shape) != (bn_partial_rows(M), 2 * K) or not bpart.is_contiguous()):
            raise ValueError("dense_bwd: bn = (x [M, K], mean [K], rstd [K], partials [bn_partial_rows(M), 2 K])")
        bnp = (_p(bx), _p(bmean), _p(brstd), _p(bpart))
    if premask is not None and (tuple(premask.shape) != (M, K) or premask.stride(1) != 1 or premask.dtype != torch.float32):
        raise ValueError("dense_bwd: premask must be [M, K] fp32 with contiguous rows")
    rider = _take_wgrad_rider(M, x.device) if defer else None
    crider = _take_cross_rider(M, x.device) if (defer and cross_rider and y_mask is None) else None
    if rider is not None or crider is not None:
        rx, rg, rdw, rdb = rider if rider is not None else (None, None, None, None)
        rK, rN = (rx.shape[1], rg.shape[1]) if rider is not None else (0, 0)
        if lib.recalgo_dense_bwd_rider_supported(_p(x), x.stride(0), _p(g), g.stride(0), _p(y_mask), _p(w), M, K, N, _p(dx), K,
                                                 _p(rx), 0 if rx is None else rx.stride(0), _p(rg), 0 if rg is None else rg.stride(0),
                                                 rK, rN):
            rws = _wgrad_workspace(rx.device, M, rK, rN, rdw) if rider is not None else None
            cargs, cdone = (None, 0, None, None, None, 0, 0, 0, None, None), None
            if crider is not None:
                cargs, cdone = _cross_rider_args(*crider)
            _lib.check(lib.recalgo_dense_bwd_rider(_p(x), x.stride(0), _p(g), g.stride(0), _p(y_mask), _p(w), M, K, N, _p(c_in),
                                                   0 if c_in is None else c_in.stride(0), float(beta), _p(dx), K, _p(dw), _p(dbias),
                                                   _p(ws), int(defer), *bnp, _p(premask), 0 if premask is None else premask.stride(0),
                                                   _p(rx), 0 if rx is None else rx.stride(0), _p(rg), 0 if rg is None else rg.stride(0),
                                                   rK, rN, _p(rdw), _p(rdb), _p(rws), *cargs, _stream(x)), "recalgo_dense_bwd_rider")
            _dense_pending.append((M, K, N, ws, dw, dbias))
            if rider is not None:
                rider_stats["wgrad"] += 1
                if int(lib.recalgo_dense_bwd_weights_workspace_bytes(M, rK, rN)) > 0:
                    _dense_pending.append((M, rK, rN, rws, rdw, rdb))
            if cdone is not None:
                rider_stats["cross"] += 1
                cdone()
            return dx
        if rider is not None:
            dense_bwd_weights(rx, rg, None, rdw, rdb, defer=True)          # (not on the vectorised tile paths: a launch of its own)
        if crider is not None:
            _cross_rider.append(crider)                                    # (left to the cross node's own backward)
    _lib.check(lib.recalgo_dense_bwd_bn(_p(x), x.stride(0), _p(g), g.stride(0), _p(y_mask), _p(w), M, K, N, _p(c_in),
                                        0 if c_in is None else c_in.stride(0), float(beta), _p(dx), K, _p(dw), _p(dbias), _p(ws),
                                        int(defer), *bnp, _p(premask), 0 if premask is None else premask.stride(0), _stream(x)),
               "recalgo_dense_bwd")
    if defer:
        _dense_pending.append((M, K, N, ws, dw, dbias))
    return dx


# A weight gradient waiting for a launch to ride in (recalgo_dense_bwd_rider): left by the fused tail's backward (its layer's
# operands are ready, the layer below runs its merged backward next), taken by the next deferred dense_bwd over the same batch;
# whatever is still here when the deferred sums are flushed (or another rider arrives) is launched on its own.
_wgrad_rider = []


def defer_wgrad_rider(x: torch.Tensor, g: torch.Tensor, dw: torch.Tensor, dbias: Optional[torch.Tensor]) -> None:
    launch_wgrad_rider()
    _wgrad_rider.append((_mat(x, "x"), _mat(g, "g"), dw, dbias))


def _take_wgrad_rider(M: int, device):
    if not _wgrad_rider:
        return None
    rx, rg, rdw, rdb = _wgrad_rider[0]
    if rx.shape[0] != M or rx.device != device or rdb is None or not rdw.is_contiguous():
        return None
    return _wgrad_rider.pop()


def launch_wgrad_rider() -> None:
    while _wgrad_rider:
        rx, rg, rdw, rdb = _wgrad_rider.pop()
        dense_bwd_weights(rx, rg, None, rdw, rdb, defer=True)


# The CrossNet backward waiting for a ride (recalgo_dense_bwd_rider's c_* arguments): left by the fused tail's backward, which
# produces the cross branch's upstream gradient long before autograd runs the cross node (created first, it runs last); taken by
# the next deferred dense_bwd WITHOUT a GradJoin of its own (the layer that shares x0 with the cross network needs the result).
_cross_rider = []
rider_stats = {"wgrad": 0, "cross": 0}       # launches that carried a rider (tests / bench read it)


def defer_cross_rider(node, g: torch.Tensor) -> None:
    """node: the _CrossFn node of the cross output whose gradient g [B, d] the caller has just produced."""
    _cross_rider.clear()
    if getattr(node, "grad_join", None) is None or not cross_riders_enabled:
        return
    w, _ = node.vars
    L = int(w.data.shape[0])
    if (g.dim() == 2 and g.is_contiguous() and g.dtype == torch.float32 and g.shape[1] == node.d and node.d % 4 == 0
            and g.data_ptr() % 16 == 0 and _lib_().recalgo_dense_bwd_cross_rider_supported(int(node.d), L)):
        _cross_rider.append((node, g))


cross_riders_enabled = True


def _take_cross_rider(M: int, device):
    if not _cross_rider:
        return None
    node, g = _cross_rider[0]
    if g.shape[0] != M or g.device != device:
        return None
    return _cross_rider.pop()


def _cross_rider_args(node, g):
    """-> (the c_* arguments of recalgo_dense_bwd_rider, what to do once the launch is out)"""
    lib = _lib_()
    w, b = node.vars
    (x0p,) = node.saved_tensors
    B, dp = x0p.shape
    L = int(w.data.shape[0])
    key = ("cross", x0p.device.type, x0p.device.index, B, dp, L, w.grad.data_ptr())
    ws = _dense_ws.get(key)
    if ws is None:
        ws = _dense_ws[key] = torch.empty(max(int(lib.recalgo_cross_bwd_workspace_bytes(B, dp, L)), 16), dtype=torch.uint8,
                                          device=x0p.device)
    dx0 = torch.empty_like(x0p)
    join = node.grad_join

    def done():
        rows, wsf = int(lib.recalgo_cross_bwd_partial_rows(B)), ws.view(torch.float32)
        jobs = [(wsf, 0, rows, 2 * L * dp, L * dp, w.grad), (wsf, L * dp, rows, 2 * L * dp, L * dp, b.grad)]
        _colsum_pending.extend(jobs)
        join.early_dx, join.early_used = dx0, False
        node.early = (g, dx0, jobs, lambda: join.early_used)
    return (_p(x0p), dp, _p(w.data), _p(b.data), _p(g), g.stride(0), dp, L, _p(dx0), _p(ws)), done


def _wgrad_workspace(device, M, K, N, dw):
    nbytes = int(_lib_().recalgo_dense_bwd_weights_workspace_bytes(M, K, N))
    key = (device.type, device.index, M, K, N, dw.data_ptr())
    ws = _dense_ws.get(key)
    if ws is None:
        ws = _dense_ws[key] = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
    return ws


def flush_dense_splits(step_dev: Optional[torch.Tensor] = None) -> None:
    """Finish every deferred sum of the step in ONE launch: the weight-gradient split reductions of `dense`, the column
    sums the loss tail left behind, and (`step_dev`, the optimizer's int64 step counter) the step increment — the
    optimizer kernel that follows then only reads the counter."""
    join_side_streams()
    launch_wgrad_rider()
    if not _dense_pending and not _colsum_pending and step_dev is None:
        return
    jobs = (_DenseSplit * max(len(_dense_pending), 1))()
    for i, (M, K, N, ws, dw, dbias) in enumerate(_dense_pending):
        jobs[i] = _DenseSplit(M, K, N, ws.data_ptr(), dw.data_ptr(), 0 if dbias is None else dbias.data_ptr())
    sums = (_ColSum * max(len(_colsum_pending), 1))()
    for i, (part, off, rows, stride, n, out) in enumerate(_colsum_pending):
        sums[i] = _ColSum(part.data_ptr() + 4 * off, out.data_ptr(), rows, stride, n)
    dev_t = _dense_pending[0][4] if _dense_pending else (_colsum_pending[0][0] if _colsum_pending else step_dev)
    nj, ns = len(_dense_pending), len(_colsum_pending)
    _dense_pending.clear()
    _colsum_pending.clear()
    _lib.check(_lib_().recalgo_dense_bwd_weights_reduce(jobs, nj, sums, ns, _p(step_dev), _stream(dev_t)),
               "recalgo_dense_bwd_weights_reduce")


def mlp_width_supported(C: int) -> bool:
    return bool(_lib_().recalgo_mlp_width_supported(int(C)))


def relu_bwd_bias_(g: torch.Tensor, y: Optional[torch.Tensor], dbias: torch.Tensor) -> torch.Tensor:
    """g2 = g * [y > 0] (g itself when y is None), dbias[:] = colsum(g2); -> g2.  g, y [rows, C]."""
    rows, C = g.shape
    lib = _lib_()
    ws = _workspace(lib.recalgo_relu_bwd_bias_workspace_bytes(rows, C), g.device)
    g2 = None if y is None else torch.empty_like(g)
    _lib.check(lib.recalgo_relu_bwd_bias(_p(g), _p(y), rows, C, _p(g2), _p(dbias), _p(ws), _stream(g)),
               "recalgo_relu_bwd_bias")
    return g if y is None else g2


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def dense1_supported(parts) -> bool:
    """The one-unit head kernel takes 1..4 contiguous fp32 [B, w] device tensors with one B."""
    return (1 <= len(parts) <= 4 and all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.is_contiguous()
                                         and t.shape[0] == parts[0].shape[0] and t.shape[1] >= 1 for t in parts))


def dense1_fwd(parts, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """[B, 1] = concat(parts, -1) @ w + bias without the concat (include/recalgo.h recalgo_dense1_fwd)."""
    import ctypes
    B = parts[0].shape[0]
    widths = (ctypes.c_int * len(parts))(*[int(t.shape[1]) for t in parts])
    out = torch.empty(B, 1, dtype=torch.float32, device=parts[0].device)
    _lib.check(_lib_().recalgo_dense1_fwd(_ptr_array(parts), widths, len(parts), B, _p(w), _p(bias), _p(out), _stream(out)),
               "recalgo_dense1_fwd")
    return out


def dense1_bwd(parts, w: torch.Tensor, g: torch.Tensor, dxs, dw: torch.Tensor, dbias: Optional[torch.Tensor]) -> None:
    import ctypes
    B, C = parts[0].shape[0], sum(int(t.shape[1]) for t in parts)
    lib = _lib_()
    widths = (ctypes.c_int * len(parts))(*[int(t.shape[1]) for t in parts])
    ws = _workspace(lib.recalgo_dense1_bwd_workspace_bytes(B, C), g.device)
    _lib.check(lib.recalgo_dense1_bwd(_ptr_array(parts), widths, len(parts), B, _p(w), _p(g), _ptr_array(dxs), _p(dw), _p(dbias),
                                      _p(ws), _stream(g)), "recalgo_dense1_bwd")


def batchnorm_train_fwd(x, gamma, beta, moving_mean, moving_var, momentum: float, eps: float, partials=None,
                        out_drop: Optional["DropSpec"] = None):
    """partials: the per-tile moments of x when its producer has already left them (dense_fwd(bn_partials=)): ONE launch
    (merge + apply) instead of two.  out_drop: the dropout BEHIND the BatchNorm rides in the store (recalgo_batchnorm_apply_drop)."""
    rows, C = x.shape
    lib = _lib_()
    y = torch.empty_like(x)
    mean = torch.empty(C, device=x.device, dtype=torch.float32)
    rstd = torch.empty(C, device=x.device, dtype=torch.float32)
    if out_drop is not None:
        out_drop.check_mask((rows, C))
        if partials is None:
            partials = torch.empty(bn_partial_rows(rows), 2 * C, device=x.device, dtype=torch.float32)
            _lib.check(lib.recalgo_batchnorm_moments(_p(x), rows, C, _p(partials), _stream(x)), "recalgo_batchnorm_moments")
        cd, _keep = _cdrop(out_drop)
        _lib.check(lib.recalgo_batchnorm_apply_drop(_p(x), _p(gamma), _p(beta), _p(partials), 1, rows, C, eps, momentum,
                                                    _p(moving_mean), _p(moving_var), _p(y), _p(mean), _p(rstd), cd, _stream(x)),
                   "recalgo_batchnorm_apply_drop")
        return y, mean, rstd
    if partials is not None:
        _lib.check(lib.recalgo_batchnorm_apply(_p(x), _p(gamma), _p(beta), _p(partials), 1, rows, C, eps, momentum,
                                               _p(moving_mean), _p(moving_var), _p(y), _p(mean), _p(rstd), _stream(x)),
                   "recalgo_batchnorm_apply")
        return y, mean, rstd
    ws = _workspace(lib.recalgo_batchnorm_workspace_bytes(rows, C), x.device)
    _lib.check(lib.recalgo_batchnorm_train_fwd(_p(x), _p(gamma), _p(beta), rows, C, eps, momentum, _p(moving_mean),
                                               _p(moving_var), _p(y), _p(mean), _p(rstd), _p(ws), _stream(x)),
               "recalgo_batchnorm_train_fwd")
    return y, mean, rstd


def batchnorm_train_bwd_act(x, gamma, mean, rstd, g, dgamma, dbeta, kind: int, z, alpha, dalpha, defer: bool,
                            sums=None, g_drop: Optional["DropSpec"] = None) -> torch.Tensor:
    """BatchNorm backward continued through the per-channel activation x = act(z, alpha) (recalgo_batchnorm_train_bwd_act):
    -> dL/dz; dgamma / dbeta / dalpha are overwritten (`defer`: dalpha by the step's deferred-sum launch)."""
    rows, C = x.shape
    lib = _lib_()
    nbytes = int(lib.recalgo_batchnorm_bwd_act_workspace_bytes(rows, C))
    dz = torch.empty_like(x)
    if defer:
        key = ("bn_act", x.device.type, x.device.index, rows, C, dalpha.data_ptr())
        ws = _dense_ws.get(key)
        if ws is None:
            ws = _dense_ws[key] = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    else:
        ws = _workspace(nbytes, x.device)
    if g_drop is not None:
        # the BatchNorm's output went through a dropout: g is read as g * keep / (1 - rate) (sums of the un-dropped g are of no use)
        cd, _keep = _cdrop(g_drop)
        _lib.check(lib.recalgo_batchnorm_train_bwd_drop(_p(x), _p(gamma), _p(mean), _p(rstd), _p(g), None, rows, C, int(kind), _p(z), _p(alpha),
                                                        _p(dz), _p(dgamma), _p(dbeta), None if defer else _p(dalpha), _p(ws), 0, 1.0,
                                                        cd, _stream(x)), "recalgo_batchnorm_train_bwd_drop")
    else:
        _lib.check(lib.recalgo_batchnorm_train_bwd_act(_p(x), _p(gamma), _p(mean), _p(rstd), _p(g), _p(sums), rows, C, int(kind), _p(z), _p(alpha),
                                                       _p(dz), _p(dgamma), _p(dbeta), None if defer else _p(dalpha), _p(ws), 0,
                                                       _stream(x)), "recalgo_batchnorm_train_bwd_act")
    if defer:
        nb = bn_partial_rows(rows)
        _colsum_pending.append((ws.view(torch.float32), nb * 2 * C, nb, C, C, dalpha))
    return dz


_syncbn_rows_checked = set()


def _syncbn_check_rows(rows: int, world: int, all_gather, device) -> None:
    """The Sync-BatchNorm kernels weigh every rank's tiles by the LOCAL row count: every rank must hold the same number of
    examples (weak scaling with drop_remainder; a ragged last batch would give wrong global statistics, or a hang when the
    tile counts differ).  Checked once per batch size (one tiny all_gather), outside a capture."""
    key = (int(rows), int(world))
    if key in _syncbn_rows_checked or (device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
        return
    got = torch.empty(world, 1, device=device, dtype=torch.float32)
    all_gather(got, torch.full((1,), float(rows), device=device))
    if not bool((got == float(rows)).all()):
        raise ValueError(f"sync_batch_norm: every rank must hold the same number of examples per step (this rank: {rows}, "
                         f"ranks: {got.reshape(-1).tolist()}); drop the ragged last batch")
    _syncbn_rows_checked.add(key)


def batchnorm_sync_fwd(x, gamma, beta, moving_mean, moving_var, momentum: float, eps: float, sync):
    """Sync-BatchNorm forward (include/recalgo.h "Sync-BatchNorm building blocks"): per-tile moments -> all_gather of the
    partial rows over the data-parallel group -> merge + apply.  sync = (world, rank, all_gather(out [world, n], in [n])).
    Every rank must hold the same number of rows (checked once per batch size)."""
    world, _, all_gather = sync
    rows, C = x.shape
    _syncbn_check_rows(rows, world, all_gather, x.device)
    lib = _lib_()
    nb = int(lib.recalgo_batchnorm_partial_rows(rows))
    local = torch.empty(nb * 2 * C, device=x.device, dtype=torch.float32)
    _lib.check(lib.recalgo_batchnorm_moments(_p(x), rows, C, _p(local), _stream(x)), "recalgo_batchnorm_moments")
    parts = torch.empty(world, nb * 2 * C, device=x.device, dtype=torch.float32)
    all_gather(parts, local)
    y = torch.empty_like(x)
    mean = torch.empty(C, device=x.device, dtype=torch.float32)
    rstd = torch.empty(C, device=x.device, dtype=torch.float32)
    _lib.check(lib.recalgo_batchnorm_apply(_p(x), _p(gamma), _p(beta), _p(parts), world, rows, C, eps, momentum,
                                           _p(moving_mean), _p(moving_var), _p(y), _p(mean), _p(rstd), _stream(x)),
               "recalgo_batchnorm_apply")
    return y, mean, rstd


def batchnorm_sync_bwd(x, gamma, mean, rstd, g, dgamma, dbeta, sync) -> torch.Tensor:
    world, rank, all_gather = sync
    rows, C = x.shape
    lib = _lib_()
    nb = int(lib.recalgo_batchnorm_partial_rows(rows))
    local = torch.empty(nb * 2 * C, device=x.device, dtype=torch.float32)
    _lib.check(lib.recalgo_batchnorm_bwd_sums(_p(x), _p(mean), _p(rstd), _p(g), rows, C, _p(local), _stream(x)),
               "recalgo_batchnorm_bwd_sums")
    parts = torch.empty(world, nb * 2 * C, device=x.device, dtype=torch.float32)
    all_gather(parts, local)
    dx = torch.empty_like(x)
    _lib.check(lib.recalgo_batchnorm_bwd_apply(_p(x), _p(gamma), _p(mean), _p(rstd), _p(g), _p(parts), world, rank, rows, C,
                                               _p(dx), _p(dgamma), _p(dbeta), 0, _stream(x)), "recalgo_batchnorm_bwd_apply")
    return dx


def batchnorm_train_bwd(x, gamma, mean, rstd, g, dgamma, dbeta, sums=None, relu_x: bool = False, relu_scale: float = 1.0,
                        g_drop: Optional["DropSpec"] = None) -> torch.Tensor:
    """sums [bn_partial_rows(rows), 2 C]: the per-tile (colsum g | colsum g * xhat) rows when g's producer has left them
    (dense_bwd(bn=)): ONE launch (merge + apply) instead of two.  relu_x: x is a ReLU output — dx is zeroed where x <= 0
    (the producing dense layer then needs no mask: nn.ReluSource); relu_scale = 1 / (1 - rate) when x also went through a
    dropout in front of this BatchNorm (dense(relu) -> dropout -> batch_norm).  g_drop: a dropout BEHIND this BatchNorm."""
    rows, C = x.shape
    lib = _lib_()
    dx = torch.empty_like(x)
    if relu_scale != 1.0 or g_drop is not None:
        ws = _workspace(lib.recalgo_batchnorm_bwd_act_workspace_bytes(rows, C), x.device)
        cd, _keep = _cdrop(g_drop)
        _lib.check(lib.recalgo_batchnorm_train_bwd_drop(_p(x), _p(gamma), _p(mean), _p(rstd), _p(g), None if g_drop is not None else _p(sums),
                                                        rows, C, -1, None, None, _p(dx), _p(dgamma), _p(dbeta), None, _p(ws),
                                                        int(relu_x), float(relu_scale), cd, _stream(x)), "recalgo_batchnorm_train_bwd_drop")
        return dx
    if sums is not None:
        _lib.check(lib.recalgo_batchnorm_bwd_apply(_p(x), _p(gamma), _p(mean), _p(rstd), _p(g), _p(sums), 1, 0, rows, C, _p(dx),
                                                   _p(dgamma), _p(dbeta), int(relu_x), _stream(x)), "recalgo_batchnorm_bwd_apply")
        return dx
    ws = _workspace(lib.recalgo_batchnorm_workspace_bytes(rows, C), x.device)
    _lib.check(lib.recalgo_batchnorm_train_bwd(_p(x), _p(gamma), _p(mean), _p(rstd), _p(g), rows, C, _p(dx), _p(dgamma),
                                               _p(dbeta), _p(ws), int(relu_x), _stream(x)), "recalgo_batchnorm_train_bwd")
    return dx


# =============================================================================================
# a14: loss tail
# =============================================================================================
# The gradient the training loop seeds the loss with (1, or 1/world under data parallelism) is known
# before the forward runs: inside `loss_seed(c)` the loss kernel bakes c into dlogit and the backward
# hands dlogit on unchanged — no ones_like fill, no scaling launch.  The caller promises to seed
# loss.backward() with exactly c (Estimator.train_step does).
_loss_seed: Optional[float] = None


class loss_seed:
    def __init__(self, c: float):
        self.c = float(c)

    def __enter__(self):
        global _loss_seed
        self.prev, _loss_seed = _loss_seed, self.c
        return self

    def __exit__(self, *exc):
        global _loss_seed
        _loss_seed = self.prev
        return False


class _SigmoidCEFn(Function):
    @staticmethod
    def forward(ctx, logits, labels):
        ctx.set_materialize_grads(False)      # no zero tensor for the (non-differentiable) probabilities
        ctx.seed = _loss_seed
        B = logits.numel()
        lg = logits.contiguous().view(-1)
        lb = labels.contiguous().view(-1).to(torch.float32)
        prob = torch.empty_like(lg)
        loss = torch.empty(1, device=lg.device, dtype=torch.float32)
        dlogit = torch.empty_like(lg)
        _lib.check(_lib_().recalgo_sigmoid_ce_fwd_bwd(
            _p(lg), _p(lb), B, 1.0 if ctx.seed is None else ctx.seed, _p(prob), _p(loss), _p(dlogit), _stream(lg)),
            "recalgo_sigmoid_ce_fwd_bwd")
        ctx.save_for_backward(dlogit)
        ctx.shape = logits.shape
        ctx.mark_non_differentiable(prob)
        return loss.view(()), prob.view(logits.shape)

    @staticmethod
    def backward(ctx, gloss, _gprob):
        (dlogit,) = ctx.saved_tensors
        if gloss is None:
            return None, None
        if ctx.seed is not None:
            return dlogit.view(ctx.shape), None
        return (dlogit * gloss).view(ctx.shape), None


def sigmoid_cross_entropy(logits: torch.Tensor, labels: torch.Tensor):
    """-> (mean loss scalar, probabilities)."""
    return _SigmoidCEFn.apply(logits, labels)


class _LogitLossFn(Function):
    """TRAIN-step tail in one launch (include/recalgo.h recalgo_logit_loss_fwd_bwd): one-unit head(s) + sigmoid-CE +
    their backward.  Requires the loss-gradient seed to be known (ops.loss_seed): the gradients are produced by the
    forward launch, the backward only hands them out.  The loss VALUE and the head's weight / bias gradients are
    column sums of per-workgroup partials, finished by the step's deferred-sum launch (flush_dense_splits)."""

    @staticmethod
    def forward(ctx, anchor, labels, heads, bias, n_addends, loss_addend, *tensors):
        # heads: [(kernel Variable, n_parts)]; tensors = parts of all heads (in order) + the addend tensors [B, 1]
        ctx.set_materialize_grads(False)
        lib = _lib_()
        n_parts = sum(n for _, n in heads)
        parts, addends = list(tensors[:n_parts]), list(tensors[n_parts:])
        B = parts[0].shape[0]
        dev = parts[0].device
        widths = [int(t.shape[1]) for t in parts]
        C = sum(widths)
        ws_w, i = [], 0
        for kernel, n in heads:                      # weight slice of every part inside its head's (sum w, 1) kernel
            off = 0
            for t in parts[i:i + n]:
                ws_w.append(kernel.data.reshape(-1)[off:off + t.shape[1]])
                off += t.shape[1]
            i += n
        rows = int(lib.recalgo_logit_loss_partial_rows(B))
        partials = torch.empty(rows, C + 2, device=dev, dtype=torch.float32)
        logit = torch.empty(B, 1, device=dev, dtype=torch.float32)
        prob, dlogit = torch.empty_like(logit), torch.empty_like(logit)
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        need = ctx.needs_input_grad[6:6 + n_parts]
        dxs = [torch.empty_like(t) if nd else None for t, nd in zip(parts, need)]
        # a part that IS the ReLU output of a dense layer gets its gradient already masked (nn.ReluSource)
        # (not one that also went through a fused dropout: the mask here does not scale — its producer masks and scales itself)
        srcs = [getattr(t, "_recalgo_relu_src", None) if (nd and getattr(t, "_recalgo_relu_scale", 1.0) == 1.0) else None
                for t, nd in zip(parts, need)]
        relu_flags = (ctypes.c_int * n_parts)(*[int(sr is not None) for sr in srcs])
        lb = labels.contiguous().view(-1).to(torch.float32)
        wi = (ctypes.c_int * n_parts)(*widths)
        _lib.check(lib.recalgo_logit_loss_fwd_bwd(
            _ptr_array(parts), _ptr_array(ws_w), wi, n_parts, None if bias is None else _p(bias.data),
            _p(addends[0].contiguous().view(-1)) if n_addends > 0 else None,
            _p(addends[1].contiguous().view(-1)) if n_addends > 1 else None,
            _p(lb), _p(loss_addend), B, float(_loss_seed), _p(logit), _p(prob), _p(dlogit), _ptr_array(dxs), relu_flags, _p(partials),
            _stream(logit)), "recalgo_logit_loss_fwd_bwd")
        for sr, dxp in zip(srcs, dxs):
            if sr is not None:
                sr.premasked = dxp
        # deferred column sums: dw of every head (contiguous columns of the partial rows), d bias, the loss value
        col, i = 0, 0
        for kernel, n in heads:
            wsum = sum(widths[i:i + n])
            if hasattr(kernel, "colsum_jobs"):           # (a derived kernel delivers its gradient itself: ops._PairKernel)
                _colsum_pending.extend(kernel.colsum_jobs(partials, col, rows, C + 2))
            else:
                _colsum_pending.append((partials, col, rows, C + 2, wsum, kernel.grad))
            col += wsum
            i += n
        if bias is not None:
            _colsum_pending.append((partials, C, rows, C + 2, 1, bias.grad))
        _colsum_pending.append((partials, C + 1, rows, C + 2, 1, loss))
        _dlogit_partials.clear()                     # sum_b dlogit[b] = column C of the partial rows (colsum_of_dlogit)
        _dlogit_partials[dlogit.data_ptr()] = (partials, C, rows, C + 2, B)
        ctx.dxs, ctx.dlogit, ctx.n_addends = dxs, dlogit, n_addends
        ctx.mark_non_differentiable(prob, logit)
        return loss.view(()), prob, logit

    @staticmethod
    def backward(ctx, gloss, _gprob, _glogit):
        if gloss is None:
            return (None,) * (6 + len(ctx.dxs) + ctx.n_addends)
        # the seed was baked into dlogit / dx by the forward launch (the caller promises to seed backward with it)
        return (None, None, None, None, None, None, *ctx.dxs, *([ctx.dlogit] * ctx.n_addends))


class _TailDenseHeadFn(Function):
    """The last hidden layer + the one-unit head + sigmoid-CE + the backward of all three down to the layer's input, in ONE
    launch (include/recalgo.h recalgo_tail_dense_head_fwd_bwd; DCN's tail, dcn.py:166-172).  As _LogitLossFn: the loss-gradient
    seed is known, the forward launch produces the gradients, the backward hands them out — and launches the layer's weight
    gradient (a batch reduction: its own launch, split partials summed by the step's deferred-sum launch)."""

    @staticmethod
    def forward(ctx, anchor, labels, head_kernel, head_bias, w3, b3, side_first, loss_addend, h2, side):
        ctx.set_materialize_grads(False)
        lib = _lib_()
        B, K2 = h2.shape
        N3 = int(w3.data.shape[1])
        Cs = 0 if side is None else int(side.shape[1])
        C = Cs + N3
        dev = h2.device
        hk = head_kernel.data.reshape(-1)
        w_side, w_h3 = (hk[:Cs], hk[Cs:]) if side_first else (hk[N3:], hk[:N3])
        rows = int(lib.recalgo_tail_partial_rows(B))
        partials = torch.empty(rows, C + 2, device=dev, dtype=torch.float32)
        logit = torch.empty(B, 1, device=dev, dtype=torch.float32)
        prob, dlogit = torch.empty_like(logit), torch.empty_like(logit)
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        d_side = torch.empty_like(side) if (side is not None and ctx.needs_input_grad[9]) else None
        dz3 = torch.empty(B, N3, device=dev, dtype=torch.float32)
        dh2 = torch.empty_like(h2)
        lb = labels.contiguous().view(-1).to(torch.float32)
        _lib.check(lib.recalgo_tail_dense_head_fwd_bwd(
            _p(h2), K2, _p(w3.data), _p(b3.data), N3, _p(side), Cs, int(bool(side_first)), _p(w_side) if Cs else None, _p(w_h3),
            None if head_bias is None else _p(head_bias.data), _p(lb), _p(loss_addend), B, float(_loss_seed), _p(logit), _p(prob),
            _p(dlogit), _p(d_side), _p(dz3), _p(dh2), _p(partials), _stream(logit)), "recalgo_tail_dense_head_fwd_bwd")
        src = getattr(h2, "_recalgo_relu_src", None)
        if src is not None:
            src.premasked = dh2                        # masked with h2 > 0 by the kernel: the producing layer skips its own mask
        if hasattr(head_kernel, "colsum_jobs"):
            _colsum_pending.extend(head_kernel.colsum_jobs(partials, 0, rows, C + 2))
        else:
            _colsum_pending.append((partials, 0, rows, C + 2, C, head_kernel.grad))
        if head_bias is not None:
            _colsum_pending.append((partials, C, rows, C + 2, 1, head_bias.grad))
        _colsum_pending.append((partials, C + 1, rows, C + 2, 1, loss))
        _dlogit_partials.clear()
        _dlogit_partials[dlogit.data_ptr()] = (partials, C, rows, C + 2, B)
        ctx.h2, ctx.dz3, ctx.dh2, ctx.d_side, ctx.vars = h2, dz3, dh2, d_side, (w3, b3)
        ctx.cross_node = getattr(side, "_recalgo_cross_node", None) if side is not None else None
        ctx.mark_non_differentiable(prob, logit)
        return loss.view(()), prob, logit

    @staticmethod
    def backward(ctx, gloss, _gprob, _glogit):
        if gloss is None:
            return (None,) * 10
        w3, b3 = ctx.vars
        # (rides in the launch of the layer below's backward, which autograd runs next: ops.dense_bwd picks it up)
        defer_wgrad_rider(ctx.h2, ctx.dz3, w3.grad, b3.grad)
        if ctx.cross_node is not None and ctx.d_side is not None:
            # ... and so does the backward of the cross network `side` came out of: autograd would run that node last
            defer_cross_rider(ctx.cross_node, ctx.d_side)
        return (None, None, None, None, None, None, None, None, ctx.dh2, ctx.d_side)


def tail_dense_head_supported(h2, units: int, side) -> bool:
    """recalgo_tail_dense_head_fwd_bwd serves this last-hidden-layer / head pair (inside a training step whose loss seed is known)."""
    if _loss_seed is None or not torch.is_grad_enabled():
        return False
    for t in (h2, side):
        if t is None:
            continue
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.is_contiguous()
                and t.data_ptr() % 16 == 0):
            return False
    if side is not None and side.shape[0] != h2.shape[0]:
        return False
    return bool(_lib_().recalgo_tail_dense_head_supported(int(h2.shape[1]), int(units), 0 if side is None else int(side.shape[1])))


def tail_dense_head(store, labels, head_kernel, head_bias, w3, b3, h2, side, side_first: bool,
                    loss_addend: Optional[torch.Tensor] = None):
    """-> (mean sigmoid-CE loss (+ loss_addend), probabilities [B, 1], logit [B, 1]) of
    logit = dense1(concat([side, relu(h2 w3 + b3)]) in the given order) + head_bias."""
    if loss_addend is not None:
        loss_addend = loss_addend.detach().reshape(1).to(torch.float32)
    return _TailDenseHeadFn.apply(store.anchor, labels, head_kernel, head_bias, w3, b3, bool(side_first), loss_addend, h2, side)


class _ConcatSumsqFn(Function):
    """torch.cat(parts, -1) with sum(out^2) * scale as a detached by-product (one launch, include/recalgo.h
    recalgo_concat_sumsq); the gradient of a part is its column block of the output's gradient."""

    @staticmethod
    def forward(ctx, scale, joins, *parts):
        lib = _lib_()
        ctx.joins = joins or {}
        parts = [t.contiguous() for t in parts]
        B = parts[0].shape[0]
        widths = [int(t.shape[1]) for t in parts]
        dev = parts[0].device
        out = torch.empty(B, sum(widths), device=dev, dtype=torch.float32)
        val = torch.empty(1, device=dev, dtype=torch.float32)
        key = ("concat_sumsq", dev.type, dev.index, B)
        ws = _dense_ws.get(key)
        if ws is None:
            ws = _dense_ws[key] = torch.zeros(int(lib.recalgo_concat_sumsq_workspace_bytes(B)), dtype=torch.uint8, device=dev)
        wi = (ctypes.c_int * len(parts))(*widths)
        _lib.check(lib.recalgo_concat_sumsq(_ptr_array(parts), wi, len(parts), B, _p(out), float(scale), _p(val), _p(ws),
                                            _stream(out)), "recalgo_concat_sumsq")
        ctx.widths = widths
        ctx.mark_non_differentiable(val)
        ctx.set_materialize_grads(False)      # (else autograd fills a zero "gradient" for val: one launch per step)
        return out, val

    @staticmethod
    def backward(ctx, g, _gv):
        if g is None:
            return (None, None) + (None,) * len(ctx.widths)
        outs, off = [], 0
        for i, w in enumerate(ctx.widths):
            gi = g[:, off:off + w]
            join = ctx.joins.get(i)
            # (a part with a second consumer whose backward kernel adds this block itself: parked, not returned)
            outs.append(None if join is not None and join.park(gi) else gi)
            off += w
        return (None, None, *outs)


def concat_sumsq(parts, scale: float, joins=None):
    """-> (concat(parts, -1) [B, C], scale * sum(concat^2) as a detached [1] tensor); 1..4 fp32 [B, w] device tensors.
    joins {part index: nn.GradJoin}: that part's gradient block is parked for its other consumer's backward kernel."""
    return _ConcatSumsqFn.apply(float(scale), joins, *parts)


_dlogit_partials = {}


def colsum_of_dlogit(g: torch.Tensor, out: torch.Tensor) -> bool:
    """out[0] = sum_b g[b] as ONE MORE JOB of the step's deferred-sum launch, when g is the d(loss)/d(logit) vector the fused
    loss tail produced (a logit addend's gradient, e.g. DeepFM's first-order logit -> its bias): the tail's partial rows
    already hold the per-workgroup sums.  False: g is something else — the caller sums it itself."""
    e = _dlogit_partials.get(g.data_ptr())
    if e is None or g.numel() != e[4] or not g.is_contiguous():
        return False
    _colsum_pending.append((e[0], e[1], e[2], e[3], 1, out))
    return True


def logit_loss_supported(parts, addends) -> bool:
    return (_loss_seed is not None and torch.is_grad_enabled() and dense1_supported(parts) and len(addends) <= 2
            and sum(int(t.shape[1]) for t in parts) <= 4096          # the kernel keeps an [8, C] tile + the weights in LDS
            and all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == 1 for t in addends))


def logit_loss(store, labels: torch.Tensor, heads, bias, parts, addends, loss_addend: Optional[torch.Tensor] = None):
    """-> (mean sigmoid-CE loss (+ loss_addend, a detached device scalar), probabilities [B,1], logit [B,1]) of
    logit = sum_heads dense1(parts) + bias + addends."""
    if loss_addend is not None:
        loss_addend = loss_addend.detach().reshape(1).to(torch.float32)
    return _LogitLossFn.apply(store.anchor, labels, heads, bias, len(addends), loss_addend, *parts, *addends)


# =============================================================================================
# a12: PReLU / Dice
# =============================================================================================
class _ActFn(Function):
    @staticmethod
    def forward(ctx, anchor, x, alpha: Variable, kind: int):
        x = x.contiguous()
        rows, C = x.shape
        y = torch.empty_like(x)
        _lib.check(_lib_().recalgo_activation_fwd(_p(x), _p(alpha.data), rows, C, kind, _p(y), _stream(x)),
                   "recalgo_activation_fwd")
        ctx.alpha, ctx.kind = alpha, kind
        ctx.in_step = _loss_seed is not None      # built inside Estimator.train_step: its optimizer runs the deferred sums
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        rows, C = x.shape
        lib = _lib_()
        gy = gy.contiguous()
        dx = torch.empty_like(x)
        nbytes = int(lib.recalgo_activation_bwd_workspace_bytes(rows, C))
        if ctx.in_step:
            # inside a training step: d(alpha) = column sum of the partial rows, a job of the step's deferred-sum launch
            key = ("act", x.device.type, x.device.index, rows, C, ctx.alpha.grad.data_ptr())
            ws = _dense_ws.get(key)
            if ws is None:
                ws = _dense_ws[key] = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
            _lib.check(lib.recalgo_activation_bwd(_p(x), _p(ctx.alpha.data), _p(gy), rows, C, ctx.kind, _p(dx), None, _p(ws),
                                                  _stream(x)), "recalgo_activation_bwd")
            _colsum_pending.append((ws.view(torch.float32), 0, int(lib.recalgo_activation_bwd_partial_rows(rows, C)), C, C,
                                    ctx.alpha.grad))
            return None, dx, None, None
        ws = _workspace(nbytes, x.device)
        _lib.check(lib.recalgo_activation_bwd(_p(x), _p(ctx.alpha.data), _p(gy), rows, C, ctx.kind, _p(dx),
                                              _p(ctx.alpha.grad), _p(ws), _stream(x)),
                   "recalgo_activation_bwd")
        return None, dx, None, None


def activation(store, x: torch.Tensor, alpha: Variable, kind: str) -> torch.Tensor:
    if store.building:
        return torch.zeros_like(x)
    return _ActFn.apply(store.anchor, x, alpha, _ACT[kind])


# =============================================================================================
# tf.layers.dropout (training mode)
# =============================================================================================
class DropSpec:
    """One training-mode tf.layers.dropout call: rate, and where its keep decisions come from — an explicit mask (parity
    tests replay the reference run's), or the counter-based hash of csrc/dropout.h keyed by (seed, call, device step counter)."""
    __slots__ = ("rate", "mask", "seed", "call", "step")

    def __init__(self, rate: float, mask: Optional[torch.Tensor], seed: int, call: int, step: Optional[torch.Tensor]):
        self.rate, self.mask, self.seed, self.call, self.step = float(rate), mask, int(seed) & 0x7FFFFFFF, int(call), step

    @property
    def scale(self) -> float:
        return 1.0 / (1.0 - self.rate)

    def check_mask(self, shape) -> None:
        if self.mask is not None and (tuple(self.mask.shape) != tuple(shape) or self.mask.dtype != torch.float32
                                      or not self.mask.is_contiguous()):
            raise ValueError("dropout: the explicit keep mask must be a contiguous fp32 tensor of the dropped tensor's shape")


class _CDrop(ctypes.Structure):              # include/recalgo.h recalgo_dropout_t
    _fields_ = [("rate", ctypes.c_double), ("keep_mask", ctypes.c_void_p), ("seed", ctypes.c_uint), ("call", ctypes.c_uint),
                ("step", ctypes.c_void_p)]


def _cdrop(d: Optional["DropSpec"]):
    """-> (byref argument or None, keep-alive)"""
    if d is None:
        return None, None
    c = _CDrop(d.rate, None if d.mask is None else d.mask.data_ptr(), d.seed, d.call, None if d.step is None else d.step.data_ptr())
    return ctypes.byref(c), c


def _dropout_launch(fn_name: str, x: torch.Tensor, d: DropSpec) -> torch.Tensor:
    y = torch.empty_like(x)
    _lib.check(getattr(_lib_(), fn_name)(_p(x), x.numel(), d.rate, _p(d.mask), d.seed, d.call, _p(d.step), _p(y), _stream(x)), fn_name)
    return y


class _DropoutFn(Function):
    @staticmethod
    def forward(ctx, x, d: DropSpec):
        ctx.d = d
        return _dropout_launch("recalgo_dropout_fwd", x.contiguous(), d)

    @staticmethod
    def backward(ctx, g):
        return _dropout_launch("recalgo_dropout_bwd", g.contiguous(), ctx.d), None


def dropout(x: torch.Tensor, d: DropSpec) -> torch.Tensor:
    """y = x * keep / (1 - rate) as its own launch each way (a dropout no neighbouring kernel can absorb)."""
    _chk(x, torch.float32, "x")
    if d.mask is not None and (d.mask.shape != x.shape or d.mask.dtype != torch.float32 or not d.mask.is_contiguous()):
        raise ValueError("dropout: the explicit keep mask must be a contiguous fp32 tensor of x's shape")
    return _DropoutFn.apply(x, d)


def dropout_keep_mask(shape, d: DropSpec, device) -> torch.Tensor:
    """The keep mask (1 / 0) the hash of `d` stands for at the CURRENT value of the step counter."""
    out = torch.empty(*shape, dtype=torch.float32, device=device)
    _lib.check(_lib_().recalgo_dropout_keep_mask(out.numel(), d.rate, d.seed, d.call, _p(d.step), _p(out), _stream(out)),
               "recalgo_dropout_keep_mask")
    return out


# =============================================================================================
# a15: TF1 Adam
# =============================================================================================
def adam_tf1_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int,
              lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
              zero_grad: bool = True, lr_t_dev: Optional[torch.Tensor] = None) -> None:
    """In-place dense TF1 Adam over flat fp32 buffers (step is 1-based).  With `lr_t_dev`
    (a 1-element device tensor maintained by adam_tf1_advance_) the launch is hipGraph
    replayable."""
    import math
    import struct
    # TF keeps lr/beta1/beta2 as float32 scalars: round them first (1 - f32(0.999) != 0.001)
    f32 = lambda x: struct.unpack("f", struct.pack("f", x))[0]
    lr_, b1_, b2_ = f32(lr), f32(beta1), f32(beta2)
    lr_t = 0.0 if lr_t_dev is not None else \
        lr_ * math.sqrt(1.0 - b2_ ** step) / (1.0 - b1_ ** step)
    _lib.check(_lib_().recalgo_adam_tf1_dense(
        _p(p), _p(g), _p(m), _p(v), p.numel(), lr_t, _p(lr_t_dev), beta1, beta2, eps,
        int(zero_grad), _stream(p)), "recalgo_adam_tf1_dense")


def adam_tf1_rows_(weight: torch.Tensor, grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor,
                   row_live: torch.Tensor, lr_t_dev: torch.Tensor, beta1: float = 0.9, beta2: float = 0.999,
                   eps: float = 1e-8, zero_grad: bool = True) -> None:
    """TF1 dense Adam over an embedding arena [rows, K] that skips rows no batch has touched yet
    (exactly the identity for them); row_live [rows] uint8 is maintained by the kernel."""
    rows, K = weight.shape
    _lib.check(_lib_().recalgo_adam_tf1_rows(
        _p(weight), _p(grad), _p(m), _p(v), _p(row_live), rows, K, 0.0, _p(lr_t_dev), beta1, beta2, eps,
        int(zero_grad), _stream(weight)), "recalgo_adam_tf1_rows")


def adam_tf1_list_(arena, lr_t_dev: torch.Tensor, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                   zero_grad: bool = True) -> None:
    """TF1 dense Adam applied to the arena's live rows only (identity elsewhere: bit-identical to
    the dense pass)."""
    _, lst, cnt = arena.live_state()
    rows, K = arena.weight.shape
    _lib.check(_lib_().recalgo_adam_tf1_list(
        _p(arena.weight), _p(arena.grad), _p(arena.m), _p(arena.v), _p(lst), _p(cnt), rows, K, 0.0, _p(lr_t_dev),
        beta1, beta2, eps, int(zero_grad), _stream(arena.weight)), "recalgo_adam_tf1_list")


def adam_tf1_advance_(step_dev: torch.Tensor, lr_t_dev: torch.Tensor, lr: float,
                      beta1: float = 0.9, beta2: float = 0.999) -> None:
    """step_dev (int64[1]) += 1; lr_t_dev (float[1]) = lr*sqrt(1-b2^t)/(1-b1^t), on device."""
    _lib.check(_lib_().recalgo_adam_tf1_advance(_p(step_dev), lr, beta1, beta2, _p(lr_t_dev),
                                                 _stream(step_dev)), "recalgo_adam_tf1_advance")


class _AdamArena(ctypes.Structure):          # include/recalgo.h recalgo_adam_arena_t
    _fields_ = [("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("live_list", ctypes.c_void_p), ("live_count", ctypes.c_void_p), ("max_rows", ctypes.c_int64),
                ("K", ctypes.c_int), ("lazy", ctypes.c_int)]


def adam_tf1_step_(flat, flat_grad, flat_m, flat_v, arenas, step_dev: torch.Tensor, ticket_dev: Optional[torch.Tensor],
                   lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, zero_grad: bool = True,
                   advance: bool = False, lazy: bool = False, plan_scans=()) -> None:
    """ONE launch: TF1 Adam over the flat dense buffer and over the live rows of every arena, lr_t derived on the device
    from step_dev (already advanced unless `advance`; include/recalgo.h recalgo_adam_tf1_step).  plan_scans: the
    recalgo_plan_scan_t records (sparse.plan_scan_record) of the scatter plans whose `apply` follows — their bucket-total
    prefix scans ride on this launch (recalgo_adam_tf1_step_plans)."""
    lib = _lib_()
    n = 0 if flat is None else flat.numel()
    for i in range(0, max(len(arenas), 1), 4):
        chunk = arenas[i:i + 4]
        arr = (_AdamArena * max(len(chunk), 1))()
        for j, a in enumerate(chunk):
            _, lst, cnt = a.live_state()
            rows, K = a.weight.shape
            arr[j] = _AdamArena(a.weight.data_ptr(), a.grad.data_ptr(), a.m.data_ptr(), a.v.data_ptr(), lst.data_ptr(),
                                cnt.data_ptr(), rows, K, int(lazy))
        first = i == 0
        if not first:
            raise NotImplementedError("more than 4 embedding arenas in one model")
        scans = None
        if plan_scans:
            scans = (type(plan_scans[0]) * len(plan_scans))(*plan_scans)
        _lib.check(lib.recalgo_adam_tf1_step_plans(_p(flat) if n else None, _p(flat_grad) if n else None, _p(flat_m) if n else None,
                                                   _p(flat_v) if n else None, n, arr, len(chunk), _p(step_dev), _p(ticket_dev),
                                                   int(advance), lr, beta1, beta2, eps, int(zero_grad), scans, len(plan_scans),
                                                   _stream(step_dev)), "recalgo_adam_tf1_step")
