"""Multi-GPU: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm).

The reference is single-process: everything here is new functionality
whose oracle is "N ranks == 1 rank on the same global batch" (tests/test_dist_gloo.py).

Partitioning
  * examples: data parallel, B_local per rank;
  * embedding arenas: ROW-sharded — global arena row r lives on rank r % N at local row r // N
    (modulo, not ranges: Zipf-hot rows are the low ids of every table and would pile onto rank 0);
    the Adam moments are sharded with the rows, so TF1's dense Adam is a local pass over the shard;
  * dense variables: replicated; their flat gradient buffer is all-reduced (one collective).

Per step and arena lookup (`ExchangePlan`)
  1. bucket the (valid) global rows by owner;  all_to_all of the bucket sizes, then of the local
     row numbers (int64; B_local*F*8 B per rank, 7/8 of it leaves the GPU);
  2. every owner gathers its rows from its shard (HIP gather kernel) and all_to_all's them back
     (B_local*F*K*4 B per rank);  the rows land in a *staged arena* [M, K] in request order;
  3. the unchanged single-GPU kernels (gather, fused DeepFM sparse path, bag mean, sequence
     gather) run on the staged arena with identity ids;
  4. backward: the kernels scatter into the staged gradient, which is all_to_all'ed back to the
     owners and scatter-added into the shard's gradient (HIP kernel).  This push runs on a second HIP stream
     (fork / join by events, capturable): the rest of the backward pass, the deferred weight-gradient sums and
     the dense all-reduce proceed on the main stream; the optimizer launch joins.
Two exchange plans implement steps 1-4:
  * `ExchangePlan`: exact bucket sizes (data-dependent all_to_all splits -> one small host sync per
    lookup; steps must be launched eagerly);
  * `StaticExchangePlan` (default under `attach_data_parallel`): every (source, owner) bucket has a
    fixed capacity of `capacity_factor` x the mean bucket size, unused slots carry id -1 / zero
    rows.  All shapes are static and nothing is read back to the host, so the whole N-GPU step is
    hipGraph-capturable; a device-side flag records a bucket overflow (the step's result is then
    invalid and the caller re-plans with a larger factor).
Both plans first DE-DUPLICATE the requests (`recalgo_dedup_rows`): a Zipf batch asks for each field's hottest row
~7 % of the time, and only the first request of a distinct row is bucketed — later requests read (forward) and
accumulate into (backward, summed locally by the scatter kernel) the same staged row, so a row crosses xGMI once
per rank and step in each direction and the owner sees one gradient row per requesting rank.  That also removes
the hot-row skew from the bucket sizes (distinct rows spread uniformly over r % N), so the static capacity is
1.5 x the mean instead of the 2 x a per-request bucketing needs.
On the 8-GPU xGMI full mesh every peer pair has its own link, so the all_to_all is link-parallel
(≈0.85 MB per link per direction at B_local=4096, F=26, K=16); the dense all-reduce (≈1.5 MB) is
latency-bound.  Each rank back-propagates loss_rank / N, so SUM collectives yield the gradient
of the global-batch mean loss.  BatchNorm uses per-replica statistics (the reference's BN is
single-device; documented deviation for N > 1).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .variables import EmbeddingArena


class ShardSpec:
    def __init__(self, rank: int, world: int, group=None, dist=None):
        self.rank, self.world, self.group = int(rank), int(world), group
        if dist is None:
            import torch.distributed as dist
        self.dist = dist

    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor, out_splits: List[int], in_splits: List[int]):
        self.dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)


class ExchangePlan:
    """Who asks whom for which rows: built once per (batch, arena row set), used for the forward
    fetch of any arena with that row layout and for the gradient push of the backward."""

    def __init__(self, rows: torch.Tensor, sh: ShardSpec, dedup=None):
        # rows: int64 [M] global arena rows, -1 = OOV / padding (nothing is exchanged for those)
        self.sh, self.M = sh, rows.numel()
        W = sh.world
        self.rep = None
        if dedup is not None:
            # only the first request of every distinct row travels; the others read its staged row
            rows, self.rep = dedup(rows.reshape(-1))
        pos = torch.nonzero(rows >= 0).squeeze(1)
        r = rows[pos]
        owner = r % W
        order = torch.argsort(owner, stable=True)
        self.send_pos = pos[order]                       # request slot of every sent row, bucket order
        send_local = (r // W)[order].contiguous()
        send_counts = torch.bincount(owner, minlength=W)
        recv_counts = torch.empty_like(send_counts)
        sh.dist.all_to_all_single(recv_counts, send_counts, group=sh.group)
        self.sc = send_counts.tolist()                   # the one host sync of the exchange
        self.rc = recv_counts.tolist()
        self.recv_local = torch.empty(sum(self.rc), dtype=torch.int64, device=rows.device)
        sh.all_to_all(self.recv_local, send_local, self.rc, self.sc)

    def fetch(self, shard_weight: torch.Tensor, local_gather) -> torch.Tensor:
        """-> [M, K] rows in request order (zero rows where the request was -1)."""
        K = shard_weight.shape[1]
        rows_out = local_gather(shard_weight, self.recv_local)             # [sum(rc), K]
        back = torch.empty(sum(self.sc), K, dtype=shard_weight.dtype, device=shard_weight.device)
        self.sh.all_to_all(back, rows_out, [c for c in self.sc], [c for c in self.rc])
        out = torch.zeros(self.M, K, dtype=shard_weight.dtype, device=shard_weight.device)
        out.index_copy_(0, self.send_pos, back)
        return out

    def staged_ids(self, rows: torch.Tensor, shape) -> torch.Tensor:
        if self.rep is None:
            return identity_ids(rows, shape)       # the staged table is in request order
        flat = rows.reshape(-1)
        return torch.where(flat >= 0, self.rep, torch.full_like(self.rep, -1)).reshape(shape)

    def push_grad(self, staged_grad: torch.Tensor, arena, local_scatter_add, owner_src=None) -> None:
        """arena.grad[owner rows] += staged_grad rows (duplicate requests were already summed into their
        representative's staged row by the local scatter; without dedup they accumulate on the owner).  `owner_src`
        (sparse.Source registered by the forward): the received rows become that lookup's gradient instead — summed per
        row and applied by the shard's owner-computes optimizer launch."""
        K = staged_grad.shape[1]
        gsend = staged_grad.index_select(0, self.send_pos)
        grecv = torch.empty(sum(self.rc), K, dtype=staged_grad.dtype, device=staged_grad.device)
        self.sh.all_to_all(grecv, gsend, [c for c in self.rc], [c for c in self.sc])
        if owner_src is not None:
            owner_src.set_grad(grecv)
        else:
            local_scatter_add(arena, self.recv_local, grecv)


class StaticExchangePlan:
    """Fixed-capacity variant of ExchangePlan: same interface, static shapes, no host sync.  The
    bucketing is one HIP pass (`recalgo_exchange_plan`); the staged table IS the all_to_all receive
    buffer [world*cap, K] and the kernels address it through `req_slot`, so neither direction needs
    an unpack copy: forward = owner gather -> all_to_all; backward = all_to_all -> owner scatter-add."""

    def __init__(self, rows: torch.Tensor, sh: ShardSpec, capacity: int, overflow: torch.Tensor, planner, dedup=None):
        self.sh, self.M, self.cap = sh, rows.numel(), int(capacity)
        rows = rows.reshape(-1)
        if dedup is None:
            self.send_local, self.req_slot = planner(rows, sh.world, self.cap, overflow)
        else:
            # bucket the distinct rows only: a duplicate request shares the bucket entry of the row's first request
            unique_rows, rep = dedup(rows)
            self.send_local, slot = planner(unique_rows, sh.world, self.cap, overflow)
            self.req_slot = slot.index_select(0, rep)
        self.recv_local = torch.empty_like(self.send_local)
        sh.dist.all_to_all_single(self.recv_local, self.send_local, group=sh.group)

    def staged_ids(self, rows: torch.Tensor, shape) -> torch.Tensor:
        return self.req_slot.reshape(shape)

    def fetch(self, shard_weight: torch.Tensor, local_gather) -> torch.Tensor:
        rows_out = local_gather(shard_weight, self.recv_local)                   # id -1 -> zero row
        back = torch.empty_like(rows_out)
        self.sh.dist.all_to_all_single(back, rows_out, group=self.sh.group)
        return back

    def push_grad(self, staged_grad: torch.Tensor, arena, local_scatter_add, owner_src=None) -> None:
        grecv = torch.empty_like(staged_grad)
        self.sh.dist.all_to_all_single(grecv, staged_grad, group=self.sh.group)
        if owner_src is not None:
            owner_src.set_grad(grecv)                                            # (see ExchangePlan.push_grad)
        else:
            local_scatter_add(arena, self.recv_local, grecv)                     # id -1 is skipped


# ---- the two local kernels of the exchange (HIP; tests substitute CPU doubles) ------------------
_zero_base = {}


def _zero(device) -> torch.Tensor:
    """int64 [1] = 0 (the row_base of a one-table lookup), one per device, made outside graph capture."""
    z = _zero_base.get(device)
    if z is None:
        z = _zero_base[device] = torch.zeros(1, dtype=torch.int64, device=device)
    return z


def hip_exchange_plan(rows: torch.Tensor, world: int, cap: int, overflow: torch.Tensor):
    """-> (send_local int64 [world*cap], req_slot int64 [M]) by include/recalgo.h recalgo_exchange_plan."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    dev, M = rows.device, rows.numel()
    send_local = torch.empty(world * cap, dtype=torch.int64, device=dev)
    req_slot = torch.empty(M, dtype=torch.int64, device=dev)
    counters = torch.empty(world, dtype=torch.int32, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(lib.recalgo_exchange_plan(p(rows.contiguous()), M, world, cap, p(send_local), None, p(req_slot), p(counters),
                                         p(overflow), st), "recalgo_exchange_plan")
    return send_local, req_slot


def hip_dedup_rows(rows: torch.Tensor):
    """-> (unique_rows int64 [M], rep int64 [M]) by include/recalgo.h recalgo_dedup_rows."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    rows = rows.contiguous()
    dev, M = rows.device, rows.numel()
    unique_rows, rep = torch.empty_like(rows), torch.empty_like(rows)
    if M:
        ws = torch.empty(int(lib.recalgo_dedup_rows_workspace_bytes(M)), dtype=torch.uint8, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        _lib.check(lib.recalgo_dedup_rows(p(rows), M, p(unique_rows), p(rep), p(ws), st), "recalgo_dedup_rows")
    return unique_rows, rep


def hip_local_gather(shard_weight: torch.Tensor, local_rows: torch.Tensor, deferred=None, step_dev=None) -> torch.Tensor:
    """Owner-side gather of the requested rows (id -1 -> zero row).  `deferred` / `step_dev` (sparse.deferred_view): rows
    whose deferred-Adam state lags are read as of the current step."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    n, K = local_rows.numel(), shard_weight.shape[1]
    out = torch.empty(n, K, dtype=torch.float32, device=shard_weight.device)
    if n:
        zero = _zero(shard_weight.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(shard_weight.device).cuda_stream)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        _lib.check(lib.recalgo_embedding_gather_fwd_deferred(p(local_rows), p(shard_weight), p(zero), n, 1, K, p(out), K, 0,
                                                             deferred, step_dev, 0, st), "recalgo_embedding_gather_fwd")
    return out


def hip_local_scatter_add(arena: EmbeddingArena, local_rows: torch.Tensor, g: torch.Tensor) -> None:
    """arena.grad[local_rows] += g on the owner (id -1 skipped), and the rows join the live-row list."""
    import ctypes
    from . import _lib, ops
    lib = _lib.load()
    shard_grad = arena.grad
    n, K = local_rows.numel(), shard_grad.shape[1]
    if n:
        zero = _zero(shard_grad.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(shard_grad.device).cuda_stream)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        g = g.contiguous()
        # the scatter marks the rows it flushes in the shard's live-row list
        _lib.check(lib.recalgo_embedding_gather_bwd(p(local_rows), p(g), p(zero), n, 1, K, K, 0, p(shard_grad),
                                                    ops._live(arena), st), "recalgo_embedding_gather_bwd")


class Sharding:
    """Attached to an EmbeddingArena (`arena.sharding`): the arena then holds only the rows
    r % world == rank (at local index r // world)."""

    def __init__(self, sh: ShardSpec, global_rows: int, local_gather=hip_local_gather,
                 local_scatter_add=hip_local_scatter_add, capacity_factor: Optional[float] = None,
                 planner=hip_exchange_plan, dedup=hip_dedup_rows):
        self.sh, self.global_rows = sh, int(global_rows)
        self.local_gather, self.local_scatter_add, self.planner = local_gather, local_scatter_add, planner
        self.dedup = dedup                              # None: every request travels
        self.capacity_factor = capacity_factor          # None: exact (dynamic) buckets
        self.overflow: Optional[torch.Tensor] = None    # device flag, sticky

    def capacity(self, M: int) -> int:
        W = self.sh.world
        if W == 1:
            return max(M, 1)
        c = int(-(-M * self.capacity_factor // W))
        return min(max((c + 7) // 8 * 8, 8), max(M, 1))

    def plan(self, rows: torch.Tensor):
        if self.capacity_factor is None:
            return ExchangePlan(rows, self.sh, self.dedup)
        if self.overflow is None:
            self.overflow = torch.zeros(1, dtype=torch.bool, device=rows.device)
        return StaticExchangePlan(rows, self.sh, self.capacity(rows.numel()), self.overflow, self.planner, self.dedup)


class StagedArena:
    """The rows one batch needs, fetched from their owners into a local table (request order for the
    exact plan, bucket order for the static plan: `plan.staged_ids` maps each request to its staged
    row).  Quacks like an EmbeddingArena for the single-GPU kernels; the gradient
    they scatter into `.grad` is pushed back to the owners by `flush_grad()` (called by the op's
    backward right after its kernel)."""

    def __init__(self, plan: ExchangePlan, arena: EmbeddingArena, store=None, training: bool = False):
        self.plan, self.arena, self.K = plan, arena, arena.K
        self.name = arena.name + "/staged"
        sd: Sharding = arena.sharding
        gather = sd.local_gather
        # OWNER side on the owner-computes path (sparse.py): the rows the peers ask this rank for are one more lookup
        # of the shard's request plan — [n, 1] local rows, -1 = unused bucket slot — caught up (deferred Adam) before the
        # owner gather reads them; the gradient rows that come back in the backward become that lookup's gradient, summed
        # per row in a fixed order and applied by the shard's optimizer launch (no float atomics on the owner either)
        self.owner_src = None
        if gather is hip_local_gather and store is not None:
            from . import sparse
            recv = plan.recv_local
            if training:
                self.owner_src = sparse.begin_lookup(arena, store, recv.reshape(-1, 1), None, None, 0, recv.numel(), 1, True)
            dv, stp = sparse.view_for(self.owner_src, arena, store)
            if dv is not None:                          # lagging rows are read as of the current step
                gather = lambda w, rows, _dv=dv, _stp=stp: hip_local_gather(w, rows, _dv, _stp)
        self.weight = plan.fetch(arena.weight, gather)
        self.tables = {"__staged__": (0, self.weight.shape[0])}
        self._grad: Optional[torch.Tensor] = None

    @property
    def grad(self) -> torch.Tensor:
        if self._grad is None:
            self._grad = torch.zeros_like(self.weight)
        return self._grad

    def table_view(self, _name):
        return self.weight

    def flush_grad(self):
        if self._grad is not None:
            sd: Sharding = self.arena.sharding
            g, self._grad = self._grad, None
            if g.is_cuda and push_overlap():
                # the push (all_to_all of the staged gradient + owner scatter-add) leaves the main stream: the rest of
                # the backward pass, the deferred weight-gradient sums and the dense all-reduce run beside it; the
                # optimizer joins (join_push_streams, called by the gradient hook after the all-reduce)
                dev = g.device
                side = _push_stream(dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    self.plan.push_grad(g, self.arena, sd.local_scatter_add, self.owner_src)
                _push_pending.append((dev, g))          # keeps the staged gradient alive until the join
            else:
                self.plan.push_grad(g, self.arena, sd.local_scatter_add, self.owner_src)


_push_streams = {}
_push_pending = []


def push_overlap() -> bool:
    """RECALGO_DP_OVERLAP=0 keeps the gradient push on the main stream (bring-up / bisecting aid)."""
    import os
    return os.environ.get("RECALGO_DP_OVERLAP", "1") != "0"


def _push_stream(device) -> "torch.cuda.Stream":
    key = (device.type, device.index)
    st = _push_streams.get(key)
    if st is None:
        st = _push_streams[key] = torch.cuda.Stream(device=device)
    return st


def join_push_streams() -> None:
    """The current stream waits for every gradient push issued since the last join (arena.grad and the live-row
    lists of the shards are complete after this)."""
    for dev in {d for d, _ in _push_pending}:
        torch.cuda.current_stream(dev).wait_stream(_push_stream(dev))
    _push_pending.clear()


def global_rows(ids: torch.Tensor, row_base: torch.Tensor) -> torch.Tensor:
    """ids [B, F] (id < 0 = OOV) + row_base [F] -> global arena rows [B*F], -1 where OOV."""
    rows = ids + row_base.unsqueeze(0)
    return torch.where(ids >= 0, rows, torch.full_like(rows, -1)).reshape(-1)


def identity_ids(rows: torch.Tensor, shape) -> torch.Tensor:
    """ids into the staged arena: slot number where a row was requested, -1 where it was not."""
    ar = torch.arange(rows.numel(), dtype=torch.int64, device=rows.device)
    return torch.where(rows >= 0, ar, torch.full_like(ar, -1)).reshape(shape)


def shard_arena_(arena: EmbeddingArena, sh: ShardSpec, local_gather=hip_local_gather,
                 local_scatter_add=hip_local_scatter_add, capacity_factor: Optional[float] = None,
                 planner=hip_exchange_plan, dedup=hip_dedup_rows) -> None:
    """Re-shard a fully materialised (replicated-at-init) arena in place: keep rows r % N == rank.
    Every rank must have built the same arena (same seed) — that is what makes N ranks == 1 rank.
    (The production order is attach_data_parallel BEFORE the build, which never materialises the whole arena.)"""
    if getattr(arena, "sharding", None) is not None:
        return
    rows = arena.weight.shape[0]
    take = lambda t: t[sh.rank::sh.world].contiguous().clone()
    arena.weight, arena.grad, arena.m, arena.v = take(arena.weight), take(arena.grad), take(arena.m), take(arena.v)
    arena.live = None           # live-row bookkeeping is rebuilt for the shard on next use
    arena.__dict__.pop("sparse", None)   # (owner-computes plan of the unsharded arena: callers sync it before re-sharding)
    arena.sharding = Sharding(sh, rows, local_gather, local_scatter_add, capacity_factor, planner, dedup)


def unshard_arena(arena: EmbeddingArena, what: str = "weight") -> torch.Tensor:
    """all_gather the shards back into the [global_rows, K] tensor (tests, checkpoints)."""
    sd: Sharding = arena.sharding
    sh = sd.sh
    if what in ("weight", "m", "v"):
        from . import sparse
        sparse.sync_arena(arena)             # deferred Adam: every row reflects all completed optimizer steps
    local = getattr(arena, what)
    n_max = (sd.global_rows + sh.world - 1) // sh.world
    pad = torch.zeros(n_max, local.shape[1], dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(sh.world)]
    sh.dist.all_gather(parts, pad, group=sh.group)
    full = torch.empty(sd.global_rows, local.shape[1], dtype=local.dtype, device=local.device)
    for r, part in enumerate(parts):
        n = (sd.global_rows - r + sh.world - 1) // sh.world
        full[r::sh.world] = part[:n]
    return full


def gather_arena_to_host(arena: EmbeddingArena, what: str = "weight", dst_rank: int = 0,
                         window_rows: int = 1 << 20) -> Optional[torch.Tensor]:
    """COLLECTIVE.  The [global_rows, K] tensor of a row-sharded arena assembled in HOST memory on `dst_rank` only (None on
    the other ranks), through a bounded device window: per round every rank contributes at most `window_rows` of its local
    rows (64 MB at K = 16), the rounds are all_gather'ed (the one collective every backend in use here offers) and only the
    writer copies them out.  No rank ever holds a whole table on its GPU — the point of sharding at construction; the
    all_gather of whole padded shards (unshard_arena) needs ~2x the table per GPU and the same again per rank on the host."""
    sd: Sharding = arena.sharding
    sh = sd.sh
    if what in ("weight", "m", "v"):
        from . import sparse
        sparse.sync_arena(arena)             # deferred Adam: every row reflects all completed optimizer steps
    local = getattr(arena, what)
    K = local.shape[1]
    n_max = (sd.global_rows + sh.world - 1) // sh.world
    writer = sh.rank == dst_rank
    full = torch.empty(sd.global_rows, K, dtype=local.dtype) if writer else None
    for r0 in range(0, n_max, window_rows):
        n = min(window_rows, n_max - r0)
        piece = torch.zeros(n, K, dtype=local.dtype, device=local.device)
        have = max(0, min(n, local.shape[0] - r0))
        if have:
            piece[:have] = local[r0:r0 + have]
        parts = [torch.empty_like(piece) for _ in range(sh.world)]
        sh.dist.all_gather(parts, piece, group=sh.group)
        if writer:
            for r, part in enumerate(parts):
                n_r = (sd.global_rows - r + sh.world - 1) // sh.world          # local rows of rank r
                cnt = max(0, min(n, n_r - r0))
                if cnt:
                    # local row l of rank r is global row l * world + r
                    full[(r0 * sh.world + r)::sh.world][:cnt] = part[:cnt].cpu()
        del parts, piece
    return full


class _ShardAtBuild:
    """Carried by the VariableStore from attach_data_parallel to the end of the build: every arena is materialised
    as this rank's rows only (EmbeddingArena.materialize(shard=...)) and gets its Sharding."""

    def __init__(self, sh: ShardSpec, **sharding_kw):
        self.sh, self.kw = sh, sharding_kw

    def attach(self, arena: EmbeddingArena) -> None:
        arena.live = None
        arena.sharding = Sharding(self.sh, arena.rows, **self.kw)


def attach_data_parallel(est, dist=None, group=None, local_gather=hip_local_gather,
                         local_scatter_add=hip_local_scatter_add, capacity_factor: Optional[float] = 1.5,
                         planner=hip_exchange_plan, dedup=hip_dedup_rows, sync_batch_norm: bool = False):
    """Make an Estimator one rank of an N-rank job: shard every embedding arena row-wise, all-reduce the flat dense
    gradient before the optimizer, scale the loss gradient by 1/N.  Call it BEFORE the first build / train call:
    the arenas are then created sharded (no rank ever holds a whole table).  Called on an already built Estimator
    it re-shards the replicated arenas in place (every rank must have built them from the same seed).
    `capacity_factor` selects the static (graph-capturable) exchange — buckets of capacity_factor x the mean number
    of DISTINCT-row requests per owner; None = exact dynamic buckets.  `dedup` = None sends every request.
    `sync_batch_norm`: training-mode BatchNorm normalises with the statistics of the GLOBAL batch (the per-tile moment
    partials of all ranks are all-gathered between the two BatchNorm launches, forward and backward) — N ranks then equal
    the reference's single-device BatchNorm on the concatenated batch; the default keeps per-replica statistics."""
    if dist is None:
        import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sh = ShardSpec(rank, world, group, dist)
    if sync_batch_norm and world > 1:
        def _all_gather(out: torch.Tensor, inp: torch.Tensor):          # out [world, n] <- inp [n] of every rank
            if hasattr(dist, "all_gather_into_tensor"):
                dist.all_gather_into_tensor(out.view(-1), inp, group=group)
            else:
                parts = [out[r] for r in range(world)]
                dist.all_gather(parts, inp, group=group)
        est.store.sync_bn = (world, rank, _all_gather)
    kw = dict(local_gather=local_gather, local_scatter_add=local_scatter_add, capacity_factor=capacity_factor,
              planner=planner, dedup=dedup)

    def sync_dense():
        # replicated dense variables must start identical
        if est.store.flat is not None and est.store.flat.numel():
            dist.broadcast(est.store.flat, src=0, group=group)

    if est._built:
        sync_dense()
        from . import sparse
        sparse.sync_store(est.store)          # deferred Adam of the unsharded arenas: finish it before slicing w / m / v
        for ar in est.store.arenas.values():
            shard_arena_(ar, sh, **kw)
    else:
        est.store.shard_at_build = _ShardAtBuild(sh, **kw)
        est._after_build.append(sync_dense)

    def grad_hook(store):
        if store.flat_grad is not None and store.flat_grad.numel():
            dist.all_reduce(store.flat_grad, op=dist.ReduceOp.SUM, group=group)
        join_push_streams()          # the gradient pushes ran beside the tail of the backward pass and the all-reduce
    est.grad_hook = grad_hook
    est.loss_grad_scale = 1.0 / world
    est.shard_spec = sh
    return est


def exchange_overflowed(est) -> bool:
    """True if any static exchange bucket overflowed ON ANY RANK since attach.  COLLECTIVE (every rank must call it at
    the same point): the per-rank device flags are max-reduced over the group first, so that all ranks see the same
    answer and raise — or carry on — together; a rank that raised alone would leave its peers blocked in their next
    all_to_all / all_gather until the RCCL timeout."""
    flags = [a.sharding.overflow for a in est.store.arenas.values()
             if getattr(a, "sharding", None) is not None and a.sharding.overflow is not None]
    if not flags:
        return False
    sh = next((getattr(a.sharding, "sh", None) for a in est.store.arenas.values() if getattr(a, "sharding", None) is not None), None)
    any_flag = torch.stack([f.reshape(-1)[0].to(torch.int32) for f in flags]).max().reshape(1)
    if sh is not None and sh.world > 1:
        sh.dist.all_reduce(any_flag, op=sh.dist.ReduceOp.MAX, group=sh.group)
    return bool(any_flag.item())


class HostStagedCollectives:
    """`torch.distributed` look-alike whose collectives bounce device tensors through host memory
    (gloo underneath).  Test and bring-up aid only: it lets N ranks share ONE GPU — RCCL refuses
    two ranks on a device — so that the N > 1 data path runs on the real HIP kernels on a 1-GPU box
    (tests/test_gpu_dist.py).  Not graph-capturable (every call synchronises), never the product path."""

    def __init__(self, dist):
        self._d = dist
        self.ReduceOp = dist.ReduceOp

    def get_world_size(self, group=None):
        return self._d.get_world_size(group)

    def get_rank(self, group=None):
        return self._d.get_rank(group)

    def barrier(self, group=None):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self._d.barrier(group=group)

    def broadcast(self, t, src, group=None):
        h = t.detach().cpu()
        self._d.broadcast(h, src=src, group=group)
        t.copy_(h)

    def all_reduce(self, t, op=None, group=None):
        h = t.detach().cpu()
        self._d.all_reduce(h, op=op if op is not None else self.ReduceOp.SUM, group=group)
        t.copy_(h)

    def all_to_all_single(self, out, inp, out_splits=None, in_splits=None, group=None):
        hin, hout = inp.detach().cpu().contiguous(), torch.empty(out.shape, dtype=out.dtype)
        self._d.all_to_all_single(hout, hin, out_splits, in_splits, group=group)
        out.copy_(hout)

    def all_gather(self, parts, t, group=None):
        hs = [p.detach().cpu() for p in parts]
        self._d.all_gather(hs, t.detach().cpu(), group=group)
        for p, h in zip(parts, hs):
            p.copy_(h)
