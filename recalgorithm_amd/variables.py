"""Parameter store that plays the role of TF1 variable scopes for the mirrored model_fns.

The reference creates parameters implicitly by name (`tf.get_variable`,
algorithm/DCN/cross_layer.py:18-19; `tf.layers.dense` auto-names `dense`, `dense_1`, ...)
inside `tf.variable_scope`s.  Here a `VariableStore` keyed by the same names owns them.

Dense parameters are packed into ONE flat fp32 buffer (and one flat gradient buffer, one m,
one v): the TF1-Adam update is a single kernel launch and the data-parallel all-reduce is a
single collective over the flat gradient.  Every parameter-owning op writes
its parameter gradient straight into its slice of the flat gradient buffer (no autograd
accumulation kernels); only activation gradients flow through torch.autograd.

Embedding tables live in `EmbeddingArena`s (see include/recalgo.h "arena").
"""
from __future__ import annotations

import os

import contextlib
import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch


class Variable:
    """A named fp32 parameter: `.data` and `.grad` are views into the store's flat buffers
    once the store is packed."""

    __slots__ = ("name", "data", "grad", "trainable")

    def __init__(self, name: str, data: torch.Tensor, trainable: bool = True):
        self.name = name
        self.data = data
        self.grad = torch.zeros_like(data) if trainable else None
        self.trainable = trainable

    @property
    def shape(self):
        return tuple(self.data.shape)

    def __repr__(self):
        return f"Variable({self.name}, shape={self.shape})"


# ---- initialisers -------------------------------------------------
def glorot_uniform(shape: Sequence[int], gen: torch.Generator) -> torch.Tensor:
    """TF default for tf.get_variable / tf.layers.dense kernels."""
    if len(shape) < 1:
        fan_in = fan_out = 1
    elif len(shape) == 1:
        fan_in = fan_out = shape[0]
    elif len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = 1
        for s in shape[:-2]:
            rf *= s
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(*shape, generator=gen) * 2 - 1) * limit


def truncated_normal(shape: Sequence[int], stddev: float, gen: torch.Generator) -> torch.Tensor:
    """TF embedding_column default: truncated_normal(0, 1/sqrt(dim)), resampling |z| > 2."""
    out = torch.randn(*shape, generator=gen)
    for _ in range(8):
        bad = out.abs() > 2
        if not bad.any():
            break
        out = torch.where(bad, torch.randn(*shape, generator=gen), out)
    return out.clamp(-2, 2) * stddev


def zeros(shape, gen=None):
    return torch.zeros(*shape)


def ones(shape, gen=None):
    return torch.ones(*shape)


class VariableStore:
    def __init__(self, device: torch.device | str = "cpu", seed: int = 42):
        self.device = torch.device(device)
        self.vars: Dict[str, Variable] = {}
        self.order: List[str] = []
        self._scope: List[str] = []
        self._auto: Dict[str, int] = {}
        self._alias: Dict[str, Tuple[str, int]] = {}   # sub-variable -> (block, index)
        self._gen = torch.Generator().manual_seed(seed)
        self.packed = False
        self.flat = self.flat_grad = self.flat_m = self.flat_v = None
        # autograd anchor: makes parameter-owning ops differentiable even when none of their
        # tensor inputs requires grad (their parameter grads are written as side effects)
        self.anchor = torch.zeros(1, device=self.device, requires_grad=True)
        self.anchor._recalgo_store = self         # (ops.anchor_store: lookups find the optimizer state through the anchor)
        self.arenas: Dict[str, "EmbeddingArena"] = {}
        self.seed = seed
        # building == True: a dry model_fn pass that only registers variables / tables
        # (input layers return zeros and launch nothing); see Estimator._build
        self.building = False
        self.shared_tables: Dict[tuple, str] = {}
        self.opt_state = None            # device-side Adam step counter / lr_t (estimator.py)
        self._drop_calls = 0             # training-mode dropout calls of the current model_fn invocation (nn.dropout)
        self._rb_cache: Dict[tuple, torch.Tensor] = {}

    def row_base_tensor(self, arena: "EmbeddingArena", table_names: Sequence[str]) -> torch.Tensor:
        """int64 device tensor of the first arena row of each named table (cached)."""
        key = (arena.name, tuple(table_names))
        t = self._rb_cache.get(key)
        if t is None:
            t = torch.tensor([arena.tables[n][0] for n in table_names], dtype=torch.int64,
                             device=self.device)
            self._rb_cache[key] = t
        return t

    def finalize(self):
        """End of the building pass: allocate the arenas and pack the dense variables."""
        plan = getattr(self, "shard_at_build", None)     # parallel.attach_data_parallel before the build
        for ar in self.arenas.values():
            ar.materialize(shard=None if plan is None else (plan.sh.rank, plan.sh.world))
            if plan is not None:
                plan.attach(ar)
        self.pack()
        self.building = False

    # -- scopes -------------------------------------------------------------------------
    @contextlib.contextmanager
    def variable_scope(self, name: str):
        self._scope.append(name)
        try:
            yield
        finally:
            self._scope.pop()

    def begin_call(self):
        """A model_fn invocation == a fresh TF graph: auto-naming counters restart."""
        self._auto.clear()
        self._scope.clear()
        self._drop_calls = 0
        from . import sparse
        sparse.new_forward(self)

    def ensure_opt_state(self) -> dict:
        """The optimizer's device-side state {step int64[1], lr_t float[1]}: created on first use (the optimizer's first
        apply_gradients, or a training-mode dropout — its keep masks are keyed by the step counter)."""
        if self.opt_state is None:
            self.opt_state = {"step": torch.zeros(1, dtype=torch.int64, device=self.device),
                              "lr_t": torch.zeros(1, dtype=torch.float32, device=self.device)}
        return self.opt_state

    def scope_name(self) -> str:
        return "/".join(self._scope)

    def full_name(self, name: str) -> str:
        return "/".join(self._scope + [name])

    def auto_name(self, base: str) -> str:
        """tf.layers auto naming inside the current scope: base, base_1, base_2, ..."""
        key = self.full_name(base)
        n = self._auto.get(key, 0)
        self._auto[key] = n + 1
        return base if n == 0 else f"{base}_{n}"

    # -- variables ------------------------------------------------------------------------
    def get_variable(self, name: str, shape: Sequence[int],
                     initializer: Optional[Callable] = None, trainable: bool = True) -> Variable:
        full = self.full_name(name)
        v = self.vars.get(full)
        if v is not None:
            if tuple(v.shape) != tuple(int(s) for s in shape):
                raise ValueError(f"variable {full}: shape {v.shape} != requested {tuple(shape)}")
            return v
        if self.packed:
            raise RuntimeError(f"variable {full} requested after the store was packed")
        shape = [int(s) for s in shape]
        init = initializer or glorot_uniform
        data = init(shape, self._gen).to(torch.float32).to(self.device).contiguous()
        v = Variable(full, data, trainable)
        self.vars[full] = v
        self.order.append(full)
        return v

    def get_variable_block(self, block_name: str, names: Sequence[str], shape: Sequence[int],
                           initializer: Optional[Callable] = None) -> Tuple[Variable, List[Variable]]:
        """Allocate len(names) same-shaped variables contiguously ([n, *shape]) so that a fused
        kernel can treat them as one tensor; the individual names remain addressable."""
        full = self.full_name(block_name)
        if full in self.vars:
            return self.vars[full], [self.vars[self.full_name(n)] for n in names]
        init = initializer or glorot_uniform
        shape = [int(s) for s in shape]
        parts = [init(shape, self._gen).to(torch.float32) for _ in names]
        data = torch.stack(parts, 0).to(self.device).contiguous()
        blk = Variable(full, data, True)
        self.vars[full] = blk
        self.order.append(full)
        subs = []
        for i, n in enumerate(names):
            sv = Variable.__new__(Variable)
            sv.name, sv.data, sv.grad, sv.trainable = self.full_name(n), blk.data[i], blk.grad[i], True
            self.vars[sv.name] = sv          # alias, not in self.order (owned by the block)
            self._alias[sv.name] = (full, i)
            subs.append(sv)
        return blk, subs

    # -- packing --------------------------------------------------------------------------
    def sync(self) -> None:
        """Every embedding arena reflects ALL completed optimizer steps.  The tables' TF1 Adam is evaluated lazily (sparse.py:
        a row's g = 0 updates are replayed when the row is next read, swept or flushed), so between steps `arena.weight / m / v`
        lag for rows no batch has touched for a while.  `named_arrays`, checkpoints, export and `unshard_arena` call this
        themselves; any OTHER direct reader of `arena.weight`, `arena.table_view(...)`, `arena.m`, `arena.v` must call it first."""
        from . import sparse
        sparse.sync_store(self)

    def housekeeping(self) -> None:
        """Cheap, sync-free maintenance the training loops call every few dozen steps."""
        for ar in self.arenas.values():
            if getattr(ar, "tracks_live_rows", False):
                ar.order_live_list()

    def pack(self):
        """Move every dense variable into one flat buffer (16-byte aligned slices)."""
        if self.packed:
            return
        owned = [self.vars[n] for n in self.order]
        offs, total = [], 0
        for v in owned:
            offs.append(total)
            total += (v.data.numel() + 3) // 4 * 4
        self.flat = torch.zeros(total, device=self.device)
        self.flat_grad = torch.zeros(total, device=self.device)
        self.flat_m = torch.zeros(total, device=self.device)
        self.flat_v = torch.zeros(total, device=self.device)
        self.trainable_mask = torch.ones(total, device=self.device)
        for v, o in zip(owned, offs):
            n = v.data.numel()
            old = v.data
            self.flat[o:o + n].copy_(old.reshape(-1))
            new = self.flat[o:o + n].view(old.shape)
            newg = self.flat_grad[o:o + n].view(old.shape)
            if not v.trainable:
                self.trainable_mask[o:o + n] = 0
            v.data, v.grad = new, newg
        for sub, (blk, idx) in self._alias.items():
            self.vars[sub].data = self.vars[blk].data[idx]
            self.vars[sub].grad = self.vars[blk].grad[idx]
        self.packed = True

    def named_arrays(self, gather: bool = False) -> Dict[str, torch.Tensor]:
        """name -> tensor of every variable and embedding table.  Tables of a row-sharded arena are not local views:
        with `gather` they are all_gather'ed (a COLLECTIVE: every rank must call), without it they raise."""
        from . import sparse
        sparse.sync_store(self)               # deferred Adam: whole tables are about to be read
        out = {n: self.vars[n].data for n in self.vars}
        for ar in self.arenas.values():
            if gather and getattr(ar, "sharding", None) is not None:
                from . import parallel
                full = parallel.unshard_arena(ar, "weight")
                for tn, (rb, vocab) in ar.tables.items():
                    out[tn] = ar.shaped(tn, full[rb:rb + vocab])
                continue
            for tn in ar.tables:
                out[tn] = ar.table_view(tn)
        return out


def named_grads(store: VariableStore) -> Dict[str, torch.Tensor]:
    """name -> gradient tensor, same keys as VariableStore.named_arrays()."""
    from . import nn, parallel, sparse
    nn.apply_parked_grads()
    parallel.join_push_streams()
    sparse.materialize_grads(store)
    out = {n: v.grad for n, v in store.vars.items() if v.grad is not None}
    for ar in store.arenas.values():
        for tn, (rb, vocab) in ar.tables.items():
            out[tn] = ar.shaped(tn, ar.grad[rb:rb + vocab])
    return out


_STORE_STACK: List[VariableStore] = []


def current_store() -> VariableStore:
    """The store model-building functions create their variables in (TF's default graph +
    variable scope stack).  Set by `use_store` (the Estimator does this around model_fn)."""
    if not _STORE_STACK:
        raise RuntimeError("no active VariableStore: wrap the call in `with use_store(store):`")
    return _STORE_STACK[-1]


@contextlib.contextmanager
def use_store(store: VariableStore):
    _STORE_STACK.append(store)
    try:
        yield store
    finally:
        _STORE_STACK.pop()


def variable_scope(name: str):
    """tf.variable_scope(name) on the current store."""
    return current_store().variable_scope(name)


def get_variable(name: str, shape, initializer=None, trainable: bool = True) -> Variable:
    """tf.get_variable(name, shape) on the current store (default init glorot-uniform, A-7)."""
    return current_store().get_variable(name, shape, initializer, trainable)


class EmbeddingArena:
    """All embedding tables of width K of one model in one [rows, K] fp32 tensor plus the
    gradient / Adam-moment arenas of the same shape (include/recalgo.h "arena")."""

    def __init__(self, name: str, K: int, device, seed: int = 43):
        self.name, self.K, self.device = name, int(K), torch.device(device)
        self.tables: Dict[str, Tuple[int, int]] = {}   # table name -> (row_base, vocab)
        self._init: Dict[str, torch.Tensor] = {}
        self.rows = 0
        self._gen = torch.Generator().manual_seed(seed)
        self.weight = self._grad = self.m = self.v = None
        self.live = self.live_list = self.live_count = None   # live-row bookkeeping (see live_state)
        self._live_rows = -1
        self._order_ws = None                                  # order_live_list's scratch
        self.trainable = True

    @property
    def grad(self) -> Optional[torch.Tensor]:
        """The gradient arena [rows, K].  On the owner-computes path (sparse.py) the row gradients of a backward pass stay
        with the lookups' gradient matrices until the optimizer consumes them; whoever READS this attribute before that
        (tests, tools, named_grads) gets them summed into the arena first (recalgo_scatter_apply, GRAD mode)."""
        plan = self.__dict__.get("sparse")
        if plan is not None and plan.sources and not plan.grad_materialized:
            from . import sparse
            sparse.materialize_arena(self)
        return self._grad

    @grad.setter
    def grad(self, value) -> None:
        self._grad = value

    def add_table(self, name: str, vocab: int, init: Optional[torch.Tensor] = None,
                  view_shape: Optional[Sequence[int]] = None) -> int:
        """`view_shape`: the shape the reference's variable of this name has when it is not [vocab, K] (FFM's
        (F-1, V, K) per-field tables occupy (F-1) * V arena rows): named_arrays / named_grads present it that way."""
        if name in self.tables:
            return self.tables[name][0]
        if view_shape is not None:
            self.__dict__.setdefault("view_shapes", {})[name] = tuple(int(x) for x in view_shape)
        if self.weight is not None:
            raise RuntimeError("arena already materialised")
        rb = self.rows
        self.tables[name] = (rb, int(vocab))
        self.rows += int(vocab)
        if init is not None:
            self._init[name] = init
        return rb

    # tables at least this large are initialised directly in HBM (a 100 M x 16 table is 6.4 GB:
    # generating it on the host and copying it over would take minutes)
    DEVICE_INIT_ROWS = 4_000_000

    # rows generated per piece by the device-side initialiser (bounds the transient to 64 MB at K = 16)
    DEVICE_INIT_CHUNK = 1 << 20

    def materialize(self, shard=None):
        """Allocate and initialise the arena.  shard = (rank, world): keep only the rows r % world == rank (at local
        index r // world) — every rank draws the SAME random stream piece by piece and keeps its own rows, so an
        N-rank job starts from exactly the single-process initial values without ever holding the whole table
        (a 100 M x 16 table + gradient + two Adam moments is 25.6 GB replicated, 3.2 GB per rank on 8 ranks)."""
        if self.weight is not None:
            return
        K = self.K
        rank, world = (0, 1) if shard is None else (int(shard[0]), int(shard[1]))
        rows = max(len(range(rank, self.rows, world)), 1)
        big = self.rows >= self.DEVICE_INIT_ROWS and self.device.type == "cuda"
        w = torch.empty(rows, K, device=self.device if big else "cpu")
        if self.rows == 0 or rows > len(range(rank, self.rows, world)):
            w.zero_()
        dgen = torch.Generator(device=self.device).manual_seed(self._gen.initial_seed() + 1) if big else None

        def keep(values: torch.Tensor, g0: int):
            # values = global rows [g0, g0 + n): store the ones this rank owns
            first = (rank - g0) % world
            mine = values.reshape(-1, K)[first::world]
            if mine.shape[0]:
                l0 = (g0 + first) // world
                w[l0:l0 + mine.shape[0]] = mine.to(w.device)

        for name, (rb, vocab) in self.tables.items():
            if name in self._init:
                keep(self._init[name], rb)
            elif big:
                # truncated_normal(0, 1/sqrt(K)) on the device: resample |z| > 2 a few times, then clamp
                for c0 in range(0, vocab, self.DEVICE_INIT_CHUNK):
                    n = min(self.DEVICE_INIT_CHUNK, vocab - c0)
                    t = torch.randn(n, K, device=self.device, generator=dgen)
                    for _ in range(4):
                        bad = t.abs() > 2
                        t = torch.where(bad, torch.randn(n, K, device=self.device, generator=dgen), t)
                    keep(t.clamp_(-2, 2).mul_(1.0 / math.sqrt(K)), rb + c0)
                    del t
            else:
                keep(truncated_normal((vocab, K), 1.0 / math.sqrt(K), self._gen), rb)
        self.weight = w.to(self.device).contiguous()
        self.grad = torch.zeros_like(self.weight)
        self.m = torch.zeros_like(self.weight)
        self.v = torch.zeros_like(self.weight)
        self._init.clear()

    def table_view(self, name: str) -> torch.Tensor:
        if getattr(self, "sharding", None) is not None:
            raise RuntimeError(f"arena {self.name} is row-sharded over {self.sharding.sh.world} ranks: its tables are not "
                               "local views (gather them with parallel.unshard_arena / VariableStore.named_arrays(gather=True))")
        rb, vocab = self.tables[name]
        return self.shaped(name, self.weight[rb:rb + vocab])

    def shaped(self, name: str, rows: torch.Tensor) -> torch.Tensor:
        vs = self.__dict__.get("view_shapes", {}).get(name)
        return rows if vs is None else rows.view(vs)

    # -- live-row bookkeeping (optimizer cost proportional to the rows ever touched; see
    #    include/recalgo.h recalgo_mark_live_rows / recalgo_adam_tf1_list) ---------------------------
    @property
    def tracks_live_rows(self) -> bool:
        return self.weight is not None and self.weight.is_cuda

    def live_state(self):
        """-> (live uint8 [rows padded to 4], live_list int32 [rows], live_count int32 [1]); built
        lazily, and rebuilt from the Adam moments when the arena was re-sharded or restored."""
        rows = self.weight.shape[0]
        if self.live is None or self._live_rows != rows:
            dev = self.weight.device
            alive = ((self.m != 0) | (self.v != 0)).any(dim=1)
            self.live = torch.zeros((rows + 3) // 4 * 4, dtype=torch.uint8, device=dev)
            self.live[:rows] = alive.to(torch.uint8)
            idx = torch.nonzero(alive).squeeze(1).to(torch.int32)
            self.live_list = torch.zeros(max(rows, 1), dtype=torch.int32, device=dev)
            self.live_list[:idx.numel()] = idx
            self.live_count = torch.tensor([idx.numel()], dtype=torch.int32, device=dev)
            self._live_rows = rows
            self._order_ws = None
        return self.live, self.live_list, self.live_count

    def order_live_list(self) -> None:
        """Housekeeping between steps, no host synchronisation: rebuild the live-row list in address
        order (include/recalgo.h recalgo_order_live_list).  recalgo_mark_live_rows appends rows in
        first-touch order, i.e. at random; the list Adam then walks HBM at random (measured: 40 us vs
        32 us address-ordered, 345 k rows x 64 B x 7 streams)."""
        if self.live is None or not self.weight.is_cuda:
            return
        import ctypes
        from . import _lib
        lib = _lib.load()
        rows = self.weight.shape[0]
        if self._order_ws is None:
            self._order_ws = torch.empty(max(int(lib.recalgo_order_live_list_workspace_bytes(rows)), 4), dtype=torch.uint8,
                                         device=self.weight.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        st = ctypes.c_void_p(torch.cuda.current_stream(self.weight.device).cuda_stream)
        _lib.check(lib.recalgo_order_live_list(p(self.live), rows, p(self.live_list), p(self.live_count), p(self._order_ws), st),
                   "recalgo_order_live_list")

    def force_all_live(self) -> None:
        """Mark every row live: the optimizer then walks the whole arena, i.e. TF1's dense Adam at its full cost (what a
        long training run converges to; bench.py's forced-dense point)."""
        live, lst, cnt = self.live_state()
        rows = self.weight.shape[0]
        live[:rows] = 1
        lst[:rows] = torch.arange(rows, dtype=torch.int32, device=lst.device)
        cnt.fill_(rows)

    def live_rows(self) -> torch.Tensor:
        """uint8 [rows]: 1 where a gradient has reached the row."""
        return self.live_state()[0][:self.weight.shape[0]]
