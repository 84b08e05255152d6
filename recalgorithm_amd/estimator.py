"""tf.estimator / tf.train look-alikes: the call surface the reference's model scripts drive
(algorithm/DeepFM/deepfm.py:231-343 and the same skeleton in every model).

  model_fn(features, labels, mode, params) -> EstimatorSpec     (unchanged contract)
  Estimator(model_fn, params, config).train / evaluate / predict, train_and_evaluate

Execution model (MI355X-first, not TF's): model_fn is run eagerly once per step on the
Estimator's VariableStore; its kernels are enqueued on one HIP stream.  For fixed-shape batches
the whole step (forward, backward, TF1-Adam) is captured once into a hipGraph and replayed
(`GraphedTrainStep`), so a step costs one graph launch instead of ~40 kernel launches.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Dict, Iterable, Optional

import torch

from . import ops
from .variables import VariableStore, use_store


class ModeKeys:
    TRAIN = "train"
    EVAL = "eval"
    PREDICT = "infer"


class EstimatorSpec:
    def __init__(self, mode, predictions=None, loss=None, train_op=None, eval_metric_ops=None,
                 export_outputs=None, training_hooks=None, **_ignored):
        self.mode, self.predictions, self.loss, self.train_op = mode, predictions, loss, train_op
        self.eval_metric_ops = eval_metric_ops
        self.export_outputs, self.training_hooks = export_outputs, training_hooks


class RunConfig:
    def __init__(self, model_dir=None, save_checkpoints_steps=None, device=None, seed=42,
                 use_hip_graph=True, **_ignored):
        self.model_dir, self.save_checkpoints_steps = model_dir, save_checkpoints_steps
        self.device, self.seed, self.use_hip_graph = device, seed, use_hip_graph


class TrainSpec:
    def __init__(self, input_fn, max_steps=None, hooks=None):
        self.input_fn, self.max_steps, self.hooks = input_fn, max_steps, hooks


class EvalSpec:
    def __init__(self, input_fn, steps=None, throttle_secs=600, exporters=None, **_ignored):
        self.input_fn, self.steps, self.throttle_secs, self.exporters = input_fn, steps, throttle_secs, exporters


# --------------------------------------------------------------------------------------------
# tf.train.AdamOptimizer (a15)
# --------------------------------------------------------------------------------------------
class TrainOp:
    def __init__(self, optimizer: "AdamOptimizer", loss: torch.Tensor, store: VariableStore):
        self.optimizer, self.loss, self.store = optimizer, loss, store

    def run(self):
        self.loss.backward()
        self.optimizer.apply_gradients(self.store)


class AdamOptimizer:
    """tf.train.AdamOptimizer(learning_rate, beta1, beta2, epsilon).minimize(loss):
    dense TF1 semantics for every variable, embedding tables included."""

    lazy_embeddings = False

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.beta1, self.beta2, self.eps = float(learning_rate), beta1, beta2, epsilon

    def minimize(self, loss, global_step=None) -> TrainOp:
        from .variables import current_store
        return TrainOp(self, loss, current_store())

    def apply_gradients(self, store: VariableStore, grad_hook: Optional[Callable] = None):
        st = store.ensure_opt_state()
        from . import nn, sparse
        arenas = [ar for ar in store.arenas.values() if ar.weight is not None and ar.trainable]
        # arenas on the owner-computes path (sparse.py): their optimizer step is fused with the row-gradient scatter
        # (recalgo_scatter_apply) — TF1 Adam with dense semantics evaluated lazily but exactly, or LazyAdam
        owned = [ar for ar in arenas if sparse.has_work(ar)]
        owned.sort(key=sparse.has_companions)     # (an arena whose lookups ride on another arena's plan comes after it)
        arenas = [ar for ar in arenas if not sparse.has_work(ar)]
        fused = all(ar.tracks_live_rows for ar in arenas) and len(arenas) <= 4 and store.device.type == "cuda"
        # parked gradients: deferred weight-gradient split sums (+ the step counter, advanced by the same launch),
        # l2-regulariser contributions (PNN weight_regularizer)
        nn.apply_parked_grads(step_dev=st["step"] if fused else None)
        if grad_hook is not None:           # data-parallel all-reduce of the flat dense grads
            grad_hook(store)
        if fused:
            # one launch: dense TF1 Adam over the flat buffer + dense-semantics TF1 Adam over the rows a gradient has ever
            # reached (the update is the identity for all others); lr_t derived on the device from the step counter
            # (+ one workgroup per owner-computes plan: the prefix scan of its bucket totals, which `place` then reads)
            scans = [r for r in (sparse.plan_scan_record(ar, self.lazy_embeddings) for ar in owned[:4]) if r is not None]
            ops.adam_tf1_step_(store.flat, store.flat_grad, store.flat_m, store.flat_v, arenas, st["step"], None,
                               self.lr, self.beta1, self.beta2, self.eps, lazy=self.lazy_embeddings, plan_scans=scans)
            for ar in owned:
                sparse.apply(ar, self.lazy_embeddings, st["step"], self.lr, self.beta1, self.beta2, self.eps)
            return
        if self.lazy_embeddings and arenas:
            raise NotImplementedError("LazyAdamOptimizer needs the fused optimizer launch (HIP device, <= 4 arenas)")
        ops.adam_tf1_advance_(st["step"], st["lr_t"], self.lr, self.beta1, self.beta2)
        for ar in owned:
            sparse.apply(ar, self.lazy_embeddings, st["step"], self.lr, self.beta1, self.beta2, self.eps)
        kw = dict(step=-1, lr=self.lr, beta1=self.beta1, beta2=self.beta2, eps=self.eps,
                  zero_grad=True, lr_t_dev=st["lr_t"])
        if store.flat is not None and store.flat.numel():
            ops.adam_tf1_(store.flat, store.flat_grad, store.flat_m, store.flat_v, **kw)
        for ar in arenas:
            if ar.tracks_live_rows:
                ops.adam_tf1_list_(ar, st["lr_t"], self.beta1, self.beta2, self.eps)
            else:
                ops.adam_tf1_(ar.weight.view(-1), ar.grad.view(-1), ar.m.view(-1), ar.v.view(-1), **kw)


class LazyAdamOptimizer(AdamOptimizer):
    """tf.contrib.opt.LazyAdamOptimizer (the reference's DIEN, /root/reference algorithm/DIEN/dien.py:328): embedding rows
    without a gradient in a step keep their weights AND moments; dense variables update as in Adam.  The six hot-path
    models use tf.train.AdamOptimizer (dense semantics on the tables): selecting this class for them
    (params["lazy_adam"] = True / bench.py --lazy-adam) is a labelled DEVIATION that makes the optimizer cost
    proportional to the rows of the batch instead of the rows ever touched."""
    lazy_embeddings = True


def get_global_step():
    return None


# --------------------------------------------------------------------------------------------
# tf.metrics.{accuracy, auc} (a14) — streaming accumulators
# --------------------------------------------------------------------------------------------
class Metric:
    def update(self):
        raise NotImplementedError

    def result(self) -> float:
        raise NotImplementedError


class AccuracyMetric(Metric):
    def __init__(self, labels, predictions):
        self.labels, self.predictions = labels, predictions
        self.correct, self.total = 0.0, 0.0

    def merge(self, other):
        self.labels, self.predictions = other.labels, other.predictions

    def update(self):
        self.correct += float((self.labels.reshape(-1) == self.predictions.reshape(-1)).sum())
        self.total += self.labels.numel()

    def result(self):
        return self.correct / max(self.total, 1.0)


class AUCMetric(Metric):
    """tf.metrics.auc defaults: 200 thresholds, trapezoidal ROC."""

    def __init__(self, labels, predictions, num_thresholds=200):
        self.labels, self.predictions, self.n = labels, predictions, num_thresholds
        eps = 1e-7
        th = [(i + 1) * 1.0 / (num_thresholds - 1) for i in range(num_thresholds - 2)]
        self.th = torch.tensor([0.0 - eps] + th + [1.0 + eps], dtype=torch.float32)
        self.tp = torch.zeros(num_thresholds, dtype=torch.float64)
        self.fp = torch.zeros_like(self.tp)
        self.fn = torch.zeros_like(self.tp)
        self.tn = torch.zeros_like(self.tp)

    def merge(self, other):
        self.labels, self.predictions = other.labels, other.predictions

    def update(self):
        p = self.predictions.detach().reshape(1, -1).float().cpu()
        y = self.labels.detach().reshape(1, -1).cpu() > 0.5
        pred_pos = p > self.th.unsqueeze(1)
        self.tp += (pred_pos & y).sum(1)
        self.fp += (pred_pos & ~y).sum(1)
        self.fn += (~pred_pos & y).sum(1)
        self.tn += (~pred_pos & ~y).sum(1)

    def result(self):
        eps = 1e-7
        tpr = (self.tp + eps) / (self.tp + self.fn + eps)
        fpr = self.fp / (self.fp + self.tn + eps)
        return float(((fpr[:-1] - fpr[1:]) * (tpr[:-1] + tpr[1:]) / 2).sum())


class metrics:  # namespace mirror of tf.metrics
    @staticmethod
    def accuracy(labels, predictions):
        m = AccuracyMetric(labels, predictions)
        return (m, m)

    @staticmethod
    def auc(labels, predictions, num_thresholds=200):
        m = AUCMetric(labels, predictions, num_thresholds)
        return (m, m)


# --------------------------------------------------------------------------------------------
# hipGraph-captured training step
# --------------------------------------------------------------------------------------------
def _tree_tensors(obj, prefix=""):
    from .feature_column import Ragged
    if isinstance(obj, torch.Tensor):
        yield prefix, obj
    elif isinstance(obj, Ragged):
        yield prefix + ".values", obj.values
        yield prefix + ".offsets", obj.offsets
    elif isinstance(obj, dict):
        for k in sorted(obj):
            yield from _tree_tensors(obj[k], f"{prefix}/{k}")


def _byte_span(t: torch.Tensor):
    """[lo, hi) byte range of the storage that tensor t covers."""
    lo = t.storage_offset() * t.element_size()
    if t.numel() == 0:
        return lo, lo
    last = sum((d - 1) * st for d, st in zip(t.shape, t.stride()))
    return lo, lo + (last + 1) * t.element_size()


class DeviceBatch(dict):
    """features whose tensors (and the labels') are views of ONE device allocation laid out as `device_span[1]` describes:
    device_span = (uint8 tensor over the whole allocation, signature) — what Estimator._to_device_one_copy produces and
    GraphedTrainStep.load moves with one copy, without looking at the individual tensors."""
    device_span = None


def _clone_tree(obj):
    """Deep copy of a (features, labels) tree of tensors / Ragged / dicts that preserves storage sharing: tensors that
    are views of one allocation become views (same relative offsets / strides) of ONE cloned allocation — of the byte
    span the views cover, not of the whole storage (a batch that is a slice of a device-resident dataset must not
    duplicate the dataset)."""
    from .feature_column import Ragged
    spans = {}

    def scan(o):
        if isinstance(o, torch.Tensor):
            lo, hi = _byte_span(o)
            key = o.untyped_storage().data_ptr()
            a, b = spans.get(key, (lo, hi))
            spans[key] = (min(a, lo), max(b, hi))
        elif isinstance(o, Ragged):
            scan(o.values); scan(o.offsets)
        elif isinstance(o, dict):
            for v in o.values():
                scan(v)
        elif isinstance(o, (tuple, list)):
            for v in o:
                scan(v)
    scan(obj)
    storages = {}

    def cl(t):
        st = t.untyped_storage()
        key = st.data_ptr()
        lo, hi = spans[key]
        lo -= lo % 16                                           # keep every view's alignment
        if key not in storages:
            part = torch.empty(0, dtype=torch.uint8, device=t.device).set_(st, lo, (max(hi - lo, 0),), (1,))
            storages[key] = part.clone().untyped_storage()
        off = t.storage_offset() - lo // t.element_size()
        return torch.empty(0, dtype=t.dtype, device=t.device).set_(storages[key], off, t.shape, t.stride())

    def walk(o):
        if isinstance(o, torch.Tensor):
            return cl(o)
        if isinstance(o, Ragged):
            return Ragged(cl(o.values), cl(o.offsets))
        if isinstance(o, dict):
            return {k: walk(v) for k, v in o.items()}
        if isinstance(o, (tuple, list)):
            return type(o)(walk(v) for v in o)
        return o
    return walk(obj)


HOUSEKEEPING_EVERY = 32      # steps between VariableStore.housekeeping() calls (live-row list ordering)
OVERFLOW_POLL_EVERY = 64     # steps between reads of the static row-exchange overflow flag (N > 1; one host sync)


def _quiesce_collectives() -> None:
    """Before a capture that contains RCCL collectives: let torch.distributed's watchdog thread retire the EAGER collectives
    issued so far.  It polls their completion events every 100 ms; HIP refuses an event query (hipErrorCapturedEvent:
    'operation not permitted on an event last recorded in a capturing stream') once the communicator's internal stream —
    on which those events were recorded — has joined a capture, and the watchdog then aborts the process (seen as a rare
    crash of the warm-up-then-capture sequence).  All eager work is complete here (the caller has synchronised), so one
    poll interval empties the watchdog's list."""
    try:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_backend() != "nccl":
            return
    except Exception:            # (custom collectives objects, e.g. the host-staged bring-up adapter: no watchdog)
        return
    import time
    torch.cuda.synchronize()
    time.sleep(0.5)


class BatchShapeMismatch(ValueError):
    """A batch whose structure / shapes are not those the step was captured on (the last partial batch of an epoch): the
    ONLY condition under which the training loop runs a step eagerly beside a captured graph — any other error of the
    replay path propagates."""


class GraphedTrainStep:
    """Capture `step_fn(features, labels)` (forward + backward + optimizer, everything enqueued
    on the current stream) into a hipGraph over static input buffers; `__call__` copies the new
    batch into the static buffers and replays.  Shapes must not change between calls."""

    def __init__(self, step_fn: Callable, features, labels, warmup: int = 3):
        self.step_fn = step_fn
        # PRIVATE static input buffers: clones of the first batch (storage-preserving: the 26 id columns of one [B, F]
        # matrix stay views of one allocation).  Capturing on the caller's tensors would overwrite the user's batch on
        # every load() and replay stale data when that tensor comes round again.
        span = getattr(features, "device_span", None)
        features, labels = _clone_tree((features, labels))
        self.static_f, self.static_l = features, labels
        self._static = list(_tree_tensors(features, "f")) + list(_tree_tensors(labels, "l"))
        # a batch that is ONE allocation with a known layout (DeviceBatch): later batches of the same layout are loaded by one
        # copy of the whole span (the clone kept the layout: every static tensor sits at its source's offset)
        self._span_plan = None
        if span is not None and self._static:
            st0 = self._static[0][1].untyped_storage()
            if (st0.nbytes() >= span[0].numel() and all(t.untyped_storage().data_ptr() == st0.data_ptr() for _, t in self._static)
                    and all(a.storage_offset() == b.storage_offset() and a.stride() == b.stride() and a.shape == b.shape
                            for (_, a), (_, b) in zip(self._static, list(_tree_tensors(span[2][0], "f")) + list(_tree_tensors(span[2][1], "l"))))):
                dst = torch.empty(0, dtype=torch.uint8, device=span[0].device).set_(st0, 0, (span[0].numel(),), (1,))
                self._span_plan = (span[1], dst)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.out = step_fn(features, labels)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        _quiesce_collectives()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: HIP calls of OTHER threads (the RCCL watchdog of torch.distributed polls events)
        # must not be treated as capture violations while this thread captures
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.out = step_fn(features, labels)
        self.warmup_steps = warmup
        owner = getattr(step_fn, "__self__", None)                # Estimator.train_step -> its VariableStore
        self._housekeeping = getattr(getattr(owner, "store", None), "housekeeping", None)
        self._calls = 0

    def load(self, features, labels):
        """Copy a new batch into the static input buffers.  Inputs that are views of one
        allocation with the same layout on both sides (the 26 id columns of one [B, F] matrix)
        are moved by a single whole-allocation copy."""
        span = getattr(features, "device_span", None)
        if span is not None and self._span_plan is not None and span[1] == self._span_plan[0]:
            _device_copy(self._span_plan[1], span[0])
            return
        new = list(_tree_tensors(features, "f")) + list(_tree_tensors(labels, "l"))
        if len(new) != len(self._static):
            raise BatchShapeMismatch("graphed step: the input structure changed")
        spans = self._static_spans()
        groups = {}
        for i, ((k0, dst), (k1, src)) in enumerate(zip(self._static, new)):
            if k0 != k1 or dst.shape != src.shape:
                raise BatchShapeMismatch(f"graphed step: input {k1} changed shape/structure")
            if dst.dtype == src.dtype and dst.stride() == src.stride() and src.device == dst.device:
                # (same shape, strides and dtype: the source covers a byte span of the same length as the destination's)
                groups.setdefault((spans[i][0], src.untyped_storage().data_ptr()), []).append((i, src.storage_offset() * spans[i][3]))
            else:
                dst.copy_(src, non_blocking=True)
        singles = []
        for items in groups.values():
            # views of one allocation on both sides with the same relative layout: ONE copy of the byte span they cover —
            # provided no OTHER static input lives inside that span (the copy overwrites the gaps between the views too)
            if len(items) > 1:
                d_lo = min(spans[i][1] for i, _ in items)
                d_hi = max(spans[i][2] for i, _ in items)
                s_lo = min(lo for _, lo in items)
                mine = {i for i, _ in items}
                st_ptr = spans[items[0][0]][0]
                alone = all(k in mine or sp[0] != st_ptr or sp[2] <= d_lo or sp[1] >= d_hi for k, sp in enumerate(spans))
                if alone and d_hi > d_lo and all(spans[i][1] - d_lo == lo - s_lo for i, lo in items):
                    d0, s0 = self._static[items[0][0]][1], new[items[0][0]][1]
                    dv = self._dst_views.get((spans[items[0][0]][0], d_lo, d_hi))
                    if dv is None:
                        dv = torch.empty(0, dtype=torch.uint8, device=d0.device).set_(d0.untyped_storage(), d_lo, (d_hi - d_lo,), (1,))
                        self._dst_views[(spans[items[0][0]][0], d_lo, d_hi)] = dv
                    sv = torch.empty(0, dtype=torch.uint8, device=s0.device).set_(s0.untyped_storage(), s_lo, (d_hi - d_lo,), (1,))
                    _device_copy(dv, sv)
                    continue
            singles.extend(i for i, _ in items)
        for i in singles:                        # (after every span copy: a per-tensor copy is never overwritten by one)
            self._static[i][1].copy_(new[i][1], non_blocking=True)

    def _static_spans(self):
        """Per static input: (storage pointer, first byte, end byte, element size) — computed once: the per-step path of a
        host-fed training loop is Python, and every tensor method call in it costs as much as a tenth of the GPU step."""
        sp = self.__dict__.get("_spans")
        if sp is None:
            sp = self._spans = [(t.untyped_storage().data_ptr(),) + _byte_span(t) + (t.element_size(),) for _, t in self._static]
            self._dst_views = {}
        return sp

    def __call__(self, features=None, labels=None):
        if features is not None:
            self.load(features, labels)
        self.graph.replay()
        self._calls += 1
        if self._housekeeping is not None and self._calls % HOUSEKEEPING_EVERY == 0:
            self._housekeeping()       # enqueued between replays on the same stream; never waits for the GPU
        return self.out


# --------------------------------------------------------------------------------------------
# Estimator
# --------------------------------------------------------------------------------------------
def _device_copy(dst: torch.Tensor, src: torch.Tensor) -> None:
    """The batch -> static input buffer copy of a captured step.  Device to device on one GPU it is the library's own copy
    kernel (recalgo_copy_bytes: ~2 us for the 0.9 MB of a 4096-example batch; the runtime's copy kernel took 7.1 us of the
    230 us DCN step, profiles/r04zz_dcn_kernel_stats.md)."""
    if dst.is_cuda and src.is_cuda and dst.device == src.device and dst.is_contiguous() and src.is_contiguous():
        from . import ops
        ops.copy_bytes(dst, src)
    else:
        dst.copy_(src, non_blocking=True)


def _host_copy(dst: torch.Tensor, src: torch.Tensor) -> None:
    """dst.copy_(src) for host tensors on the training loop's thread.  Same dtype, both contiguous: ONE memmove — a batch's
    id matrix is ~1 MB, for which Tensor.copy_ starts an intra-op parallel region over every core of the host (128 threads
    on the GPU box); next to the reader's decode threads that costs milliseconds per call (measured: 1.6 ms against 0.08 ms),
    the whole difference between a host-fed loop at 0.5 M and at 8 M examples/s."""
    if (dst.dtype == src.dtype and dst.numel() == src.numel() and dst.is_contiguous() and src.is_contiguous()
            and dst.device.type == "cpu" and src.device.type == "cpu"):
        import ctypes
        ctypes.memmove(dst.data_ptr(), src.data_ptr(), dst.numel() * dst.element_size())
    else:
        dst.copy_(src)


class Estimator:
    def __init__(self, model_fn, params=None, config: Optional[RunConfig] = None, model_dir=None):
        self.model_fn, self.params = model_fn, params or {}
        self.config = config or RunConfig(model_dir=model_dir)
        dev = self.config.device
        if dev is None:
            if not torch.cuda.is_available():
                raise RuntimeError("recalgorithm_amd needs a HIP device (no CPU fallback)")
            dev = "cuda"
        self.device = torch.device(dev)
        self.store = VariableStore(self.device, seed=self.config.seed)
        self.global_step = 0
        self._built = False
        self._after_build = []           # callables run once at the end of the build (parallel.attach_data_parallel)
        self.grad_hook = None     # set by parallel wrappers (dense-grad all-reduce)
        self.loss_grad_scale = None   # 1/world under data parallelism (parallel.attach_data_parallel)
        self._seed_grad, self._seed_value = None, None
        self._pinned = {}                # (shape, dtype) -> ring of pinned staging buffers (_h2d)

    # -- plumbing -------------------------------------------------------------------------
    _STAGING_RING = 8

    def _h2d(self, fill, shape, dtype, dst: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Host -> device through a small ring of PERSISTENT pinned staging buffers (one ring per dtype, buffers grow to
        the largest batch seen): `fill(view)` writes the batch into a view of a pinned buffer, the copy to the device is
        asynchronous.  (tensor.pin_memory() per batch allocates and frees page-locked memory every step — on this stack
        milliseconds per call, an order of magnitude more than the step; a pageable source makes the copy synchronous.)
        A buffer is reused only after the copy that read it last has completed (its event)."""
        shape = tuple(int(d) for d in shape)
        numel = 1
        for d in shape:
            numel *= d
        ring = self._pinned.get(dtype)
        if ring is None:
            ring = self._pinned[dtype] = {"bufs": [None] * self._STAGING_RING, "events": [None] * self._STAGING_RING, "i": 0}
        i = ring["i"]
        ring["i"] = (i + 1) % self._STAGING_RING
        if ring["events"][i] is not None:
            ring["events"][i].synchronize()
        buf = ring["bufs"][i]
        if buf is None or buf.numel() < numel:
            cap = 1 << max(int(numel - 1).bit_length(), 10)
            buf = ring["bufs"][i] = torch.empty(cap, dtype=dtype, pin_memory=True)
        view = buf[:numel].view(shape)
        fill(view)
        if dst is None:
            dev = view.to(self.device, non_blocking=True)
        else:                                # (an existing device buffer of this shape: no allocation, the same asynchronous copy)
            dev = dst
            dev.copy_(view, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        ring["events"][i] = ev
        return dev

    def _to_device(self, features, labels):
        from .feature_column import Ragged

        def mv(x):
            if isinstance(x, torch.Tensor):
                if x.device.type == "cpu" and self.device.type == "cuda" and x.numel() and not x.is_pinned():
                    return self._h2d(lambda buf: _host_copy(buf, x), x.shape, x.dtype)
                return x.to(self.device, non_blocking=True)
            if isinstance(x, Ragged):
                return Ragged(mv(x.values), mv(x.offsets))
            if isinstance(x, dict):
                return {k: mv(v) for k, v in x.items()}
            return x

        one = self._to_device_one_copy(features, labels)
        if one is not None:
            return one
        if isinstance(features, dict):
            features = self._pack_host_columns(features)
        return mv(features), mv(labels)

    def _to_device_one_copy(self, features, labels):
        """The batch of the native reader whose features are exactly the columns of ONE [B, F] id matrix (io/native.py
        PackedBatch) with [B, 1] float labels: matrix and labels go through ONE staging buffer and ONE host-to-device copy, and
        arrive as views of one device allocation (id matrix, then the labels) — the layout a captured step's input load moves
        with a single copy (GraphedTrainStep.load).  None: not that shape of batch."""
        pb = self._packed_host_batch(features, labels)
        if pb is None:
            return None
        mat, keys, labs, nid, total, sig, fill = pb
        B = mat.shape[0]
        dev = self._h2d(fill, (total,), torch.uint8)
        dmat = dev[:nid].view(torch.int64).view(mat.shape)
        feats = DeviceBatch()
        for j, k in enumerate(keys):
            feats[k] = dmat[:, j]
        out_l = {k: dev[nid + 4 * B * i:nid + 4 * B * (i + 1)].view(torch.float32).view(v.shape) for i, (k, v) in enumerate(labs)}
        feats.device_span = (dev, sig, (dict(feats), out_l))
        return feats, out_l

    def _packed_host_batch(self, features, labels):
        """-> (id matrix, keys, [(label key, tensor)], id bytes, total bytes, layout signature, fill(pinned uint8 view)) of a
        native-reader batch whose features are exactly the columns of ONE host [B, F] id matrix with [B, 1] float labels;
        None for any other batch."""
        packed = getattr(features, "packed_ids", None)
        if self.device.type != "cuda" or packed is None or not isinstance(labels, dict) or not labels:
            return None
        mat, keys = packed
        if len(keys) < 2 or set(features) != set(keys) or mat.dtype != torch.int64 or not mat.is_contiguous() or mat.device.type != "cpu":
            return None
        B = mat.shape[0]
        labs = list(labels.items())
        if not all(isinstance(v, torch.Tensor) and v.device.type == "cpu" and v.dtype == torch.float32 and v.numel() == B
                   for _, v in labs):
            return None
        nid = mat.numel() * 8
        total = nid + 4 * B * len(labs)
        sig = ("ids+labels", int(total), int(B), tuple(keys), tuple(k for k, _ in labs), tuple(tuple(v.shape) for _, v in labs))

        def fill(buf):
            _host_copy(buf[:nid].view(torch.int64).view(mat.shape), mat)
            for i, (_, v) in enumerate(labs):
                _host_copy(buf[nid + 4 * B * i:nid + 4 * B * (i + 1)].view(torch.float32), v.reshape(-1))
        return mat, keys, labs, nid, total, sig, fill

    def feed_step(self, graphed: "GraphedTrainStep", features, labels, pb=None):
        """One replay of a captured step on a HOST batch.  A native-reader batch with the layout the step was captured on goes
        from its pinned staging buffer STRAIGHT into the graph's static input span — one asynchronous copy; no intermediate
        device tensor, no per-column views, no device-to-device load (together they were 0.3 ms of host time per step, more
        than the GPU step).  Any other batch: `_to_device` + `graphed(features, labels)` as before."""
        plan = graphed._span_plan
        if plan is not None:
            pb = pb if pb is not None else self._packed_host_batch(features, labels)     # (the training loop passes its own)
            if pb is not None and pb[5] == plan[0] and plan[1].numel() == pb[4]:
                self._h2d(pb[6], (pb[4],), torch.uint8, dst=plan[1])
                return graphed()
        features, labels = self._to_device(features, labels)
        return graphed(features, labels)

    def _pack_host_columns(self, features: dict, force: bool = False) -> dict:
        """A decoded batch arrives as one host tensor per feature (26 id vectors + 16 dense columns for the
        WeChat-shaped data): moved one by one that is ~40 tiny host-to-device copies per step, more than the
        GPU step itself.  Host-resident single-valued id vectors ([B] int64) are packed into ONE [B, F] matrix
        (columns in sorted key order, the order fc.input_layer consumes them in) and the [B, 1] float columns
        into one [B, n] matrix; each is copied once and the features become column views of it — which is also
        the layout the gather kernels read in place (feature_column._as_matrix).  Device-resident inputs are
        left alone."""
        if self.device.type == "cpu" and not force:
            return features
        groups = {"ids": [], "dense": []}
        for k in sorted(features):
            v = features[k]
            if not isinstance(v, torch.Tensor) or v.device.type != "cpu":
                continue
            if v.dtype == torch.int64 and v.dim() == 1:
                groups["ids"].append(k)
            elif v.dtype == torch.float32 and v.dim() == 2 and v.shape[1] == 1:
                groups["dense"].append(k)
        out = dict(features)
        packed = getattr(features, "packed_ids", None)
        if packed is not None and len(packed[1]) >= 2 and packed[1] == groups["ids"] and all(
                features[k].data_ptr() == packed[0][:, j].data_ptr() and features[k].stride() == packed[0][:, j].stride()
                for j, k in enumerate(packed[1])):
            # the native reader already decoded the id features into ONE [B, F] matrix in this order (io/native.py
            # PackedBatch): one contiguous copy instead of re-stacking F strided column views
            mat = packed[0]
            dev = self._h2d(lambda buf: _host_copy(buf, mat), tuple(mat.shape), torch.int64) if self.device.type == "cuda" \
                else mat.to(self.device)
            for j, k in enumerate(packed[1]):
                out[k] = dev[:, j]
            groups["ids"] = []
        for kind, keys in groups.items():
            keys = [k for k in keys if features[k].shape[0] == features[keys[0]].shape[0]]
            if len(keys) < 2:
                continue
            cols_ = [features[k] for k in keys]
            B_ = cols_[0].shape[0]
            if self.device.type == "cuda":
                if kind == "ids":                                                          # [B, F]
                    dev = self._h2d(lambda buf: torch.stack(cols_, dim=1, out=buf), (B_, len(cols_)), torch.int64)
                else:                                                                      # [B, n]
                    dev = self._h2d(lambda buf: torch.cat(cols_, dim=1, out=buf), (B_, len(cols_)), torch.float32)
            else:
                host = torch.stack(cols_, dim=1) if kind == "ids" else torch.cat(cols_, dim=1)
                dev = host.to(self.device)
            for j, k in enumerate(keys):
                out[k] = dev[:, j] if kind == "ids" else dev[:, j:j + 1]
        return out

    def _call_model_fn(self, features, labels, mode) -> EstimatorSpec:
        with use_store(self.store):
            self.store.begin_call()
            return self.model_fn(features, labels, mode, self.params)

    def _build(self, features, labels, mode):
        if self._built:
            return
        self.store.building = True
        with torch.no_grad():
            self._call_model_fn(features, labels, mode)
        self.store.finalize()
        self._built = True
        self._maybe_restore()
        for fn in self._after_build:
            fn()

    def build(self, features, labels=None, mode=ModeKeys.TRAIN):
        features, labels = self._to_device(features, labels)
        self._build(features, labels, mode)
        return self

    def train_step(self, features, labels):
        """One eager training step on device-resident inputs; returns the loss tensor."""
        from . import ops
        seed = 1.0 if self.loss_grad_scale is None else float(self.loss_grad_scale)
        with ops.loss_seed(seed):
            spec = self._call_model_fn(features, labels, ModeKeys.TRAIN)
        op = spec.train_op
        if self._seed_grad is None or float(self._seed_value) != seed or self._seed_grad.device != op.loss.device:
            self._seed_grad, self._seed_value = torch.full_like(op.loss.detach(), seed), seed   # made once, reused
        op.loss.backward(self._seed_grad)
        op.optimizer.apply_gradients(self.store, self.grad_hook)
        return spec.loss.detach()

    # -- public API -----------------------------------------------------------------------
    def train(self, input_fn, steps=None, max_steps=None, hooks=None, log_every=100):
        it = iter(input_fn())
        graphed = None
        t0 = time.time()
        n = 0
        while True:
            if max_steps is not None and self.global_step >= max_steps:
                break
            if steps is not None and n >= steps:
                break
            try:
                features, labels = next(it)
            except StopIteration:
                break
            loss = None
            pb = self._packed_host_batch(features, labels) if graphed is not None else None
            if pb is not None:
                host = (features, labels)
                try:
                    loss = self.feed_step(graphed, features, labels, pb)     # (host batch -> the graph's input span, one copy)
                except BatchShapeMismatch:       # last partial batch: eager step
                    features, labels = self._to_device(*host)
                    loss = self.train_step(features, labels)
            else:
                features, labels = self._to_device(features, labels)
                self._build(features, labels, ModeKeys.TRAIN)
            # the first two steps run eagerly (they also warm the allocator and hipBLASLt);
            # from the third fixed-shape batch on, the step is one hipGraph replay
            if loss is not None:
                pass
            elif self.config.use_hip_graph and _static_batch(features) and n >= 2:
                try:
                    if graphed is None:
                        graphed = GraphedTrainStep(self.train_step, features, labels, warmup=0)
                        loss = graphed()
                    else:
                        loss = graphed(features, labels)
                except BatchShapeMismatch:       # last partial batch: eager step
                    loss = self.train_step(features, labels)
            else:
                loss = self.train_step(features, labels)
                if n % HOUSEKEEPING_EVERY == HOUSEKEEPING_EVERY - 1:
                    self.store.housekeeping()
            self.global_step += 1
            n += 1
            if log_every and self.global_step % log_every < 1:
                print(f"[recalgo] step {self.global_step} loss {float(loss):.6f} "
                      f"({n / max(time.time() - t0, 1e-9):.1f} steps/s)", flush=True)
            if self.global_step % OVERFLOW_POLL_EVERY == 0:
                self._check_exchange_overflow()
            sc = self.config.save_checkpoints_steps
            if sc and self.config.model_dir and self.global_step % sc == 0:
                self.save_checkpoint()
        self._check_exchange_overflow()
        if self.config.model_dir:
            self.save_checkpoint()
        return self

    @torch.no_grad()
    def evaluate(self, input_fn, steps=None) -> Dict[str, float]:
        agg: Dict[str, Metric] = {}
        loss_sum, nb = 0.0, 0
        for i, (features, labels) in enumerate(input_fn()):
            if steps is not None and i >= steps:
                break
            features, labels = self._to_device(features, labels)
            self._build(features, labels, ModeKeys.EVAL)
            spec = self._call_model_fn(features, labels, ModeKeys.EVAL)
            for k, (m, _) in (spec.eval_metric_ops or {}).items():
                if k in agg:
                    agg[k].merge(m)
                else:
                    agg[k] = m
                agg[k].update()
            loss_sum += float(spec.loss)
            nb += 1
        out = {k: m.result() for k, m in agg.items()}
        out["loss"] = loss_sum / max(nb, 1)
        out["global_step"] = self.global_step
        return out

    @torch.no_grad()
    def predict(self, input_fn) -> Iterable[Dict[str, object]]:
        for features, labels in _with_labels(input_fn()):
            features, labels = self._to_device(features, labels)
            self._build(features, labels, ModeKeys.PREDICT)
            spec = self._call_model_fn(features, None, ModeKeys.PREDICT)
            preds = {k: v.detach().cpu().numpy() for k, v in spec.predictions.items()}
            B = len(next(iter(preds.values())))
            for b in range(B):
                yield {k: v[b] for k, v in preds.items()}

    # -- checkpoints (RunConfig(model_dir, save_checkpoints_steps)) -------------------------
    def _ckpt_path(self):
        return os.path.join(self.config.model_dir, "model.ckpt.pt")

    def save_checkpoint(self):
        """Every rank must call (collective when the arenas are row-sharded): the shards are gathered, rank 0 writes."""
        self._check_exchange_overflow()
        state, writer = collect_checkpoint_state(self.store, self.global_step)
        if writer:
            os.makedirs(self.config.model_dir, exist_ok=True)
            tmp = self._ckpt_path() + ".tmp"
            torch.save(state, tmp)
            os.replace(tmp, self._ckpt_path())
        sh = getattr(self, "shard_spec", None)
        if sh is not None and sh.world > 1:
            sh.dist.barrier(group=sh.group)            # nobody resumes from a half-written file

    # -- variables by their reference (TF) names --------------------------------
    # The mirror keeps every variable under the reference's TF name with the reference's shape, with one
    # exception: the (sum V, 1) first-order kernel of DeepFM / FwFM (`.../fm_first_order_dense/kernel`) is
    # stored as one slice per indicator column (`<kernel name>/<column key>`).  TF's input_layer lays the
    # indicator columns out sorted by column name (`<key>_indicator`), which fixes the row order.
    def _kernel_slices(self, arrays, name):
        subs = [k for k in arrays if k.startswith(name + "/")]
        return sorted(subs, key=lambda k: k[len(name) + 1:] + "_indicator")

    def export_variables(self) -> dict:
        """{reference TF variable name: numpy array} of the built model (per-column first-order slices are
        concatenated back into the reference's (sum V, 1) kernel)."""
        import numpy as np
        if not self._built:
            raise RuntimeError("export_variables: call build(features, labels) first")
        arrays = self.store.named_arrays()
        out, merged = {}, {}
        for k, v in arrays.items():
            base = k.rsplit("/", 1)[0]
            if base.endswith("/kernel") and base not in arrays:
                merged.setdefault(base, None)
            else:
                out[k] = v.detach().cpu().numpy().copy()
        for base in merged:
            out[base] = np.concatenate([arrays[k].detach().cpu().numpy() for k in self._kernel_slices(arrays, base)], 0)
        return out

    def load_variables(self, values, strict: bool = True):
        """Assign variables from {reference TF variable name: array} — e.g. the tensors of a TF-1.14
        checkpoint of the reference script dumped with `tf.train.load_checkpoint(dir).get_tensor(name)`
        (scripts/tf_ckpt_to_npz.py) — so that reference-trained weights run on these kernels.  Optimizer
        slots (`.../Adam`, `.../Adam_1`), `global_step` and `beta*_power` entries are ignored.  Returns the
        names assigned; with `strict`, a shape mismatch or a model variable without a value raises."""
        if not self._built:
            raise RuntimeError("load_variables: call build(features, labels) first")
        arrays = self.store.named_arrays()
        done = set()
        with torch.no_grad():
            for name, val in values.items():
                if name.endswith("/Adam") or name.endswith("/Adam_1") or name in ("global_step", "beta1_power", "beta2_power"):
                    continue
                t = torch.as_tensor(val).to(torch.float32)
                if name in arrays:
                    if tuple(arrays[name].shape) != tuple(t.shape):
                        if strict:
                            raise ValueError(f"load_variables: {name} has shape {tuple(t.shape)}, the model wants {tuple(arrays[name].shape)}")
                        continue
                    arrays[name].copy_(t)
                    done.add(name)
                    continue
                subs = self._kernel_slices(arrays, name)
                if subs and sum(arrays[k].shape[0] for k in subs) != t.shape[0]:
                    if strict:
                        raise ValueError(f"load_variables: {name} has {t.shape[0]} rows, the model's columns add up to "
                                         f"{sum(arrays[k].shape[0] for k in subs)}")
                    continue
                if subs:
                    row = 0
                    for k in subs:
                        n = arrays[k].shape[0]
                        arrays[k].copy_(t[row:row + n].reshape(arrays[k].shape))
                        row += n
                        done.add(k)
                elif strict and not name.endswith(("moving_mean", "moving_variance")):
                    raise KeyError(f"load_variables: the model has no variable {name}")
            blocks = {b for b, _ in self.store._alias.values()}     # fused blocks are covered by their named parts
            missing = [k for k in arrays if k not in done and k not in blocks]
            if strict and missing:
                raise KeyError(f"load_variables: no value for {missing[:5]}{' ...' if len(missing) > 5 else ''}")
        for a in self.store.arenas.values():
            a.live = None                    # liveness is rebuilt from the (unchanged) moments on next use
        return sorted(done)

    def load_tf_checkpoint(self, path: str, strict: bool = True):
        """Load the variables of a TensorFlow checkpoint written by the reference script — `path` is a model_dir (its
        `checkpoint` state file names the newest one) or a checkpoint prefix (`.../model.ckpt-10000`) — read natively
        (io/tf_checkpoint.py: no TensorFlow needed; format restated, see its header).  Returns the global step stored
        in the file (0 if none); optimizer slots are ignored, as in load_variables."""
        from .io import tf_checkpoint
        prefix = path
        if os.path.isdir(path):
            prefix = tf_checkpoint.latest_checkpoint(path)
            if prefix is None:
                raise FileNotFoundError(f"load_tf_checkpoint: no `checkpoint` state file in {path}")
        # optimizer slots (3x the embedding tables in host memory) are never read: load_variables ignores them anyway
        names = [n for n in tf_checkpoint.list_variables(prefix)
                 if not (n.endswith("/Adam") or n.endswith("/Adam_1") or n in ("beta1_power", "beta2_power"))]
        values = tf_checkpoint.read_checkpoint(prefix, names=names)
        self.load_variables(values, strict=strict)
        return int(values["global_step"]) if "global_step" in values else 0

    def save_tf_checkpoint(self, prefix: str) -> str:
        """Write the model's variables (reference names / shapes) + global_step as a TensorFlow V2 checkpoint
        (`<prefix>.index`, `<prefix>.data-00000-of-00001`, `checkpoint`): the hand-back to the reference's tooling."""
        import numpy as np
        from .io import tf_checkpoint
        arrays = dict(self.export_variables())
        arrays["global_step"] = np.array(int(self.global_step), dtype=np.int64)
        tf_checkpoint.write_checkpoint(prefix, arrays)
        return prefix

    def _maybe_restore(self):
        md = self.config.model_dir
        if not md or not os.path.exists(self._ckpt_path()):
            return
        state = torch.load(self._ckpt_path(), map_location="cpu", weights_only=True)
        self.global_step = restore_checkpoint_state(self.store, state, self.device, where=self._ckpt_path())

    def _check_exchange_overflow(self):
        """Row-sharded arenas with the static (graph-capturable) exchange drop the requests that do not fit a bucket
        (the staged row reads as zeros, its gradient is discarded) and only raise a device flag: poll it."""
        if getattr(self, "shard_spec", None) is None:
            return
        from .parallel import exchange_overflowed
        if exchange_overflowed(self):
            raise RuntimeError(
                "row exchange bucket overflow: a (source, owner) bucket received more requests than its fixed capacity, "
                "so some embedding rows were read as zeros and their gradients dropped since the last check.  Re-attach "
                "with a larger capacity_factor (attach_data_parallel(..., capacity_factor=world) can never overflow) or "
                "with capacity_factor=None (exact, eager exchange).")


def collect_checkpoint_state(store: VariableStore, global_step: int):
    """-> (state dict, this rank writes it).  COLLECTIVE when arenas are row-sharded (parallel.attach_data_parallel): the
    weight / m / v shards are assembled into whole tables in the HOST memory of rank 0 only, through a bounded device
    window (parallel.gather_arena_to_host) — no rank ever holds a whole table on its GPU and the other ranks copy nothing
    to their hosts; the file is independent of the number of ranks (restore happens before re-sharding)."""
    from . import parallel, sparse
    sparse.sync_store(store)                                   # deferred Adam: every row reflects all completed steps
    sharded = {n: a for n, a in store.arenas.items() if getattr(a, "sharding", None) is not None}
    writer = all(a.sharding.sh.rank == 0 for a in sharded.values())
    variables = {n: v.data.detach().cpu() for n, v in store.vars.items()} if writer else {}
    arena_m, arena_v = {}, {}
    for n, a in store.arenas.items():
        if n in sharded:
            full = {what: parallel.gather_arena_to_host(a, what, 0) for what in ("weight", "m", "v")}
            if writer:
                for tn, (rb, vocab) in a.tables.items():
                    variables[tn] = a.shaped(tn, full["weight"][rb:rb + vocab])
                arena_m[n], arena_v[n] = full["m"], full["v"]
        else:
            if writer:
                for tn in a.tables:
                    variables[tn] = a.table_view(tn).detach().cpu()
                arena_m[n], arena_v[n] = a.m.cpu(), a.v.cpu()
    if not writer:
        return None, False
    state = {
        "global_step": int(global_step),
        "variables": variables,
        "flat_m": None if store.flat_m is None else store.flat_m.cpu(),
        "flat_v": None if store.flat_v is None else store.flat_v.cpu(),
        "arena_m": arena_m, "arena_v": arena_v,
        "opt_step": None if store.opt_state is None else int(store.opt_state["step"]),
    }
    return state, True


def _same_device(a, b) -> bool:
    """torch.device('cuda') and torch.device('cuda:0') name the same GPU when 0 is the current device."""
    a, b = torch.device(a), torch.device(b)
    if a.type != b.type:
        return False
    if a.type != "cuda":
        return True
    cur = torch.cuda.current_device()
    return (cur if a.index is None else a.index) == (cur if b.index is None else b.index)


def restore_checkpoint_state(store: VariableStore, state: dict, device, where: str = "checkpoint") -> int:
    """Load `state` into a built store; returns the global step.  The file holds whole tables (it is independent of
    the number of ranks): a row-sharded arena takes its own rows r % world == rank from them.  A variable that is
    missing from the file or whose shape changed (another vocabulary, other hidden_units) makes the restore fail:
    resuming the step counter and Adam's bias correction on a partly re-initialised model is never what the caller
    wants."""
    saved = state["variables"]
    arrays = {n: v.data for n, v in store.vars.items()}
    table_shape = {}
    for a in store.arenas.values():
        for tn, (rb, vocab) in a.tables.items():
            table_shape[tn] = a.__dict__.get("view_shapes", {}).get(tn, (vocab, a.K))
    blocks = {b for b, _ in store._alias.values()}
    problems = []
    for k, shape in list((k, tuple(t.shape)) for k, t in arrays.items()) + list(table_shape.items()):
        if k in blocks:
            continue                                           # fused blocks are covered by their named parts
        if k not in saved:
            problems.append(f"{k}: not in the file")
        elif tuple(saved[k].shape) != tuple(shape):
            problems.append(f"{k}: file has {tuple(saved[k].shape)}, model wants {tuple(shape)}")
    for n, a in store.arenas.items():
        sd = getattr(a, "sharding", None)
        full = (sd.global_rows if sd is not None else a.m.shape[0], a.K)
        for slot in ("arena_m", "arena_v"):
            if n not in state.get(slot, {}) or tuple(state[slot][n].shape) != full:
                problems.append(f"{slot}[{n}]: missing or shape differs")
    if state.get("flat_m") is not None and store.flat_m is not None and state["flat_m"].shape != store.flat_m.shape:
        problems.append(f"dense Adam moments: file has {tuple(state['flat_m'].shape)}, model wants {tuple(store.flat_m.shape)}")
    if problems:
        raise RuntimeError(f"{where} does not match the model ({len(problems)} problem(s)): " + "; ".join(problems[:8])
                           + (" ..." if len(problems) > 8 else ""))
    unused = [k for k in saved if k not in arrays and k not in table_shape]
    if unused:
        import warnings
        warnings.warn(f"{where}: {len(unused)} saved variable(s) the model does not have were ignored: {unused[:5]}")
    with torch.no_grad():
        for k, t in arrays.items():
            if k not in blocks:
                t.copy_(saved[k])
        if state.get("flat_m") is not None and store.flat_m is not None:
            store.flat_m.copy_(state["flat_m"])
            store.flat_v.copy_(state["flat_v"])
        for n, a in store.arenas.items():
            sd = getattr(a, "sharding", None)
            rank, world = (0, 1) if sd is None else (sd.sh.rank, sd.sh.world)
            for tn, (rb, vocab) in a.tables.items():
                first = (rank - rb) % world                    # first row of the table this rank owns
                mine = saved[tn].reshape(vocab, a.K)[first::world]
                l0 = (rb + first) // world
                a.weight[l0:l0 + mine.shape[0]].copy_(mine)
            n_local = len(range(rank, state["arena_m"][n].shape[0], world))
            a.m[:n_local].copy_(state["arena_m"][n][rank::world])
            a.v[:n_local].copy_(state["arena_v"][n][rank::world])
            a.live = None            # rebuilt from the restored moments on next use
            from . import sparse
            sparse.reset(a)          # (deferred-Adam bookkeeping likewise)
    if state.get("opt_step") is not None:
        if store.opt_state is not None and _same_device(store.opt_state["step"].device, device):
            # in place: the captured step, and the arenas' deferred-Adam plans (sparse.sync_arena), hold THIS tensor
            store.opt_state["step"].fill_(int(state["opt_step"]))
        else:
            store.opt_state = {
                "step": torch.tensor([state["opt_step"]], dtype=torch.int64, device=device),
                "lr_t": torch.zeros(1, device=device)}
        from . import sparse
        for ar in store.arenas.values():       # (a plan made before the restore must not sync against a stale counter)
            plan = sparse.plan_of(ar)
            if plan is not None:
                plan.step_dev = store.opt_state["step"]
    return int(state["global_step"])


def _static_batch(features) -> bool:
    from .feature_column import Ragged
    return all(isinstance(v, torch.Tensor) for v in features.values()) and \
        not any(isinstance(v, Ragged) for v in features.values())


def _with_labels(it):
    for item in it:
        if isinstance(item, tuple) and len(item) == 2:
            yield item
        else:
            yield item, None


def train_and_evaluate(estimator: Estimator, train_spec: TrainSpec, eval_spec: EvalSpec):
    """tf.estimator.train_and_evaluate (deepfm.py:323): train to max_steps, evaluate, then hand the evaluation result to
    the EvalSpec's exporters (deepfm.py:309-321: BestExporter -> <model_dir>/export/<name>/<timestamp>)."""
    estimator.train(train_spec.input_fn, max_steps=train_spec.max_steps)
    result = estimator.evaluate(eval_spec.input_fn, steps=eval_spec.steps)
    exporters = eval_spec.exporters or []
    if not isinstance(exporters, (list, tuple)):
        exporters = [exporters]
    base = estimator.config.model_dir
    for ex in exporters:
        if base is None:
            raise ValueError("train_and_evaluate: exporters need RunConfig(model_dir=...)")
        ex.export(estimator, os.path.join(base, "export", ex.name), estimator._ckpt_path(), result, True)
    return result
