"""tf.layers.{dense, batch_normalization, dropout} look-alikes for the mirrored model_fns.

These are the "context" MLP of the models.  `dense` runs on the hand-written fp32-MFMA
kernels of csrc/dense.hip (bias + ReLU fused in the forward, the ReLU mask and the bias gradient fused in the
backward); layers wider than DENSE_MAX_K inputs (FiBiNET's 9600 -> 512) run on hipBLASLt through torch.  Parameter
gradients are written straight into the flat gradient buffer (variables.py).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch.autograd import Function

from .variables import Variable, current_store, glorot_uniform, ones, zeros


def _hip(x: torch.Tensor, C: int) -> bool:
    """The fused HIP glue kernels serve contiguous fp32 device tensors of a supported width."""
    from . import ops
    return x.is_cuda and x.dtype == torch.float32 and ops.mlp_width_supported(C)


DENSE_MAX_K = 4096
FUSED_TAIL = True           # dense(last_hidden=True) may hand the layer to the fused layer + head + loss kernel (tests switch it off to compare)


def _mfma_dense(in_features: int = 0) -> bool:
    """The hand-written fp32-MFMA kernels with fused epilogues (csrc/dense.hip) serve layers of up to DENSE_MAX_K input
    features.  Wider layers are plain large GEMMs whose gradient tiles have short reductions (FiBiNET's 9600 -> 512: 150
    column tiles of 16 chunks each): the per-tile prologue / epilogue of the 64 x 64 engine then costs 25 % — the step is
    1.546 ms on it against 1.342 ms with the library kernels (round 5, same box; round 2: 1.57 vs 1.42) — those go to
    hipBLASLt through torch."""
    return in_features <= DENSE_MAX_K


class GradJoin:
    """A tensor consumed by two branches (DCN's x0 feeds the cross network and the MLP, dcn.py:157-166) receives the
    SUM of the two input gradients — in autograd an extra elementwise launch.  With a GradJoin shared by the two ops,
    the branch whose backward runs first (`dense`: it is created later in the forward) parks its input gradient here
    instead of returning it, and the other branch's backward kernel adds it in its own epilogue
    (`recalgo_cross_bwd`'s g_x0_extra) and returns the total.  If the order is ever the other way round, or the
    consumer cannot take it, both gradients are returned normally and autograd adds them."""

    def __init__(self):
        self.pending: Optional[torch.Tensor] = None
        self.consumer_done = False
        # the OTHER direction (ops.defer_cross_rider): the cross network's backward already ran — as a rider of an upper dense
        # layer's launch — and left its dx0 here; the MLP's first layer adds it in its input-gradient epilogue (beta * C, beta = 1)
        self.early_dx: Optional[torch.Tensor] = None
        self.early_used = False

    def take_early(self) -> Optional[torch.Tensor]:
        t, self.early_dx = self.early_dx, None
        if t is not None:
            self.early_used = True
        return t

    def park(self, dx: torch.Tensor) -> bool:
        if self.consumer_done or self.pending is not None:
            return False
        self.pending = dx
        return True

    def take(self) -> Optional[torch.Tensor]:
        t, self.pending, self.consumer_done = self.pending, None, True
        return t


class BNLink:
    """A training-mode BatchNorm whose output feeds a dense layer (tf.layers.batch_normalization -> tf.layers.dense): the
    dense layer's backward kernel can leave the two column sums BatchNorm's backward starts with (ops.dense_bwd(bn=)).  The
    BatchNorm attaches a BNLink to its output tensor; a dense layer that finds one on its input fills `sums` in its backward
    and records which gradient tensor they belong to; the BatchNorm's backward uses them only if that IS the gradient it is
    given (another consumer of the BatchNorm output, or anything in between, makes autograd hand over a different tensor)."""

    def __init__(self, x, mean, rstd):
        self.x, self.mean, self.rstd = x, mean, rstd
        self.sums, self.grad_ptr = None, 0

    def new_sums(self) -> torch.Tensor:
        from . import ops
        rows, C = self.x.shape
        self.sums = torch.empty(ops.bn_partial_rows(rows), 2 * C, device=self.x.device, dtype=torch.float32)
        return self.sums

    def take(self, g: torch.Tensor):
        sums, self.sums = self.sums, None
        return sums if (sums is not None and g.data_ptr() == self.grad_ptr and g.is_contiguous()) else None


def _attach_link(out: torch.Tensor) -> torch.Tensor:
    """the BNLink the node's forward made (ctx.link), hung on the output tensor for the consumer to find"""
    link = getattr(out.grad_fn, "link", None) if out.grad_fn is not None else None
    if link is not None:
        out._recalgo_bn_link = link
    return out


def _bn_link_of(x: torch.Tensor):
    link = getattr(x, "_recalgo_bn_link", None)
    return link if (link is not None and x.dim() == 2 and x.is_contiguous() and tuple(x.shape) == tuple(link.x.shape)) else None


class ReluSource:
    """Rides on the output tensor of a tf.layers.dense(..., relu) (`_recalgo_relu_src`).  A consumer whose backward kernel can
    mask its input gradient with that tensor itself — the next dense layer (recalgo_dense_bwd_bn dx_relu_mask), the fused
    loss tail (recalgo_logit_loss_fwd_bwd relu_parts) — does so and leaves the gradient tensor here; the producing layer's
    backward skips its own mask (the mask loads of both of its GEMMs) when the gradient autograd hands it IS that tensor.
    Any other consumer of the activation makes autograd sum into a new tensor (the reference held here keeps the engine from
    accumulating in place): the pointer differs and the mask is applied as before — idempotent on the pre-masked share."""
    __slots__ = ("premasked",)

    def __init__(self):
        self.premasked = None

    def take(self, g: torch.Tensor) -> bool:
        pm, self.premasked = self.premasked, None
        return pm is not None and pm.data_ptr() == g.data_ptr() and pm.shape == g.shape and pm.stride() == g.stride()


class _DenseFn(Function):
    @staticmethod
    def forward(ctx, anchor, x, kernel: Variable, bias: Optional[Variable], relu: bool, input_l2: float = 0.0,
                grad_join: Optional[GradJoin] = None, bn_partials: Optional[torch.Tensor] = None,
                relu_src: Optional[ReluSource] = None, drop=None):
        ctx.input_l2 = float(input_l2)
        ctx.drop_scale = 1.0 if drop is None else float(drop.scale)
        ctx.grad_join = grad_join
        ctx.bn_link = _bn_link_of(x)
        ctx.relu_src = relu_src
        ctx.x_relu_src = getattr(x, "_recalgo_relu_src", None) if (x.dim() == 2 and x.is_contiguous()) else None
        if getattr(x, "_recalgo_relu_scale", 1.0) != 1.0:
            ctx.x_relu_src = None              # (only the BatchNorm backward scales while it masks: the producer does it itself)
        x2 = x.reshape(-1, x.shape[-1])
        ctx.hip = x2.is_cuda and x2.dtype == torch.float32 and _mfma_dense(x2.shape[1])
        if ctx.hip:
            from . import ops
            if x2.stride(1) != 1:
                x2 = x2.contiguous()
            # GEMM + bias + ReLU in one launch
            # (drop: the tf.layers.dropout behind the layer applied by the epilogue — y IS the dropped tensor, y > 0 <=> relu > 0 and kept)
            y = ops.dense_fwd(x2, kernel.data, None if bias is None else bias.data, relu, bn_partials=bn_partials, drop=drop)
        elif drop is not None:
            raise ValueError("dense(drop=): only with the hand-written kernels (the caller checks nn.dense_drop_supported)")
        elif bias is not None and relu and x2.is_cuda:
            y = torch._addmm_activation(bias.data, x2, kernel.data)     # GEMM + bias + ReLU epilogue (hipBLASLt)
        else:
            y = torch.addmm(bias.data, x2, kernel.data) if bias is not None else x2 @ kernel.data
            if relu:
                y = torch.relu_(y)
        ctx.vars = (kernel, bias)
        ctx.relu = relu
        ctx.xshape = x.shape
        ctx.save_for_backward(x2, y if relu else None)
        return y.view(*x.shape[:-1], kernel.data.shape[1])

    @staticmethod
    def backward(ctx, g):
        from . import ops
        kernel, bias = ctx.vars
        x2, y = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1])
        if ctx.hip:
            if g2.stride(1) != 1 or (y is not None and g2.stride() != y.stride()):
                g2 = g2.contiguous()
            # the ReLU mask rides on the staging of g in both GEMMs, the bias gradient on the weight-gradient one; the
            # split partials of the weight gradient are summed by ONE launch per backward pass (ops.flush_dense_splits)
            mask = y if ctx.relu else None
            if mask is not None and ctx.relu_src is not None and ctx.relu_src.take(g2):
                mask = None                        # the consumer's backward kernel masked (and, behind a fused dropout, scaled) its dx with y already
            elif ctx.drop_scale != 1.0:
                g2 = g2 * ctx.drop_scale           # (no consumer did it: d/d relu of relu * keep / (1 - rate), the keep part is the mask y > 0)
            db = None if bias is None else bias.grad
            if ctx.needs_input_grad[1]:
                # input and weight gradient in ONE launch
                link = ctx.bn_link if ctx.grad_join is None else None
                src = ctx.x_relu_src if ctx.grad_join is None else None
                c_in, beta = (x2, ctx.input_l2) if ctx.input_l2 else (None, 0.0)
                early = None
                if ctx.grad_join is not None and c_in is None and ctx.grad_join.early_dx is not None:
                    e = ctx.grad_join.early_dx
                    if tuple(e.shape) == tuple(x2.shape) and e.is_contiguous():
                        early = ctx.grad_join.take_early()       # the other consumer's input gradient, already computed: dx += it
                        c_in, beta = early, 1.0
                dx = ops.dense_bwd(x2, g2, mask, kernel.data, kernel.grad, db, c_in=c_in, beta=beta, defer=True,
                                   bn=None if link is None else (link.x, link.mean, link.rstd, link.new_sums()),
                                   premask=None if src is None else x2, cross_rider=ctx.grad_join is None).view(ctx.xshape)
                if src is not None:
                    src.premasked = dx             # x is the ReLU output of the layer below: its gradient is masked here
                if link is not None:
                    link.grad_ptr = dx.data_ptr()
                if ctx.grad_join is not None and early is None and ctx.grad_join.park(dx):
                    dx = None                      # added by the other consumer of x in its backward kernel
                return None, dx, None, None, None, None, None, None, None, None
            # (no input gradient wanted: the first layer over a non-differentiable input.  Tried in round 2: the weight
            # gradient on a second stream beside the dgrad chain — DCN 0.319 vs 0.268 ms; removed)
            ops.dense_bwd_weights(x2, g2, mask, kernel.grad, db, defer=True)
            return None, None, None, None, None, None, None, None, None, None
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        if bias is not None and _hip(g2, g2.shape[1]):
            # one fused pass: ReLU mask + bias gradient (csrc/mlp.hip)
            g2 = ops.relu_bwd_bias_(g2, y if ctx.relu else None, bias.grad)
        else:
            if ctx.relu:
                g2 = g2 * (y > 0)
            if bias is not None:
                torch.sum(g2, dim=0, out=bias.grad)
        torch.mm(x2.t(), g2, out=kernel.grad)
        if ctx.input_l2:
            # d/dx of the activity regulariser (input_l2 / 2) * sum(x^2) rides on the GEMM as its beta * C term
            dx = torch.addmm(x2, g2, kernel.data.t(), beta=ctx.input_l2)
        else:
            dx = g2 @ kernel.data.t()
        return None, dx.view(ctx.xshape), None, None, None, None, None, None, None, None


class _Dense1Fn(Function):
    """tf.layers.dense(tf.concat(parts, -1), 1): one fused pass each way (csrc/mlp.hip dense1_*)."""

    @staticmethod
    def forward(ctx, anchor, kernel: Variable, bias: Optional[Variable], *parts):
        from . import ops
        ctx.vars = (kernel, bias)
        ctx.save_for_backward(*parts)
        return ops.dense1_fwd(parts, kernel.data, None if bias is None else bias.data)

    @staticmethod
    def backward(ctx, g):
        from . import ops
        kernel, bias = ctx.vars
        parts = ctx.saved_tensors
        dxs = [torch.empty_like(t) if need else None for t, need in zip(parts, ctx.needs_input_grad[3:])]
        ops.dense1_bwd(parts, kernel.data, g.contiguous(), dxs, kernel.grad, None if bias is None else bias.grad)
        return (None, None, None, *dxs)


class LazyDense:
    """`tf.layers.dense(x, units, relu)` — the LAST hidden layer of a TRAIN step — not evaluated yet: when its only consumer
    turns out to be the one-unit head in front of the loss (dcn.py:166-172), model_tail.finish_model_fn runs layer, head, loss and
    their backward as ONE kernel (ops.tail_dense_head).  Any other use goes through `materialize()`: the layer as `dense` runs it."""

    def __init__(self, x, kernel: Variable, bias: Variable, grad_join=None):
        self.x, self.kernel, self.bias, self.grad_join = x, kernel, bias, grad_join
        self._out = None

    @property
    def shape(self):
        return (*self.x.shape[:-1], int(self.kernel.data.shape[1]))

    def dim(self):
        return self.x.dim()

    @property
    def is_cuda(self):
        return self.x.is_cuda

    def materialize(self) -> torch.Tensor:
        if self._out is None:
            store = current_store()
            src = ReluSource()
            self._out = _DenseFn.apply(store.anchor, self.x, self.kernel, self.bias, True, 0.0, self.grad_join, None, src, None)
            self._out._recalgo_relu_src = src
        return self._out


def _resolved(t):
    return t.materialize() if isinstance(t, LazyDense) else t


class LazyConcat:
    """`tf.concat(values, axis=-1)` whose consumer is a one-unit `dense`: the head kernel reads the
    parts in place, so the [B, sum(widths)] copy (and the two slice copies of its gradient) never
    happen.  Any other use goes through `materialize()`."""

    def __init__(self, values):
        self.parts = list(values)

    @property
    def shape(self):
        return (*self.parts[0].shape[:-1], sum(int(t.shape[-1]) for t in self.parts))

    def materialize(self) -> torch.Tensor:
        return torch.cat([_resolved(t) for t in self.parts], dim=-1)


class LazyLogit:
    """`tf.layers.dense(x, 1)` (and sums of such heads and of [B, 1] tensors) whose consumer is the loss of a TRAIN
    step: model_tail.finish_model_fn hands the un-evaluated sum to ONE kernel that computes the logit, the
    probabilities, the loss and the whole backward of this tail (ops.logit_loss).  Any other use goes through
    `materialize()`, i.e. the separate head kernel."""

    def __init__(self, heads=(), tensors=()):
        self.heads = list(heads)          # [(kernel Variable, bias Variable | None, [parts])]
        self.tensors = list(tensors)      # [B, 1] addends

    @property
    def shape(self):
        B = (self.heads[0][2][0] if self.heads else self.tensors[0]).shape[0]
        return (B, 1)

    def __add__(self, other):
        if isinstance(other, LazyLogit):
            return LazyLogit(self.heads + other.heads, self.tensors + other.tensors)
        if isinstance(other, torch.Tensor):
            return LazyLogit(self.heads, self.tensors + [other])
        return NotImplemented

    __radd__ = __add__

    def tail(self):
        """(side part | None, LazyDense, side_first) when this is ONE head over [side, last hidden layer] (either order) with
        nothing else added and the fused layer + head + loss kernel serves the shapes; else None."""
        from . import ops
        if len(self.heads) != 1 or self.tensors:
            return None
        kernel, _, parts = self.heads[0]
        lazies = [i for i, t in enumerate(parts) if isinstance(t, LazyDense)]
        if len(lazies) != 1 or len(parts) > 2 or hasattr(kernel, "apply_head"):
            return None
        ld = parts[lazies[0]]
        side = parts[1 - lazies[0]] if len(parts) == 2 else None
        if not ops.tail_dense_head_supported(ld.x, ld.shape[-1], side):
            return None
        return side, ld, lazies[0] == 1

    def resolve(self):
        """the un-evaluated last hidden layer (LazyDense), if any, run as an ordinary dense layer"""
        self.heads = [(k, b, [_resolved(t) for t in ps]) for k, b, ps in self.heads]
        return self

    def fusable(self) -> bool:
        from . import ops
        self.resolve()
        parts = [t for _, _, ps in self.heads for t in ps]
        return (len(self.heads) >= 1 and sum(b is not None for _, b, _ in self.heads) <= 1
                and ops.logit_loss_supported(parts, self.tensors))

    def materialize(self) -> torch.Tensor:
        store = current_store()
        out = None
        for kernel, bias, parts in self.resolve().heads:
            t = kernel.apply_head(parts) if hasattr(kernel, "apply_head") else _Dense1Fn.apply(store.anchor, kernel, bias, *parts)
            out = t if out is None else out + t
        for t in self.tensors:
            out = t if out is None else out + t
        return out


def concat(values, axis: int = -1):
    """tf.concat along the last axis; lazy (see LazyConcat) when every value is a 2-D device tensor."""
    values = list(values)
    if axis in (-1, values[0].dim() - 1) and len(values) <= 4 and all(t.dim() == 2 and t.is_cuda for t in values):
        return LazyConcat(values)
    return torch.cat([_resolved(t) for t in values], dim=axis)


def dense_drop_supported(x, units: int) -> bool:
    """dense(drop=) — the dropout behind a ReLU layer applied by the layer's own epilogue — is served for what the hand-written
    forward kernel serves: a 2-D fp32 device tensor of up to DENSE_MAX_K features, outside Sync-BatchNorm."""
    store = current_store()
    return (not store.building and isinstance(x, torch.Tensor) and x.dim() == 2 and x.is_cuda and x.dtype == torch.float32
            and _mfma_dense(x.shape[1]) and int(units) % 4 == 0 and x.shape[0] * int(units) < (1 << 32)
            and getattr(getattr(store.anchor, "_recalgo_store", None), "sync_bn", None) is None)


def dense(x, units, activation: Optional[str] = None,
          use_bias: bool = True, name: Optional[str] = None, input_l2: float = 0.0,
          grad_join: Optional[GradJoin] = None, bn_stats: bool = False, drop=None, last_hidden: bool = False) -> torch.Tensor:
    """tf.layers.dense(x, units, activation=None|relu, use_bias, name).  `units` may be a
    str (the reference passes FLAGS.hidden_units.split(','), deepfm.py:286; quirk B-2).
    Variables: <scope>/<name>/kernel (glorot-uniform), <scope>/<name>/bias (zeros).
    `input_l2` = c (not a TF argument): the caller adds the loss term (c / 2) * sum(x^2) as a VALUE only
    (`l2_value`) and this layer adds its gradient c * x to the input gradient inside the dgrad GEMM
    (beta * C epilogue) — c must already contain the loss-gradient seed (ops.loss_seed).
    `bn_stats` (not a TF argument): a training-mode tf.layers.batch_normalization consumes this layer's output next (the
    dense -> [dropout] -> batch_norm order of deepfm.py:207-211): the GEMM's epilogue leaves the batch moments of its tiles
    and `batch_normalization` runs without a moments pass of its own (it finds them attached to the tensor it is given —
    a dropout in between makes a new tensor, and the BatchNorm layer computes its own moments as before).
    `last_hidden` (not a TF argument): this ReLU layer's output goes to the one-unit head in front of the loss (dcn.py:166-172);
    in a TRAIN step it is returned un-evaluated (LazyDense) for the fused layer + head + loss kernel."""
    store = current_store()
    units = int(units)
    if isinstance(x, LazyDense):
        x = x.materialize()
    name = name or store.auto_name("dense")
    with store.variable_scope(name):
        kernel = store.get_variable("kernel", (x.shape[-1], units), glorot_uniform)
        bias = store.get_variable("bias", (units,), zeros) if use_bias else None
    if activation not in (None, "relu"):
        raise ValueError(f"unsupported activation {activation}")
    parts = x.parts if isinstance(x, LazyConcat) else [x]
    if units == 1 and activation is None:
        from . import ops
        parts = [t if (isinstance(t, LazyDense) or t.is_contiguous()) else t.contiguous() for t in parts]
        if any(isinstance(t, LazyDense) for t in parts):
            lazy = LazyLogit([(kernel, bias, parts)])
            if not store.building and lazy.tail() is not None:
                return lazy                                   # TRAIN step: layer + head + loss + their backward in one launch
            parts = lazy.resolve().heads[0][2]
        if ops.dense1_supported(parts):
            if ops.logit_loss_supported(parts, []) and not store.building:
                return LazyLogit([(kernel, bias, parts)])     # TRAIN step: evaluated together with the loss
            return _Dense1Fn.apply(store.anchor, kernel, bias, *parts)
        x = LazyConcat(parts) if isinstance(x, LazyConcat) else parts[0]
    if isinstance(x, LazyConcat):
        x = x.materialize()
    if (last_hidden and FUSED_TAIL and activation == "relu" and use_bias and not store.building and input_l2 == 0.0 and not bn_stats and drop is None
            and grad_join is None and isinstance(x, torch.Tensor) and getattr(x, "_recalgo_relu_src", None) is not None
            and getattr(x, "_recalgo_relu_scale", 1.0) == 1.0):
        from . import ops
        if ops.tail_dense_head_supported(x, units, None):
            return LazyDense(x, kernel, bias, grad_join)
    bn_part = None
    if (bn_stats and not store.building and x.dim() == 2 and x.is_cuda and x.dtype == torch.float32 and units % 4 == 0
            and _mfma_dense(x.shape[1]) and getattr(getattr(store.anchor, "_recalgo_store", None), "sync_bn", None) is None):
        from . import ops
        bn_part = torch.empty(ops.bn_partial_rows(x.shape[0]), 2 * units, device=x.device, dtype=torch.float32)
    if drop is not None and (activation != "relu" or not dense_drop_supported(x, units)):
        raise ValueError("dense(drop=) needs activation='relu' and nn.dense_drop_supported(x, units)")
    relu_src = ReluSource() if (activation == "relu" and not store.building) else None
    out = _DenseFn.apply(store.anchor, x, kernel, bias, activation == "relu", input_l2, grad_join, bn_part, relu_src, drop)
    if bn_part is not None:
        out._recalgo_bn_partials = bn_part
    if relu_src is not None:
        out._recalgo_relu_src = relu_src
        if drop is not None:
            out._recalgo_relu_scale = float(drop.scale)      # (a consumer that masks its dx with this tensor also scales it)
    return out


def dense_relu_dropout_bn(x, units, dropout_rate, batch_norm: bool, training: bool) -> torch.Tensor:
    """One hidden layer of the DeepFM / PNN / FiBiNET / NFM MLPs (/root/reference algorithm/DeepFM/deepfm.py:207-211):
        net = tf.layers.dense(net, unit, activation=tf.nn.relu)
        if "dropout_rate" in params and 0.0 < params["dropout_rate"] < 1.0: net = tf.layers.dropout(net, rate, training=...)
        if params["batch_norm"]: net = tf.layers.batch_normalization(net, training=...)
    Same variables, scopes and arithmetic as the three calls.  In a training step with both the dropout and the BatchNorm on,
    the dropout costs no launch: the dense layer's epilogue applies it (and leaves the BatchNorm's tile moments of the DROPPED
    tensor), and the BatchNorm's backward — which masks its input gradient with that tensor anyway — scales it by 1 / (1 - rate)."""
    rate = float(dropout_rate) if dropout_rate is not None else 0.0
    drop_on = bool(training) and 0.0 < rate < 1.0
    if drop_on and batch_norm and isinstance(x, torch.Tensor) and dense_drop_supported(x, units):
        d = drop_spec((x.shape[0], int(units)), rate, x.device)
        net = dense(x, units, activation="relu", bn_stats=True, drop=d)
        return batch_normalization(net, training=True)
    net = dense(x, units, activation="relu", bn_stats=bool(batch_norm) and bool(training) and not drop_on)
    if 0.0 < rate < 1.0:
        net = dropout(net, rate, training=training)
    if batch_norm:
        net = batch_normalization(net, training=training)
    return net


def dense_with(x: torch.Tensor, kernel: Variable, bias: Optional[Variable] = None, relu: bool = False) -> torch.Tensor:
    """tf.matmul(x, kernel) (+ bias) (+ relu) on variables the model created itself with tf.get_variable (AFM's attention
    network, afm.py:168-190): the same kernels as `dense` (one-unit outputs go through the head kernel)."""
    store = current_store()
    if kernel.data.shape[1] == 1 and not relu:
        from . import ops
        xc = x if x.is_contiguous() else x.contiguous()
        if ops.dense1_supported([xc]):
            return _Dense1Fn.apply(store.anchor, kernel, bias, xc)
    return _DenseFn.apply(store.anchor, x, kernel, bias, relu, 0.0, None)


def l2_value(x: torch.Tensor, half_coeff: float) -> torch.Tensor:
    """half_coeff * sum(x^2) as a detached scalar (its gradient is taken care of by `dense(input_l2=)`)."""
    with torch.no_grad():
        v = x.reshape(-1)
        return torch.dot(v, v) * half_coeff


class _BatchNormTrainFn(Function):
    @staticmethod
    def forward(ctx, anchor, x, gamma: Variable, beta: Variable, mmean: Variable, mvar: Variable,
                momentum: float, eps: float, partials: Optional[torch.Tensor] = None):
        ctx.vars = (gamma, beta)
        ctx.hip = x.dim() == 2 and x.is_contiguous() and _hip(x, x.shape[1])
        ctx.x_relu_src = getattr(x, "_recalgo_relu_src", None) if ctx.hip else None
        ctx.x_relu_scale = float(getattr(x, "_recalgo_relu_scale", 1.0)) if ctx.x_relu_src is not None else 1.0
        # Sync-BatchNorm (parallel.attach_data_parallel(sync_batch_norm=True)): statistics over the GLOBAL batch
        ctx.sync = getattr(getattr(anchor, "_recalgo_store", None), "sync_bn", None)
        if ctx.sync is not None and not ctx.hip:
            raise NotImplementedError("sync_batch_norm needs the HIP BatchNorm kernels (2-D contiguous input, C % 4 == 0)")
        if ctx.hip:
            from . import ops
            if ctx.sync is not None:
                y, mean, rstd = ops.batchnorm_sync_fwd(x, gamma.data, beta.data, mmean.data, mvar.data, momentum, eps, ctx.sync)
            else:
                y, mean, rstd = ops.batchnorm_train_fwd(x, gamma.data, beta.data, mmean.data, mvar.data, momentum, eps,
                                                        partials=partials)
                ctx.link = BNLink(x, mean, rstd)       # (attached to the output by batch_normalization())
            ctx.save_for_backward(x, mean, rstd)
            return y
        mean = x.mean(dim=0)
        xc = x - mean
        var = (xc * xc).mean(dim=0)            # biased, tf.nn.moments
        rstd = torch.rsqrt(var + eps)
        xhat = xc * rstd
        y = xhat * gamma.data + beta.data
        # moving stats: assign_moving_average, decay = momentum (TF default 0.99)
        mmean.data.mul_(momentum).add_(mean, alpha=1 - momentum)
        mvar.data.mul_(momentum).add_(var, alpha=1 - momentum)
        ctx.save_for_backward(xhat, rstd)
        return y

    @staticmethod
    def backward(ctx, g):
        gamma, beta = ctx.vars
        if ctx.hip:
            from . import ops
            x, mean, rstd = ctx.saved_tensors
            if ctx.sync is not None:
                dx = ops.batchnorm_sync_bwd(x, gamma.data, mean, rstd, g.contiguous(), gamma.grad, beta.grad, ctx.sync)
            else:
                # x is the ReLU output of a dense layer: this kernel masks the gradient it writes (it reads x anyway) and that
                # layer's backward runs without mask loads (ReluSource)
                src = ctx.x_relu_src
                dx = ops.batchnorm_train_bwd(x, gamma.data, mean, rstd, g.contiguous(), gamma.grad, beta.grad,
                                             sums=ctx.link.take(g), relu_x=src is not None, relu_scale=ctx.x_relu_scale)
                if src is not None:
                    src.premasked = dx
            return None, dx, None, None, None, None, None, None, None, None
        xhat, rstd = ctx.saved_tensors
        B = g.shape[0]
        dbeta = g.sum(dim=0)
        dgamma = (g * xhat).sum(dim=0)
        gamma.grad.copy_(dgamma)
        beta.grad.copy_(dbeta)
        dx = (gamma.data * rstd / B) * (B * g - dbeta - xhat * dgamma)
        return None, dx, None, None, None, None, None, None, None, None


class _BatchNormInferFn(Function):
    @staticmethod
    def forward(ctx, x, scale, shift):
        ctx.save_for_backward(scale)
        return x * scale + shift

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        return g * scale, None, None


def batch_normalization(x: torch.Tensor, training: bool = False,
                        momentum: float = 0.99, epsilon: float = 1e-3,
                        name: Optional[str] = None) -> torch.Tensor:
    """tf.layers.batch_normalization on (B, C)."""
    store = current_store()
    name = name or store.auto_name("batch_normalization")
    C = x.shape[-1]
    with store.variable_scope(name):
        gamma = store.get_variable("gamma", (C,), ones)
        beta = store.get_variable("beta", (C,), zeros)
        mmean = store.get_variable("moving_mean", (C,), zeros, trainable=False)
        mvar = store.get_variable("moving_variance", (C,), ones, trainable=False)
    if training:
        # (batch moments left behind by the producing dense layer's epilogue, see dense(bn_stats=))
        pre = getattr(x, "_recalgo_bn_partials", None) if (x.dim() == 2 and x.is_contiguous()) else None
        return _attach_link(_BatchNormTrainFn.apply(store.anchor, x, gamma, beta, mmean, mvar, momentum, epsilon, pre))
    inv = torch.rsqrt(mvar.data + epsilon) * gamma.data
    return _BatchNormInferFn.apply(x, inv, beta.data - mmean.data * inv)


class _DenseActBNFn(Function):
    """dense -> prelu | dice -> batch_normalization(training=True) as five launches instead of nine: the GEMM's epilogue applies
    the activation and leaves the BatchNorm tile moments (ops.dense_fwd_act), the BatchNorm backward continues through the
    activation (ops.batchnorm_train_bwd_act); same arithmetic per element as the three layers one after the other."""

    @staticmethod
    def forward(ctx, anchor, x, kernel: Variable, bias: Variable, alpha: Variable, kind: int, gamma: Variable, beta: Variable,
                mmean: Variable, mvar: Variable, momentum: float, eps: float, input_l2: float, drop=None):
        from . import ops
        x2 = x if x.stride(1) == 1 else x.contiguous()
        M, N = x2.shape[0], kernel.data.shape[1]
        partials = torch.empty(ops.bn_partial_rows(M), 2 * N, device=x.device, dtype=torch.float32)
        z, y = ops.dense_fwd_act(x2, kernel.data, bias.data, kind, alpha.data, partials)
        # drop: the tf.layers.dropout BEHIND the BatchNorm (din.py:233-236) rides in the BatchNorm's store, and in the loads of its backward
        out, mean, rstd = ops.batchnorm_train_fwd(y, gamma.data, beta.data, mmean.data, mvar.data, momentum, eps, partials=partials,
                                                  out_drop=drop)
        ctx.drop = drop
        # (the next dense layer's backward may leave this BN's sums, see _attach_link — not behind a dropout: they would be sums of
        # the un-dropped gradient)
        ctx.link = BNLink(y, mean, rstd) if drop is None else None
        ctx.in_link = _bn_link_of(x)                                 # (... and this one's those of the BatchNorm before it)
        ctx.vars, ctx.kind, ctx.input_l2 = (kernel, bias, alpha, gamma, beta), kind, float(input_l2)
        ctx.in_step = ops._loss_seed is not None      # Estimator.train_step: its optimizer runs the deferred column sums
        ctx.save_for_backward(x2, z, y, mean, rstd)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import ops
        kernel, bias, alpha, gamma, beta = ctx.vars
        x2, z, y, mean, rstd = ctx.saved_tensors
        dz = ops.batchnorm_train_bwd_act(y, gamma.data, mean, rstd, g.contiguous(), gamma.grad, beta.grad, ctx.kind, z, alpha.data,
                                         alpha.grad, defer=ctx.in_step, sums=None if ctx.link is None else ctx.link.take(g),
                                         g_drop=ctx.drop)
        dx = None
        if ctx.needs_input_grad[1]:
            link = ctx.in_link
            dx = ops.dense_bwd(x2, dz, None, kernel.data, kernel.grad, bias.grad, c_in=x2 if ctx.input_l2 else None,
                               beta=ctx.input_l2, defer=True,
                               bn=None if link is None else (link.x, link.mean, link.rstd, link.new_sums()))
            if link is not None:
                link.grad_ptr = dx.data_ptr()
        else:
            ops.dense_bwd_weights(x2, dz, None, kernel.grad, bias.grad, defer=True)
        return (None, dx) + (None,) * 12


def dense_activation_bn(x: torch.Tensor, units, kind: str, act_name, batch_norm: bool, training: bool,
                        input_l2: float = 0.0, momentum: float = 0.99, epsilon: float = 1e-3, dropout_rate=None) -> torch.Tensor:
    """One hidden layer of DIN's `fcn` scope (/root/reference algorithm/DIN/din.py:227-236):
        net = tf.layers.dense(net, units, activation=None); net = dice | prelu (net, name=act_name)
        if batch_norm: net = tf.layers.batch_normalization(net, training=training)
        if "dropout_rate" in params and 0.0 < rate < 1.0: net = tf.layers.dropout(net, rate, training=training)   (dropout_rate=)
    Variables, scopes and arithmetic are those of the calls; in a training step on the GPU they run as ONE autograd node
    (_DenseActBNFn: the dropout in the BatchNorm's store and in the loads of its backward), otherwise as the separate layers."""
    rate = float(dropout_rate) if dropout_rate is not None else 0.0
    drop_on = bool(training) and 0.0 < rate < 1.0
    from . import ops
    store = current_store()
    units = int(units)
    dname = store.auto_name("dense")
    fused = bool(batch_norm and training and not store.building and x.dim() == 2 and x.is_cuda and x.dtype == torch.float32
                 and units % 4 == 0 and _mfma_dense(x.shape[1]) and torch.is_grad_enabled()
                 and getattr(getattr(store.anchor, "_recalgo_store", None), "sync_bn", None) is None)
    if not fused:
        net = dense(x, units, activation=None, name=dname, input_l2=input_l2)
        alpha = store.get_variable(f"{kind}_alpha_{act_name}", (units,), ones)
        net = ops.activation(store, net, alpha, kind)
        net = batch_normalization(net, training=training, momentum=momentum, epsilon=epsilon) if batch_norm else net
        return dropout(net, rate, training=training) if 0.0 < rate < 1.0 else net
    with store.variable_scope(dname):
        kernel = store.get_variable("kernel", (x.shape[-1], units), glorot_uniform)
        bias = store.get_variable("bias", (units,), zeros)
    alpha = store.get_variable(f"{kind}_alpha_{act_name}", (units,), ones)
    with store.variable_scope(store.auto_name("batch_normalization")):
        gamma = store.get_variable("gamma", (units,), ones)
        beta = store.get_variable("beta", (units,), zeros)
        mmean = store.get_variable("moving_mean", (units,), zeros, trainable=False)
        mvar = store.get_variable("moving_variance", (units,), ones, trainable=False)
    d = drop_spec((x.shape[0], units), rate, x.device) if (drop_on and x.shape[0] * units < (1 << 32)) else None
    out = _attach_link(_DenseActBNFn.apply(store.anchor, x, kernel, bias, alpha, ops._ACT[kind], gamma, beta, mmean, mvar, momentum,
                                           epsilon, input_l2, d))
    return dropout(out, rate, training=training) if (drop_on and d is None) else out


DROPOUT_KEEP_MASKS: list = []      # test hook: keep masks consumed (FIFO) by the next training-mode dropout calls
DROPOUT_SPECS: list = []           # test hook: when not None, every training-mode dropout call appends its ops.DropSpec


def drop_spec(x_shape, rate: float, device):
    """The ops.DropSpec of the next training-mode dropout call of the current model_fn invocation: an injected keep mask
    (DROPOUT_KEEP_MASKS), else the hash stream (store seed + rank, call index, the optimizer's device step counter)."""
    from . import ops
    store = current_store()
    call = store._drop_calls
    store._drop_calls += 1
    mask = None
    if DROPOUT_KEEP_MASKS:
        mask = DROPOUT_KEEP_MASKS.pop(0).to(device=device, dtype=torch.float32).reshape(tuple(x_shape)).contiguous()
    rank = 0                       # (data parallel: every rank drops its own slice of the global batch with its own stream)
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank = dist.get_rank()
    except Exception:
        pass
    d = ops.DropSpec(rate, mask, store.seed * 1000003 + 7919 * rank, call, store.ensure_opt_state()["step"])
    if DROPOUT_SPECS is not None:
        DROPOUT_SPECS.append(d)
        del DROPOUT_SPECS[:-64]
    return d


def dropout(x: torch.Tensor, rate: float, training: bool = False) -> torch.Tensor:
    """tf.layers.dropout: keep prob 1-rate, scaled by 1/(1-rate); identity when not training.  TF's random stream
    cannot be reproduced: parity tests inject the keep mask the golden recorded through DROPOUT_KEEP_MASKS; otherwise the
    keep decisions are the counter-based hash of csrc/dropout.h (a new mask per optimizer step, also under hipGraph replay)."""
    if not training or rate <= 0.0:
        return x
    store = current_store()
    if store.building:
        return x
    from . import ops
    return ops.dropout(x, drop_spec(x.shape, rate, x.device))


class _L2RegFn(Function):
    @staticmethod
    def forward(ctx, anchor, scale: float, variables):
        ctx.scale, ctx.vars = scale, variables
        return sum((v.data * v.data).sum() for v in variables) * (0.5 * scale)

    @staticmethod
    def backward(ctx, g):
        # runs before the kernels that OVERWRITE these variables' grads?  No: the parameter-owning
        # ops overwrite .grad during their own backward, in autograd order.  To be order independent
        # the regulariser's contribution is parked and added by `apply_parked_grads` (called by the
        # optimizer before the update).
        for v in ctx.vars:
            _PARKED.append((v, ctx.scale, g))
        return None, None, None


_PARKED = []


def apply_parked_grads(step_dev=None):
    """Finish the gradients that backward left parked (called once per step, after backward, by the optimizer and by
    `variables.named_grads`): the deferred weight-gradient split sums of `dense` (ops.flush_dense_splits), then
    grad += scale * g * w for every parked l2 term."""
    from . import ops
    ops.flush_dense_splits(step_dev)
    while _PARKED:
        v, scale, g = _PARKED.pop()
        v.grad.add_(v.data * (scale * g))


def l2_regularization(scale: float, variables) -> torch.Tensor:
    """sum_v tf.contrib.layers.l2_regularizer(scale)(v) = scale * sum(v^2) / 2."""
    return _L2RegFn.apply(current_store().anchor, float(scale), tuple(variables))
