"""oracle/tf1_shim/tensorflow — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

A minimal, eager, torch-CPU stand-in for the part of the TensorFlow-1.14 API that the
reference's hot-path files call (listed by `grep -o "tf\\.[A-Za-z_.]*"` over
/root/reference/algorithm/{DeepFM,DCN,xDeepFM,DIN,FiBiNET,PNN} and utils.py).  Its only purpose
is to let `oracle/gen_golden.py` import and execute the reference's OWN, UNMODIFIED sources
(`import tensorflow as tf` resolves here when oracle/tf1_shim is first on sys.path) so that the
golden vectors in tests/golden/ pin the reference's *composition* of primitives: which op, in
which order, on which axis, under which variable name, with which quirk.

What it does NOT pin: TensorFlow's kernels themselves.  Each primitive below restates the
documented TF-1.14 behaviour ([TF-ext]); TF cannot be installed here.

Execution model: every tf.Tensor is a `T` wrapping a torch tensor (float64 by default so the
vectors are the exact-arithmetic anchor for both the fp32 HIP kernels and the fp32/fp64
oracle); variables are torch leaves, gradients come from torch.autograd when the reference's
`optimizer.minimize(loss)` train_op is run.
"""
from __future__ import annotations

import collections
import contextlib
import itertools
import math
import types

import torch

FLOAT = torch.float64          # dtype standing in for tf.float32 (see module docstring)

float32 = "float32"
float64 = "float64"
int32 = "int32"
int64 = "int64"
string = "string"
bool = "bool"  # noqa: A001  (tf.bool)
AUTO_REUSE = "AUTO_REUSE"
Tensor = None  # replaced below (tf.Tensor is used in type comments only)


# ---------------------------------------------------------------------------------------------
# shapes
# ---------------------------------------------------------------------------------------------
class Dimension(int):
    @property
    def value(self):
        return int(self)


class TensorShape(tuple):
    def as_list(self):
        return [int(d) for d in self]

    def __getitem__(self, i):
        r = tuple.__getitem__(self, i)
        return TensorShape(r) if isinstance(i, slice) else r


def _shape_of(t):
    return TensorShape(Dimension(s) for s in t.shape)


def _raw(x):
    return x.t if isinstance(x, T) else x


def _int(x):
    if isinstance(x, T):
        return int(x.t.item())
    return int(x)


# ---------------------------------------------------------------------------------------------
# tensors
# ---------------------------------------------------------------------------------------------
class T:
    __array_priority__ = 100

    def __init__(self, t, name=None):
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(t)
        if t.is_floating_point() and t.dtype != FLOAT:
            t = t.to(FLOAT)
        self.t = t
        self.name = name

    shape = property(lambda self: _shape_of(self.t))
    dtype = property(lambda self: self.t.dtype)

    def get_shape(self):
        return _shape_of(self.t)

    def numpy(self):
        return self.t.detach().cpu().numpy()

    def _b(self, other, fn, rev=False):
        o = _raw(other)
        if not isinstance(o, torch.Tensor):
            o = torch.as_tensor(o, dtype=self.t.dtype if self.t.is_floating_point() else None)
        return T(fn(o, self.t) if rev else fn(self.t, o))

    __add__ = lambda s, o: s._b(o, torch.add)
    __radd__ = lambda s, o: s._b(o, torch.add, True)
    __sub__ = lambda s, o: s._b(o, torch.sub)
    __rsub__ = lambda s, o: s._b(o, torch.sub, True)
    __mul__ = lambda s, o: s._b(o, torch.mul)
    __rmul__ = lambda s, o: s._b(o, torch.mul, True)
    __truediv__ = lambda s, o: s._b(o, torch.div)
    __rtruediv__ = lambda s, o: s._b(o, torch.div, True)
    __neg__ = lambda s: T(-s.t)
    __pow__ = lambda s, o: T(s.t ** _raw(o))

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        idx = tuple(_int(i) if isinstance(i, (T, Dimension)) else i for i in idx)
        return T(self.t[idx])

    def __repr__(self):
        return f"<tf1_shim.T shape={tuple(self.t.shape)} dtype={self.t.dtype} name={self.name}>"


Tensor = T


class SparseTensor:
    def __init__(self, indices, values, dense_shape):
        self.indices, self.values, self.dense_shape = indices, values, dense_shape


class Variable(T):
    def __init__(self, t, name, trainable=True):
        super().__init__(t, name)
        self.t.requires_grad_(trainable)
        self.trainable = trainable


# ---------------------------------------------------------------------------------------------
# graph state: variables, scopes, collections, auto-naming
# ---------------------------------------------------------------------------------------------
class GraphKeys:
    UPDATE_OPS = "update_ops"
    REGULARIZATION_LOSSES = "regularization_losses"


class _Graph:
    def __init__(self, seed=0):
        self.vars = collections.OrderedDict()
        self.scope = []
        self.uid = collections.Counter()
        self.collections = collections.defaultdict(list)
        self.gen = torch.Generator().manual_seed(seed)
        self.shared_tables = {}
        self.layer_outputs = collections.OrderedDict()     # name -> T, for golden dumps


_G = _Graph()


def reset_default_graph(seed=0):
    global _G
    _G = _Graph(seed)


def get_default_graph():
    return _G


def _scope_name():
    return "/".join(_G.scope)


def _unique(base):
    """TF1 default_name uniquification inside the current variable scope: base, base_1, ..."""
    key = (_scope_name(), base)
    n = _G.uid[key]
    _G.uid[key] += 1
    return base if n == 0 else f"{base}_{n}"


@contextlib.contextmanager
def variable_scope(name, reuse=None, default_name=None, **_kw):
    _G.scope.append(name if name is not None else _unique(default_name))
    try:
        yield
    finally:
        _G.scope.pop()


name_scope = variable_scope


def glorot_uniform_initializer():
    def init(shape):
        shape = tuple(shape)
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
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(*shape, generator=_G.gen, dtype=FLOAT) * 2 - 1) * lim
    return init


def constant_initializer(value):
    return lambda shape: torch.full(tuple(shape), float(value), dtype=FLOAT)


def zeros_initializer():
    return constant_initializer(0.0)


def ones_initializer():
    return constant_initializer(1.0)


def truncated_normal_initializer(mean=0.0, stddev=1.0):
    def init(shape):
        out = torch.randn(*shape, generator=_G.gen, dtype=FLOAT)
        for _ in range(16):
            bad = out.abs() > 2
            if not bad.any():
                break
            out = torch.where(bad, torch.randn(*shape, generator=_G.gen, dtype=FLOAT), out)
        return out.clamp(-2, 2) * stddev + mean
    return init


def _norm_shape(shape):
    if shape is None:
        return ()
    if isinstance(shape, (int, Dimension)):
        return (int(shape),)
    return tuple(int(s) for s in shape)


def get_variable(name, shape=None, dtype=None, initializer=None, regularizer=None, trainable=True, **_kw):
    """tf.get_variable: default initializer glorot_uniform [TF-ext A-7]; an existing name is
    returned as is (the only reuse in the reference is AUTO_REUSE, din_attention.py:21-23)."""
    full = "/".join(_G.scope + [name])
    v = _G.vars.get(full)
    if v is None:
        init = initializer or glorot_uniform_initializer()
        v = Variable(init(_norm_shape(shape)).clone(), full, trainable)
        _G.vars[full] = v
    # a regulariser is attached when the variable is created in a graph; gen_golden.py re-runs
    # model_fn per mode on the same variables ("new graph" = cleared collections), so the
    # registration is per (graph pass, variable)
    if regularizer is not None and full not in _G.collections["__regularized__"]:
        _G.collections["__regularized__"].append(full)
        r = regularizer(v)
        if r is not None:
            _G.collections[GraphKeys.REGULARIZATION_LOSSES].append(r)
    return v


def get_collection(key, scope=None):
    return list(_G.collections[key])


def global_variables_initializer():
    return None


@contextlib.contextmanager
def control_dependencies(_ops):
    yield


# ---------------------------------------------------------------------------------------------
# math
# ---------------------------------------------------------------------------------------------
def _t(x):
    return x if isinstance(x, T) else T(torch.as_tensor(x))


def constant(value, dtype=None, shape=None, name=None):
    return T(torch.as_tensor(value))


def zeros(shape, dtype=None):
    return T(torch.zeros(*_norm_shape(shape), dtype=FLOAT))


def ones_like(x):
    return T(torch.ones_like(_raw(x)))


def random_normal(shape, **_kw):
    return T(torch.randn(*_norm_shape(shape), generator=_G.gen, dtype=FLOAT))


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    A, B = _raw(a), _raw(b)
    if transpose_a:
        A = A.transpose(-1, -2)
    if transpose_b:
        B = B.transpose(-1, -2)
    return T(torch.matmul(A, B))


def multiply(a, b, name=None):
    return _t(a) * b


def scalar_mul(scalar, x, name=None):
    """tf.scalar_mul: scalar * x (fwfm.py:156)."""
    return T(_raw(scalar) * _raw(x))


def _batch_dot(x, y, axes=None):
    """[TF-ext] tf.keras.backend.batch_dot on two (B, K) tensors with axes=1: the per-example inner
    product, kept 2-D -> (B, 1) (Keras expands a rank-1 result; fwfm.py:157)."""
    a, b = _raw(x), _raw(y)
    if a.dim() != 2 or b.dim() != 2 or axes not in (1, (1, 1), [1, 1]):
        raise NotImplementedError("tf1_shim: batch_dot is restated for (B, K) x (B, K), axes=1 only")
    return T((a * b).sum(dim=1, keepdim=True))


def add(a, b, name=None):
    return _t(a) + b


def add_n(xs, name=None):
    xs = list(xs)
    out = _raw(xs[0])
    for x in xs[1:]:           # sequential accumulation in list order
        out = out + _raw(x)
    return T(out)


def square(x, name=None):
    return T(_raw(x) * _raw(x))


def sigmoid(x, name=None):
    return T(torch.sigmoid(_raw(x)))


def maximum(a, b):
    a, b = _raw(a), _raw(b)
    ref = a if isinstance(a, torch.Tensor) else b
    return T(torch.maximum(torch.as_tensor(a, dtype=ref.dtype), torch.as_tensor(b, dtype=ref.dtype)))


def minimum(a, b):
    a, b = _raw(a), _raw(b)
    ref = a if isinstance(a, torch.Tensor) else b
    return T(torch.minimum(torch.as_tensor(a, dtype=ref.dtype), torch.as_tensor(b, dtype=ref.dtype)))


def _axes(axis):
    if axis is None:
        return None
    if isinstance(axis, (list, tuple)):
        return tuple(int(a) for a in axis)
    return int(axis)


def reduce_sum(x, axis=None, keepdims=False, name=None):
    x = _raw(x)
    return T(x.sum() if axis is None else x.sum(dim=_axes(axis), keepdim=keepdims))


def reduce_mean(x, axis=None, keepdims=False, name=None):
    x = _raw(x)
    return T(x.mean() if axis is None else x.mean(dim=_axes(axis), keepdim=keepdims), name)


def concat(values, axis, name=None):
    return T(torch.cat([_raw(v) for v in values], dim=int(axis)))


def stack(values, axis=0, name=None):
    return T(torch.stack([_raw(v) for v in values], dim=int(axis)))


def reshape(x, shape, name=None):
    return T(_raw(x).reshape(tuple(_int(s) for s in shape)))


def transpose(x, perm=None, name=None):
    x = _raw(x)
    if perm is None:
        perm = list(reversed(range(x.dim())))
    return T(x.permute(*[int(p) for p in perm]))


def expand_dims(x, axis, name=None):
    return T(_raw(x).unsqueeze(int(axis)))


def squeeze(x, axis=None, name=None):
    x = _raw(x)
    return T(x.squeeze() if axis is None else x.squeeze(int(axis)))


def tile(x, multiples, name=None):
    return T(_raw(x).repeat(*[_int(m) for m in multiples]))


def einsum(equation, *inputs):
    return T(torch.einsum(equation.replace(" ", ""), *[_raw(i) for i in inputs]))


class _ShapeVec(list):
    """tf.shape(x): indexable, elements usable as ints."""


def _safe_embedding_lookup_sparse(embedding_weights, sparse_ids, sparse_weights=None, combiner="mean", **_kw):
    """[TF-ext A-3] tf.nn.safe_embedding_lookup_sparse on a rank-2 SparseTensor of ids: per row, the
    `combiner` (default mean: sequential sum in id order / count) of the looked-up rows; rows without ids
    give zeros; ids < 0 are dropped (ffm.py:156-157)."""
    if sparse_weights is not None or combiner != "mean":
        raise NotImplementedError("tf1_shim: safe_embedding_lookup_sparse is restated for combiner='mean', no weights")
    w = _raw(embedding_weights)
    idx, vals = _raw(sparse_ids.indices), _raw(sparse_ids.values)
    B = int(_raw(sparse_ids.dense_shape)[0]) if not isinstance(sparse_ids.dense_shape, (list, tuple)) else int(sparse_ids.dense_shape[0])
    out = torch.zeros(B, w.shape[1], dtype=w.dtype)
    cnt = torch.zeros(B, dtype=w.dtype)
    rows = []
    for n in range(vals.shape[0]):          # in SparseTensor order (row-major): the sum order of SparseSegmentMean
        if int(vals[n]) < 0:
            continue
        rows.append((int(idx[n, 0]), int(vals[n])))
    acc = [None] * B
    for b, v in rows:
        acc[b] = w[v] if acc[b] is None else acc[b] + w[v]
        cnt[b] += 1
    parts = [torch.zeros(w.shape[1], dtype=w.dtype) if a is None else a / cnt[b] for b, a in enumerate(acc)]
    return T(torch.stack(parts, 0))


def shape(x, name=None, out_type=None):
    return _ShapeVec(int(s) for s in _raw(x).shape)


def cast(x, dtype, name=None):
    x = _raw(x)
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(x)
    if dtype in (float32, float64):
        return T(x.to(FLOAT))
    if dtype in (int32, int64):
        return T(x.to(torch.int64))
    return T(x)


def to_float(x, name=None):
    return cast(x, float32)


def greater_equal(a, b):
    return T(_raw(a) >= _raw(b))


def not_equal(a, b):
    return T(_raw(a) != _raw(b))


def where(condition, x=None, y=None):
    if x is None and y is None:          # coordinates of the true elements, row-major, int64 (utils.py:58)
        return T(torch.nonzero(_raw(condition)))
    return T(torch.where(_raw(condition), _raw(x), _raw(y)))


def sequence_mask(lengths, maxlen=None, dtype=None):
    """[TF-ext] mask[b, t] = t < lengths[b]."""
    L = _raw(lengths).reshape(-1).to(torch.int64)
    n = _int(maxlen) if maxlen is not None else int(L.max().item())
    return T(torch.arange(n).unsqueeze(0) < L.unsqueeze(1))


def matrix_band_part(x, num_lower, num_upper):
    """[TF-ext] keep the band: (0, -1) upper triangle, (0, 0) diagonal."""
    x = _raw(x)
    n, m = x.shape[-2], x.shape[-1]
    i = torch.arange(n).unsqueeze(1)
    j = torch.arange(m).unsqueeze(0)
    keep = ((num_lower < 0) | ((i - j) <= num_lower)) & ((num_upper < 0) | ((j - i) <= num_upper))
    return T(torch.where(keep, x, torch.zeros_like(x)))


# ---------------------------------------------------------------------------------------------
# tf.nn
# ---------------------------------------------------------------------------------------------
def _relu(x, name=None):
    return T(torch.relu(_raw(x)))


def _softmax(x, axis=-1, name=None):
    return T(torch.softmax(_raw(x), dim=int(axis)))


def _sigmoid_ce(labels=None, logits=None, name=None):
    """[TF-ext A-9] max(x,0) - x*z + log(1 + exp(-|x|))."""
    x, z = _raw(logits), _raw(labels).to(FLOAT)
    return T(torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-x.abs())))


def _l2_loss(x, name=None):
    x = _raw(x)
    return T((x * x).sum() / 2)


def _conv1d(value, filters, stride, padding, name=None):
    """[TF-ext A-11] width-1 filter, stride 1, VALID: (B, W, C) x (1, C, N) -> (B, W, N)."""
    v, f = _raw(value), _raw(filters)
    assert f.shape[0] == 1 and stride == 1 and padding == "VALID", "only the reference's use is restated"
    return T(torch.matmul(v, f[0]))


nn = types.SimpleNamespace(relu=_relu, softmax=_softmax, sigmoid=sigmoid,
                           sigmoid_cross_entropy_with_logits=_sigmoid_ce, l2_loss=_l2_loss, conv1d=_conv1d,
                           safe_embedding_lookup_sparse=_safe_embedding_lookup_sparse)


# ---------------------------------------------------------------------------------------------
# tf.layers
# ---------------------------------------------------------------------------------------------
def _dense(inputs, units, activation=None, use_bias=True, name=None, reuse=None, **_kw):
    """tf.layers.dense: <scope>/<name>/{kernel,bias}; kernel glorot-uniform, bias zeros; `units`
    may be a str (TF calls int() on it — the reference passes split(',') strings, quirk B-2)."""
    x = _raw(inputs)
    units = int(units)
    lname = name if name is not None else _unique("dense")
    with variable_scope(lname):
        kernel = get_variable("kernel", (x.shape[-1], units))
        bias = get_variable("bias", (units,), initializer=zeros_initializer()) if use_bias else None
    y = torch.matmul(x, kernel.t)
    if bias is not None:
        y = y + bias.t
    out = T(y)
    return activation(out) if activation is not None else out


def _batch_normalization(inputs, training=False, center=True, scale=True, momentum=0.99, epsilon=1e-3,
                         name=None, **_kw):
    """[TF-ext A-8] moving_mean 0 / moving_variance 1; training=False when omitted (Dice relies
    on that, activations.py:34); training: biased batch variance, moving stats updated through
    the UPDATE_OPS collection."""
    x = _raw(inputs)
    C = x.shape[-1]
    lname = name if name is not None else _unique("batch_normalization")
    with variable_scope(lname):
        gamma = get_variable("gamma", (C,), initializer=ones_initializer()) if scale else None
        beta = get_variable("beta", (C,), initializer=zeros_initializer()) if center else None
        mm = get_variable("moving_mean", (C,), initializer=zeros_initializer(), trainable=False)
        mv = get_variable("moving_variance", (C,), initializer=ones_initializer(), trainable=False)
    if training:
        red = tuple(range(x.dim() - 1))
        mean = x.mean(dim=red)
        var = ((x - mean) ** 2).mean(dim=red)

        def update(mean=mean.detach(), var=var.detach()):
            with torch.no_grad():
                mm.t.mul_(momentum).add_(mean * (1 - momentum))
                mv.t.mul_(momentum).add_(var * (1 - momentum))
        _G.collections[GraphKeys.UPDATE_OPS].append(update)
    else:
        mean, var = mm.t, mv.t
    y = (x - mean) * torch.rsqrt(var + epsilon)
    if gamma is not None:
        y = y * gamma.t
    if beta is not None:
        y = y + beta.t
    return T(y)


def _dropout(inputs, rate=0.5, training=False, **_kw):
    """[TF-ext A-8] tf.layers.dropout: identity unless training; in training each element is kept with
    probability 1 - rate and scaled by 1 / (1 - rate).  TF's random stream cannot be reproduced, so the keep mask
    is drawn from a generator seeded per call and RECORDED (graph collection "__dropout_masks__", call order):
    gen_golden.py stores the masks next to the outputs and the oracle / mirror are checked with the same masks."""
    if not (training and 0.0 < rate < 1.0):
        return inputs
    x = _raw(inputs)
    masks = _G.collections["__dropout_masks__"]
    g = torch.Generator().manual_seed(977 * 1000003 + len(masks))
    keep = (torch.rand(x.shape, generator=g, dtype=torch.float64) >= rate).to(x.dtype)
    masks.append(keep)
    return T(x * keep / (1.0 - rate))


def _flatten(inputs, name=None):
    x = _raw(inputs)
    return T(x.reshape(x.shape[0], -1))


layers = types.SimpleNamespace(dense=_dense, batch_normalization=_batch_normalization, dropout=_dropout,
                               flatten=_flatten)


# ---------------------------------------------------------------------------------------------
# tf.metrics / tf.train / tf.estimator / misc namespaces
# ---------------------------------------------------------------------------------------------
def _accuracy(labels, predictions):
    v = T((_raw(labels) == _raw(predictions)).to(FLOAT).mean())
    return (v, v)


def _auc(labels, predictions, num_thresholds=200):
    """[TF-ext A-9] 200-threshold trapezoidal ROC AUC."""
    eps = 1e-7
    p = _raw(predictions).detach().reshape(1, -1)
    y = _raw(labels).detach().reshape(1, -1) > 0.5
    th = torch.tensor([0.0 - eps] + [(i + 1) / (num_thresholds - 1) for i in range(num_thresholds - 2)] + [1.0 + eps],
                      dtype=p.dtype).unsqueeze(1)
    pp = p > th
    tp, fp = (pp & y).sum(1).double(), (pp & ~y).sum(1).double()
    fn, tn = (~pp & y).sum(1).double(), (~pp & ~y).sum(1).double()
    tpr, fpr = (tp + eps) / (tp + fn + eps), fp / (fp + tn + eps)
    v = T(((fpr[:-1] - fpr[1:]) * (tpr[:-1] + tpr[1:]) / 2).sum())
    return (v, v)


metrics = types.SimpleNamespace(accuracy=_accuracy, auc=_auc)


class _TrainOp:
    """optimizer.minimize(loss): run() = autodiff + UPDATE_OPS + TF1 Adam on every trainable
    variable ([TF-ext A-10]: dense update, also for embedding tables)."""

    def __init__(self, opt, loss):
        self.opt, self.loss = opt, loss
        self.grads = None

    def run(self):
        tv = [v for v in _G.vars.values() if v.trainable]
        for v in tv:
            v.t.grad = None
        _raw(self.loss).backward()
        self.grads = {v.name: (torch.zeros_like(v.t) if v.t.grad is None else v.t.grad.clone()) for v in tv}
        for u in _G.collections[GraphKeys.UPDATE_OPS]:
            u()
        o = self.opt
        o.step += 1
        f32 = lambda x: float(torch.tensor(x, dtype=torch.float32))     # TF keeps hyper-parameters as float32
        lr, b1, b2, eps = f32(o.lr), f32(o.beta1), f32(o.beta2), f32(o.eps)
        lr_t = lr * math.sqrt(1 - b2 ** o.step) / (1 - b1 ** o.step)
        with torch.no_grad():
            for v in tv:
                g = self.grads[v.name]
                m = o.m.setdefault(v.name, torch.zeros_like(v.t))
                s = o.v.setdefault(v.name, torch.zeros_like(v.t))
                m.mul_(b1).add_(g * (1 - b1))
                s.mul_(b2).add_(g * g * (1 - b2))
                v.t.sub_(lr_t * m / (s.sqrt() + eps))
        return self.grads


class _Adam:
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, **_kw):
        self.lr, self.beta1, self.beta2, self.eps = learning_rate, beta1, beta2, epsilon
        self.step, self.m, self.v = 0, {}, {}

    def minimize(self, loss, global_step=None, **_kw):
        return _TrainOp(self, loss)


class _Hook:
    def __init__(self, *a, **k):
        self.args, self.kwargs = a, k


train = types.SimpleNamespace(AdamOptimizer=_Adam, get_global_step=lambda: None, LoggingTensorHook=_Hook,
                              ProfilerHook=_Hook)


class _ModeKeys:
    TRAIN, EVAL, PREDICT = "train", "eval", "infer"


class _EstimatorSpec:
    def __init__(self, mode, predictions=None, loss=None, train_op=None, eval_metric_ops=None,
                 export_outputs=None, training_hooks=None, **_kw):
        self.mode, self.predictions, self.loss, self.train_op = mode, predictions, loss, train_op
        self.eval_metric_ops, self.export_outputs, self.training_hooks = eval_metric_ops, export_outputs, training_hooks


def _not_restated(name):
    def f(*a, **k):
        raise NotImplementedError(f"tf1_shim: {name} (driver / IO plumbing) is not restated")
    return f


estimator = types.SimpleNamespace(
    ModeKeys=_ModeKeys, EstimatorSpec=_EstimatorSpec,
    export=types.SimpleNamespace(PredictOutput=lambda outputs: outputs,
                                 build_parsing_serving_input_receiver_fn=_not_restated("build_parsing_serving_input_receiver_fn")),
    Estimator=_not_restated("Estimator"), RunConfig=_not_restated("RunConfig"), TrainSpec=_not_restated("TrainSpec"),
    EvalSpec=_not_restated("EvalSpec"), BestExporter=_not_restated("BestExporter"),
    train_and_evaluate=_not_restated("train_and_evaluate"))

summary = types.SimpleNamespace(scalar=lambda *a, **k: None)
logging = types.SimpleNamespace(set_verbosity=lambda *_: None, INFO=20)
data = types.SimpleNamespace(TFRecordDataset=_not_restated("TFRecordDataset"),
                             experimental=types.SimpleNamespace(AUTOTUNE=-1))
parse_example = _not_restated("parse_example")
Session = _not_restated("Session")


def _l2_regularizer(scale, scope=None):
    """[TF-ext] tf.contrib.layers.l2_regularizer: scale 0.0 -> no regulariser; else scale * l2_loss(w)."""
    if float(scale) == 0.0:
        return lambda _w: None
    return lambda w: T(float(scale) * (w.t * w.t).sum() / 2)


# ---------------------------------------------------------------------------------------------
# tf.app.flags
# ---------------------------------------------------------------------------------------------
class _FlagValues:
    def __init__(self):
        object.__setattr__(self, "_d", {})

    def __getattr__(self, k):
        try:
            return self._d[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self._d[k] = v


class _Flags:
    FLAGS = _FlagValues()

    @classmethod
    def _define(cls, name, default, *_a, **_k):
        cls.FLAGS._d.setdefault(name, default)

    DEFINE_string = DEFINE_integer = DEFINE_float = DEFINE_boolean = DEFINE_bool = _define

    @classmethod
    def DEFINE_enum(cls, name, default, enum_values, *_a, **_k):
        cls.FLAGS._d.setdefault(name, default)


app = types.SimpleNamespace(flags=_Flags, run=lambda *a, **k: None)
keras = types.SimpleNamespace(backend=types.SimpleNamespace(batch_dot=_batch_dot))

from . import feature_column  # noqa: E402
from .feature_column import sequence_input_layer as _sequence_input_layer  # noqa: E402

contrib = types.SimpleNamespace(layers=types.SimpleNamespace(l2_regularizer=_l2_regularizer),
                                feature_column=types.SimpleNamespace(sequence_input_layer=_sequence_input_layer))
compat = types.SimpleNamespace(v1=types.SimpleNamespace(get_collection=get_collection))
