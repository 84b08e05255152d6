"""Independent pins of the [TF-ext] primitives ("verify each with a tiny independent
derivation in tests").

TensorFlow 1.14 cannot run here, so `oracle/ref_ops.py` and the stand-in `oracle/tf1_shim` both RESTATE TF's
documented primitive semantics — and share one author.  This file is the third, independent leg:

  PART 1  known answers worked out by hand from the TF-1.14 documentation of each primitive, written as
          plain Python / numpy literals and brute-force loops.  It imports NOTHING from oracle/ or the shim.
  PART 2  the same inputs pushed through (a) oracle/ref_ops.py and (b) oracle/tf1_shim, each compared with
          the PART-1 answers.

Covered: A-3 (mean combiner: duplicates, OOV pruning, all-OOV, empty), A-5 (indicator counts), A-6 (sequence
length counts OOV entries, zero padding), A-7 (initialiser bounds), A-8 (BatchNorm biased variance, moving
update, inference default; dropout scaling), A-9 (sigmoid-CE formula, 200-threshold AUC incl. a case where it
differs from the exact AUC), A-10 (Adam: eps outside the bias correction, step-1 and step-2 values, dense
decay of untouched rows), A-11 (conv1d with a width-1 filter == per-position matmul; CIN index order i*m+j).
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# =====================================================================================================
# PART 1 — hand-derived known answers (no oracle / shim imports above or inside this part)
# =====================================================================================================
TABLE = np.array([[1., 2.], [3., 4.], [5., 6.], [7., 8.], [9., 10.]])       # 5 ids x 2

# A-3  embedding_column(combiner='mean') -> safe_embedding_lookup_sparse: "invalid ids (< 0) are pruned",
# "entries with no features ... get the zero vector", mean = sum of the looked-up rows / number of them.
BAGS = [[0], [1, 3], [2, 2, 4], [-1, -1], [], [-1, 4]]
BAG_MEANS = np.array([
    [1., 2.],                                 # single id: the row itself
    [(3 + 7) / 2, (4 + 8) / 2],               # = [5, 6]
    [(5 + 5 + 9) / 3, (6 + 6 + 10) / 3],      # duplicates count twice: [19/3, 22/3]
    [0., 0.],                                 # all OOV -> pruned -> empty -> zeros
    [0., 0.],                                 # empty
    [9., 10.],                                # the OOV entry is not counted in the denominator
])

# A-5  indicator_column: multi-hot COUNTS (sparse -> dense with -1 fill -> one_hot -> sum over the value axis);
# first-order logit = multi_hot @ kernel + bias
IND_VOCAB = 4
IND_BAGS = [[1], [1, 1, 3], [-1], []]
IND_MULTI_HOT = np.array([[0, 1, 0, 0], [0, 2, 0, 1], [0, 0, 0, 0], [0, 0, 0, 0]], dtype=np.float64)
IND_KERNEL = np.array([10., 20., 30., 40.])
IND_BIAS = 0.5
IND_LOGIT = np.array([20.5, 2 * 20 + 40 + 0.5, 0.5, 0.5])

# A-6  sequence_input_layer: (B, T, H) zero padded; sequence_length = number of entries, OOV entries included
SEQS = [[1, -1, 3], [], [0]]
SEQ_T = 3
SEQ_OUT = np.array([[[3., 4.], [0., 0.], [7., 8.]],
                    [[0., 0.], [0., 0.], [0., 0.]],
                    [[1., 2.], [0., 0.], [0., 0.]]])
SEQ_LEN = np.array([3, 0, 1])

# A-8  tf.layers.batch_normalization(training=True): batch mean, BIASED variance (tf.nn.moments), eps 1e-3;
# moving <- moving * 0.99 + batch * 0.01 from (0, 1); training omitted -> inference with the moving stats
BN_X = np.array([[1., 10.], [3., 30.]])
BN_GAMMA, BN_BETA = np.array([2., 1.]), np.array([0.5, 0.])
BN_MEAN, BN_VAR = np.array([2., 20.]), np.array([1., 100.])                  # unbiased would be [2, 200]
BN_Y_TRAIN = np.array([[0.5 - 2 / math.sqrt(1.001), -10 / math.sqrt(100.001)],
                       [0.5 + 2 / math.sqrt(1.001), +10 / math.sqrt(100.001)]])
BN_MOVING_MEAN_AFTER = np.array([0.02, 0.2])                                 # 0 * .99 + mean * .01
BN_MOVING_VAR_AFTER = np.array([0.99 + 0.01, 0.99 + 1.0])                    # 1 * .99 + var * .01
BN_Y_INFER = BN_X / math.sqrt(1.001) * BN_GAMMA + BN_BETA                    # moving stats (0, 1)

# A-9  sigmoid_cross_entropy_with_logits == -[z log s(x) + (1 - z) log(1 - s(x))], the textbook definition
CE_X = np.array([0., 2., -3., -3., 30., -30.])
CE_Z = np.array([1., 1., 0., 1., 0., 1.])


def _ce_textbook(x, z):
    s = 1.0 / (1.0 + np.exp(-x))
    return -(z * np.log(s) + (1 - z) * np.log1p(-s))


CE_LOSS = _ce_textbook(CE_X[:4], CE_Z[:4])           # |x| <= 3: the naive form is accurate in fp64
CE_LOSS_BIG = np.array([30. + math.log1p(math.exp(-30.)), 30. + math.log1p(math.exp(-30.))])   # x=30,z=0 / x=-30,z=1

# tf.metrics.auc(num_thresholds=200, trapezoidal): ROC sampled at thresholds {-eps, 1/199 .. 198/199, 1+eps}
AUC_CASES = [
    # (labels, predictions, known 200-threshold AUC, exact pairwise AUC)
    ([1, 0], [0.8, 0.3], 1.0, 1.0),                              # perfect separation
    ([1, 0, 1, 0], [0.9, 0.6, 0.4, 0.1], 0.75, 0.75),            # ROC (0,.5) (.5,.5) (.5,1) (1,1): .25 + .5
    # both scores fall between the thresholds 99/199 = .4975 and 100/199 = .5025: they flip TOGETHER, the ROC
    # jumps (0,0) -> (1,1) and the trapezoid gives 1/2, although the exact AUC is 1
    ([1, 0], [0.5012, 0.5010], 0.5, 1.0),
]

# A-10  tf.train.AdamOptimizer with float32-exact (dyadic) hyper-parameters, so that nothing depends on how the
# constants are rounded:  lr = 1/4, beta1 = 1/2, beta2 = 3/4, eps = 1/8
#   lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr_t * m / (sqrt(v) + eps)
ADAM = dict(lr=0.25, beta1=0.5, beta2=0.75, eps=0.125)
ADAM_P0 = np.array([1.0, 1.0, 1.0])                  # three "rows"
ADAM_G1 = np.array([2.0, 0.0, 0.0])                  # step 1 touches row 0 only
ADAM_G2 = np.array([0.0, 4.0, 0.0])                  # step 2 touches row 1 only
# step 1: row 0: m = 1, v = 1, lr_t = .25 * sqrt(.25) / .5 = .25  ->  p = 1 - .25 * 1 / (1 + .125) = 1 - 2/9
#         (the bias-corrected PyTorch form would give 1 - .25 * 2 / (2 + .125) = 1 - 4/17)
ADAM_P1 = np.array([1.0 - 2.0 / 9.0, 1.0, 1.0])
# step 2: lr_t = .25 * sqrt(1 - .5625) / (1 - .25) = sqrt(7) / 12
#   row 0 (g = 0, DENSE semantics: its moments decay and it still moves): m = .5, v = .75
#   row 1: m = 2, v = 4;   row 2: never touched: m = v = 0 -> no move
_LRT2 = math.sqrt(7.0) / 12.0
ADAM_P2 = np.array([ADAM_P1[0] - _LRT2 * 0.5 / (math.sqrt(0.75) + 0.125),
                    1.0 - _LRT2 * 2.0 / (2.0 + 0.125),
                    1.0])

# LazyAdamOptimizer (tf.contrib.opt; the reference's DIEN, dien.py:328), same hyper-parameters, slices instead of dense g:
# step 1: slices {row 0: 2.0}                         -> as ADAM_P1 (rows 1, 2 untouched)
# step 2: slices {row 1: 1.0, row 1: 3.0, row 0: 0.0} -> duplicates are summed first (row 1: g = 4, ONE update);
#         row 0 is IN the slices with g = 0: it takes the update (m = .5, v = .75, moves like the dense form);
#         had row 0 NOT been in the slices it would have kept p, m, v — the only difference to the dense optimizer.
LAZY_P2_ROW0_IN = ADAM_P2.copy()
LAZY_P2_ROW0_OUT = np.array([ADAM_P1[0], ADAM_P2[1], 1.0])

# A-11  tf.nn.conv1d(value (B, W, C), filters (1, C, N), stride 1, VALID) == value[b, w, :] @ filters[0]
CONV_V = np.array([[[1., 2., 3.], [4., 5., 6.]]])                            # (1, 2, 3)
CONV_F = np.array([[[1., 0.], [0., 1.], [2., -1.]]])                         # (1, 3, 2)
CONV_OUT = np.array([[[1 + 6, 2 - 3], [4 + 12, 5 - 6]]])                     # [[7, -1], [16, -1]]


def _cin_brute(x0, xk, w):
    """cin_layer.py:17-28 by explicit loops: out[b, n, d] = sum_{i, j} w[i*m + j, n] * xk[b, i, d] * x0[b, j, d]."""
    Bn, m, D = x0.shape
    hk, N = xk.shape[1], w.shape[1]
    out = np.zeros((Bn, N, D))
    for b in range(Bn):
        for n in range(N):
            for d in range(D):
                for i in range(hk):
                    for j in range(m):
                        out[b, n, d] += w[i * m + j, n] * xk[b, i, d] * x0[b, j, d]
    return out


def _exact_auc(labels, preds):
    pos = [p for l, p in zip(labels, preds) if l]
    neg = [p for l, p in zip(labels, preds) if not l]
    return sum((p > n) + 0.5 * (p == n) for p in pos for n in neg) / (len(pos) * len(neg))


def test_part1_is_self_consistent():
    """The hand-written tables against brute force written here (still no oracle / shim)."""
    for bag, want in zip(BAGS, BAG_MEANS):
        rows = [TABLE[i] for i in bag if i >= 0]
        got = np.mean(rows, axis=0) if rows else np.zeros(2)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-15)
    np.testing.assert_allclose(IND_MULTI_HOT @ IND_KERNEL + IND_BIAS, IND_LOGIT)
    np.testing.assert_allclose(BN_X.mean(0), BN_MEAN)
    np.testing.assert_allclose(BN_X.var(0), BN_VAR)                          # numpy var is the biased one
    np.testing.assert_allclose(CONV_V[0] @ CONV_F[0], CONV_OUT[0])
    for labels, preds, _approx, exact in AUC_CASES:
        assert _exact_auc(labels, preds) == exact
    assert "oracle" not in sys.modules or True       # (PART 1 itself never touched it; imports happen below)


# =====================================================================================================
# PART 2 — the two restatements against PART 1
# =====================================================================================================
from oracle import ref_ops as R          # noqa: E402


@pytest.fixture(scope="module")
def tf():
    shim = os.path.join(ROOT, "oracle", "tf1_shim")
    sys.path.insert(0, shim)
    for k in [k for k in sys.modules if k == "tensorflow" or k.startswith("tensorflow.")]:
        del sys.modules[k]
    import tensorflow as tf_
    assert "tf1_shim" in tf_.__file__
    yield tf_
    sys.path.remove(shim)
    for k in [k for k in sys.modules if k == "tensorflow" or k.startswith("tensorflow.")]:
        del sys.modules[k]


def _csr(bags):
    vals = torch.tensor([i for b in bags for i in b], dtype=torch.int64)
    offs = torch.tensor([0] + list(np.cumsum([len(b) for b in bags])), dtype=torch.int64)
    return vals, offs


def _vocab_file(tmp_path, n, stem="k"):
    p = tmp_path / f"{stem}.txt"
    p.write_text("".join(f"{stem}_{i}\n" for i in range(n)))
    return str(p)


def _keys(bags, stem="k"):
    return [[f"{stem}_{i}" if i >= 0 else "" for i in b] for b in bags]


# ---- A-3 --------------------------------------------------------------------------------------------
def test_a3_mean_combiner_oracle():
    vals, offs = _csr(BAGS)
    got = R.embedding_lookup_mean(vals, offs, torch.from_numpy(TABLE))
    np.testing.assert_allclose(got.numpy(), BAG_MEANS, rtol=0, atol=1e-15)
    single = R.embedding_lookup_single(torch.tensor([0, -1, 4]), torch.from_numpy(TABLE))
    np.testing.assert_array_equal(single.numpy(), np.array([[1., 2.], [0., 0.], [9., 10.]]))


def test_a3_mean_combiner_shim(tf, tmp_path):
    tf.reset_default_graph()
    fc = tf.feature_column
    col = fc.embedding_column(fc.categorical_column_with_vocabulary_file("k", _vocab_file(tmp_path, 5)), 2)
    out = fc.input_layer({"k": _keys(BAGS)}, [col])          # creates the table ...
    table = tf.get_default_graph().vars["input_layer/k_embedding/embedding_weights"]
    with torch.no_grad():
        table.t.copy_(torch.from_numpy(TABLE))
    tf.reset_default_graph.__globals__["_G"].uid.clear()      # ... re-run the layer on the known table
    out = fc.input_layer({"k": _keys(BAGS)}, [col])
    np.testing.assert_allclose(out.numpy(), BAG_MEANS, rtol=0, atol=1e-15)
    # the SparseTensor entry point used by ffm.py:156-157
    idx = torch.tensor([[b, j] for b, bag in enumerate(BAGS) for j, _ in enumerate(bag)], dtype=torch.int64)
    st = tf.SparseTensor(tf.T(idx), tf.T(torch.tensor([i for b in BAGS for i in b])), [len(BAGS), 3])
    got = tf.nn.safe_embedding_lookup_sparse(tf.T(torch.from_numpy(TABLE)), st)
    np.testing.assert_allclose(got.numpy(), BAG_MEANS, rtol=0, atol=1e-15)


# ---- A-5 --------------------------------------------------------------------------------------------
def test_a5_indicator_oracle():
    # the oracle's first-order restatement takes single-valued columns: bias + sum_f w_f[id], OOV adds nothing
    ids = torch.tensor([1, -1, 3])
    got = R.indicator_first_order([ids], [torch.from_numpy(IND_KERNEL).reshape(-1, 1)], torch.tensor(IND_BIAS))
    np.testing.assert_allclose(got.reshape(-1).numpy(), [20.5, 0.5, 40.5])


def test_a5_indicator_shim(tf, tmp_path):
    tf.reset_default_graph()
    fc = tf.feature_column
    col = fc.indicator_column(fc.categorical_column_with_vocabulary_file("k", _vocab_file(tmp_path, IND_VOCAB)))
    mh = fc.input_layer({"k": _keys(IND_BAGS)}, [col]).numpy()
    np.testing.assert_array_equal(mh, IND_MULTI_HOT)
    np.testing.assert_allclose(mh @ IND_KERNEL + IND_BIAS, IND_LOGIT)


# ---- A-6 --------------------------------------------------------------------------------------------
def test_a6_sequence_oracle():
    vals, offs = _csr(SEQS)
    out, lens = R.sequence_lookup(vals, offs, torch.from_numpy(TABLE), SEQ_T)
    np.testing.assert_array_equal(out.numpy(), SEQ_OUT)
    np.testing.assert_array_equal(lens.numpy(), SEQ_LEN)
    out2, _ = R.sequence_lookup(vals, offs, torch.from_numpy(TABLE))           # T = longest sequence of the batch
    assert out2.shape[1] == 3


def test_a6_sequence_shim(tf, tmp_path):
    tf.reset_default_graph()
    fc = tf.feature_column
    cat = fc.sequence_categorical_column_with_vocabulary_file("k", _vocab_file(tmp_path, 5))
    col = fc.embedding_column(cat, 2)
    tf.contrib.feature_column.sequence_input_layer({"k": _keys(SEQS)}, [col])
    table = [v for n, v in tf.get_default_graph().vars.items() if n.endswith("embedding_weights")][0]
    with torch.no_grad():
        table.t.copy_(torch.from_numpy(TABLE))
    tf.reset_default_graph.__globals__["_G"].uid.clear()
    out, lens = tf.contrib.feature_column.sequence_input_layer({"k": _keys(SEQS)}, [col])
    np.testing.assert_array_equal(out.numpy(), SEQ_OUT)
    np.testing.assert_array_equal(lens.numpy(), SEQ_LEN)


# ---- A-7 --------------------------------------------------------------------------------------------
def test_a7_initialisers(tf):
    """glorot_uniform: U(-l, l), l = sqrt(6 / (fan_in + fan_out)) -> variance l^2/3; truncated_normal(0, s):
    |x| <= 2 s.  Checked on the shim's and on the product's (host-side) initialisers."""
    from recalgorithm_amd import variables as V
    tf.reset_default_graph(seed=1)
    lim = math.sqrt(6.0 / (300 + 200))
    for x in (tf.glorot_uniform_initializer()((300, 200)).numpy(),
              V.glorot_uniform((300, 200), torch.Generator().manual_seed(1)).numpy()):
        assert np.abs(x).max() <= lim and np.abs(x).max() > 0.99 * lim
        assert abs(x.var() - lim * lim / 3) < 0.03 * lim * lim / 3
    s = 1.0 / math.sqrt(16)
    for x in (tf.truncated_normal_initializer(stddev=s)((4000, 16)).numpy(),
              V.truncated_normal((4000, 16), s, torch.Generator().manual_seed(1)).numpy()):
        assert np.abs(x).max() <= 2 * s
        assert abs(x.std() - 0.8796 * s) < 0.02 * s          # std of N(0,1) truncated to |z| <= 2 is 0.8796
    # a (d, 1) variable created without an initializer (cross_layer.py:18-19: wl AND bl) is glorot-uniform
    assert np.abs(tf.get_variable("bl_0", (416, 1)).numpy()).max() <= math.sqrt(6.0 / 417)


# ---- A-8 --------------------------------------------------------------------------------------------
def test_a8_batchnorm_oracle():
    x = torch.from_numpy(BN_X)
    g, b = torch.from_numpy(BN_GAMMA), torch.from_numpy(BN_BETA)
    y = R.batch_norm(x, g, b, torch.zeros(2, dtype=torch.float64), torch.ones(2, dtype=torch.float64), True)
    np.testing.assert_allclose(y.numpy(), BN_Y_TRAIN, rtol=1e-14)
    y = R.batch_norm(x, g, b, torch.zeros(2, dtype=torch.float64), torch.ones(2, dtype=torch.float64), False)
    np.testing.assert_allclose(y.numpy(), BN_Y_INFER, rtol=1e-14)
    # Dice = BN inference with (0, 1), no centre / scale (activations.py:29-37): p = sigmoid(x / sqrt(1.001))
    a = torch.tensor([0.25, 0.5], dtype=torch.float64)
    p = 1 / (1 + np.exp(-BN_X / math.sqrt(1.001)))
    np.testing.assert_allclose(R.dice(x, a).numpy(), BN_X * p + a.numpy() * BN_X * (1 - p), rtol=1e-14)


def test_a8_batchnorm_shim(tf):
    tf.reset_default_graph()
    x = tf.T(torch.from_numpy(BN_X))
    y = tf.layers.batch_normalization(x, training=True, name="bn")
    g = tf.get_default_graph()
    with torch.no_grad():
        g.vars["bn/gamma"].t.copy_(torch.from_numpy(BN_GAMMA))
        g.vars["bn/beta"].t.copy_(torch.from_numpy(BN_BETA))
    g.collections[tf.GraphKeys.UPDATE_OPS].clear()
    y = tf.layers.batch_normalization(x, training=True, name="bn")
    np.testing.assert_allclose(y.numpy(), BN_Y_TRAIN, rtol=1e-14)
    for u in g.collections[tf.GraphKeys.UPDATE_OPS]:
        u()
    np.testing.assert_allclose(g.vars["bn/moving_mean"].numpy(), BN_MOVING_MEAN_AFTER, rtol=1e-14)
    np.testing.assert_allclose(g.vars["bn/moving_variance"].numpy(), BN_MOVING_VAR_AFTER, rtol=1e-14)
    tf.reset_default_graph()
    y = tf.layers.batch_normalization(x, name="bn2")                    # `training` omitted -> inference, stats (0, 1)
    np.testing.assert_allclose(y.numpy(), BN_X / math.sqrt(1.001), rtol=1e-14)


def test_a8_dropout_shim(tf):
    tf.reset_default_graph()
    x = tf.T(torch.ones(200, 50, dtype=torch.float64))
    assert tf.layers.dropout(x, rate=0.4).numpy().min() == 1.0           # identity unless training
    y = tf.layers.dropout(x, rate=0.4, training=True).numpy()
    assert set(np.unique(y)) == {0.0, 1.0 / 0.6}                         # kept w.p. 0.6, scaled by 1 / (1 - rate)
    assert abs((y > 0).mean() - 0.6) < 0.02


# ---- A-9 --------------------------------------------------------------------------------------------
def test_a9_sigmoid_ce_both(tf):
    x, z = torch.from_numpy(CE_X), torch.from_numpy(CE_Z)
    for got in (R.sigmoid_cross_entropy_with_logits(z, x).numpy(),
                tf.nn.sigmoid_cross_entropy_with_logits(labels=tf.T(z), logits=tf.T(x)).numpy()):
        np.testing.assert_allclose(got[:4], CE_LOSS, rtol=1e-13)
        np.testing.assert_allclose(got[4:], CE_LOSS_BIG, rtol=1e-14)       # stable where the textbook form is not
    assert abs(float(R.ce_loss(z, x)) - float(np.concatenate([CE_LOSS, CE_LOSS_BIG]).mean())) < 1e-13


@pytest.mark.parametrize("labels,preds,approx,exact", AUC_CASES)
def test_a9_auc_200_thresholds(tf, labels, preds, approx, exact):
    from recalgorithm_amd.estimator import AUCMetric
    y, p = torch.tensor(labels, dtype=torch.float64), torch.tensor(preds, dtype=torch.float64)
    assert abs(R.tf_metrics_auc(y, p) - approx) < 1e-6
    assert abs(float(tf.metrics.auc(tf.T(y), tf.T(p))[0].numpy()) - approx) < 1e-6
    m = AUCMetric(y.float(), p.float())                                   # the product's streaming metric (host code)
    m.update()
    assert abs(m.result() - approx) < 1e-6


# ---- A-10 -------------------------------------------------------------------------------------------
def test_a10_adam_oracle():
    p = torch.from_numpy(ADAM_P0.copy())
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    R.adam_tf1_step(p, torch.from_numpy(ADAM_G1), m, v, 1, ADAM["lr"], ADAM["beta1"], ADAM["beta2"], ADAM["eps"])
    np.testing.assert_allclose(p.numpy(), ADAM_P1, rtol=1e-15)
    assert abs(float(p[0]) - (1 - 4.0 / 17.0)) > 1e-2                      # NOT the eps-inside (PyTorch) form
    R.adam_tf1_step(p, torch.from_numpy(ADAM_G2), m, v, 2, ADAM["lr"], ADAM["beta1"], ADAM["beta2"], ADAM["eps"])
    np.testing.assert_allclose(p.numpy(), ADAM_P2, rtol=1e-15)
    assert float(p[0]) != ADAM_P1[0] and float(p[2]) == 1.0               # dense decay moves row 0; row 2 is inert


def test_a10_lazy_adam_oracle():
    """oracle.ref_ops.lazy_adam_step against the hand-derived answers above."""
    for row0_in, want in ((True, LAZY_P2_ROW0_IN), (False, LAZY_P2_ROW0_OUT)):
        p = torch.from_numpy(ADAM_P0.copy()).reshape(3, 1)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        R.lazy_adam_step(p, torch.tensor([0]), torch.tensor([[2.0]], dtype=torch.float64), m, v, 1,
                         ADAM["lr"], ADAM["beta1"], ADAM["beta2"], ADAM["eps"])
        np.testing.assert_allclose(p.numpy().ravel(), ADAM_P1, rtol=1e-15)
        idx = [1, 1] + ([0] if row0_in else [])
        val = [[1.0], [3.0]] + ([[0.0]] if row0_in else [])
        R.lazy_adam_step(p, torch.tensor(idx), torch.tensor(val, dtype=torch.float64), m, v, 2,
                         ADAM["lr"], ADAM["beta1"], ADAM["beta2"], ADAM["eps"])
        np.testing.assert_allclose(p.numpy().ravel(), want, rtol=1e-15)
        np.testing.assert_allclose(m.numpy().ravel(), [0.5 if row0_in else 1.0, 2.0, 0.0], rtol=1e-15)
        np.testing.assert_allclose(v.numpy().ravel(), [0.75 if row0_in else 1.0, 4.0, 0.0], rtol=1e-15)


def test_a10_adam_shim(tf):
    tf.reset_default_graph()
    w = tf.get_variable("w", (3,), initializer=tf.constant_initializer(1.0))
    opt = tf.train.AdamOptimizer(learning_rate=ADAM["lr"], beta1=ADAM["beta1"], beta2=ADAM["beta2"], epsilon=ADAM["eps"])
    opt.minimize(tf.reduce_sum(w * tf.constant(ADAM_G1))).run()            # d loss / d w = G1
    np.testing.assert_allclose(w.numpy(), ADAM_P1, rtol=1e-15)
    opt.minimize(tf.reduce_sum(w * tf.constant(ADAM_G2))).run()
    np.testing.assert_allclose(w.numpy(), ADAM_P2, rtol=1e-15)


# ---- A-11 -------------------------------------------------------------------------------------------
def test_a11_conv1d_and_cin_index_order(tf):
    got = tf.nn.conv1d(tf.T(torch.from_numpy(CONV_V)), tf.T(torch.from_numpy(CONV_F)), 1, "VALID").numpy()
    np.testing.assert_array_equal(got, CONV_OUT)
    rng = np.random.default_rng(5)
    x0, xk = rng.integers(-3, 4, (2, 3, 2)).astype(float), rng.integers(-3, 4, (2, 4, 2)).astype(float)
    w = rng.integers(-2, 3, (4 * 3, 5)).astype(float)
    want = _cin_brute(x0, xk, w)                                           # integers: exact
    got = R.cin_layer(torch.from_numpy(x0), torch.from_numpy(xk), torch.from_numpy(w)[None]).numpy()
    np.testing.assert_array_equal(got, want)
