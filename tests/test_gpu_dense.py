"""-m gpu parity of the fp32-MFMA dense kernels (csrc/dense.hip: forward with fused bias + ReLU and an optional
second operand pair; input gradient with the fused ReLU mask and beta * C term; weight gradient with the fused
mask, bias gradient and deterministic split over the batch) against an fp64 restatement of
tf.layers.dense / its autodiff, at the BASELINE MLP shapes (4096 x 416 -> 512 -> 256 -> 128) and at ragged /
unaligned shapes (the reference's default DCN width 82; IPNN's 351 Gram features)."""
import pytest
import torch

from recalgorithm_amd import ops
from tests.util import assert_close

pytestmark = pytest.mark.gpu

SHAPES = [(4096, 416, 512), (4096, 512, 256), (4096, 256, 128), (37, 82, 50), (130, 351, 1024), (1, 4, 4),
          (300, 48, 8), (65, 33, 65), (512, 9600, 64)]


def _ref_fwd(x, w, b, relu, x2=None, w2=None, dtype=torch.float64):
    """tf.layers.dense restated; dtype = float32 gives the reference arithmetic's own rounding (the `ref32` of
    tests/util.assert_close: the kernel may not leave materially more elements outside the strict bound than this)."""
    y = x.to(dtype) @ w.to(dtype)
    if x2 is not None:
        y = y + x2.to(dtype) @ w2.to(dtype)
    if b is not None:
        y = y + b.to(dtype)
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("M,K,N", SHAPES)
@pytest.mark.parametrize("relu,use_bias", [(True, True), (False, False)])
def test_dense_fwd_bwd(dev, M, K, N, relu, use_bias):
    gen = torch.Generator().manual_seed(M * 31 + K * 7 + N)
    x = torch.randn(M, K, generator=gen)
    w = torch.randn(K, N, generator=gen) / K ** 0.5
    b = torch.randn(N, generator=gen) * 0.1 if use_bias else None
    g = torch.randn(M, N, generator=gen)
    xd, wd, gd = x.to(dev), w.to(dev), g.to(dev)
    bd = None if b is None else b.to(dev)
    y = ops.dense_fwd(xd, wd, bd, relu)
    ref = _ref_fwd(x, w, b, relu)
    # strict guard against the fp32 arithmetic of the reference (blocked library GEMM) within the tile engine's domain:
    # layers wider than 4096 inputs (FiBiNET's 9600) run on hipBLASLt in the product path (nn.py); the engine's
    # K = 9600 case is two 4800-term fmaf chains per output and is only held to the tolerance itself
    r32 = _ref_fwd(x, w, b, relu, dtype=torch.float32) if K <= 4096 else None
    assert_close(y, ref, what=f"dense fwd {M}x{K}x{N}", reduced=True, ref32=r32)
    # backward with the mask taken from the kernel's own forward output (what nn._DenseFn does)
    mask = (y.cpu() > 0).double() if relu else torch.ones_like(ref)
    g2 = g.double() * mask
    g2f = g2.float()
    dx = ops.dense_bwd_input(gd, y if relu else None, wd)
    assert_close(dx, g2 @ w.double().t(), what="dense dgrad", reduced=True, ref32=g2f @ w.t())
    dw = torch.empty(K, N, device=dev)
    db = torch.empty(N, device=dev) if use_bias else None
    ops.dense_bwd_weights(xd, gd, y if relu else None, dw, db)
    assert_close(dw, x.double().t() @ g2, what="dense wgrad", reduced=True, ref32=x.t() @ g2f)
    if use_bias:
        assert_close(db, g2.sum(0), what="dense dbias", reduced=True, ref32=g2f.sum(0))
    # deterministic: a second run is bit-identical (fixed-order split sum, no atomics)
    dw2 = torch.empty_like(dw)
    ops.dense_bwd_weights(xd, gd, y if relu else None, dw2, None)
    assert torch.equal(dw, dw2)


def test_dense_two_operand_pairs_and_beta_c(dev):
    """The PNN form relu(x w + x2 w2 + b) (pnn.py:139-181) and the dgrad's fused beta * C term (DIN's
    mini-batch-aware regulariser, din.py:254-257), accumulate mode."""
    gen = torch.Generator().manual_seed(5)
    M, K, K2, N = 700, 416, 351, 1024
    x, x2 = torch.randn(M, K, generator=gen), torch.randn(M, K2, generator=gen)
    w, w2 = torch.randn(K, N, generator=gen) / K ** 0.5, torch.randn(K2, N, generator=gen) / K2 ** 0.5
    b = torch.randn(N, generator=gen)
    y = ops.dense_fwd(x.to(dev), w.to(dev), b.to(dev), True, x2=x2.to(dev), w2=w2.to(dev))
    assert_close(y, _ref_fwd(x, w, b, True, x2, w2), what="two-pair fwd", reduced=True,
                 ref32=_ref_fwd(x, w, b, True, x2, w2, dtype=torch.float32))
    g = torch.randn(M, N, generator=gen)
    c = torch.randn(M, K, generator=gen)
    dx = ops.dense_bwd_input(g.to(dev), None, w.to(dev), c_in=c.to(dev), beta=0.25)
    assert_close(dx, g.double() @ w.double().t() + 0.25 * c.double(), what="dgrad + beta*C", reduced=True,
                 ref32=g @ w.t() + 0.25 * c)
    base = torch.randn(M, K, generator=gen)
    out = base.to(dev).clone()
    ops.dense_bwd_input(g.to(dev), None, w.to(dev), out=out, accumulate=True)
    assert_close(out, base.double() + g.double() @ w.double().t(), what="dgrad accumulate", reduced=True, ref32=base + g @ w.t())


def test_dense_strided_views(dev):
    """Row-strided operands (a column block of a wider activation matrix) are read in place."""
    gen = torch.Generator().manual_seed(9)
    big = torch.randn(260, 200, generator=gen).to(dev)
    x = big[:, 40:168]                         # ld 200, base offset 160 B: aligned, vector path
    xu = big[:, 41:169]                        # base offset 164 B: unaligned -> element loads
    w = (torch.randn(128, 64, generator=gen) / 11).to(dev)
    for v in (x, xu):
        y = ops.dense_fwd(v, w, None, False)
        assert_close(y, v.cpu().double() @ w.cpu().double(), what="strided fwd", reduced=True, ref32=v.cpu() @ w.cpu())


def test_dense_matches_exact_fmaf_order(dev):
    """v_mfma_f32_32x32x2_f32 is an exact k-ordered fmaf chain: integer-valued operands give exact results."""
    gen = torch.Generator().manual_seed(1)
    x = torch.randint(-4, 5, (200, 96), generator=gen).float()
    w = torch.randint(-4, 5, (96, 72), generator=gen).float()
    y = ops.dense_fwd(x.to(dev), w.to(dev), None, False)
    assert torch.equal(y.cpu(), x @ w)


@pytest.mark.parametrize("M,K,N", [(4096, 416, 512), (4096, 256, 128), (2048, 100, 36)])
def test_dense_wgrad_split_handoff_stress(dev, M, K, N):
    """The batch-split partials of the weight gradient go through a workspace and are summed in fixed order by a
    second launch.  Re-using ONE workspace over many launches with fresh data: a partial read before its producer's
    launch has completed, or a stale line, shows up as an O(1) relative error.  Also: bit-identical to itself
    (fixed summation order, no atomics)."""
    gen = torch.Generator(device=dev).manual_seed(123)
    dw, db = torch.empty(K, N, device=dev), torch.empty(N, device=dev)
    dw2, db2 = torch.empty_like(dw), torch.empty_like(db)
    for it in range(25):
        x = torch.randn(M, K, device=dev, generator=gen)
        g = torch.randn(M, N, device=dev, generator=gen)
        y = torch.randn(M, N, device=dev, generator=gen)
        ops.dense_bwd_weights(x, g, y, dw, db)
        ops.dense_bwd_weights(x, g, y, dw2, db2)
        g2 = (g * (y > 0)).double()
        ref = x.double().t() @ g2
        g2f = (g * (y > 0)).cpu()
        assert_close(dw, ref, what=f"wgrad launch {it}", reduced=True, ref32=x.cpu().t() @ g2f)
        assert_close(db, g2.sum(0), what=f"dbias launch {it}", reduced=True, ref32=g2f.sum(0))
        assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize("M,K,N", [(4096, 416, 512), (4096, 256, 128), (300, 82, 50)])
def test_dense_merged_bwd_is_bit_identical_to_the_two_launches(dev, M, K, N):
    """recalgo_dense_bwd (input + weight gradient tiles in one grid) runs the same tiles as the two separate launches."""
    gen = torch.Generator(device=dev).manual_seed(M + N)
    x, g, y = (torch.randn(M, c, device=dev, generator=gen) for c in (K, N, N))
    w = torch.randn(K, N, device=dev, generator=gen) / K ** 0.5
    c_in = torch.randn(M, K, device=dev, generator=gen)
    dw, db, dw2, db2 = torch.empty(K, N, device=dev), torch.empty(N, device=dev), torch.empty(K, N, device=dev), torch.empty(N, device=dev)
    dx = ops.dense_bwd(x, g, y, w, dw, db, c_in=c_in, beta=0.5)
    dx2 = ops.dense_bwd_input(g, y, w, c_in=c_in, beta=0.5)
    ops.dense_bwd_weights(x, g, y, dw2, db2)
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize("M,K,N", [(4096, 416, 512), (4096, 512, 256), (300, 82, 100), (65, 48, 8), (64, 33, 68)])
def test_dense_fwd_leaves_the_batchnorm_moments_of_its_tiles(dev, M, K, N):
    """recalgo_dense_fwd_bn: the epilogue's per-tile (mean, sum of squared deviations) rows are what recalgo_batchnorm_moments
    computes from y in a pass of its own (same layout, same two-pass definition), y itself is unchanged, and BatchNorm fed
    with them equals BatchNorm computing its own moments (dense -> batch_normalization, deepfm.py:207-211)."""
    import ctypes
    from recalgorithm_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=gen).to(dev)
    w = (torch.randn(K, N, generator=gen) / K ** 0.5).to(dev)
    b = (torch.randn(N, generator=gen) * 0.1).to(dev)
    nb = ops.bn_partial_rows(M)
    part = torch.full((nb, 2 * N), float("nan"), device=dev)
    y = ops.dense_fwd(x, w, b, True, bn_partials=part)
    y0 = ops.dense_fwd(x, w, b, True)
    assert torch.equal(y, y0)
    want = torch.empty(nb, 2 * N, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(lib.recalgo_batchnorm_moments(p(y0), M, N, p(want), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "moments")
    assert_close(part[:, :N], want[:, :N].double(), what="tile means from the GEMM epilogue")
    assert_close(part[:, N:], want[:, N:].double(), what="tile M2 from the GEMM epilogue", reduced=True)
    # against the definition, in fp64
    yd = y0.double().cpu()
    for t in (0, nb - 1):
        rows = yd[t * 64:min(M, t * 64 + 64)]
        assert_close(part[t, :N], rows.mean(0), what=f"tile {t} mean vs fp64", reduced=True)
        assert_close(part[t, N:], ((rows - rows.mean(0)) ** 2).sum(0), what=f"tile {t} M2 vs fp64", reduced=True)
    gamma, beta = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
    mm1, mv1, mm2, mv2 = torch.zeros(N, device=dev), torch.ones(N, device=dev), torch.zeros(N, device=dev), torch.ones(N, device=dev)
    o1, mean1, rstd1 = ops.batchnorm_train_fwd(y0, gamma, beta, mm1, mv1, 0.99, 1e-3)
    o2, mean2, rstd2 = ops.batchnorm_train_fwd(y0, gamma, beta, mm2, mv2, 0.99, 1e-3, partials=part)
    assert_close(o2, o1.double(), what="BatchNorm on the epilogue's moments", reduced=True)
    assert_close(mean2, mean1.double(), what="batch mean")
    assert_close(rstd2, rstd1.double(), what="batch rstd")
    assert_close(mv2, mv1.double(), what="moving variance")


@pytest.mark.parametrize("kind", ["dice", "prelu"])
@pytest.mark.parametrize("M,K,N", [(4096, 416, 512), (300, 82, 100), (65, 48, 8)])
def test_dense_activation_batchnorm_as_one_node(dev, M, K, N, kind):
    """The hidden layer of DIN's fcn scope (dense -> dice | prelu -> batch_normalization, din.py:262-266) as the fused launches
    (recalgo_dense_fwd_act_bn, recalgo_batchnorm_train_bwd_act) against the three layers' own kernels one after the other —
    and both against the definition in fp64."""
    import ctypes
    from recalgorithm_amd import _lib
    lib = _lib.load()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    gen = torch.Generator().manual_seed(M + N + len(kind))
    x = torch.randn(M, K, generator=gen).to(dev)
    w = (torch.randn(K, N, generator=gen) / K ** 0.5).to(dev)
    b = (torch.randn(N, generator=gen) * 0.1).to(dev)
    alpha = (torch.rand(N, generator=gen) * 0.5 + 0.1).to(dev)
    gamma, beta = (torch.rand(N, generator=gen) + 0.5).to(dev), torch.randn(N, generator=gen).to(dev)
    g = torch.randn(M, N, generator=gen).to(dev)
    k = ops._ACT[kind]
    nb = ops.bn_partial_rows(M)
    # ---- forward ----
    part = torch.full((nb, 2 * N), float("nan"), device=dev)
    z, y = ops.dense_fwd_act(x, w, b, k, alpha, part)
    z0 = ops.dense_fwd(x, w, b, False)
    assert torch.equal(z, z0)
    y0 = torch.empty_like(z0)
    _lib.check(lib.recalgo_activation_fwd(p(z0), p(alpha), M, N, k, p(y0), st()), "act fwd")
    assert_close(y, y0.double(), what=f"{kind} in the GEMM epilogue vs the elementwise kernel")
    zd = z0.double().cpu()
    ad = alpha.double().cpu()
    if kind == "dice":
        px = torch.sigmoid(zd / (1 + 1e-3) ** 0.5)
        yd = zd * px + ad * zd * (1 - px)
    else:
        yd = zd.clamp(min=0) + ad * zd.clamp(max=0)
    assert_close(y, yd, what=f"{kind} in the GEMM epilogue vs fp64", ref32=y0)
    want = torch.empty(nb, 2 * N, device=dev)
    _lib.check(lib.recalgo_batchnorm_moments(p(y), M, N, p(want), st()), "moments")
    assert_close(part[:, :N], want[:, :N].double(), what="tile means of the activation from the GEMM epilogue")
    assert_close(part[:, N:], want[:, N:].double(), what="tile M2 of the activation from the GEMM epilogue", reduced=True)
    mm, mv = torch.zeros(N, device=dev), torch.ones(N, device=dev)
    out, mean, rstd = ops.batchnorm_train_fwd(y, gamma, beta, mm, mv, 0.99, 1e-3, partials=part)
    # ---- backward ----
    dgamma, dbeta, dalpha = (torch.full((N,), float("nan"), device=dev) for _ in range(3))
    dz = ops.batchnorm_train_bwd_act(y, gamma, mean, rstd, g, dgamma, dbeta, k, z, alpha, dalpha, defer=False)
    dgamma0, dbeta0, dalpha0 = (torch.empty(N, device=dev) for _ in range(3))
    dy0 = ops.batchnorm_train_bwd(y, gamma, mean, rstd, g, dgamma0, dbeta0)
    dz0 = torch.empty_like(z)
    ws = torch.empty(max(int(lib.recalgo_activation_bwd_workspace_bytes(M, N)), 16), dtype=torch.uint8, device=dev)
    _lib.check(lib.recalgo_activation_bwd(p(z), p(alpha), p(dy0), M, N, k, p(dz0), p(dalpha0), p(ws), st()), "act bwd")
    assert torch.equal(dgamma, dgamma0) and torch.equal(dbeta, dbeta0)
    assert_close(dz, dz0.double(), what=f"d(z) through BatchNorm and {kind} in one launch vs two")
    assert_close(dalpha, dalpha0.double(), what=f"d({kind} alpha) from the fused backward", reduced=True)
    # the deferred form leaves the same partial rows for the step's column-sum launch
    dalpha1 = torch.full((N,), float("nan"), device=dev)
    dz1 = ops.batchnorm_train_bwd_act(y, gamma, mean, rstd, g, dgamma, dbeta, k, z, alpha, dalpha1, defer=True)
    ops.flush_dense_splits()
    assert torch.equal(dz1, dz)
    assert_close(dalpha1, dalpha.double(), what=f"d({kind} alpha) summed by the deferred launch", reduced=True)
    # fp64 definition of the whole chain
    yv = yd.to(dev).requires_grad_(False)
    zt = zd.clone().requires_grad_(True)
    at = ad.clone().requires_grad_(True)
    if kind == "dice":
        pxt = torch.sigmoid(zt / (1 + 1e-3) ** 0.5)
        yt = zt * pxt + at * zt * (1 - pxt)
    else:
        yt = zt.clamp(min=0) + at * zt.clamp(max=0)
    mu = yt.mean(0)
    var = ((yt - mu) ** 2).mean(0)
    ot = (yt - mu) / torch.sqrt(var + 1e-3) * gamma.double().cpu() + beta.double().cpu()
    ot.backward(g.double().cpu())

    def chain32():
        z3, a3 = zd.float().clone().requires_grad_(True), ad.float().clone().requires_grad_(True)
        if kind == "dice":
            p3 = torch.sigmoid(z3 / (1 + 1e-3) ** 0.5)
            y3 = z3 * p3 + a3 * z3 * (1 - p3)
        else:
            y3 = z3.clamp(min=0) + a3 * z3.clamp(max=0)
        m3 = y3.mean(0)
        v3 = ((y3 - m3) ** 2).mean(0)
        o3 = (y3 - m3) / torch.sqrt(v3 + 1e-3) * gamma.float().cpu() + beta.float().cpu()
        o3.backward(g.float().cpu())
        return o3.detach(), z3.grad, a3.grad
    o32, dz32, da32 = chain32()
    assert_close(out, ot.detach(), what="fused layer output vs fp64", reduced=True, ref32=o32)
    assert_close(dz, zt.grad, what="fused layer d(z) vs fp64", reduced=True, ref32=dz32)
    assert_close(dalpha, at.grad, what="fused layer d(alpha) vs fp64", reduced=True, ref32=da32)


@pytest.mark.parametrize("M,K,N", [(4096, 512, 256), (4096, 256, 128), (300, 100, 52), (65, 8, 48)])
def test_dense_bwd_leaves_the_batchnorm_backward_sums(dev, M, K, N):
    """recalgo_dense_bwd_bn: a layer whose input is a BatchNorm's output (batch_normalization -> dense, deepfm.py:207-211)
    leaves, per 64-row tile of its input gradient, colsum(dx) and colsum(dx * xhat) — what recalgo_batchnorm_bwd_sums computes
    from dx in a pass of its own — with dx / dw / dbias unchanged; BatchNorm's backward fed with them equals the two-launch one."""
    import ctypes
    from recalgorithm_amd import _lib
    lib = _lib.load()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    gen = torch.Generator().manual_seed(M + K)
    bn_in = torch.randn(M, K, generator=gen).to(dev)                   # the BatchNorm's input
    gamma, beta = (torch.rand(K, generator=gen) + 0.5).to(dev), torch.randn(K, generator=gen).to(dev)
    mm, mv = torch.zeros(K, device=dev), torch.ones(K, device=dev)
    x, mean, rstd = ops.batchnorm_train_fwd(bn_in, gamma, beta, mm, mv, 0.99, 1e-3)       # = the dense layer's input
    w = (torch.randn(K, N, generator=gen) / K ** 0.5).to(dev)
    g = torch.randn(M, N, generator=gen).to(dev)
    y = torch.relu(torch.randn(M, N, generator=gen)).to(dev)
    nb = ops.bn_partial_rows(M)
    for mask in (None, y):
        dw0, db0 = torch.empty(K, N, device=dev), torch.empty(N, device=dev)
        dx0 = ops.dense_bwd(x, g, mask, w, dw0, db0)
        ops.flush_dense_splits()
        dw1, db1 = torch.empty(K, N, device=dev), torch.empty(N, device=dev)
        sums = torch.full((nb, 2 * K), float("nan"), device=dev)
        dx1 = ops.dense_bwd(x, g, mask, w, dw1, db1, bn=(bn_in, mean, rstd, sums))
        ops.flush_dense_splits()
        assert torch.equal(dx1, dx0) and torch.equal(dw1, dw0) and torch.equal(db1, db0)
        want = torch.empty(nb, 2 * K, device=dev)
        _lib.check(lib.recalgo_batchnorm_bwd_sums(p(bn_in), p(mean), p(rstd), p(dx0), M, K, p(want), st()), "bwd_sums")
        assert_close(sums[:, :K], want[:, :K].double(), what="tile colsum(dx) from the dgrad epilogue", reduced=True)
        assert_close(sums[:, K:], want[:, K:].double(), what="tile colsum(dx * xhat) from the dgrad epilogue", reduced=True)
        if K % 4 == 0:
            dg0, dbt0, dg1, dbt1 = (torch.empty(K, device=dev) for _ in range(4))
            d0 = ops.batchnorm_train_bwd(bn_in, gamma, mean, rstd, dx0, dg0, dbt0)
            d1 = ops.batchnorm_train_bwd(bn_in, gamma, mean, rstd, dx0, dg1, dbt1, sums=sums)
            assert_close(d1, d0.double(), what="BatchNorm backward on the epilogue's sums", reduced=True)
            assert_close(dg1, dg0.double(), what="d(gamma) on the epilogue's sums", reduced=True)
            assert_close(dbt1, dbt0.double(), what="d(beta) on the epilogue's sums", reduced=True)


@pytest.mark.parametrize("M,K,N,with_bn", [(4096, 512, 256, False), (300, 100, 52, False), (65, 82, 50, False), (4096, 256, 128, True)])
def test_dense_bwd_masks_its_input_gradient_with_the_relu_output_below(dev, M, K, N, with_bn):
    """recalgo_dense_bwd_bn(dx_relu_mask=): dx = (g * [y > 0]) W^T zeroed where the mask tensor is <= 0 (the row-wise
    epilogue, the element-wise one for widths that are not float4-addressable, the BatchNorm-sums epilogue); dW / db unchanged."""
    gen = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=gen).clamp(min=0).to(dev)       # a ReLU output: the mask IS the layer's input
    w = (torch.randn(K, N, generator=gen) / K ** 0.5).to(dev)
    g, y = torch.randn(M, N, generator=gen).to(dev), torch.randn(M, N, generator=gen).to(dev)
    dw, db = torch.zeros(K, N, device=dev), torch.zeros(N, device=dev)
    bn = None
    if with_bn:
        bx = torch.randn(M, K, generator=gen).to(dev)
        mean, rstd = bx.mean(0), 1.0 / (bx.var(0, unbiased=False) + 1e-3).sqrt()
        part = torch.zeros(ops.bn_partial_rows(M), 2 * K, device=dev)
        bn = (bx, mean, rstd, part)
    dx = ops.dense_bwd(x, g, y, w, dw, db, defer=True, bn=bn, premask=x)
    ops.flush_dense_splits()
    g2 = (g * (y > 0)).double()
    ref = (g2 @ w.double().t()) * (x > 0)
    g2f = (g * (y > 0)).cpu()
    assert_close(dx, ref, what="dense dgrad, masked by the layer's input", reduced=True, ref32=(g2f @ w.cpu().t()) * (x > 0).cpu())
    assert bool((dx[x <= 0] == 0).all())
    assert_close(dw, x.double().t() @ g2, what="dense wgrad beside the masked dgrad", reduced=True, ref32=x.cpu().t() @ g2f)
    if with_bn:
        xh = (bx.double() - mean.double()) * rstd.double()
        s = part.double().view(-1, 2, K).sum(0)
        assert_close(s[0], ref.sum(0), what="bn sums of the masked dx", reduced=True)
        assert_close(s[1], (ref * xh).sum(0), what="bn sums of the masked dx * xhat", reduced=True)


@pytest.mark.parametrize("second_consumer", [False, True])
def test_relu_source_protocol(dev, second_consumer):
    """nn.ReluSource: in a chain dense(relu) -> dense the upper layer's backward masks dx with its input and the lower layer
    runs WITHOUT a mask; as soon as the activation has another consumer autograd hands the lower layer a summed tensor and
    it applies its mask as before.  Either way the gradients are those of the definition."""
    from recalgorithm_amd import nn
    from recalgorithm_amd.variables import Variable, VariableStore
    gen = torch.Generator().manual_seed(5 + int(second_consumer))
    M, K, H, N = 512, 96, 128, 64
    x = torch.randn(M, K, generator=gen)
    w1, b1 = torch.randn(K, H, generator=gen) / K ** 0.5, torch.randn(H, generator=gen) * 0.1
    w2, b2 = torch.randn(H, N, generator=gen) / H ** 0.5, torch.randn(N, generator=gen) * 0.1
    gy, gz = torch.randn(M, N, generator=gen), torch.randn(M, H, generator=gen)
    store = VariableStore(dev)
    k1, c1, k2, c2 = (Variable(n, t.to(dev)) for n, t in (("k1", w1), ("b1", b1), ("k2", w2), ("b2", b2)))
    xd = x.to(dev).requires_grad_(True)
    src = nn.ReluSource()
    y1 = nn._DenseFn.apply(store.anchor, xd, k1, c1, True, 0.0, None, None, src)
    y1._recalgo_relu_src = src
    y2 = nn._DenseFn.apply(store.anchor, y1, k2, c2, False, 0.0, None, None, None)
    loss = (y2 * gy.to(dev)).sum()
    if second_consumer:
        loss = loss + (y1 * gz.to(dev)).sum()
    calls, real = [], ops.dense_bwd

    def spy(x_, g_, mask, *a, **k):
        calls.append((tuple(g_.shape), mask is not None, k.get("premask") is not None))
        return real(x_, g_, mask, *a, **k)
    ops.dense_bwd = spy
    try:
        loss.backward()
    finally:
        ops.dense_bwd = real
    ops.flush_dense_splits()
    # upper layer: no mask of its own (no ReLU), masks its dx; lower layer: masked by the upper one unless it has a second consumer
    assert calls[0] == ((M, N), False, True)
    assert calls[1] == ((M, H), second_consumer, False)
    P = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    r1 = torch.relu(P[0] @ P[1] + P[2])
    rl = ((r1 @ P[3] + P[4]) * gy.double()).sum() + ((r1 * gz.double()).sum() if second_consumer else 0.0)
    rl.backward()
    for got, want, what in ((xd.grad, P[0].grad, "dx"), (k1.grad, P[1].grad, "dW1"), (c1.grad, P[2].grad, "db1"),
                            (k2.grad, P[3].grad, "dW2"), (c2.grad, P[4].grad, "db2")):
        assert_close(got, want, what=f"relu-source chain {what}", reduced=True)


@pytest.mark.parametrize("with_sums", [False, True])
def test_batchnorm_backward_masks_dx_with_a_relu_input(dev, with_sums):
    """recalgo_batchnorm_{train_bwd, bwd_apply}(dx_relu = 1): the BatchNorm whose input is a ReLU output writes
    dx * [x > 0] — bit-equal to masking the plain backward's dx afterwards; dgamma / dbeta unchanged."""
    gen = torch.Generator().manual_seed(3 + int(with_sums))
    M, C = 1000, 96
    x = torch.randn(M, C, generator=gen).clamp(min=0).to(dev)
    g = torch.randn(M, C, generator=gen).to(dev)
    gamma = (torch.rand(C, generator=gen) + 0.5).to(dev)
    mean, rstd = x.mean(0), 1.0 / (x.var(0, unbiased=False) + 1e-3).sqrt()
    sums = None
    if with_sums:                              # the per-tile partial rows a dense layer's dgrad epilogue leaves
        xh = (x - mean) * rstd
        nb = ops.bn_partial_rows(M)
        sums = torch.zeros(nb, 2 * C, device=dev)
        for t in range(nb):
            sl = slice(64 * t, min(64 * (t + 1), M))
            sums[t, :C], sums[t, C:] = g[sl].sum(0), (g[sl] * xh[sl]).sum(0)
    dg0, db0, dg1, db1 = (torch.empty(C, device=dev) for _ in range(4))
    plain = ops.batchnorm_train_bwd(x, gamma, mean, rstd, g, dg0, db0, sums=sums)
    masked = ops.batchnorm_train_bwd(x, gamma, mean, rstd, g, dg1, db1, sums=sums, relu_x=True)
    assert torch.equal(masked, plain * (x > 0))
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
