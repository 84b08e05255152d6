// Glue of the context MLP around the library GEMMs (tf.layers.dense / batch_normalization of
// every model_fn, e.g. algorithm/DeepFM/deepfm.py:206-212), gfx950.  The GEMMs themselves stay on
// hipBLASLt; what is fused here is the memory-bound work between them, which torch would run as
// 3 (ReLU backward + bias gradient) to 8 (BatchNorm training forward / backward) separate launches:
//   relu_bwd_bias        g2 = g * [y > 0],  dbias = colsum(g2)                      (dense backward)
//   batchnorm_train_fwd  batch mean / biased variance (Chan-merged per-block moments, as accurate
//                        as tf.nn.moments' two-pass form), moving-stat update (momentum 0.99),
//                        y = (x - mean) * rsqrt(var + eps) * gamma + beta                      (2 launches)
//   batchnorm_train_bwd  dbeta = colsum(g), dgamma = colsum(g * xhat),
//                        dx = gamma * rstd / B * (B*g - dbeta - xhat * dgamma)                (2 launches)
// (BatchNorm first ran as 3 + 3 launches — moments, merge, apply / partial sums, column sums, apply; the second stages
//  now ride in the prologue of the apply kernels: at 4-5 us per launch they were a tenth of the DeepFM step.)
// All three are HBM-bound streams over [rows, C] fp32 with column reductions.  Tile = 64 rows x 64
// columns per workgroup (16 float4 column groups x 16 row lanes, 4 rows per thread, held in
// registers): a [4096, 512] layer is 512 workgroups, and only rows/64 partial rows per column are
// left for the fixed-order (deterministic) second stage.  (The first version used 8-row blocks:
// its second stages walked rows/8 partial rows and cost 10-22 us per layer.)  Requires C % 4 == 0.
#include "common.h"
#include "act.h"
#include "dropout.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTileRows = 64;

__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) {
    return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) {
    return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}

// sum of the 16 row lanes of column group cl in fixed order (every thread gets the total)
__device__ __forceinline__ float4 tile_colsum(float4 v, float4* sh, unsigned cl) {
    __syncthreads();
    sh[threadIdx.x] = v;
    __syncthreads();
    float4 t = sh[cl];
#pragma unroll
    for (unsigned k = 1; k < 16; ++k) t = f4_add(t, sh[k * 16 + cl]);
    return t;
}

// Tile = 64 rows x 64 columns (16 float4 column groups x 16 row lanes, 4 rows per thread): a
// [4096, 512] layer is 512 workgroups and only rows/64 partial rows are left for the fixed-order
// column sum (the first version reduced rows/8 partial rows: its colsum pass cost 9-11 us per layer).
__global__ __launch_bounds__(kThreads) void relu_bwd_bias_kernel(const float4* __restrict__ g,
                                                                 const float4* __restrict__ y, unsigned rows,
                                                                 unsigned C4, float4* __restrict__ g_out,
                                                                 float4* __restrict__ partials) {
    __shared__ float4 sh[kThreads];
    const unsigned cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const unsigned c4 = blockIdx.x * 16 + cl;
    const unsigned r0 = blockIdx.y * kTileRows;
    float4 acc = f4_zero();
    if (c4 < C4) {
#pragma unroll
        for (unsigned k = 0; k < kTileRows / 16; ++k) {
            const unsigned r = r0 + rl + 16 * k;
            if (r < rows) {
                float4 v = g[(size_t)r * C4 + c4];
                if (y) {
                    const float4 yy = y[(size_t)r * C4 + c4];
                    v = make_float4(yy.x > 0.f ? v.x : 0.f, yy.y > 0.f ? v.y : 0.f, yy.z > 0.f ? v.z : 0.f,
                                    yy.w > 0.f ? v.w : 0.f);
                    g_out[(size_t)r * C4 + c4] = v;
                }
                acc = f4_add(acc, v);
            }
        }
    }
    acc = tile_colsum(acc, sh, cl);
    if (rl == 0 && c4 < C4) partials[(size_t)blockIdx.y * C4 + c4] = acc;
}

// ---- BatchNorm ------------------------------------------------------------------------------
// per-block moments: partials[blk][0:C] = block mean, [C:2C] = block M2 (sum of squared deviations
// from the tile mean); tile row counts are implied by (rows, kTileRows)
__global__ __launch_bounds__(kThreads) void bn_moments_kernel(const float4* __restrict__ x, unsigned rows,
                                                              unsigned C4, float4* __restrict__ partials) {
    __shared__ float4 sh[kThreads];
    const unsigned cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const unsigned c4 = blockIdx.x * 16 + cl;
    const unsigned r0 = blockIdx.y * kTileRows, r1 = min(rows, r0 + kTileRows);
    const float n = (float)(r1 - r0);
    float4 v[kTileRows / 16];
    bool has[kTileRows / 16];
    float4 s = f4_zero();
#pragma unroll
    for (unsigned k = 0; k < kTileRows / 16; ++k) {
        const unsigned r = r0 + rl + 16 * k;
        has[k] = c4 < C4 && r < r1;
        v[k] = has[k] ? x[(size_t)r * C4 + c4] : f4_zero();
        s = f4_add(s, v[k]);
    }
    const float4 mean = f4_scale(tile_colsum(s, sh, cl), 1.0f / n);
    float4 m2 = f4_zero();
#pragma unroll
    for (unsigned k = 0; k < kTileRows / 16; ++k)
        if (has[k]) {
            const float4 d = f4_sub(v[k], mean);
            m2 = f4_add(m2, f4_mul(d, d));
        }
    m2 = tile_colsum(m2, sh, cl);
    if (rl == 0 && c4 < C4) {
        partials[(size_t)blockIdx.y * 2 * C4 + c4] = mean;
        partials[(size_t)blockIdx.y * 2 * C4 + C4 + c4] = m2;
    }
}

// Chan merge of the block moments (fixed order: 16 interleaved block groups, then the groups) -> mean, rstd of the
// 64 columns of this workgroup, then y = (x - mean) * rstd * gamma + beta on its 64 x 64 tile.  Every workgroup of a
// column block repeats the merge (nblk x 2 x 64 floats of L2-resident partials — cheaper than a launch in between);
// the workgroups of tile row 0 record mean / rstd for the backward pass and update the moving statistics.
constexpr unsigned kBnPre = 8;
__global__ __launch_bounds__(kThreads) void bn_finalize_apply_kernel(
    const float4* __restrict__ x, const float4* __restrict__ gamma, const float4* __restrict__ beta,
    const float4* __restrict__ partials, unsigned nblk, unsigned nblk_local, unsigned rows, unsigned C4, float eps,
    float momentum, float4* __restrict__ moving_mean, float4* __restrict__ moving_var, float4* __restrict__ save_mean,
    float4* __restrict__ save_rstd, float4* __restrict__ y, recalgo_drop::Spec out_drop) {
    // out_drop: the tf.layers.dropout that follows the BatchNorm (dense -> dice | prelu -> batch_norm -> dropout, din.py:227-236):
    // y := y * keep / (1 - rate) in the store (csrc/dropout.h; element index row * C + col)
    // nblk = world * nblk_local partial rows (Sync-BatchNorm: the tiles of all ranks, rank major; every rank holds `rows`
    // examples); world = 1: nblk == nblk_local
    __shared__ float4 sh[16][17];
    const unsigned cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const unsigned c4 = blockIdx.x * 16 + cl;
    const bool ok = c4 < C4;
    const float inv_rows = 1.0f / ((float)rows * (float)(nblk / nblk_local));
    // ONE memory round trip instead of three: the tile's x rows and this thread's partial rows (both moments) are requested
    // together, before the first reduction — the launch is a latency chain (5.9 us in the step for 2 us of traffic), and neither
    // the second pass over the partials nor the x loads depend on the first pass's result, only their arithmetic does.
    // (up to kBnPre partial rows per thread in registers: nblk <= 128, i.e. B <= 8192 on one rank; more: the loops as before)
    const unsigned r0 = blockIdx.y * kTileRows;
    float4 xv[kTileRows / 16];
#pragma unroll
    for (unsigned k = 0; k < kTileRows / 16; ++k) {
        const unsigned r = r0 + rl + 16 * k;
        xv[k] = (ok && r < rows) ? x[(size_t)r * C4 + c4] : f4_zero();
    }
    const bool pre = nblk <= 16 * kBnPre;
    float4 pm[kBnPre], pq[kBnPre];
    if (pre && ok) {
#pragma unroll
        for (unsigned u = 0; u < kBnPre; ++u) {
            const unsigned b = rl + 16 * u;
            if (b < nblk) {
                pm[u] = partials[(size_t)b * 2 * C4 + c4];
                pq[u] = partials[(size_t)b * 2 * C4 + C4 + c4];
            }
        }
    }
    float4 acc = f4_zero();
    if (ok) {
        if (pre) {
#pragma unroll
            for (unsigned u = 0; u < kBnPre; ++u) {
                const unsigned b = rl + 16 * u;
                if (b < nblk) {
                    const unsigned bl = b % nblk_local;
                    const float nb = (float)(min(rows, (bl + 1) * kTileRows) - bl * kTileRows);
                    acc = f4_fma(pm[u], nb, acc);
                }
            }
        } else {
#pragma unroll 4                                                // (the partial rows are independent loads: several in flight)
            for (unsigned b = rl; b < nblk; b += 16) {
                const unsigned bl = b % nblk_local;
                const float nb = (float)(min(rows, (bl + 1) * kTileRows) - bl * kTileRows);
                acc = f4_fma(partials[(size_t)b * 2 * C4 + c4], nb, acc);
            }
        }
    }
    sh[rl][cl] = acc;
    __syncthreads();
    float4 mean = sh[0][cl];
#pragma unroll
    for (int g = 1; g < 16; ++g) mean = f4_add(mean, sh[g][cl]);
    mean = f4_scale(mean, inv_rows);
    __syncthreads();
    acc = f4_zero();
    if (ok) {
        if (pre) {
#pragma unroll
            for (unsigned u = 0; u < kBnPre; ++u) {
                const unsigned b = rl + 16 * u;
                if (b < nblk) {
                    const unsigned bl = b % nblk_local;
                    const float nb = (float)(min(rows, (bl + 1) * kTileRows) - bl * kTileRows);
                    const float4 d = f4_sub(pm[u], mean);
                    acc = f4_add(acc, f4_fma(f4_mul(d, d), nb, pq[u]));
                }
            }
        } else {
#pragma unroll 4
            for (unsigned b = rl; b < nblk; b += 16) {
                const unsigned bl = b % nblk_local;
                const float nb = (float)(min(rows, (bl + 1) * kTileRows) - bl * kTileRows);
                const float4 d = f4_sub(partials[(size_t)b * 2 * C4 + c4], mean);
                acc = f4_add(acc, f4_fma(f4_mul(d, d), nb, partials[(size_t)b * 2 * C4 + C4 + c4]));
            }
        }
    }
    sh[rl][cl] = acc;
    __syncthreads();
    float4 var = sh[0][cl];
#pragma unroll
    for (int g = 1; g < 16; ++g) var = f4_add(var, sh[g][cl]);
    var = f4_scale(var, inv_rows);                              // biased (tf.nn.moments)
    const float4 rstd = make_float4(rsqrtf(var.x + eps), rsqrtf(var.y + eps), rsqrtf(var.z + eps), rsqrtf(var.w + eps));
    if (!ok) return;
    if (blockIdx.y == 0 && rl == 0) {
        save_mean[c4] = mean;
        save_rstd[c4] = rstd;
        if (moving_mean) {                                      // assign_moving_average, decay = momentum
            const float k = 1.f - momentum;
            moving_mean[c4] = f4_fma(mean, k, f4_scale(moving_mean[c4], momentum));
            moving_var[c4] = f4_fma(var, k, f4_scale(moving_var[c4], momentum));
        }
    }
    const float4 sc = f4_mul(rstd, gamma[c4]), bt = beta[c4];
    const bool dropping = recalgo_drop::enabled(out_drop);
    const recalgo_drop::Key dkey = dropping ? recalgo_drop::make_key(out_drop) : recalgo_drop::Key{0u, 0u};
#pragma unroll
    for (unsigned k = 0; k < kTileRows / 16; ++k) {
        const unsigned r = r0 + rl + 16 * k;
        if (r < rows) {
            const float4 xh = f4_sub(xv[k], mean);
            float4 o = make_float4(fmaf(xh.x, sc.x, bt.x), fmaf(xh.y, sc.y, bt.y), fmaf(xh.z, sc.z, bt.z), fmaf(xh.w, sc.w, bt.w));
            if (dropping) o = f4_mul(o, recalgo_drop::factor4(out_drop, dkey, (r * C4 + c4) * 4u));
            y[(size_t)r * C4 + c4] = o;
        }
    }
}

// partial[blk][0:C] = colsum(g), [C:2C] = colsum(g * xhat)
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_kernel(const float4* __restrict__ x,
                                                                 const float4* __restrict__ g,
                                                                 const float4* __restrict__ mean,
                                                                 const float4* __restrict__ rstd, unsigned rows,
                                                                 unsigned C4, float4* __restrict__ partials,
                                                                 recalgo_drop::Spec g_drop) {
    // g_drop: the BatchNorm's output went through a dropout (see bn_finalize_apply_kernel): its gradient is g * keep / (1 - rate)
    __shared__ float4 sh[kThreads];
    const unsigned cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const unsigned c4 = blockIdx.x * 16 + cl;
    const unsigned r0 = blockIdx.y * kTileRows;
    const bool dropping = recalgo_drop::enabled(g_drop);
    const recalgo_drop::Key dkey = dropping ? recalgo_drop::make_key(g_drop) : recalgo_drop::Key{0u, 0u};
    float4 sg = f4_zero(), sgx = f4_zero();
    if (c4 < C4) {
        const float4 mu = mean[c4], rs4 = rstd[c4];
#pragma unroll
        for (unsigned k = 0; k < kTileRows / 16; ++k) {
            const unsigned r = r0 + rl + 16 * k;
            if (r < rows) {
                float4 gv = g[(size_t)r * C4 + c4];
                if (dropping) gv = f4_mul(gv, recalgo_drop::factor4(g_drop, dkey, (r * C4 + c4) * 4u));
                const float4 xh = f4_mul(f4_sub(x[(size_t)r * C4 + c4], mu), rs4);
                sg = f4_add(sg, gv);
                sgx = f4_add(sgx, f4_mul(gv, xh));
            }
        }
    }
    sg = tile_colsum(sg, sh, cl);
    sgx = tile_colsum(sgx, sh, cl);
    if (rl == 0 && c4 < C4) {
        partials[(size_t)blockIdx.y * 2 * C4 + c4] = sg;
        partials[(size_t)blockIdx.y * 2 * C4 + C4 + c4] = sgx;
    }
}

// dbeta / dgamma = fixed-order column sums of the partial rows (16 interleaved row groups, then the groups), repeated by
// every workgroup of a column block (see bn_finalize_apply_kernel) and recorded by tile row 0; then
// dx = gamma * rstd * (g - (dbeta + xhat * dgamma) / rows) on the 64 x 64 tile.
// ACT = 1 + RECALGO_ACT_PRELU / 1 + RECALGO_ACT_DICE: x = act(z, alpha) came out of a per-channel activation (the dense ->
// dice -> batch_norm layers of din.py:262-266) and the pass continues through it — dx is then dL/dz and act_partials[tile
// row][C] the tile's terms of dL/dalpha (the layout and order of act_bwd_tile_kernel in tail.hip) — instead of a launch and a
// round trip of dL/dx in between.
template <int ACT>
__global__ __launch_bounds__(kThreads) void bn_bwd_sum_apply_kernel(
    const float4* __restrict__ x, const float4* __restrict__ g, const float4* __restrict__ gamma,
    const float4* __restrict__ mean, const float4* __restrict__ rstd, const float4* __restrict__ partials,
    unsigned nblk, unsigned nblk_local, unsigned rank, unsigned rows, unsigned C4, float4* __restrict__ dbeta,
    float4* __restrict__ dgamma, float4* __restrict__ dx, const float4* __restrict__ act_z,
    const float4* __restrict__ act_alpha, float4* __restrict__ act_partials, int dx_relu, float dx_scale,
    recalgo_drop::Spec g_drop) {
    // dx_scale (with dx_relu): x is a ReLU output that went through a dropout BEFORE this BatchNorm (dense(relu) -> dropout ->
    // batch_norm, deepfm.py:207-211; x = relu * keep / (1 - rate), so x > 0 <=> relu > 0 and kept): dx := dx / (1 - rate) there.
    // g_drop: a dropout AFTER this BatchNorm (see bn_bwd_reduce_kernel)
    // dx_relu: x IS a ReLU output (tf.layers.dense(..., relu) -> tf.layers.batch_normalization, deepfm.py:206-211): dx is zeroed where
    // x <= 0, so that the dense layer's backward gets its gradient already masked (see recalgo_dense_bwd_bn dx_relu_mask)
    // nblk = world * nblk_local partial rows (see bn_finalize_apply_kernel).  dx uses the sums over ALL ranks' tiles;
    // dbeta / dgamma get THIS rank's share (the data-parallel all-reduce of the dense gradients adds the ranks up)
    __shared__ float4 sh[2][16][17];
    const unsigned cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const unsigned c4 = blockIdx.x * 16 + cl;
    const bool ok = c4 < C4;
    // (as bn_finalize_apply_kernel: the tile's operands are requested together with the partial rows, before the reduction —
    // one memory round trip instead of two)
    const unsigned r0 = blockIdx.y * kTileRows;
    float4 xq[kTileRows / 16], gq[kTileRows / 16], zq[kTileRows / 16];
#pragma unroll
    for (unsigned t = 0; t < kTileRows / 16; ++t) {
        const unsigned r = r0 + rl + 16 * t;
        const bool in = ok && r < rows;
        xq[t] = in ? x[(size_t)r * C4 + c4] : f4_zero();
        gq[t] = in ? g[(size_t)r * C4 + c4] : f4_zero();
        if (ACT) zq[t] = in ? act_z[(size_t)r * C4 + c4] : f4_zero();
    }
    float4 sb = f4_zero(), sg = f4_zero();
    if (ok) {
#pragma unroll 4
        for (unsigned b = rl; b < nblk; b += 16) {
            sb = f4_add(sb, partials[(size_t)b * 2 * C4 + c4]);
            sg = f4_add(sg, partials[(size_t)b * 2 * C4 + C4 + c4]);
        }
    }
    sh[0][rl][cl] = sb;
    sh[1][rl][cl] = sg;
    __syncthreads();
    float4 db = sh[0][0][cl], dg = sh[1][0][cl];
#pragma unroll
    for (int q = 1; q < 16; ++q) {
        db = f4_add(db, sh[0][q][cl]);
        dg = f4_add(dg, sh[1][q][cl]);
    }
    if (nblk == nblk_local) {
        if (ok && blockIdx.y == 0 && rl == 0) {
            dbeta[c4] = db;
            dgamma[c4] = dg;
        }
    } else if (blockIdx.y == 0) {                              // (uniform over the workgroup)
        __syncthreads();
        float4 lb = f4_zero(), lg = f4_zero();
        if (ok)
            for (unsigned b = rank * nblk_local + rl; b < (rank + 1) * nblk_local; b += 16) {
                lb = f4_add(lb, partials[(size_t)b * 2 * C4 + c4]);
                lg = f4_add(lg, partials[(size_t)b * 2 * C4 + C4 + c4]);
            }
        sh[0][rl][cl] = lb;
        sh[1][rl][cl] = lg;
        __syncthreads();
        if (ok && rl == 0) {
            float4 tb = sh[0][0][cl], tg = sh[1][0][cl];
#pragma unroll
            for (int q = 1; q < 16; ++q) {
                tb = f4_add(tb, sh[0][q][cl]);
                tg = f4_add(tg, sh[1][q][cl]);
            }
            dbeta[c4] = tb;
            dgamma[c4] = tg;
        }
    }
    if (!ok && ACT == 0) return;
    const float inv_rows = 1.0f / ((float)rows * (float)(nblk / nblk_local));
    const bool dropping = recalgo_drop::enabled(g_drop);
    const recalgo_drop::Key dkey = dropping ? recalgo_drop::make_key(g_drop) : recalgo_drop::Key{0u, 0u};
    float4 da = f4_zero();
    if (ok) {
        const float4 mu = mean[c4], rs4 = rstd[c4];
        const float4 k = f4_mul(gamma[c4], rs4);
        const float4 al = ACT ? act_alpha[c4] : f4_zero();
#pragma unroll
        for (unsigned t = 0; t < kTileRows / 16; ++t) {
            const unsigned r = r0 + rl + 16 * t;
            if (r < rows) {
                const float4 xv = xq[t];
                const float4 xh = f4_mul(f4_sub(xv, mu), rs4);
                float4 gv = gq[t];
                if (dropping) gv = f4_mul(gv, recalgo_drop::factor4(g_drop, dkey, (r * C4 + c4) * 4u));
                float4 d =
                    make_float4(k.x * (gv.x - inv_rows * (db.x + xh.x * dg.x)), k.y * (gv.y - inv_rows * (db.y + xh.y * dg.y)),
                                k.z * (gv.z - inv_rows * (db.z + xh.z * dg.z)), k.w * (gv.w - inv_rows * (db.w + xh.w * dg.w)));
                if (ACT) {
                    const float4 z = zq[t];
                    float4 t4;
                    d = make_float4(recalgo_act::bwd<ACT == 1 + RECALGO_ACT_DICE>(z.x, al.x, d.x, t4.x),
                                    recalgo_act::bwd<ACT == 1 + RECALGO_ACT_DICE>(z.y, al.y, d.y, t4.y),
                                    recalgo_act::bwd<ACT == 1 + RECALGO_ACT_DICE>(z.z, al.z, d.z, t4.z),
                                    recalgo_act::bwd<ACT == 1 + RECALGO_ACT_DICE>(z.w, al.w, d.w, t4.w));
                    da = f4_add(da, t4);
                }
                if (dx_relu) d = make_float4(xv.x > 0.f ? d.x * dx_scale : 0.f, xv.y > 0.f ? d.y * dx_scale : 0.f,
                                             xv.z > 0.f ? d.z * dx_scale : 0.f, xv.w > 0.f ? d.w * dx_scale : 0.f);
                dx[(size_t)r * C4 + c4] = d;
            }
        }
    }
    if (ACT) {
        __syncthreads();                                       // (sh: the column sums above are consumed)
        float4* shf = &sh[0][0][0];
        shf[threadIdx.x] = da;
        __syncthreads();
        if (rl == 0 && ok) {
            float4 t = shf[cl];
#pragma unroll
            for (unsigned q = 1; q < 16; ++q) t = f4_add(t, shf[q * 16 + cl]);
            act_partials[(size_t)blockIdx.y * C4 + c4] = t;
        }
    }
}

inline bool width_ok(int C) { return C >= 4 && C % 4 == 0; }
inline int nblk_of(int rows) { return cdiv(rows, kTileRows); }

// ---------------------------------------------------------------------------------------
// One-unit dense head over a (virtual) concatenation of up to 4 row-major inputs:
//   logit[b] = bias + sum_p <x_p[b, :], w[off_p : off_p + width_p]>
// the `tf.concat([...]) -> tf.layers.dense(., 1)` tail every model ends with.  As library calls it
// is a concat copy + split-K GEMV (2 launches) forward and a reduce, a GEMV, a rank-1 GEMM and two
// slice copies backward — ~10 launches of ~5 us for 9 MB of traffic.  Here: one pass each way.
// ---------------------------------------------------------------------------------------
constexpr int kHeadMaxParts = 4;
struct HeadParts {
    const float* x[kHeadMaxParts];
    float* dx[kHeadMaxParts];
    int width[kHeadMaxParts];
    int n;
};

// one wave per example; lanes stride the columns (each wave load = 256 contiguous bytes)
__global__ __launch_bounds__(256) void dense1_fwd_kernel(HeadParts P, int B, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63;
    if (b >= B) return;
    float acc = 0.f;
    int off = 0;
    for (int p = 0; p < P.n; ++p) {
        const float* __restrict__ xr = P.x[p] + (size_t)b * P.width[p];
        for (int j = lane; j < P.width[p]; j += 64) acc = fmaf(xr[j], w[off + j], acc);
        off += P.width[p];
    }
    acc = wave_sum(acc);
    if (lane == 0) out[b] = acc + (bias ? bias[0] : 0.f);
}

// workgroup = kHeadRows examples x 256 columns of the virtual concat (grid.x = column chunk, grid.y =
// row tile); thread = one column: dx[b, c] = g[b] w[c] is written and dw[c] += g[b] x[b, c]
// accumulated in the same pass over x (8 independent loads in flight per thread).  Per-tile partial
// rows of dw (and of db in column C) are summed in fixed order by colsum16.
constexpr int kHeadRows = 32;
constexpr int kHeadMaxPartialRows = 1024;     // the fixed-order column sum behind walks partial rows / 16 per thread
// rows per workgroup: one 32-row tile up to 32 K examples, more tiles per workgroup beyond (AFM's attention net runs this
// head over B * pairs = 1.3 M rows: 41 K partial rows made the column sum 330 us)
inline int head_rows_per_block(int B) {
    const int tiles = cdiv(B > 0 ? B : 1, kHeadRows);
    return cdiv(tiles, kHeadMaxPartialRows) * kHeadRows;
}
__global__ __launch_bounds__(256) void dense1_bwd_kernel(HeadParts P, int B, int C, int rows_per_block,
                                                         const float* __restrict__ w, const float* __restrict__ g,
                                                         float* __restrict__ partials) {
    __shared__ float gs[kHeadRows];
    const int c = blockIdx.x * 256 + threadIdx.x;          // column of the concat
    int p = 0, off = 0;
    if (c < C)
        while (c >= off + P.width[p]) off += P.width[p++];
    const int W = c < C ? P.width[p] : 1, j = c - off;
    const float wj = c < C ? w[c] : 0.f;
    float acc = 0.f, sgs = 0.f;
    const int row_end = min(B, (int)(blockIdx.y + 1) * rows_per_block);
    for (int b0 = blockIdx.y * rows_per_block; b0 < row_end; b0 += kHeadRows) {
        const int nb = min(kHeadRows, row_end - b0);
        __syncthreads();
        if ((int)threadIdx.x < kHeadRows) gs[threadIdx.x] = (int)threadIdx.x < nb ? g[b0 + threadIdx.x] : 0.f;
        __syncthreads();
        if (c < C) {
            const float* __restrict__ xp = P.x[p] + (size_t)b0 * W + j;
            float* __restrict__ dxp = P.dx[p] ? P.dx[p] + (size_t)b0 * W + j : nullptr;
            int r = 0;
            for (; r + 8 <= nb; r += 8) {
                float xv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) xv[k] = xp[(size_t)(r + k) * W];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    acc = fmaf(gs[r + k], xv[k], acc);
                    if (dxp) dxp[(size_t)(r + k) * W] = gs[r + k] * wj;
                }
            }
            for (; r < nb; ++r) {
                acc = fmaf(gs[r], xp[(size_t)r * W], acc);
                if (dxp) dxp[(size_t)r * W] = gs[r] * wj;
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0)
            for (int r = 0; r < nb; ++r) sgs += gs[r];
    }
    float* __restrict__ prow = partials + (size_t)blockIdx.y * (C + 1);
    if (c < C) prow[c] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) prow[C] = sgs;
}


// ---------------------------------------------------------------------------------------
// The whole logit / loss tail of a TRAIN step in one pass (algorithm/DeepFM/deepfm.py:206-217,234-235 and the same
// tail in every model_fn): logit = sum_p <x_p, w_p> + bias + addends, prob = sigmoid, mean sigmoid-CE, and — the
// loss-gradient seed being known — d loss / d logit, dx_p = dlogit * w_p, and per-workgroup partial sums of
// dw_p = sum_b dlogit_b x_p[b, :], d bias and the loss.  Was: dense1_fwd + sigmoid_ce + dense1_bwd + colsum16
// (4 launches of 5-7 us on 9 MB of traffic).  Workgroup = kTailRows examples: phase 1 one wave per example (dot
// products, lanes stride the columns), phase 2 one thread per column (the tile is re-read from L1/L2).
// partial row layout: [dw over the C concatenated columns | d bias | loss / B]  (C + 2 floats); the rows are summed
// in fixed order by the deferred-sum launch (recalgo_dense_bwd_weights_reduce).
// ---------------------------------------------------------------------------------------
constexpr int kTailRows = 4;      // (A/B on one box, DCN step: 4 rows 0.2147 ms, 8 rows 0.2161, 16 rows slower still — more, smaller workgroups
                                  //  hide the load -> dot -> store chain of a row better than the extra partial rows cost)
struct TailArgs {
    const float* x[kHeadMaxParts];
    const float* w[kHeadMaxParts];      // one weight vector per part (xDeepFM sums three one-unit heads)
    float* dx[kHeadMaxParts];           // may be null per part
    int width[kHeadMaxParts];
    int relu[kHeadMaxParts];            // part p is a ReLU output: dx_p := 0 where x_p <= 0 (see recalgo_dense_bwd_bn)
    int n;
    const float* bias;                  // [1] or null
    const float* addend[2];             // [B] extra logit terms or null (DeepFM: FM first / second order)
    const float* labels;                // [B]
    const float* loss_addend;           // device scalar added to the loss value (a regulariser's VALUE), or null
    float grad_scale;
    float* logit; float* prob; float* dlogit;       // [B]
    float* partials;                    // [gridDim.x][C + 2]
    int B, C;
};

__global__ __launch_bounds__(256) void logit_loss_kernel(TailArgs P) {
    // dynamic LDS: the workgroup's [kTailRows][C] tile of the (virtually concatenated) inputs + the C weights: the
    // inputs are read from HBM once and serve both the dot products and the weight-gradient column pass
    extern __shared__ __attribute__((aligned(16))) float tail_smem[];
    __shared__ float s_dl[kTailRows], s_loss[kTailRows];
    const int C = P.C;
    float* tile = tail_smem;                      // [kTailRows][C]
    float* wcat = tail_smem + kTailRows * C;      // [C]
    const int b0 = blockIdx.x * kTailRows;
    const int nb = min(kTailRows, P.B - b0);
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float invB = 1.0f / (float)P.B;
    constexpr int RW = kTailRows / 4;
    {
        int off = 0;
        for (int p = 0; p < P.n; ++p) {
            const int W = P.width[p];
            for (int j = threadIdx.x; j < W; j += 256) wcat[off + j] = P.w[p][j];
            if ((W & 3) == 0 && (reinterpret_cast<uintptr_t>(P.x[p]) & 15) == 0) {
                // float4 rows, all of a wave's loads of a pass in flight before the first LDS store (a scalar
                // load -> store loop over a 416-float row was 7 dependent round trips per row)
                const int W4 = W >> 2;
                const float4* __restrict__ x4 = reinterpret_cast<const float4*>(P.x[p]);
                for (int j0 = (int)lane; j0 < W4; j0 += 128) {
                    float4 v[RW][2];
#pragma unroll
                    for (int i = 0; i < RW; ++i)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int r = wave + 4 * i, j = j0 + 64 * u;
                            v[i][u] = (r < nb && j < W4) ? x4[(size_t)(b0 + r) * W4 + j] : f4_zero();
                        }
#pragma unroll
                    for (int i = 0; i < RW; ++i)
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int r = wave + 4 * i, j = j0 + 64 * u;
                            if (r < nb && j < W4) {
                                float* d = tile + r * C + off + 4 * j;
                                if (((r * C + off) & 3) == 0) {
                                    *reinterpret_cast<float4*>(d) = v[i][u];
                                } else {
                                    d[0] = v[i][u].x; d[1] = v[i][u].y; d[2] = v[i][u].z; d[3] = v[i][u].w;
                                }
                            }
                        }
                }
            } else {
#pragma unroll
                for (int i = 0; i < RW; ++i) {
                    const int r = wave + 4 * i;
                    if (r < nb) {
                        const float* __restrict__ xr = P.x[p] + (size_t)(b0 + r) * W;
                        for (int j = lane; j < W; j += 64) tile[r * C + off + j] = xr[j];
                    }
                }
            }
            off += W;
        }
    }
    __syncthreads();
    // phase 1: wave w owns rows w, w + 4, ...
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int r = wave + 4 * i;
        float d = 0.f, ls = 0.f;
        float acc = 0.f;
        if (r < nb)
            for (int c = lane; c < C; c += 64) acc = fmaf(tile[r * C + c], wcat[c], acc);
        const float dot = wave_sum(acc);
        if (r < nb) {
            const int b = b0 + r;
            float x = dot + (P.bias ? P.bias[0] : 0.f);
            if (P.addend[0]) x += P.addend[0][b];
            if (P.addend[1]) x += P.addend[1][b];
            const float z = P.labels[b];
            const float e = expf(-fabsf(x));
            ls = fmaxf(x, 0.f) - x * z + log1pf(e);               // tf.nn.sigmoid_cross_entropy_with_logits
            const float rr = e / (1.0f + e);
            const float pr = x >= 0.f ? 1.0f / (1.0f + e) : rr;
            d = (((x >= 0.f ? 1.0f : 0.f) - z) + (x >= 0.f ? -rr : rr)) * P.grad_scale * invB;
            if (lane == 0) {
                P.logit[b] = x;
                P.prob[b] = pr;
                P.dlogit[b] = d;
            }
        }
        if (lane == 0) { s_dl[r] = d; s_loss[r] = ls; }
    }
    __syncthreads();
    // phase 2: one thread per column of the concatenation
    float* __restrict__ prow = P.partials + (size_t)blockIdx.x * (C + 2);
    for (int c = threadIdx.x; c < C; c += 256) {
        int p = 0, off = 0;
        while (c >= off + P.width[p]) off += P.width[p++];
        const int W = P.width[p], j = c - off;
        float* __restrict__ dxp = P.dx[p] ? P.dx[p] + (size_t)b0 * W + j : nullptr;
        const float wj = wcat[c];
        const bool relu = P.relu[p] != 0;
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < kTailRows; ++r) {
            if (r < nb) {
                const float xv = tile[r * C + c];
                acc = fmaf(s_dl[r], xv, acc);
                if (dxp) dxp[(size_t)r * W] = (relu && !(xv > 0.f)) ? 0.f : s_dl[r] * wj;
            }
        }
        prow[c] = acc;
    }
    if (threadIdx.x == 0) {
        float sd = 0.f, sl = 0.f;
        for (int r = 0; r < kTailRows; ++r) { sd += s_dl[r]; sl += s_loss[r]; }
        prow[C] = sd;
        prow[C + 1] = sl * invB + ((blockIdx.x == 0 && P.loss_addend) ? P.loss_addend[0] : 0.f);
    }
}

}  // namespace

static int head_parts(const float* const* x_parts, float* const* dx_parts, const int* widths, int n_parts, int B,
                      HeadParts* P) {
    if (n_parts < 1 || n_parts > kHeadMaxParts || x_parts == nullptr || widths == nullptr) return -1;
    int C = 0;
    P->n = n_parts;
    for (int p = 0; p < kHeadMaxParts; ++p) {
        const bool on = p < n_parts;
        if (on && ((B > 0 && x_parts[p] == nullptr) || widths[p] < 1)) return -1;
        P->x[p] = on ? x_parts[p] : nullptr;
        P->dx[p] = on && dx_parts ? dx_parts[p] : nullptr;
        P->width[p] = on ? widths[p] : 0;
        C += P->width[p];
    }
    return C;
}

RECALGO_EXPORT int recalgo_dense1_fwd(const float* const* x_parts, const int* widths, int n_parts, int B,
                                      const float* w, const float* bias, float* out, recalgo_stream_t stream) {
    HeadParts P;
    const int C = head_parts(x_parts, nullptr, widths, n_parts, B, &P);
    RECALGO_REQUIRE(C > 0 && B >= 0 && w != nullptr && (B == 0 || out != nullptr));
    if (B == 0) return 0;
    hipLaunchKernelGGL(dense1_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, as_stream(stream), P, B, w, bias, out);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int64_t recalgo_dense1_bwd_workspace_bytes(int B, int C) {
    return ((int64_t)cdiv(B > 0 ? B : 1, kHeadRows) * (C + 1) + 1) * (int64_t)sizeof(float);   // + 1 scratch float
}

RECALGO_EXPORT int recalgo_dense1_bwd(const float* const* x_parts, const int* widths, int n_parts, int B,
                                      const float* w, const float* g, float* const* dx_parts, float* dw, float* dbias,
                                      void* workspace, recalgo_stream_t stream) {
    HeadParts P;
    const int C = head_parts(x_parts, dx_parts, widths, n_parts, B, &P);
    RECALGO_REQUIRE(C > 0 && B >= 0 && w != nullptr && dw != nullptr);
    hipStream_t st = as_stream(stream);
    if (B == 0) {
        (void)hipMemsetAsync(dw, 0, (size_t)C * sizeof(float), st);
        if (dbias) (void)hipMemsetAsync(dbias, 0, sizeof(float), st);
        RECALGO_RETURN_LAST();
    }
    RECALGO_REQUIRE(g != nullptr && workspace != nullptr);
    const int rpb = head_rows_per_block(B);
    const int blocks = cdiv(B, rpb);
    float* partials = static_cast<float*>(workspace);
    hipLaunchKernelGGL(dense1_bwd_kernel, dim3(cdiv(C, 256), blocks), dim3(256), 0, st, P, B, C, rpb, w, g, partials);
    // columns [0, C) -> dw, column C -> dbias (or the scratch float behind the partial rows)
    launch_colsum16(partials, (unsigned)blocks, (unsigned)(C + 1), dw, (unsigned)C,
                    dbias ? dbias : partials + (size_t)blocks * (C + 1), st);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_mlp_width_supported(int C) { return width_ok(C) ? 1 : 0; }

RECALGO_EXPORT int64_t recalgo_relu_bwd_bias_workspace_bytes(int rows, int C) {
    if (rows <= 0 || !width_ok(C)) return 0;
    return (int64_t)nblk_of(rows) * C * (int64_t)sizeof(float);
}

RECALGO_EXPORT int recalgo_relu_bwd_bias(const float* g, const float* y, int rows, int C, float* g_out, float* dbias,
                                         void* workspace, recalgo_stream_t stream) {
    RECALGO_REQUIRE(rows > 0 && width_ok(C) && g && dbias && workspace && ((y == nullptr) == (g_out == nullptr)));
    hipStream_t st = as_stream(stream);
    const int nb = nblk_of(rows);
    float* partials = static_cast<float*>(workspace);
    hipLaunchKernelGGL(relu_bwd_bias_kernel, dim3(cdiv(C / 4, 16), nb), dim3(kThreads), 0, st, reinterpret_cast<const float4*>(g),
                       reinterpret_cast<const float4*>(y), (unsigned)rows, (unsigned)(C / 4),
                       reinterpret_cast<float4*>(g_out), reinterpret_cast<float4*>(partials));
    launch_colsum16(partials, (unsigned)nb, (unsigned)C, dbias, (unsigned)C, static_cast<float*>(nullptr), st);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int64_t recalgo_batchnorm_workspace_bytes(int rows, int C) {
    if (rows <= 0 || !width_ok(C)) return 0;
    return (int64_t)nblk_of(rows) * 2 * C * (int64_t)sizeof(float);
}

RECALGO_EXPORT int recalgo_batchnorm_train_fwd(const float* x, const float* gamma, const float* beta, int rows, int C,
                                               float eps, float momentum, float* moving_mean, float* moving_var,
                                               float* y, float* save_mean, float* save_rstd, void* workspace,
                                               recalgo_stream_t stream) {
    RECALGO_REQUIRE(rows > 0 && width_ok(C) && x && gamma && beta && y && save_mean && save_rstd && workspace);
    RECALGO_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr));
    hipStream_t st = as_stream(stream);
    const int nb = nblk_of(rows);
    const unsigned C4 = C / 4;
    float* partials = static_cast<float*>(workspace);
    hipLaunchKernelGGL(bn_moments_kernel, dim3(cdiv(C4, 16), nb), dim3(kThreads), 0, st, reinterpret_cast<const float4*>(x),
                       (unsigned)rows, C4, reinterpret_cast<float4*>(partials));
    hipLaunchKernelGGL(bn_finalize_apply_kernel, dim3(cdiv(C4, 16), nb), dim3(kThreads), 0, st,
                       reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(gamma),
                       reinterpret_cast<const float4*>(beta), reinterpret_cast<const float4*>(partials), (unsigned)nb,
                       (unsigned)nb, (unsigned)rows, C4, eps, momentum, reinterpret_cast<float4*>(moving_mean),
                       reinterpret_cast<float4*>(moving_var), reinterpret_cast<float4*>(save_mean),
                       reinterpret_cast<float4*>(save_rstd), reinterpret_cast<float4*>(y), recalgo_drop::disabled());
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_batchnorm_train_bwd(const float* x, const float* gamma, const float* save_mean,
                                               const float* save_rstd, const float* g, int rows, int C, float* dx,
                                               float* dgamma, float* dbeta, void* workspace, int dx_relu,
                                               recalgo_stream_t stream) {
    return recalgo_batchnorm_train_bwd_act(x, gamma, save_mean, save_rstd, g, nullptr, rows, C, RECALGO_ACT_NONE, nullptr, nullptr, dx,
                                           dgamma, dbeta, nullptr, workspace, dx_relu, stream);
}

RECALGO_EXPORT int64_t recalgo_batchnorm_bwd_act_workspace_bytes(int rows, int C) {
    if (rows <= 0 || !width_ok(C)) return 0;
    return (int64_t)nblk_of(rows) * 3 * C * (int64_t)sizeof(float);
}

RECALGO_EXPORT int recalgo_batchnorm_train_bwd_act(const float* x, const float* gamma, const float* save_mean,
                                                   const float* save_rstd, const float* g, const float* sums, int rows, int C,
                                                   int act_kind,
                                                   const float* act_z, const float* act_alpha, float* dx, float* dgamma,
                                                   float* dbeta, float* dalpha, void* workspace, int dx_relu,
                                                   recalgo_stream_t stream) {
    return recalgo_batchnorm_train_bwd_drop(x, gamma, save_mean, save_rstd, g, sums, rows, C, act_kind, act_z, act_alpha, dx, dgamma, dbeta,
                                            dalpha, workspace, dx_relu, 1.0f, nullptr, stream);
}

RECALGO_EXPORT int recalgo_batchnorm_train_bwd_drop(const float* x, const float* gamma, const float* save_mean,
                                                    const float* save_rstd, const float* g, const float* sums, int rows, int C,
                                                    int act_kind, const float* act_z, const float* act_alpha, float* dx,
                                                    float* dgamma, float* dbeta, float* dalpha, void* workspace, int dx_relu,
                                                    float dx_scale, const recalgo_dropout_t* g_drop, recalgo_stream_t stream) {
    RECALGO_REQUIRE(recalgo_drop::abi_ok(g_drop) && (g_drop == nullptr || (sums == nullptr && (int64_t)rows * C < ((int64_t)1 << 32))));
    const recalgo_drop::Spec gd = recalgo_drop::from_abi(g_drop);
    RECALGO_REQUIRE(rows > 0 && width_ok(C) && x && gamma && save_mean && save_rstd && g && dx && dgamma && dbeta &&
                    workspace);
    RECALGO_REQUIRE(act_kind == RECALGO_ACT_NONE ||
                    ((act_kind == RECALGO_ACT_PRELU || act_kind == RECALGO_ACT_DICE) && act_z && act_alpha && !dx_relu));
    hipStream_t st = as_stream(stream);
    const int nb = nblk_of(rows);
    const unsigned C4 = C / 4;
    float* ws = static_cast<float*>(workspace);
    float* act_partials = ws + (size_t)nb * 2 * C;              // [nb][C]: the tile rows' terms of dalpha
    // sums: the partial rows (colsum g | colsum g * xhat per 64-row tile) are already there — left by the epilogue of the
    // kernel that produced g (recalgo_dense_bwd_bn) — and the pass over g and x that computes them is skipped
    const float* partials = sums ? sums : ws;
    if (sums == nullptr)
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(cdiv(C4, 16), nb), dim3(kThreads), 0, st, reinterpret_cast<const float4*>(x),
                           reinterpret_cast<const float4*>(g), reinterpret_cast<const float4*>(save_mean),
                           reinterpret_cast<const float4*>(save_rstd), (unsigned)rows, C4, reinterpret_cast<float4*>(ws), gd);
#define RECALGO_BN_APPLY(ACT)                                                                                                   \
    hipLaunchKernelGGL(bn_bwd_sum_apply_kernel<ACT>, dim3(cdiv(C4, 16), nb), dim3(kThreads), 0, st,                             \
                       reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(g),                                  \
                       reinterpret_cast<const float4*>(gamma), reinterpret_cast<const float4*>(save_mean),                      \
                       reinterpret_cast<const float4*>(save_rstd), reinterpret_cast<const float4*>(partials), (unsigned)nb,     \
                       (unsigned)nb, 0u, (unsigned)rows, C4, reinterpret_cast<float4*>(dbeta), reinterpret_cast<float4*>(dgamma), \
                       reinterpret_cast<float4*>(dx), reinterpret_cast<const float4*>(act_z),                                   \
                       reinterpret_cast<const float4*>(act_alpha), reinterpret_cast<float4*>(act_partials), dx_relu, dx_scale, gd)
    if (act_kind == RECALGO_ACT_NONE) RECALGO_BN_APPLY(0);
    else if (act_kind == RECALGO_ACT_PRELU) RECALGO_BN_APPLY(1 + RECALGO_ACT_PRELU);
    else RECALGO_BN_APPLY(1 + RECALGO_ACT_DICE);
#undef RECALGO_BN_APPLY
    // dalpha == NULL: the caller sums the nb partial rows (a job of the step's deferred-sum launch)
    if (act_kind != RECALGO_ACT_NONE && dalpha)
        launch_colsum16(act_partials, (unsigned)nb, (unsigned)C, dalpha, (unsigned)C, static_cast<float*>(nullptr), st);
    RECALGO_RETURN_LAST();
}

// ---- Sync-BatchNorm building blocks: the two launches of each direction as separate entry points, so that the per-tile
// partials of all ranks can be all-gathered in between (include/recalgo.h) ---------------------------------------------
RECALGO_EXPORT int recalgo_batchnorm_partial_rows(int rows) { return rows > 0 ? nblk_of(rows) : 0; }

RECALGO_EXPORT int recalgo_batchnorm_moments(const float* x, int rows, int C, float* partials, recalgo_stream_t stream) {
    RECALGO_REQUIRE(rows > 0 && width_ok(C) && x && partials);
    const unsigned C4 = C / 4;
    hipLaunchKernelGGL(bn_moments_kernel, dim3(cdiv(C4, 16), nblk_of(rows)), dim3(kThreads), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(x), (unsigned)rows, C4, reinterpret_cast<float4*>(partials));
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_batchnorm_apply(const float* x, const float* gamma, const float* beta, const float* partials,
                                           int world, int rows, int C, float eps, float momentum, float* moving_mean,
                                           float* moving_var, float* y, float* save_mean, float* save_rstd,
                                           recalgo_stream_t stream) {
    return recalgo_batchnorm_apply_drop(x, gamma, beta, partials, world, rows, C, eps, momentum, moving_mean, moving_var, y, save_mean,
                                        save_rstd, nullptr, stream);
}

RECALGO_EXPORT int recalgo_batchnorm_apply_drop(const float* x, const float* gamma, const float* beta, const float* partials,
                                                int world, int rows, int C, float eps, float momentum, float* moving_mean,
                                                float* moving_var, float* y, float* save_mean, float* save_rstd,
                                                const recalgo_dropout_t* out_drop, recalgo_stream_t stream) {
    RECALGO_REQUIRE(recalgo_drop::abi_ok(out_drop) && (out_drop == nullptr || (int64_t)rows * C < ((int64_t)1 << 32)));
    RECALGO_REQUIRE(rows > 0 && world >= 1 && width_ok(C) && x && gamma && beta && partials && y && save_mean && save_rstd);
    RECALGO_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr));
    const int nb = nblk_of(rows);
    const unsigned C4 = C / 4;
    hipLaunchKernelGGL(bn_finalize_apply_kernel, dim3(cdiv(C4, 16), nb), dim3(kThreads), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(gamma),
                       reinterpret_cast<const float4*>(beta), reinterpret_cast<const float4*>(partials),
                       (unsigned)(nb * world), (unsigned)nb, (unsigned)rows, C4, eps, momentum,
                       reinterpret_cast<float4*>(moving_mean), reinterpret_cast<float4*>(moving_var),
                       reinterpret_cast<float4*>(save_mean), reinterpret_cast<float4*>(save_rstd), reinterpret_cast<float4*>(y),
                       recalgo_drop::from_abi(out_drop));
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_batchnorm_bwd_sums(const float* x, const float* save_mean, const float* save_rstd, const float* g,
                                              int rows, int C, float* partials, recalgo_stream_t stream) {
    RECALGO_REQUIRE(rows > 0 && width_ok(C) && x && save_mean && save_rstd && g && partials);
    const unsigned C4 = C / 4;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(cdiv(C4, 16), nblk_of(rows)), dim3(kThreads), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(g),
                       reinterpret_cast<const float4*>(save_mean), reinterpret_cast<const float4*>(save_rstd), (unsigned)rows, C4,
                       reinterpret_cast<float4*>(partials), recalgo_drop::disabled());
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_batchnorm_bwd_apply(const float* x, const float* gamma, const float* save_mean, const float* save_rstd,
                                               const float* g, const float* partials, int world, int rank, int rows, int C,
                                               float* dx, float* dgamma, float* dbeta, int dx_relu, recalgo_stream_t stream) {
    RECALGO_REQUIRE(rows > 0 && world >= 1 && rank >= 0 && rank < world && width_ok(C));
    RECALGO_REQUIRE(x && gamma && save_mean && save_rstd && g && partials && dx && dgamma && dbeta);
    const int nb = nblk_of(rows);
    const unsigned C4 = C / 4;
    hipLaunchKernelGGL(bn_bwd_sum_apply_kernel<0>, dim3(cdiv(C4, 16), nb), dim3(kThreads), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(g),
                       reinterpret_cast<const float4*>(gamma), reinterpret_cast<const float4*>(save_mean),
                       reinterpret_cast<const float4*>(save_rstd), reinterpret_cast<const float4*>(partials),
                       (unsigned)(nb * world), (unsigned)nb, (unsigned)rank, (unsigned)rows, C4,
                       reinterpret_cast<float4*>(dbeta), reinterpret_cast<float4*>(dgamma), reinterpret_cast<float4*>(dx),
                       static_cast<const float4*>(nullptr), static_cast<const float4*>(nullptr), static_cast<float4*>(nullptr), dx_relu, 1.0f,
                       recalgo_drop::disabled());
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int64_t recalgo_logit_loss_partial_rows(int B) { return (int64_t)cdiv(B > 0 ? B : 1, kTailRows); }

RECALGO_EXPORT int recalgo_logit_loss_fwd_bwd(const float* const* x_parts, const float* const* w_parts, const int* widths,
                                              int n_parts, const float* bias, const float* addend0, const float* addend1,
                                              const float* labels, const float* loss_addend, int B, float grad_scale,
                                              float* logit, float* prob,
                                              float* dlogit, float* const* dx_parts, const int* relu_parts, float* partials,
                                              recalgo_stream_t stream) {
    RECALGO_REQUIRE(n_parts >= 1 && n_parts <= kHeadMaxParts && x_parts && w_parts && widths && B > 0);
    RECALGO_REQUIRE(labels && logit && prob && dlogit && partials);
    TailArgs P;
    P.n = n_parts; P.C = 0;
    for (int p = 0; p < kHeadMaxParts; ++p) {
        const bool on = p < n_parts;
        if (on) RECALGO_REQUIRE(x_parts[p] != nullptr && w_parts[p] != nullptr && widths[p] >= 1);
        P.x[p] = on ? x_parts[p] : nullptr;
        P.w[p] = on ? w_parts[p] : nullptr;
        P.dx[p] = on && dx_parts ? dx_parts[p] : nullptr;
        P.width[p] = on ? widths[p] : 0;
        P.relu[p] = (on && relu_parts) ? relu_parts[p] : 0;
        P.C += P.width[p];
    }
    P.bias = bias; P.addend[0] = addend0; P.addend[1] = addend1; P.labels = labels; P.loss_addend = loss_addend; P.grad_scale = grad_scale;
    P.logit = logit; P.prob = prob; P.dlogit = dlogit; P.partials = partials; P.B = B;
    const size_t smem = (size_t)(kTailRows + 1) * P.C * sizeof(float);
    RECALGO_REQUIRE(smem <= 150 * 1024);
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&logit_loss_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(logit_loss_kernel, dim3(cdiv(B, kTailRows)), dim3(256), smem, as_stream(stream), P);
    RECALGO_RETURN_LAST();
}
