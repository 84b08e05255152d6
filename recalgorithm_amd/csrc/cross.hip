// K4: DCN CrossNet (algorithm/DCN/cross_layer.py:4-26 stacked by dcn.py:157-160), gfx950.
//
//   x_{l+1} = x0 * (x_l . w_l) + b_l + x_l ,   x_0 = x0                      (reference form)
//
// Because x_l . w_l is a scalar, the stack has a closed form that the fused kernels use:
//   x_l = c_l * x0 + B_l ,  B_l = sum_{j<l} b_j ,  c_0 = 1
//   p_l = x0 . w_l  (per example)      beta_l = B_l . w_l  (batch constant)
//   s_l = c_l * p_l + beta_l ,  c_{l+1} = c_l + s_l          =>  out = c_L * x0 + B_L
// All L dot products are independent, so a wave needs ONE batched shuffle reduction per example
// instead of L dependent ones, and the batch reductions of the backward collapse to
//   A_l = sum_b dp_l[b] * x0[b,:]   (L vectors),   G = sum_b g[b,:],   T_l = sum_b dbeta_l[b]
//   dw_l = A_l + T_l * B_l ,   db_j = G + sum_{l>j} T_l * w_l .
// Same math as the reference, different fp32 evaluation order (parity: 1e-5 rel vs fp64 oracle).
//
// HBM-bound: forward x0 in / out out (2*d*4 B per example), backward x0, g in / dx0 out
// (3*d*4 B); w, b stay in L1/L2.  One wave owns one example: its d floats are NV float4 per lane.
//
// The single-layer entry points (reference signature cross_layer(x0, xl, index) with an
// arbitrary xl) use the direct form.
#include "common.h"
#include "cross_bwd.h"

namespace {

constexpr int kFwdThreads = 256;
constexpr int kBwdThreads = 512;               // 8 waves: <= 256 VGPRs each, no spills
constexpr int kBwdWaves = kBwdThreads / 64;
constexpr int kMaxPartialRows = 256;

using recalgo_cross::wave_sum_n;

// ---------------------------------------------------------------------------------------------
// fused stack, forward
// ---------------------------------------------------------------------------------------------
// Optional: x0 is not read but GATHERED (fc.input_layer over F single-valued embedding columns of width K, dcn.py:152-153: the
// wave that computes an example's stack has the example's row in registers — it fetches the row's pieces from the embedding
// arena itself and writes x0 for the other consumers of the input layer (the MLP branch, the backward) on the way: the
// separate gather launch of a DCN step (5.4 us at its launch floor) disappears.
struct GatherSrc {
    const int64_t* ids;        // [B][F], id < 0: zero row;  nullptr: x0 is an input
    const float4* arena;       // [rows][K / 4]
    const int64_t* row_base;   // [F]
    unsigned F, K4;
};

template <int NV, int L>
__global__ __launch_bounds__(kFwdThreads) void cross_stack_fwd_kernel(
    float* __restrict__ x0, unsigned x_stride, const float4* __restrict__ w,
    const float4* __restrict__ b, unsigned B, unsigned d4, float* __restrict__ out,
    unsigned out_stride, GatherSrc gs) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * kFwdThreads + threadIdx.x) >> 6;
    const unsigned nwaves = (gridDim.x * kFwdThreads) >> 6;

    // lane-local slices of w_l and of the prefix sums B_l; beta_l partials are batch constants
    float4 wv[L][NV], Bv[NV];
    float red[2 * L];
#pragma unroll
    for (int v = 0; v < NV; ++v) Bv[v] = f4_zero();
#pragma unroll
    for (int l = 0; l < L; ++l) {
        float acc = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            unsigned idx = lane + v * 64;
            wv[l][v] = idx < d4 ? w[l * d4 + idx] : f4_zero();
            acc += f4_dot(Bv[v], wv[l][v]);
            if (idx < d4) Bv[v] = f4_add(Bv[v], b[l * d4 + idx]);
        }
        red[L + l] = acc;                      // partial of beta_l = B_l . w_l
    }

    for (unsigned ex = wave; ex < B; ex += nwaves) {
        float4* xr = reinterpret_cast<float4*>(x0 + (size_t)ex * x_stride);
        float4 xv[NV];
        if (gs.ids != nullptr) {
            long long id[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const unsigned idx = lane + v * 64;
                id[v] = idx < d4 ? gs.ids[(size_t)ex * gs.F + idx / gs.K4] : -1;
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const unsigned idx = lane + v * 64, f = idx / gs.K4, q = idx - f * gs.K4;
                xv[v] = id[v] >= 0 ? gs.arena[(size_t)(gs.row_base[f] + id[v]) * gs.K4 + q] : f4_zero();
                if (idx < d4) xr[idx] = xv[v];
            }
        } else {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                unsigned idx = lane + v * 64;
                xv[v] = idx < d4 ? xr[idx] : f4_zero();
            }
        }
        float r[2 * L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            float acc = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) acc += f4_dot(xv[v], wv[l][v]);
            r[l] = acc;
            r[L + l] = red[L + l];
        }
        wave_sum_n<2 * L>(r);
        float c = 1.f;
#pragma unroll
        for (int l = 0; l < L; ++l) c += fmaf(c, r[l], r[L + l]);   // c_{l+1} = c_l + (c_l p_l + beta_l)
        float4* orow = reinterpret_cast<float4*>(out + (size_t)ex * out_stride);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            unsigned idx = lane + v * 64;
            if (idx < d4) orow[idx] = f4_fma(xv[v], c, Bv[v]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fused stack, backward: csrc/cross_bwd.h
// ---------------------------------------------------------------------------------------------
template <int NV, int L>
__global__ __launch_bounds__(kBwdThreads) void cross_stack_bwd_kernel(
    const float* __restrict__ x0, unsigned x_stride, const float4* __restrict__ w,
    const float4* __restrict__ b, const float* __restrict__ g, unsigned g_stride,
    const float* __restrict__ g_x0_extra, unsigned B, unsigned d4, float* __restrict__ dx0,
    float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [kBwdWaves][d] (+ [kBwdWaves + 1][L])
    recalgo_cross::cross_stack_bwd_block<NV, L, kBwdWaves>(x0, x_stride, w, b, g, g_stride, g_x0_extra, B, d4, dx0, partials, blockIdx.x,
                                                          gridDim.x, smem);
}

// ---------------------------------------------------------------------------------------------
// single layer, direct form (xl distinct from x0)
// ---------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(kFwdThreads) void cross_layer_fwd_kernel(
    const float* __restrict__ x0, const float* __restrict__ xl_in, unsigned x_stride,
    const float4* __restrict__ w, const float4* __restrict__ b, unsigned B, unsigned d4,
    float* __restrict__ out, unsigned out_stride) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * kFwdThreads + threadIdx.x) >> 6;
    const unsigned nwaves = (gridDim.x * kFwdThreads) >> 6;
    for (unsigned ex = wave; ex < B; ex += nwaves) {
        const float4* xr = reinterpret_cast<const float4*>(x0 + (size_t)ex * x_stride);
        const float4* lr = reinterpret_cast<const float4*>(xl_in + (size_t)ex * x_stride);
        float4 x0v[NV], xl[NV];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            unsigned idx = lane + v * 64;
            x0v[v] = idx < d4 ? xr[idx] : f4_zero();
            xl[v] = idx < d4 ? lr[idx] : f4_zero();
            if (idx < d4) s += f4_dot(xl[v], w[idx]);
        }
        s = wave_sum(s);
        float4* orow = reinterpret_cast<float4*>(out + (size_t)ex * out_stride);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            unsigned idx = lane + v * 64;
            // reference order: (x0 * s + b) + xl      cross_layer.py:22-24
            if (idx < d4) orow[idx] = f4_add(f4_fma(x0v[v], s, b[idx]), xl[v]);
        }
    }
}

// partial row layout: [dw | db]  (2*d floats)
template <int NV>
__global__ __launch_bounds__(kBwdThreads) void cross_layer_bwd_kernel(
    const float* __restrict__ x0, const float* __restrict__ xl_in, unsigned x_stride,
    const float4* __restrict__ w, const float* __restrict__ g, unsigned g_stride, unsigned B,
    unsigned d4, float* __restrict__ dx0, float* __restrict__ dxl, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned lane = threadIdx.x & 63;
    const unsigned wib = threadIdx.x >> 6;
    const unsigned wave = blockIdx.x * kBwdWaves + wib;
    const unsigned nwaves = gridDim.x * kBwdWaves;
    const unsigned d = d4 * 4;
    float4 dwacc[NV], dbacc[NV], wv[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        unsigned idx = lane + v * 64;
        dwacc[v] = dbacc[v] = f4_zero();
        wv[v] = idx < d4 ? w[idx] : f4_zero();
    }
    for (unsigned ex = wave; ex < B; ex += nwaves) {
        const float4* xr = reinterpret_cast<const float4*>(x0 + (size_t)ex * x_stride);
        const float4* lr = reinterpret_cast<const float4*>(xl_in + (size_t)ex * x_stride);
        const float4* gr = reinterpret_cast<const float4*>(g + (size_t)ex * g_stride);
        float4 x0v[NV], xl[NV], gv[NV];
        float r[2] = {0.f, 0.f};
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            unsigned idx = lane + v * 64;
            x0v[v] = idx < d4 ? xr[idx] : f4_zero();
            xl[v] = idx < d4 ? lr[idx] : f4_zero();
            gv[v] = idx < d4 ? gr[idx] : f4_zero();
            r[0] += f4_dot(xl[v], wv[v]);     // s = xl . w
            r[1] += f4_dot(gv[v], x0v[v]);    // t = g . x0
        }
        wave_sum_n<2>(r);
        float4* o0 = reinterpret_cast<float4*>(dx0 + (size_t)ex * x_stride);
        float4* ol = reinterpret_cast<float4*>(dxl + (size_t)ex * x_stride);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            unsigned idx = lane + v * 64;
            dwacc[v] = f4_fma(xl[v], r[1], dwacc[v]);
            dbacc[v] = f4_add(dbacc[v], gv[v]);
            if (idx < d4) {
                o0[idx] = f4_scale(gv[v], r[0]);
                ol[idx] = f4_fma(wv[v], r[1], gv[v]);
            }
        }
    }
    float* prow = partials + (size_t)blockIdx.x * 2 * d;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            unsigned idx = lane + v * 64;
            if (idx < d4)
                *reinterpret_cast<float4*>(smem + (size_t)wib * d + idx * 4) = pass == 0 ? dwacc[v] : dbacc[v];
        }
        __syncthreads();
        for (unsigned j = threadIdx.x; j < d; j += kBwdThreads) {
            float acc = 0.f;
#pragma unroll
            for (int wv_ = 0; wv_ < kBwdWaves; ++wv_) acc += smem[(size_t)wv_ * d + j];
            prow[(size_t)pass * d + j] = acc;
        }
    }
}

// plain column sums of [nrows][ncols] partials (ncols % 4 == 0); workgroup = 16 float4 column
// lanes x 16 row slices
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ partials, unsigned nrows,
                                                      unsigned ncols, float* __restrict__ out0,
                                                      float* __restrict__ out1, unsigned split) {
    __shared__ float4 sh[16][16];
    const unsigned cl = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const unsigned col4 = blockIdx.x * 16 + cl;
    float4 acc = f4_zero();
    if (col4 * 4 < ncols) {
#pragma unroll 8
        for (unsigned r = slice; r < nrows; r += 16)
            acc = f4_add(acc, *reinterpret_cast<const float4*>(partials + (size_t)r * ncols + col4 * 4));
    }
    sh[slice][cl] = acc;
    __syncthreads();
    if (slice == 0 && col4 * 4 < ncols) {
        float4 s = f4_zero();
#pragma unroll
        for (int k = 0; k < 16; ++k) s = f4_add(s, sh[k][cl]);
        unsigned col = col4 * 4;
        if (col < split) *reinterpret_cast<float4*>(out0 + col) = s;
        else *reinterpret_cast<float4*>(out1 + (col - split)) = s;
    }
}

inline int bwd_grid(int B) {
    int need = cdiv(B, kBwdWaves);
    return need < 1 ? 1 : (need > kMaxPartialRows ? kMaxPartialRows : need);
}

template <int NV, int L>
int launch_stack(bool fwd, const float* x0, int x_stride, const float* w, const float* b, const float* g,
                 int g_stride, const float* gx, int B, int d, float* out, int out_stride, float* dw,
                 float* db, float* partials, hipStream_t st, int defer, const GatherSrc* gs) {
    const float4* w4 = reinterpret_cast<const float4*>(w);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    if (fwd) {
        hipLaunchKernelGGL((cross_stack_fwd_kernel<NV, L>), dim3(cdiv(B, kFwdThreads / 64)), dim3(kFwdThreads),
                           0, st, const_cast<float*>(x0), (unsigned)x_stride, w4, b4, (unsigned)B, (unsigned)(d / 4), out,
                           (unsigned)out_stride, gs ? *gs : GatherSrc{nullptr, nullptr, nullptr, 1u, 1u});
        return (int)hipGetLastError();
    }
    const int grid = bwd_grid(B);
    const size_t smem = ((size_t)kBwdWaves * d + (size_t)(kBwdWaves + 1) * L) * sizeof(float);
    hipLaunchKernelGGL((cross_stack_bwd_kernel<NV, L>), dim3(grid), dim3(kBwdThreads), smem, st, x0,
                       (unsigned)x_stride, w4, b4, g, (unsigned)g_stride, gx, (unsigned)B, (unsigned)(d / 4),
                       out, partials);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || defer) return (int)e;
    const unsigned ncols = 2u * L * d;
    hipLaunchKernelGGL(colsum4_kernel, dim3(cdiv((int64_t)ncols, 64)), dim3(256), 0, st, partials, (unsigned)grid, ncols, dw,
                       db, (unsigned)(L * d));
    return (int)hipGetLastError();
}

template <int NV>
int dispatch_stack_L(int L, bool fwd, const float* x0, int x_stride, const float* w, const float* b,
                     const float* g, int g_stride, const float* gx, int B, int d, float* out,
                     int out_stride, float* dw, float* db, float* partials, hipStream_t st, int defer, const GatherSrc* gs) {
    switch (L) {
#define CASE_L(LL)                                                                                      \
    case LL:                                                                                            \
        return launch_stack<NV, LL>(fwd, x0, x_stride, w, b, g, g_stride, gx, B, d, out, out_stride, dw, \
                                    db, partials, st, defer, gs);
        CASE_L(1) CASE_L(2) CASE_L(3) CASE_L(4) CASE_L(5) CASE_L(6)
#undef CASE_L
        default: return (int)hipErrorInvalidValue;
    }
}

int dispatch_stack(int L, bool fwd, const float* x0, int x_stride, const float* w, const float* b,
                   const float* g, int g_stride, const float* gx, int B, int d, float* out, int out_stride,
                   float* dw, float* db, float* partials, hipStream_t st, int defer = 0, const GatherSrc* gs = nullptr) {
    const int nv = cdiv(d / 4, 64);
    if (nv <= 1) return dispatch_stack_L<1>(L, fwd, x0, x_stride, w, b, g, g_stride, gx, B, d, out, out_stride, dw, db, partials, st, defer, gs);
    if (nv <= 2) return dispatch_stack_L<2>(L, fwd, x0, x_stride, w, b, g, g_stride, gx, B, d, out, out_stride, dw, db, partials, st, defer, gs);
    if (nv <= 4) return dispatch_stack_L<4>(L, fwd, x0, x_stride, w, b, g, g_stride, gx, B, d, out, out_stride, dw, db, partials, st, defer, gs);
    return (int)hipErrorInvalidValue;
}

}  // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
RECALGO_EXPORT int recalgo_cross_fwd(const float* x0, int x_stride, const float* w, const float* b,
                                     int B, int d, int L, float* out, int out_stride,
                                     recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && d > 0 && d % 4 == 0 && d <= 1024 && L >= 1 && L <= 6);
    RECALGO_REQUIRE(x_stride % 4 == 0 && out_stride % 4 == 0 && x_stride >= d && out_stride >= d);
    if (B == 0) return 0;
    return dispatch_stack(L, true, x0, x_stride, w, b, nullptr, 0, nullptr, B, d, out, out_stride, nullptr,
                          nullptr, nullptr, as_stream(stream));
}

RECALGO_EXPORT int recalgo_gather_cross_fwd(const int64_t* ids, const float* arena, const int64_t* row_base, int B, int F, int K,
                                            const float* w, const float* b, int L, float* x0, int x_stride, float* out,
                                            int out_stride, recalgo_stream_t stream) {
    const int d = F * K;
    RECALGO_REQUIRE(B >= 0 && F >= 1 && K >= 4 && K % 4 == 0 && d <= 1024 && L >= 1 && L <= 6);
    RECALGO_REQUIRE(ids != nullptr && arena != nullptr && row_base != nullptr && x0 != nullptr && out != nullptr);
    RECALGO_REQUIRE(x_stride % 4 == 0 && out_stride % 4 == 0 && x_stride >= d && out_stride >= d);
    RECALGO_REQUIRE((reinterpret_cast<uintptr_t>(arena) & 15) == 0 && (reinterpret_cast<uintptr_t>(x0) & 15) == 0);
    if (B == 0) return 0;
    const GatherSrc gs{ids, reinterpret_cast<const float4*>(arena), row_base, (unsigned)F, (unsigned)(K / 4)};
    return dispatch_stack(L, true, x0, x_stride, w, b, nullptr, 0, nullptr, B, d, out, out_stride, nullptr, nullptr, nullptr,
                          as_stream(stream), 0, &gs);
}

RECALGO_EXPORT int64_t recalgo_cross_bwd_workspace_bytes(int B, int d, int L) {
    if (B <= 0 || d <= 0 || L <= 0) return 0;
    return (int64_t)bwd_grid(B) * 2 * L * d * (int64_t)sizeof(float);
}

RECALGO_EXPORT int recalgo_cross_bwd_partial_rows(int B) { return B > 0 ? bwd_grid(B) : 0; }

RECALGO_EXPORT int recalgo_cross_bwd(const float* x0, int x_stride, const float* w, const float* b,
                                     const float* g, int g_stride, const float* g_x0_extra, int B,
                                     int d, int L, float* dx0, float* dw, float* db, void* workspace,
                                     int defer_reduce, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B > 0 && d > 0 && d % 4 == 0 && d <= 1024 && L >= 1 && L <= 6);
    RECALGO_REQUIRE(x_stride % 4 == 0 && g_stride % 4 == 0 && x_stride >= d && g_stride >= d);
    RECALGO_REQUIRE(workspace != nullptr && (defer_reduce || (dw != nullptr && db != nullptr)));
    return dispatch_stack(L, false, x0, x_stride, w, b, g, g_stride, g_x0_extra, B, d, dx0, x_stride, dw, db,
                          static_cast<float*>(workspace), as_stream(stream), defer_reduce);
}

RECALGO_EXPORT int recalgo_cross_layer_fwd(const float* x0, const float* xl, int x_stride,
                                           const float* w, const float* b, int B, int d, float* out,
                                           int out_stride, recalgo_stream_t stream) {
    RECALGO_REQUIRE(xl != nullptr && B >= 0 && d > 0 && d % 4 == 0 && d <= 2048);
    RECALGO_REQUIRE(x_stride % 4 == 0 && out_stride % 4 == 0 && x_stride >= d && out_stride >= d);
    if (B == 0) return 0;
    const int nv = cdiv(d / 4, 64);
    hipStream_t st = as_stream(stream);
#define LAUNCH(NV)                                                                                        \
    hipLaunchKernelGGL(cross_layer_fwd_kernel<NV>, dim3(cdiv(B, kFwdThreads / 64)), dim3(kFwdThreads), 0, st, \
                       x0, xl, (unsigned)x_stride, reinterpret_cast<const float4*>(w),                    \
                       reinterpret_cast<const float4*>(b), (unsigned)B, (unsigned)(d / 4), out,           \
                       (unsigned)out_stride)
    if (nv <= 1) LAUNCH(1);
    else if (nv <= 2) LAUNCH(2);
    else if (nv <= 4) LAUNCH(4);
    else LAUNCH(8);
#undef LAUNCH
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_cross_layer_bwd(const float* x0, const float* xl, int x_stride,
                                           const float* w, const float* b, const float* g,
                                           int g_stride, int B, int d, float* dx0, float* dxl,
                                           float* dw, float* db, void* workspace,
                                           recalgo_stream_t stream) {
    (void)b;
    RECALGO_REQUIRE(xl != nullptr && dxl != nullptr && workspace != nullptr);
    RECALGO_REQUIRE(B > 0 && d > 0 && d % 4 == 0 && d <= 2048);
    RECALGO_REQUIRE(x_stride % 4 == 0 && g_stride % 4 == 0 && x_stride >= d && g_stride >= d);
    const int nv = cdiv(d / 4, 64);
    const int grid = bwd_grid(B);
    hipStream_t st = as_stream(stream);
    float* partials = static_cast<float*>(workspace);
    size_t smem = (size_t)kBwdWaves * d * sizeof(float);
#define LAUNCH(NV)                                                                                      \
    hipLaunchKernelGGL(cross_layer_bwd_kernel<NV>, dim3(grid), dim3(kBwdThreads), smem, st, x0, xl,     \
                       (unsigned)x_stride, reinterpret_cast<const float4*>(w), g, (unsigned)g_stride,   \
                       (unsigned)B, (unsigned)(d / 4), dx0, dxl, partials)
    if (nv <= 1) LAUNCH(1);
    else if (nv <= 2) LAUNCH(2);
    else if (nv <= 4) LAUNCH(4);
    else LAUNCH(8);
#undef LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(colsum4_kernel, dim3(cdiv(2 * (int64_t)d, 64)), dim3(256), 0, st, partials, (unsigned)grid,
                       (unsigned)(2 * d), dw, db, (unsigned)d);
    RECALGO_RETURN_LAST();
}
