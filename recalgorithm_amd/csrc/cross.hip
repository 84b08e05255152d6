// K4: DCN CrossNet, all L layers fused (algorithm/DCN/cross_layer.py:4-26, dcn.py:157-160).
//
// HBM-bound: per example the forward moves x0 in and x_L out (2*d*4 B), the backward x0 and g
// in and dx0 out (3*d*4 B); w, b (2*L*d floats) stay in L1/L2.  One wave owns one example:
// its d floats sit in registers as NV float4 per lane (lane j holds float4 j, j+64, ...), the
// per-layer scalar x_l.w_l is a 64-lane shuffle reduction.
#include "common.h"

namespace {

template <int NV>
__global__ __launch_bounds__(256) void cross_fwd_kernel(
    const float* __restrict__ x0, const float* __restrict__ xl_in, unsigned x_stride,
    const float4* __restrict__ w,
    const float4* __restrict__ b, unsigned B, unsigned d4, unsigned L, float* __restrict__ out,
    unsigned out_stride) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
    for (unsigned ex = wave; ex < B; ex += nwaves) {
        const float4* xr = reinterpret_cast<const float4*>(x0 + (size_t)ex * x_stride);
        float4 x0v[NV], xl[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            unsigned idx = lane + v * 64;
            x0v[v] = idx < d4 ? xr[idx] : f4_zero();
            xl[v] = x0v[v];
            if (xl_in && idx < d4)
                xl[v] = reinterpret_cast<const float4*>(xl_in + (size_t)ex * x_stride)[idx];
        }
        for (unsigned l = 0; l < L; ++l) {
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                unsigned idx = lane + v * 64;
                if (idx < d4) s += f4_dot(xl[v], w[l * d4 + idx]);
            }
            s = wave_sum(s);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                unsigned idx = lane + v * 64;
                if (idx < d4) {
                    float4 bb = b[l * d4 + idx];
                    // reference order: (x0 * s + b) + xl      cross_layer.py:22-24
                    xl[v] = f4_add(f4_fma(x0v[v], s, bb), xl[v]);
                }
            }
        }
        float4* orow = reinterpret_cast<float4*>(out + (size_t)ex * out_stride);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            unsigned idx = lane + v * 64;
            if (idx < d4) orow[idx] = xl[v];
        }
    }
}

// Backward.  Template on L so that the recomputed x_l and the dw/db accumulators are register
// arrays with static indices.  A 1024-thread workgroup (16 waves) keeps the number of
// per-workgroup dw/db partials small; they are reduced across workgroups by a second,
// deterministic kernel.
constexpr int kBwdThreads = 1024;
constexpr int kBwdWaves = kBwdThreads / 64;

template <int NV, int L>
__global__ __launch_bounds__(kBwdThreads) void cross_bwd_kernel(
    const float* __restrict__ x0, const float* __restrict__ xl_in, unsigned x_stride,
    const float4* __restrict__ w,
    const float4* __restrict__ b, const float* __restrict__ g, unsigned g_stride,
    const float* __restrict__ g_x0_extra, unsigned B, unsigned d4, float* __restrict__ dx0,
    float* __restrict__ dxl /* only with xl_in */,
    float* __restrict__ partials /* [gridDim.x][2][L][d] */) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [kBwdWaves][d]
    const unsigned lane = threadIdx.x & 63;
    const unsigned wib = threadIdx.x >> 6;
    const unsigned wave = blockIdx.x * kBwdWaves + wib;
    const unsigned nwaves = gridDim.x * kBwdWaves;
    const unsigned d = d4 * 4;

    float4 dwacc[L][NV], dbacc[L][NV];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int v = 0; v < NV; ++v) dwacc[l][v] = dbacc[l][v] = f4_zero();

    for (unsigned ex = wave; ex < B; ex += nwaves) {
        const float4* xr = reinterpret_cast<const float4*>(x0 + (size_t)ex * x_stride);
        const float4* gr = reinterpret_cast<const float4*>(g + (size_t)ex * g_stride);
        float4 x0v[NV], gv[NV], xs[L][NV];
        float s[L];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            unsigned idx = lane + v * 64;
            x0v[v] = idx < d4 ? xr[idx] : f4_zero();
            gv[v] = idx < d4 ? gr[idx] : f4_zero();
            xs[0][v] = x0v[v];
            if (xl_in)
                xs[0][v] = idx < d4 ? reinterpret_cast<const float4*>(xl_in + (size_t)ex * x_stride)[idx]
                                    : f4_zero();
        }
        // recompute the forward: xs[l] = x_l, s[l] = x_l . w_l
#pragma unroll
        for (int l = 0; l < L; ++l) {
            float t = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                unsigned idx = lane + v * 64;
                if (idx < d4) t += f4_dot(xs[l][v], w[l * d4 + idx]);
            }
            s[l] = wave_sum(t);
            if (l + 1 < L) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    unsigned idx = lane + v * 64;
                    xs[l + 1][v] = idx < d4
                                       ? f4_add(f4_fma(x0v[v], s[l], b[l * d4 + idx]), xs[l][v])
                                       : f4_zero();
                }
            }
        }
        // reverse sweep (Cross)
        float4 dx0acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) dx0acc[v] = f4_zero();
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            float t = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) t += f4_dot(gv[v], x0v[v]);
            t = wave_sum(t);  // g . x0
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                unsigned idx = lane + v * 64;
                dwacc[l][v] = f4_fma(xs[l][v], t, dwacc[l][v]);
                dbacc[l][v] = f4_add(dbacc[l][v], gv[v]);
                dx0acc[v] = f4_fma(gv[v], s[l], dx0acc[v]);
                if (idx < d4) gv[v] = f4_fma(w[l * d4 + idx], t, gv[v]);
            }
        }
        float4* orow = reinterpret_cast<float4*>(dx0 + (size_t)ex * x_stride);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            unsigned idx = lane + v * 64;
            if (idx < d4) {
                float4 r = dx0acc[v];
                if (xl_in)
                    reinterpret_cast<float4*>(dxl + (size_t)ex * x_stride)[idx] = gv[v];
                else
                    r = f4_add(r, gv[v]);
                if (g_x0_extra)
                    r = f4_add(r, reinterpret_cast<const float4*>(g_x0_extra + (size_t)ex * x_stride)[idx]);
                orow[idx] = r;
            }
        }
    }

    // workgroup reduction of dw, then db, through LDS in fixed wave order (deterministic)
    // one [kBwdWaves][d] LDS tile per (pass, layer): 64 KiB at d = 1024
    float* pblk = partials + (size_t)blockIdx.x * 2 * L * d;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
            __syncthreads();
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                unsigned idx = lane + v * 64;
                if (idx < d4)
                    *reinterpret_cast<float4*>(smem + (size_t)wib * d + idx * 4) =
                        pass == 0 ? dwacc[l][v] : dbacc[l][v];
            }
            __syncthreads();
            for (unsigned j = threadIdx.x; j < d; j += kBwdThreads) {
                float acc = 0.f;
#pragma unroll
                for (int wv = 0; wv < kBwdWaves; ++wv) acc += smem[(size_t)wv * d + j];
                pblk[((size_t)pass * L + l) * d + j] = acc;
            }
        }
    }
}

__global__ __launch_bounds__(256) void cross_reduce_partials_kernel(
    const float* __restrict__ partials, unsigned nblk, unsigned Ld, float* __restrict__ dw,
    float* __restrict__ db) {
    unsigned j = blockIdx.x * 256 + threadIdx.x;
    if (j >= 2 * Ld) return;
    float acc = 0.f;
#pragma unroll 8
    for (unsigned k = 0; k < nblk; ++k) acc += partials[(size_t)k * 2 * Ld + j];
    if (j < Ld) dw[j] = acc; else db[j - Ld] = acc;
}

inline int cross_bwd_grid(int B) {
    int need = cdiv(B, kBwdWaves);
    return need < 256 ? (need < 1 ? 1 : need) : 256;
}

template <int NV, int L>
int launch_cross_bwd(const float* x0, const float* xl_in, int x_stride, const float* w, const float* b,
                     const float* g, int g_stride, const float* gx, int B, int d, float* dx0,
                     float* dxl, float* partials, hipStream_t st) {
    size_t smem = (size_t)kBwdWaves * d * sizeof(float);
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_bwd_kernel<NV, L>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((cross_bwd_kernel<NV, L>), dim3(cross_bwd_grid(B)), dim3(kBwdThreads), smem, st,
                       x0, xl_in, (unsigned)x_stride, reinterpret_cast<const float4*>(w),
                       reinterpret_cast<const float4*>(b), g, (unsigned)g_stride, gx, (unsigned)B,
                       (unsigned)(d / 4), dx0, dxl, partials);
    return (int)hipGetLastError();
}

template <int NV>
int dispatch_cross_bwd_L(int L, const float* x0, const float* xl_in, int x_stride, const float* w,
                         const float* b, const float* g, int g_stride, const float* gx, int B, int d,
                         float* dx0, float* dxl, float* partials, hipStream_t st) {
    switch (L) {
#define CASE_L(LL) \
    case LL: return launch_cross_bwd<NV, LL>(x0, xl_in, x_stride, w, b, g, g_stride, gx, B, d, dx0, dxl, partials, st);
        CASE_L(1) CASE_L(2) CASE_L(3) CASE_L(4) CASE_L(5) CASE_L(6)
#undef CASE_L
        default: return (int)hipErrorInvalidValue;
    }
}

}  // namespace

namespace {
int cross_fwd_impl(const float* x0, const float* xl_in, int x_stride, const float* w, const float* b,
                   int B, int d, int L, float* out, int out_stride, recalgo_stream_t stream);
}
RECALGO_EXPORT int recalgo_cross_fwd(const float* x0, int x_stride, const float* w, const float* b,
                                     int B, int d, int L, float* out, int out_stride,
                                     recalgo_stream_t stream) {
    return cross_fwd_impl(x0, nullptr, x_stride, w, b, B, d, L, out, out_stride, stream);
}
RECALGO_EXPORT int recalgo_cross_layer_fwd(const float* x0, const float* xl, int x_stride,
                                           const float* w, const float* b, int B, int d, float* out,
                                           int out_stride, recalgo_stream_t stream) {
    RECALGO_REQUIRE(xl != nullptr);
    return cross_fwd_impl(x0, xl, x_stride, w, b, B, d, 1, out, out_stride, stream);
}
namespace {
int cross_fwd_impl(const float* x0, const float* xl_in, int x_stride, const float* w, const float* b,
                   int B, int d, int L, float* out, int out_stride, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && d > 0 && d % 4 == 0 && d <= 2048 && L >= 1 && L <= 8);
    RECALGO_REQUIRE(x_stride % 4 == 0 && out_stride % 4 == 0 && x_stride >= d && out_stride >= d);
    if (B == 0) return 0;
    const int d4 = d / 4;
    const int nv = cdiv(d4, 64);
    const int blocks = cdiv(B, 4);  // 4 waves per 256-thread workgroup, one example per wave
    hipStream_t st = as_stream(stream);
#define LAUNCH_FWD(NV)                                                                             \
    hipLaunchKernelGGL(cross_fwd_kernel<NV>, dim3(blocks), dim3(256), 0, st, x0, xl_in,            \
                       (unsigned)x_stride, reinterpret_cast<const float4*>(w),                     \
                       reinterpret_cast<const float4*>(b),                                         \
                       (unsigned)B, (unsigned)d4, (unsigned)L, out, (unsigned)out_stride)
    if (nv <= 1) LAUNCH_FWD(1);
    else if (nv <= 2) LAUNCH_FWD(2);
    else if (nv <= 4) LAUNCH_FWD(4);
    else LAUNCH_FWD(8);
#undef LAUNCH_FWD
    RECALGO_RETURN_LAST();
}
}  // namespace

RECALGO_EXPORT int64_t recalgo_cross_bwd_workspace_bytes(int B, int d, int L) {
    if (B <= 0 || d <= 0 || L <= 0) return 0;
    return (int64_t)cross_bwd_grid(B) * 2 * L * d * (int64_t)sizeof(float);
}

namespace {
int cross_bwd_impl(const float* x0, const float* xl_in, int x_stride, const float* w, const float* b,
                   const float* g, int g_stride, const float* g_x0_extra, int B, int d, int L,
                   float* dx0, float* dxl, float* dw, float* db, void* workspace,
                   recalgo_stream_t stream);
}
RECALGO_EXPORT int recalgo_cross_bwd(const float* x0, int x_stride, const float* w, const float* b,
                                     const float* g, int g_stride, const float* g_x0_extra, int B,
                                     int d, int L, float* dx0, float* dw, float* db, void* workspace,
                                     recalgo_stream_t stream) {
    return cross_bwd_impl(x0, nullptr, x_stride, w, b, g, g_stride, g_x0_extra, B, d, L, dx0, nullptr,
                          dw, db, workspace, stream);
}
RECALGO_EXPORT int recalgo_cross_layer_bwd(const float* x0, const float* xl, int x_stride,
                                           const float* w, const float* b, const float* g,
                                           int g_stride, int B, int d, float* dx0, float* dxl,
                                           float* dw, float* db, void* workspace,
                                           recalgo_stream_t stream) {
    RECALGO_REQUIRE(xl != nullptr && dxl != nullptr);
    return cross_bwd_impl(x0, xl, x_stride, w, b, g, g_stride, nullptr, B, d, 1, dx0, dxl, dw, db,
                          workspace, stream);
}
namespace {
int cross_bwd_impl(const float* x0, const float* xl_in, int x_stride, const float* w, const float* b,
                   const float* g, int g_stride, const float* g_x0_extra, int B, int d, int L,
                   float* dx0, float* dxl, float* dw, float* db, void* workspace,
                   recalgo_stream_t stream) {
    RECALGO_REQUIRE(B > 0 && d > 0 && d % 4 == 0 && d <= 1024 && L >= 1 && L <= 6);
    RECALGO_REQUIRE(x_stride % 4 == 0 && g_stride % 4 == 0 && x_stride >= d && g_stride >= d);
    RECALGO_REQUIRE(workspace != nullptr);
    RECALGO_REQUIRE((size_t)kBwdWaves * d * sizeof(float) <= 150 * 1024);
    hipStream_t st = as_stream(stream);
    float* partials = static_cast<float*>(workspace);
    const int nv = cdiv(d / 4, 64);
    int rc;
    if (nv <= 1) rc = dispatch_cross_bwd_L<1>(L, x0, xl_in, x_stride, w, b, g, g_stride, g_x0_extra, B, d, dx0, dxl, partials, st);
    else if (nv <= 2) rc = dispatch_cross_bwd_L<2>(L, x0, xl_in, x_stride, w, b, g, g_stride, g_x0_extra, B, d, dx0, dxl, partials, st);
    else rc = dispatch_cross_bwd_L<4>(L, x0, xl_in, x_stride, w, b, g, g_stride, g_x0_extra, B, d, dx0, dxl, partials, st);
    if (rc != 0) return rc;
    const unsigned Ld = (unsigned)(L * d);
    hipLaunchKernelGGL(cross_reduce_partials_kernel, dim3(cdiv(2 * (int64_t)Ld, 256)), dim3(256), 0, st,
                       partials, (unsigned)cross_bwd_grid(B), Ld, dw, db);
    RECALGO_RETURN_LAST();
}
}  // namespace
