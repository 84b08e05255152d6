// Shared device/host helpers for the recalgo HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/recalgo.h"

#define RECALGO_EXPORT extern "C" __attribute__((visibility("default")))

// Every entry point returns hipError_t as int (0 == hipSuccess) and never throws.
#define RECALGO_RETURN_LAST()                  \
    do {                                       \
        hipError_t e__ = hipGetLastError();    \
        return (int)e__;                       \
    } while (0)

#define RECALGO_REQUIRE(cond)                          \
    do {                                               \
        if (!(cond)) return (int)hipErrorInvalidValue; \
    } while (0)

static inline hipStream_t as_stream(recalgo_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- wave64 helpers -------------------------------------------------------
// Sum over all 64 lanes, result in every lane.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Sum over aligned groups of W lanes (W = 4, 8, 16, 32), result in every lane of the group.
template <int W>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 f4_scale(float4 a, float s) {
    return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
}
__device__ __forceinline__ float4 f4_fma(float4 a, float s, float4 c) {
    return make_float4(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z), fmaf(a.w, s, c.w));
}
__device__ __forceinline__ float f4_dot(float4 a, float4 b) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// Hardware fp32 atomic add (global_atomic_add_f32); build uses -munsafe-fp-atomics.
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// ---- deterministic column sum of partial rows ---------------------------------------------------
// out[c] = sum_{r < S} partials[r * n + c].  256 threads = 16 columns x 16 row groups: a thread adds
// rows rg, rg+16, ... in order, then the 16 groups are added in order (fixed summation order, and
// S/16 instead of S dependent loads per thread — the one-thread-per-column form of this loop cost
// 60-240 us for S = 256..1024 partial rows).  Columns [0, n0) go to out0, the rest to out1.
namespace {
__global__ __launch_bounds__(256) void colsum16_kernel(const float* __restrict__ partials, unsigned S, unsigned n,
                                                       float* __restrict__ out0, unsigned n0,
                                                       float* __restrict__ out1) {
    __shared__ float sh[16][17];
    const unsigned cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const unsigned c = blockIdx.x * 16 + cl;
    float acc = 0.f;
    if (c < n) {
        // independent loads, eight in flight per thread, added in row order (see dense_sum_slabs_kernel)
        const float* col = partials + c;
        unsigned r = rg;
        for (; r + 16 * 7 < S; r += 16 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = col[(size_t)(r + 16 * u) * n];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; r < S; r += 16) acc += col[(size_t)r * n];
    }
    sh[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && c < n) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += sh[g][cl];
        if (c < n0) out0[c] = t;
        else out1[c - n0] = t;
    }
}
inline void launch_colsum16(const float* partials, unsigned S, unsigned n, float* out0, unsigned n0, float* out1,
                            hipStream_t st) {
    hipLaunchKernelGGL(colsum16_kernel, dim3((n + 15) / 16), dim3(256), 0, st, partials, S, n, out0, n0, out1);
}
}  // namespace
