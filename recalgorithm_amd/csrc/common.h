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
