// tf.layers.dropout's keep decision as a counter-based hash: keep(element) is a pure function of (seed, call, step, flat
// element index), so the forward kernel, the backward kernel and every fused epilogue that applies the same dropout
// (dense.hip, mlp.hip) agree without a stored mask, and a captured training step draws a new mask per replay (the step
// counter lives on the device and is advanced inside the graph).  Two rounds of a 32-bit finalizer (murmur3's and a
// second set of odd constants) over the index, keyed before and between the rounds.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace recalgo_drop {

struct Spec {
    const float* mask;     // explicit keep mask (1 = keep, 0 = drop), same flat layout as the tensor; NULL: the hash below
    const int64_t* step;   // device step counter (NULL: 0)
    uint32_t seed, call;   // stream of this dropout layer: store seed (+ rank), index of the call inside the model_fn
    uint32_t threshold;    // keep iff hash >= threshold, threshold = rate * 2^32
    float scale;           // 1 / (1 - rate)
};

__host__ __device__ __forceinline__ uint32_t threshold_of(double rate) {
    const double t = rate * 4294967296.0;
    return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}

__host__ __device__ __forceinline__ bool enabled(const Spec& s) { return s.scale != 0.f; }     // (scale = 1 / (1 - rate) >= 1 when on)
inline Spec disabled() {
    Spec s;
    s.mask = nullptr; s.step = nullptr; s.seed = s.call = 0; s.threshold = 0; s.scale = 0.f;
    return s;
}
// the C-ABI form (include/recalgo.h recalgo_dropout_t) -> Spec; NULL: disabled
template <class T>
inline Spec from_abi(const T* d) {
    if (d == nullptr) return disabled();
    Spec s;
    s.mask = d->keep_mask; s.step = d->step; s.seed = d->seed; s.call = d->call;
    s.threshold = threshold_of(d->rate);
    s.scale = (float)(1.0 / (1.0 - d->rate));
    return s;
}
template <class T>
inline bool abi_ok(const T* d) {
    return d == nullptr || (d->rate > 0.0 && d->rate < 1.0 && (reinterpret_cast<uintptr_t>(d->keep_mask) & 15) == 0);
}

struct Key {
    uint32_t k0, k1;
};
__device__ __forceinline__ Key make_key(const Spec& s) {
    const uint64_t st = s.step ? (uint64_t)*s.step : 0ull;
    Key k;
    k.k0 = s.seed * 0x9E3779B1u + (uint32_t)st * 0x85EBCA77u + 0x165667B1u;
    k.k1 = s.call * 0xC2B2AE3Du + (uint32_t)(st >> 32) * 0x27D4EB2Fu + 0x9E3779B9u;
    return k;
}
__device__ __forceinline__ uint32_t hash(Key k, uint32_t idx) {
    uint32_t h = idx ^ k.k0;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    h += k.k1;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
// multiplier of element idx: scale if kept, 0 if dropped
__device__ __forceinline__ float factor(const Spec& s, Key k, uint32_t idx) {
    if (s.mask) return s.mask[idx] > 0.f ? s.scale : 0.f;
    return hash(k, idx) >= s.threshold ? s.scale : 0.f;
}
__device__ __forceinline__ float4 factor4(const Spec& s, Key k, uint32_t idx) {      // idx % 4 == 0, mask 16-byte aligned
    if (s.mask) {
        const float4 m = *reinterpret_cast<const float4*>(s.mask + idx);
        return make_float4(m.x > 0.f ? s.scale : 0.f, m.y > 0.f ? s.scale : 0.f, m.z > 0.f ? s.scale : 0.f, m.w > 0.f ? s.scale : 0.f);
    }
    return make_float4(hash(k, idx) >= s.threshold ? s.scale : 0.f, hash(k, idx + 1) >= s.threshold ? s.scale : 0.f,
                       hash(k, idx + 2) >= s.threshold ? s.scale : 0.f, hash(k, idx + 3) >= s.threshold ? s.scale : 0.f);
}

}  // namespace recalgo_drop
