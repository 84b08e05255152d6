// tf.layers.dropout(x, rate, training=True) (algorithm/DeepFM/deepfm.py:208-209, DIN/din.py:235-236, FiBiNET/fibinet.py:193-194,
// PNN/pnn.py:188-189, NFM/nfm.py:170): y = x * keep / (1 - rate), keep ~ Bernoulli(1 - rate) per element.  The keep decision
// is the counter-based hash of dropout.h (or an explicit mask: parity tests replay the masks of the reference run), so the
// backward needs no stored mask: dx = g * keep / (1 - rate) from the same key.  HBM-bound elementwise passes, float4.
#include "common.h"
#include "dropout.h"

namespace {

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, int64_t n, recalgo_drop::Spec s,
                                                      float* __restrict__ y) {
    const recalgo_drop::Key k = recalgo_drop::make_key(s);
    const int64_t n4 = n / 4, T = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += T) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 f = recalgo_drop::factor4(s, k, (uint32_t)(4 * i));
        reinterpret_cast<float4*>(y)[i] = make_float4(v.x * f.x, v.y * f.y, v.z * f.z, v.w * f.w);
    }
    const int64_t t = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) y[t] = x[t] * recalgo_drop::factor(s, k, (uint32_t)t);
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(int64_t n, recalgo_drop::Spec s, float* __restrict__ out) {
    const recalgo_drop::Key k = recalgo_drop::make_key(s);
    const int64_t T = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += T)
        out[i] = recalgo_drop::factor(s, k, (uint32_t)i) > 0.f ? 1.f : 0.f;
}

inline bool spec_ok(double rate, const float* x, const float* mask, int64_t n) {
    // float4 paths: base pointers 16-byte aligned (every tensor of the library is); 32-bit element index
    return rate > 0.0 && rate < 1.0 && n >= 0 && n < (int64_t)1 << 32 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(mask) & 15) == 0;
}
inline recalgo_drop::Spec make_spec(double rate, const float* mask, unsigned seed, unsigned call, const int64_t* step) {
    recalgo_drop::Spec s;
    s.mask = mask; s.step = step; s.seed = seed; s.call = call;
    s.threshold = recalgo_drop::threshold_of(rate);
    s.scale = (float)(1.0 / (1.0 - rate));      // TF: keep_prob and its reciprocal are Python doubles, cast to x.dtype once
    return s;
}
inline unsigned blocks_for(int64_t n) {
    const int64_t want = (n / 4 + 255) / 256;
    return (unsigned)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
}

}  // namespace

RECALGO_EXPORT int recalgo_dropout_fwd(const float* x, int64_t n, double rate, const float* keep_mask, unsigned seed, unsigned call,
                                       const int64_t* step, float* y, recalgo_stream_t stream) {
    RECALGO_REQUIRE(spec_ok(rate, x, keep_mask, n) && (reinterpret_cast<uintptr_t>(y) & 15) == 0);
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), x, n,
                       make_spec(rate, keep_mask, seed, call, step), y);
    RECALGO_RETURN_LAST();
}

// the backward is the same map applied to the gradient
RECALGO_EXPORT int recalgo_dropout_bwd(const float* g, int64_t n, double rate, const float* keep_mask, unsigned seed, unsigned call,
                                       const int64_t* step, float* dx, recalgo_stream_t stream) {
    return recalgo_dropout_fwd(g, n, rate, keep_mask, seed, call, step, dx, stream);
}

RECALGO_EXPORT int recalgo_dropout_keep_mask(int64_t n, double rate, unsigned seed, unsigned call, const int64_t* step, float* out,
                                             recalgo_stream_t stream) {
    RECALGO_REQUIRE(rate > 0.0 && rate < 1.0 && n >= 0 && n < (int64_t)1 << 32);
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), n,
                       make_spec(rate, nullptr, seed, call, step), out);
    RECALGO_RETURN_LAST();
}
