// TF1 Adam on one element, and the deferred-exact form of its dense semantics (see sparse.hip): the pieces shared by the
// optimizer kernels (sparse.hip, tail.hip) and the forward lookups (embed.hip).
//
// The update (tf.train.AdamOptimizer; /root/reference algorithm/DeepFM/deepfm.py:246-250):
//     m = b1 * m + (1 - b1) * g;  v = b2 * v + (1 - b2) * g * g;  p -= lr_t * m / (sqrt(v) + eps)
// is evaluated as  p = fma(-(lr_t * m), rcp(sqrt(v) + eps), p)  with the hardware's v_sqrt_f32 / v_rcp_f32 (1 ulp each;
// the step is within 3 ulp of the correctly rounded quotient, i.e. ~4e-7 relative on a quantity that moves p by ~lr —
// far inside the 1e-5 the parity tests hold an Adam step to).  Why not IEEE division and sqrt: a row whose state is valid
// for step s < target takes the g = 0 updates of steps s+1 .. target one after the other (no closed form reproduces the
// fp32 roundings of the dense pass), and the correctly rounded forms cost 45 instructions per replayed step against 9 —
// the replay of a DCN batch's ~35 k lagging rows was 12 us of pure VALU time per step.  EVERY Adam update of the
// library goes through adam1(), so the deferred form stays bit-identical to the dense pass (tests/test_gpu_sparse.py).
#pragma once
#include "common.h"

namespace recalgo_deferred {

constexpr unsigned kLrRing = RECALGO_LR_RING;           // power of two

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float lr_t, float b1, float b2, float eps) {
    m = fmaf(b1, m, (1.f - b1) * g);
    v = fmaf(b2, v, (1.f - b2) * g * g);
    p = fmaf(-(lr_t * m), __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + eps), p);
}
__device__ __forceinline__ void vadam(float4& p, const float4 g, float4& m, float4& v, float lr_t, float b1, float b2, float eps) {
    adam1(p.x, g.x, m.x, v.x, lr_t, b1, b2, eps);
    adam1(p.y, g.y, m.y, v.y, lr_t, b1, b2, eps);
    adam1(p.z, g.z, m.z, v.z, lr_t, b1, b2, eps);
    adam1(p.w, g.w, m.w, v.w, lr_t, b1, b2, eps);
}
__device__ __forceinline__ void vadam(float& p, const float g, float& m, float& v, float lr_t, float b1, float b2, float eps) {
    adam1(p, g, m, v, lr_t, b1, b2, eps);
}
__device__ __forceinline__ float4 vzero(const float4*) { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float vzero(const float*) { return 0.f; }

// replay the g = 0 updates of steps s+1 .. target on one lane's piece (float4 or float) of a row
template <typename V>
__device__ __forceinline__ void replay(V& w, V& m, V& v, int s, int target, const float* lr_ring, float b1, float b2, float eps) {
    for (int j = s + 1; j <= target; ++j)
        vadam(w, vzero(static_cast<const V*>(nullptr)), m, v, lr_ring[(unsigned)j & (kLrRing - 1)], b1, b2, eps);
}
__device__ __forceinline__ void replay1(float& w, float& m, float& v, int s, int target, const float* lr_ring, float b1, float b2, float eps) {
    replay<float>(w, m, v, s, target, lr_ring, b1, b2, eps);
}

// The replay of a whole WAVE's rows: every lane brings (w, m, v) from its own s to `target` (s >= target: nothing to do).
// The step index runs wave-uniformly from the smallest s of the wave — so lr_t(j) is ONE scalar per iteration, read from a
// window of the ring held one entry per lane (v_readlane: no memory access inside the loop) — and a lane joins in when
// the loop reaches its own s.  Same operations in the same order as replay(): bit-identical.
struct LrWindow {
    float win;             // lane i: lr_t(w0 + i)
    int w0;
};
__device__ __forceinline__ LrWindow lr_window(const float* lr_ring, int target) {
    LrWindow W;
    W.w0 = target - 63;
    const int j = W.w0 + (int)(threadIdx.x & 63);
    W.win = j >= 1 ? lr_ring[(unsigned)j & (kLrRing - 1)] : 0.f;
    return W;
}
__device__ __forceinline__ int wave_min_int(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ void replay_wave(float& w, float& m, float& v, int s, int target, const LrWindow& W, const float* lr_ring,
                                            float b1, float b2, float eps) {
    const int smin = __builtin_amdgcn_readfirstlane(wave_min_int(s < target ? s : target));
    for (int j = smin + 1; j <= target; ++j) {
        const float lr = j >= W.w0 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, W.win), j - W.w0))
                                   : lr_ring[(unsigned)j & (kLrRing - 1)];
        if (j > s) adam1(w, 0.f, m, v, lr, b1, b2, eps);
    }
}

// Read-only view for the forward lookups: a lookup that meets a row whose state lags replays the missed steps in
// registers and uses the result WITHOUT writing it back (the optimizer's `apply` / the sweep do the real catch-up).
// Hot rows are current, so the extra loads (m, v) and the loop only run for rows the batch has not seen for a while.
struct ReadView {
    const float* m; const float* v;        // arena based, like last_step
    const int* last_step;                  // nullptr: no deferred state, plain lookup
    const float* lr_ring;
    const long long* step;                 // rows are brought to step[0] + step_off
    int step_off;
    float b1, b2, eps;
    long long row_offset;                  // arena row of row 0 of the table the kernel indexes
};

// w = the piece (index q of KV per row) of table row `row` as the lookup read it; returns it as of the target step
template <typename V>
__device__ __forceinline__ V current_piece(const ReadView& D, V w, long long row, unsigned q, unsigned KV) {
    if (D.last_step == nullptr) return w;
    const long long ar = row + D.row_offset;
    const int s = D.last_step[ar];
    const int target = (int)(D.step[0] + D.step_off);
    if (s <= 0 || s >= target) return w;
    V m = reinterpret_cast<const V*>(D.m)[(size_t)ar * KV + q], v = reinterpret_cast<const V*>(D.v)[(size_t)ar * KV + q];
    replay(w, m, v, s, target, D.lr_ring, D.b1, D.b2, D.eps);
    return w;
}

}  // namespace recalgo_deferred
