// Deferred exact TF1 Adam (see sparse.hip): the pieces shared by the optimizer kernels (sparse.hip) and the forward
// lookups (embed.hip).  A row whose state is valid for step s < target takes the g = 0 updates of steps s+1 .. target:
//     m = b1 * m;  v = b2 * v;  w -= lr_t(j) * m / (sqrt(v) + eps)
// with exactly the fp32 operations of the dense pass (recalgo_adam_tf1_dense / _step), so the replay is bit-identical to it.
#pragma once
#include "common.h"

namespace recalgo_deferred {

constexpr unsigned kLrRing = RECALGO_LR_RING;           // power of two

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float lr_t, float b1, float b2, float eps) {
    m = fmaf(b1, m, (1.f - b1) * g);
    v = fmaf(b2, v, (1.f - b2) * g * g);
    p -= lr_t * m / (sqrtf(v) + eps);
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
// the same for one float, with the two addends of adam1 that vanish for g = 0 evaluated once: (1 - b1) * 0 and
// (1 - b2) * 0 * 0 are the SAME values adam1 forms each step, so every fma below has adam1's operands bit for bit
__device__ __forceinline__ void replay1(float& w, float& m, float& v, int s, int target, const float* lr_ring, float b1, float b2, float eps) {
    const float z = 0.f;
    const float c1 = (1.f - b1) * z, c2 = (1.f - b2) * z * z;
    for (int j = s + 1; j <= target; ++j) {
        m = fmaf(b1, m, c1);
        v = fmaf(b2, v, c2);
        w -= lr_ring[(unsigned)j & (kLrRing - 1)] * m / (sqrtf(v) + eps);
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
