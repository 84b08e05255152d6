// PReLU / Dice on one element and their derivatives (/root/reference algorithm/DIN/activations.py:4-37): shared by the
// elementwise kernels (tail.hip), the dense layer's epilogue (dense.hip) and the BatchNorm backward that continues through
// the activation (mlp.hip).
#pragma once
#include "common.h"

namespace recalgo_act {

constexpr float kDiceInvStd = 0.99950037468777310f;  // 1/sqrt(1 + 1e-3): Dice's BatchNorm never trains (stats 0, 1; quirk B-5)

__device__ __forceinline__ float prelu(float x, float a) { return fmaxf(0.f, x) + a * fminf(0.f, x); }
__device__ __forceinline__ float dice(float x, float a) {
    const float px = 1.0f / (1.0f + expf(-x * kDiceInvStd));
    return x * px + a * x * (1.0f - px);
}
// g = dL/dy -> dL/dx (returned) and this element's term of dL/dalpha
template <bool DICE>
__device__ __forceinline__ float bwd(float x, float a, float g, float& dalpha) {
    if (DICE) {
        const float px = 1.0f / (1.0f + expf(-x * kDiceInvStd));
        const float dpx = px * (1.0f - px) * kDiceInvStd;
        dalpha = g * x * (1.0f - px);
        return g * (px + a * (1.0f - px) + x * dpx * (1.0f - a));            // y = x*px + a*x*(1-px)
    }
    dalpha = g * fminf(0.f, x);
    return g * (x > 0.f ? 1.0f : (x < 0.f ? a : 0.f));
}

}  // namespace recalgo_act
