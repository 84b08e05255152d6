// CrossNet, fused L-layer backward: the workgroup body, shared by its own kernel (csrc/cross.hip) and by the launches it rides in
// (csrc/dense.hip).  /root/reference algorithm/DCN/cross_layer.py:10-25, dcn.py:157-160.
#pragma once
#include "common.h"

namespace recalgo_cross {

// batched 64-lane butterfly: reduces N independent values with N shuffles in flight per stage
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float t[N];
#pragma unroll
        for (int i = 0; i < N; ++i) t[i] = __shfl_xor(v[i], o, 64);
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] += t[i];
    }
}

// ---------------------------------------------------------------------------------------------
// fused stack, backward.  Per-wave register accumulators (A_l, G, T_l) -> fixed-order LDS
// reduction per workgroup -> this workgroup's share of dw / db as one partial row in global -> column sum
// (colsum4_kernel here, or a job of the step's deferred-sum launch: recalgo_dense_bwd_weights_reduce).
// partial row layout: [dw_0 .. dw_{L-1} | db_0 .. db_{L-1}]   (2*L*d floats)
// ---------------------------------------------------------------------------------------------
// One workgroup's share (WAVES waves; workgroup `block` of `nblocks`, which is also its partial row); smem: WAVES * d +
// (WAVES + 1) * L floats.  A __device__ function so that it can also run as a RIDER inside another kernel's launch
// (csrc/dense.hip dense_bwd_rider_kernel).
template <int NV, int L, int WAVES>
__device__ __forceinline__ void cross_stack_bwd_block(
    const float* __restrict__ x0, unsigned x_stride, const float4* __restrict__ w,
    const float4* __restrict__ b, const float* __restrict__ g, unsigned g_stride,
    const float* __restrict__ g_x0_extra, unsigned B, unsigned d4, float* __restrict__ dx0,
    float* __restrict__ partials, unsigned block, unsigned nblocks, float* smem) {
    constexpr int kBwdWaves = WAVES, kBwdThreads = WAVES * 64;
    const unsigned lane = threadIdx.x & 63;
    const unsigned wib = threadIdx.x >> 6;
    const unsigned wave = block * kBwdWaves + wib;
    const unsigned nwaves = nblocks * kBwdWaves;
    const unsigned d = d4 * 4;

    float4 wv[L][NV];
    float beta_part[L];
    {
        float4 Bv[NV];
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
            beta_part[l] = acc;
        }
    }
    float4 A[L][NV], G[NV];
    float T[L];
#pragma unroll
    for (int v = 0; v < NV; ++v) G[v] = f4_zero();
#pragma unroll
    for (int l = 0; l < L; ++l) {
        T[l] = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) A[l][v] = f4_zero();
    }

    for (unsigned ex = wave; ex < B; ex += nwaves) {
        const float4* xr = reinterpret_cast<const float4*>(x0 + (size_t)ex * x_stride);
        const float4* gr = reinterpret_cast<const float4*>(g + (size_t)ex * g_stride);
        float4 xv[NV], gv[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            unsigned idx = lane + v * 64;
            xv[v] = idx < d4 ? xr[idx] : f4_zero();
            gv[v] = idx < d4 ? gr[idx] : f4_zero();
        }
        float r[2 * L + 1];
        {
            float q = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) q += f4_dot(gv[v], xv[v]);
            r[2 * L] = q;                                   // dc_L = g . x0
        }
#pragma unroll
        for (int l = 0; l < L; ++l) {
            float acc = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) acc += f4_dot(xv[v], wv[l][v]);
            r[l] = acc;
            r[L + l] = beta_part[l];
        }
        wave_sum_n<2 * L + 1>(r);
        float c[L + 1];
        c[0] = 1.f;
#pragma unroll
        for (int l = 0; l < L; ++l) c[l + 1] = c[l] + fmaf(c[l], r[l], r[L + l]);
        // reverse scalar sweep
        float dc = r[2 * L];
        float dp[L];
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            T[l] += dc;                       // dbeta_l = dc_{l+1}
            dp[l] = dc * c[l];
            dc = dc * (1.f + r[l]);
        }
        float4* orow = reinterpret_cast<float4*>(dx0 + (size_t)ex * x_stride);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            unsigned idx = lane + v * 64;
            float4 o = f4_scale(gv[v], c[L]);
#pragma unroll
            for (int l = 0; l < L; ++l) {
                o = f4_fma(wv[l][v], dp[l], o);
                A[l][v] = f4_fma(xv[v], dp[l], A[l][v]);
            }
            G[v] = f4_add(G[v], gv[v]);
            if (idx < d4) {
                if (g_x0_extra)
                    o = f4_add(o, reinterpret_cast<const float4*>(g_x0_extra + (size_t)ex * x_stride)[idx]);
                orow[idx] = o;
            }
        }
    }

    // workgroup reduction in fixed wave order (deterministic).  The partial row already holds this workgroup's share of
    // the FINAL gradients — dw_l = A_l + T_l * B_l (B_l = sum_{j<l} b_j), db_j = G + sum_{l>j} T_l * w_l are linear in
    // (A, G, T) — so what remains is a plain column sum over the partial rows (colsum4_kernel, or one job of the
    // step's deferred-sum launch).  Row layout: [dw_0 .. dw_{L-1} | db_0 .. db_{L-1}]  (2*L*d floats).
    const unsigned row_len = 2 * L * d;
    float* prow = partials + (size_t)block * row_len;
    float* sT = smem + (size_t)kBwdWaves * d;                 // [kBwdWaves][L] per-wave T, then [L] their sum
    const float* wf = reinterpret_cast<const float*>(w);
    const float* bf = reinterpret_cast<const float*>(b);
    if (lane == 0) {
#pragma unroll
        for (int l = 0; l < L; ++l) sT[wib * L + l] = T[l];   // T is wave-uniform
    }
    __syncthreads();
    if (threadIdx.x < L) {
        float acc = 0.f;
        for (int wv_ = 0; wv_ < kBwdWaves; ++wv_) acc += sT[wv_ * L + threadIdx.x];
        sT[kBwdWaves * L + threadIdx.x] = acc;
    }
    __syncthreads();
    float Tw[L];
#pragma unroll
    for (int l = 0; l < L; ++l) Tw[l] = sT[kBwdWaves * L + l];
#pragma unroll
    for (int vec = 0; vec <= L; ++vec) {
        __syncthreads();
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            unsigned idx = lane + v * 64;
            if (idx < d4) {
                float4 val = G[v];
#pragma unroll
                for (int l = 0; l < L; ++l)
                    if (vec == l) val = A[l][v];
                *reinterpret_cast<float4*>(smem + (size_t)wib * d + idx * 4) = val;
            }
        }
        __syncthreads();
        for (unsigned j = threadIdx.x; j < d; j += kBwdThreads) {
            float acc = 0.f;
#pragma unroll
            for (int wv_ = 0; wv_ < kBwdWaves; ++wv_) acc += smem[(size_t)wv_ * d + j];
            if (vec < L) {
                float Bl = 0.f;                               // B_vec[j] = sum_{jj < vec} b_jj[j]
#pragma unroll
                for (int jj = 0; jj < L; ++jj)
                    if (jj < vec) Bl += bf[(size_t)jj * d + j];
                prow[(size_t)vec * d + j] = fmaf(Bl, Tw[vec], acc);
            } else {
                float tail = 0.f;                             // db_jj = G + sum_{l > jj} T_l * w_l[j], built from the top
#pragma unroll
                for (int jj = L - 1; jj >= 0; --jj) {
                    prow[(size_t)(L + jj) * d + j] = acc + tail;
                    tail = fmaf(Tw[jj], wf[(size_t)jj * d + j], tail);
                }
            }
        }
    }
}


}  // namespace recalgo_cross
