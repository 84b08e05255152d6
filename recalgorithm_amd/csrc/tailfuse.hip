// The tail of a DCN training step in ONE launch (algorithm/DCN/dcn.py:163-172 + the loss tail every model_fn shares,
// deepfm.py:214-238): the LAST hidden layer, the one-unit head over concat([cross_out, h3]), sigmoid, the mean
// cross-entropy and — the loss-gradient seed being known — the backward of all of it down to the layer's input:
//     h3 = relu(h2 w3 + b3)                  logit = <side, w_side> + <h3, w_h3> + bias            prob, loss, dlogit
//     d_side = dlogit w_side                 dz3 = dlogit w_h3 * [h3 > 0]                          dh2 = (dz3 w3^T) * [h2 > 0]
// Rounds 2-5 ran this as three launches (recalgo_dense_fwd 8.5 us, recalgo_logit_loss_fwd_bwd 10.5 us, recalgo_dense_bwd
// 10.4 us in the step) whose fixed costs and latency chains are most of their time; h3 and the head's gradient wrt h3 never
// leave the chip here.  The layer's WEIGHT gradient (h2^T dz3, a reduction over the batch) stays a launch of its own
// (recalgo_dense_bwd_weights on the dz3 this kernel writes).
//
// The kernel is one latency chain per workgroup (loads -> GEMM -> row sums -> loss -> GEMM -> stores), so it is built to keep
// that chain short rather than to keep the matrix pipe busy:
//   * 16 examples per workgroup (B = 4096: 256 workgroups = one per CU), 8 waves, v_mfma_f32_16x16x4_f32;
//   * every global load of the forward — the h2 tile, the side tile, the head's weights and the wave's whole slice of w3 —
//     is requested before the first wait; the backward GEMM's slice of w3 is requested before the head / loss phase;
//   * forward: wave w owns columns [16 w, 16 w + 16) of h3; A = the h2 tile in LDS (one ds_read_b128 per four steps), B in
//     registers straight from global / L2, two accumulator chains (a dependent 16x16x4 issues every 40 cycles, not 32);
//   * head: row sums of the accumulator registers over 16-lane rows (DPP), the side part from the LDS copy of the side tile —
//     which the backward reads again for d_side and dw_side instead of going back to global memory;
//   * backward: dz3 goes to global (for the weight gradient) and through LDS becomes the A operand of dh2 = dz3 w3^T, whose
//     B operand — w3[j][n .. n + 3] — is ONE float4 load per four steps; wave w owns K2 / 8 columns of dh2.
// Partial row per workgroup, laid out as recalgo_logit_loss_fwd_bwd's: [dw over the concatenated head columns | d bias | loss].
#include "common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kTR = 16;                  // examples per workgroup
constexpr int kN3 = 128;                 // units of the last hidden layer (8 waves x 16 columns)
constexpr int kThreads = 512;

// scripts/tailfuse_lab.hip builds this file with -DTAILFUSE_TIMELINE: shader-clock totals per phase of wave 0 of workgroup 0
#ifdef TAILFUSE_TIMELINE
__device__ unsigned long long tailfuse_tl[16];
// (time stamps stay in registers until the end: a store here would wait for every load in flight)
#define TF_TL_BEGIN() unsigned long long tl_t_[12]; tl_t_[11] = clock64()
#define TF_TL(i)                                                 \
    do {                                                         \
        __builtin_amdgcn_sched_barrier(0);                       \
        tl_t_[i] = clock64();                                    \
        __builtin_amdgcn_sched_barrier(0);                       \
    } while (0)
#define TF_TL_END()                                                                      \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                           \
        tailfuse_tl[0] += tl_t_[0] - tl_t_[11];                                          \
        for (int i_ = 1; i_ < 11; ++i_) tailfuse_tl[i_] += tl_t_[i_] - tl_t_[i_ - 1];    \
    }
#else
#define TF_TL_BEGIN()
#define TF_TL(i)
#define TF_TL_END()
#endif

struct TailFuseArgs {
    const float* h2; const float* w3; const float* b3;
    const float* side; const float* w_side; const float* w_h3; const float* head_bias;
    const float* labels; const float* loss_addend;
    float* logit; float* prob; float* dlogit; float* d_side; float* dz3; float* dh2; float* partials;
    int B, Cs, col_side, col_h3;         // col_*: first column of the part inside the head's concatenation (and the partial row)
    float grad_scale;
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float f4_at(const float4& v, int x) { return x == 0 ? v.x : (x == 1 ? v.y : (x == 2 ? v.z : v.w)); }

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a DPP row, in every lane of the row: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
// (VALU operand modifiers: no trip through the LDS crossbar, which __shfl_xor's ds_bpermute takes)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
// sum over the wave, as a wave-uniform value: row sums, then the four rows' lane 0 through the scalar unit
__device__ __forceinline__ float wave_sum_uniform(float v) {
    v = row16_sum(v);
    const int i = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48)));
}

template <int KC>                         // K2 = 128 KC
__global__ __launch_bounds__(kThreads) void tail_dense_head_kernel(TailFuseArgs P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K2 = 128 * KC;
    constexpr int HS = K2 + 4, ZS = kN3 + 4;              // row strides: conflict-free ds_read_b128 per 8-lane group
    const int Cs = P.Cs, C = Cs + kN3, SS = Cs + 4;
    float* Hs = smem;                                     // [kTR][HS]   the h2 tile
    float* Zs = Hs + kTR * HS;                            // [kTR][ZS]   dz3
    float* Ss = Zs + kTR * ZS;                            // [kTR][SS]   the side tile
    float* wcat = Ss + kTR * SS;                          // [Cs | 128]  w_side | w_h3
    float* rowpart = wcat + ((C + 3) & ~3);               // [8][kTR]    <h3, w_h3> per wave's 16 columns
    float* rowside = rowpart + 8 * kTR;                   // [kTR]
    float* s_dl = rowside + kTR;                          // [kTR]
    float* s_loss = s_dl + kTR;                           // [kTR]
    const int b0 = blockIdx.x * kTR;
    const int nb = min(kTR, P.B - b0);
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, l16 = lane & 15;
    const unsigned col = wave * 16 + l16;                 // this lane's column of h3
    const float invB = 1.0f / (float)P.B;
    TF_TL_BEGIN();

    // ---- every global load of the forward, requested before the first wait ----
    constexpr int NH = kTR * (K2 / 4) / kThreads;         // float4 of the h2 tile per thread (KC)
    float4 hv[NH];
#pragma unroll
    for (int u = 0; u < NH; ++u) {
        const int idx = (int)tid + u * kThreads, r = idx / (K2 / 4), c4 = idx % (K2 / 4);
        hv[u] = r < nb ? reinterpret_cast<const float4*>(P.h2 + (size_t)(b0 + r) * K2)[c4] : f4_zero();
    }
    const int Cs4 = Cs / 4;
    float4 sv[8];                                         // kTR * Cs4 <= 16 * 256 = 8 x 512
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int idx = (int)tid + u * kThreads;
        if (idx < kTR * Cs4) {
            const int r = idx / Cs4, c4 = idx - r * Cs4;
            sv[u] = r < nb ? reinterpret_cast<const float4*>(P.side + (size_t)(b0 + r) * Cs)[c4] : f4_zero();
        }
    }
    const float bias3 = P.b3[col];
    float wstage[3];                                      // (Cs + 128) / 512 <= 3 head weights per thread
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int j = (int)tid + u * kThreads;
        wstage[u] = j < Cs ? P.w_side[j] : (j < C ? P.w_h3[j - Cs] : 0.f);
    }
    float lab = 0.f, hb = 0.f;
    if (tid < (unsigned)nb) lab = P.labels[b0 + tid];
    if (tid < (unsigned)kTR && P.head_bias) hb = P.head_bias[0];
    // (the tile loads above are waited for first: in-order return, so the B operand is requested after them)
    // B operand: step x of group g of chunk c multiplies k = 128 c + 16 g + 4 kq + x
    // (the first 128 k now; the rest once the tiles have landed — the vector memory pipe moves 64 B / clk per CU, and the
    // workgroup's whole w3 in front of the barrier delays the tiles everything waits for)
    float bf[KC][8][4];
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int x = 0; x < 4; ++x) bf[0][g][x] = P.w3[(size_t)(16 * g + 4 * kq + x) * kN3 + col];
#pragma unroll
    for (int u = 0; u < NH; ++u) {
        const int idx = (int)tid + u * kThreads, r = idx / (K2 / 4), c4 = idx % (K2 / 4);
        *reinterpret_cast<float4*>(Hs + r * HS + 4 * c4) = hv[u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int idx = (int)tid + u * kThreads;
        if (idx < kTR * Cs4) {
            const int r = idx / Cs4, c4 = idx - r * Cs4;
            *reinterpret_cast<float4*>(Ss + r * SS + 4 * c4) = sv[u];
        }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int j = (int)tid + u * kThreads;
        if (j < C) wcat[j] = wstage[u];
    }
    __syncthreads();
#pragma unroll
    for (int c = 1; c < KC; ++c)
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int x = 0; x < 4; ++x) bf[c][g][x] = P.w3[(size_t)(128 * c + 16 * g + 4 * kq + x) * kN3 + col];
    TF_TL(0);
    // ---- the side part of the logit: wave w takes rows 2 w, 2 w + 1 (LDS only) ----
    if (Cs > 0) {
        const float4* w4 = reinterpret_cast<const float4*>(wcat);
        float d0 = 0.f, d1 = 0.f;
        for (int c4 = (int)lane; c4 < Cs4; c4 += 64) {
            const float4 ww = w4[c4];
            d0 += f4_dot(*reinterpret_cast<const float4*>(Ss + (2 * wave) * SS + 4 * c4), ww);
            d1 += f4_dot(*reinterpret_cast<const float4*>(Ss + (2 * wave + 1) * SS + 4 * c4), ww);
        }
        d0 = wave_sum_uniform(d0);
        d1 = wave_sum_uniform(d1);
        if (lane == 0) { rowside[2 * wave] = d0; rowside[2 * wave + 1] = d1; }
    } else if (tid < (unsigned)kTR) {
        rowside[tid] = 0.f;
    }
    TF_TL(1);
    // ---- forward: h3[16 rows][16 columns of this wave] ----
    f32x4 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc0[r] = bias3; acc1[r] = 0.f; }
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(Hs + l16 * HS + 128 * c + 16 * g + 4 * kq);
            if (g & 1) {
#pragma unroll
                for (int x = 0; x < 4; ++x) acc1 = mfma16(f4_at(a, x), bf[c][g][x], acc1);
            } else {
#pragma unroll
                for (int x = 0; x < 4; ++x) acc0 = mfma16(f4_at(a, x), bf[c][g][x], acc0);
            }
        }
    TF_TL(2);
    // ---- B operand of the backward GEMM: tile t of this wave, step x of group g multiplies n = 16 g + 4 kq + x ----
    // (requested now, consumed after the head / loss phase)
    float4 bb[KC][8];
#pragma unroll
    for (int t = 0; t < KC; ++t)
#pragma unroll
        for (int g = 0; g < 8; ++g)
            bb[t][g] = *reinterpret_cast<const float4*>(P.w3 + (size_t)(wave * (16 * KC) + 16 * t + l16) * kN3 + 16 * g + 4 * kq);
    // accumulator register r of lane (l16, kq): row 4 kq + r, column `col`
    float h[4];
    const float wn = wcat[Cs + col];
    {
        float s[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            h[r] = fmaxf(acc0[r] + acc1[r], 0.f);
            s[r] = h[r] * wn;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] += dpp_mov<0xB1>(s[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] += dpp_mov<0x4E>(s[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] += dpp_mov<0x141>(s[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] += dpp_mov<0x140>(s[r]);
        if (l16 == 0) *reinterpret_cast<float4*>(rowpart + wave * kTR + 4 * kq) = make_float4(s[0], s[1], s[2], s[3]);
    }
    TF_TL(3);
    __syncthreads();
    TF_TL(4);
    // ---- logit, probability, loss, d loss / d logit: one thread per row ----
    if (tid < (unsigned)kTR) {
        const int r = (int)tid;
        float d = 0.f, ls = 0.f;
        if (r < nb) {
            const int b = b0 + r;
            float x = ((rowpart[r] + rowpart[kTR + r]) + (rowpart[2 * kTR + r] + rowpart[3 * kTR + r])) +
                      ((rowpart[4 * kTR + r] + rowpart[5 * kTR + r]) + (rowpart[6 * kTR + r] + rowpart[7 * kTR + r]));
            x += rowside[r];
            x += hb;
            const float z = lab;
            const float e = expf(-fabsf(x));
            ls = fmaxf(x, 0.f) - x * z + log1pf(e);               // tf.nn.sigmoid_cross_entropy_with_logits
            const float rr = e / (1.0f + e);
            const float pr = x >= 0.f ? 1.0f / (1.0f + e) : rr;
            d = (((x >= 0.f ? 1.0f : 0.f) - z) + (x >= 0.f ? -rr : rr)) * P.grad_scale * invB;
            P.logit[b] = x;
            P.prob[b] = pr;
            P.dlogit[b] = d;
        }
        s_dl[r] = d;
        s_loss[r] = ls;
    }
    __syncthreads();
    TF_TL(5);
    float* __restrict__ prow = P.partials + (size_t)blockIdx.x * (C + 2);
    // ---- dz3 (the A operand of the backward GEMM) and dw_h3 first: the GEMM waits on nothing else ----
    {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (int)kq + r;
            const float dl = s_dl[row];
            a = fmaf(dl, h[r], a);
            const float dz = h[r] > 0.f ? dl * wn : 0.f;
            Zs[row * ZS + col] = dz;
            if (row < nb) P.dz3[(size_t)(b0 + row) * kN3 + col] = dz;
        }
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        if (kq == 0) prow[P.col_h3 + col] = a;
    }
    TF_TL(6);
    // ---- head backward on the side part: d_side, dw_side (from the LDS copy of the side tile) ----
    // (the two waves of a SIMD take it on opposite sides of the backward GEMM: one's stores run under the other's MFMAs)
    auto side_backward = [&]() {
        if (Cs > 0) {
            const float4* w4 = reinterpret_cast<const float4*>(wcat);
            if (P.d_side) {
                for (int idx = tid; idx < kTR * Cs4; idx += kThreads) {
                    const int r = idx / Cs4, c4 = idx - r * Cs4;
                    if (r < nb) reinterpret_cast<float4*>(P.d_side + (size_t)(b0 + r) * Cs)[c4] = f4_scale(w4[c4], s_dl[r]);
                }
            }
            for (int c = tid; c < Cs; c += kThreads) {
                float a = 0.f;
#pragma unroll
                for (int r = 0; r < kTR; ++r) a = fmaf(s_dl[r], Ss[r * SS + c], a);
                prow[P.col_side + c] = a;
            }
        }
    };
    if (wave >= 4) side_backward();
    if (tid == 0) {
        float sd = 0.f, sl = 0.f;
        for (int r = 0; r < kTR; ++r) { sd += s_dl[r]; sl += s_loss[r]; }
        prow[C] = sd;
        prow[C + 1] = sl * invB + ((blockIdx.x == 0 && P.loss_addend) ? P.loss_addend[0] : 0.f);
    }
    TF_TL(7);
    __syncthreads();
    TF_TL(8);
    // ---- dh2 = (dz3 w3^T) * [h2 > 0]: wave w owns columns [16 KC w, 16 KC (w + 1)), K = the 128 units ----
    f32x4 dacc[KC];
#pragma unroll
    for (int t = 0; t < KC; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dacc[t][r] = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const float4 a = *reinterpret_cast<const float4*>(Zs + l16 * ZS + 16 * g + 4 * kq);
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int t = 0; t < KC; ++t) dacc[t] = mfma16(f4_at(a, x), f4_at(bb[t][g], x), dacc[t]);
    }
    if (wave < 4) side_backward();
    TF_TL(9);
#pragma unroll
    for (int t = 0; t < KC; ++t) {
        const int j = (int)wave * (16 * KC) + 16 * t + (int)l16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (int)kq + r;
            if (row < nb) P.dh2[(size_t)(b0 + row) * K2 + j] = Hs[row * HS + j] > 0.f ? dacc[t][r] : 0.f;
        }
    }
    TF_TL(10);
    TF_TL_END();
}

inline size_t tail_smem(int K2, int Cs) {
    const int C = Cs + kN3;
    return (size_t)(kTR * (K2 + 4) + kTR * (kN3 + 4) + kTR * (Cs + 4) + ((C + 3) & ~3) + 8 * kTR + 3 * kTR) * sizeof(float);
}

template <int KC>
int launch_tail(const TailFuseArgs& P, int Cs, hipStream_t s) {
    const size_t smem = tail_smem(128 * KC, Cs);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_dense_head_kernel<KC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(tail_dense_head_kernel<KC>, dim3(cdiv(P.B, kTR)), dim3(kThreads), smem, s, P);
    RECALGO_RETURN_LAST();
}

}  // namespace

RECALGO_EXPORT int recalgo_tail_partial_rows(int B) { return B > 0 ? cdiv(B, kTR) : 0; }

RECALGO_EXPORT int recalgo_tail_dense_head_supported(int K2, int N3, int Cs) {
    return (N3 == kN3 && K2 >= 128 && K2 <= 512 && K2 % 128 == 0 && Cs >= 0 && Cs <= 1024 && Cs % 4 == 0) ? 1 : 0;
}

RECALGO_EXPORT int recalgo_tail_dense_head_fwd_bwd(const float* h2, int K2, const float* w3, const float* b3, int N3, const float* side,
                                                   int Cs, int side_first, const float* w_side, const float* w_h3,
                                                   const float* head_bias, const float* labels, const float* loss_addend, int B,
                                                   float grad_scale, float* logit, float* prob, float* dlogit, float* d_side,
                                                   float* dz3, float* dh2, float* partials, recalgo_stream_t stream) {
    RECALGO_REQUIRE(recalgo_tail_dense_head_supported(K2, N3, Cs) && B > 0);
    RECALGO_REQUIRE(h2 && w3 && b3 && w_h3 && labels && logit && prob && dlogit && dz3 && dh2 && partials);
    RECALGO_REQUIRE(Cs == 0 || (side && w_side));
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    RECALGO_REQUIRE(al(h2) && al(w3) && al(side) && al(d_side));
    TailFuseArgs P;
    P.h2 = h2; P.w3 = w3; P.b3 = b3; P.side = Cs ? side : nullptr; P.w_side = w_side; P.w_h3 = w_h3; P.head_bias = head_bias;
    P.labels = labels; P.loss_addend = loss_addend; P.logit = logit; P.prob = prob; P.dlogit = dlogit; P.d_side = d_side;
    P.dz3 = dz3; P.dh2 = dh2; P.partials = partials; P.B = B; P.Cs = Cs;
    P.col_side = side_first ? 0 : kN3;
    P.col_h3 = side_first ? Cs : 0;
    P.grad_scale = grad_scale;
    hipStream_t s = as_stream(stream);
    switch (K2 / 128) {
        case 1: return launch_tail<1>(P, Cs, s);
        case 2: return launch_tail<2>(P, Cs, s);
        case 3: return launch_tail<3>(P, Cs, s);
        default: return launch_tail<4>(P, Cs, s);
    }
}
