// K9: DIN attention (algorithm/DIN/din_attention.py:4-43), gfx950.
//
//   x_t = [q, k_t, q - k_t, q * k_t]                                   (4H)
//   s_t = f3(relu(f2(relu(f1 x_t))))            dense 4H -> 64 -> 32 -> 1, with biases
//   softmax branch : s_t <- (t < len ? s_t : -2^32+1) / sqrt(H);  w = softmax_t(s)
//   default branch : w_t = s_t * [t < len]
//   out = sum_t w_t k_t
//
// FLOP-bound (fp32): per history row 2*(2H*64 + 64*32) MACs forward, ~3.5x that backward.  The
// MLP over the T rows of one example is a chain of small GEMMs with M = 64 rows (T <= 64, zero
// padded), so it runs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32; same peak as the VALU on
// gfx950, but an MFMA needs 2 LDS dwords per lane per 64 cycles where the register-resident VALU
// formulation needed one broadcast ds_read_b128 per 4 FMAs and was LDS-bound, and its backward
// spilled 1000+ VGPRs).  One wave owns one example:
//   f1 is factored so that its q-only part is computed once per example:
//       f1 x = q (W1a + W1c) + k (W1b - W1c) + (q*k) W1d  =  cq + X Wx,   X = [k | q*k]  (64 x 2H)
//   H1 = relu(X Wx + cq)      64 MFMAs      (A = X from LDS, B = Wx from LDS, C initialised with cq)
//   H2 = relu(H1 W2 + b2)     64 MFMAs      (H1 goes through LDS: accumulator layout -> A layout)
//   s  = H2 W3 + b3           VALU, lane = row
// backward (recomputes the forward; nothing but q, k is read):
//   dW2 += H1^T dH2  (64)   dH1 = (dH2 W2^T) * [H1>0]  (64)   dWx += X^T dH1  (64)   dX = dH1 Wx^T  (64)
// Weights are staged once per persistent workgroup in LDS with odd row strides (33 / 65 floats) so
// that both the row-major (B operand) and the transposed (B operand of the backward) fragment
// reads are bank-conflict free.  Weight-gradient accumulators stay in registers across the
// examples of a wave, then: fixed-order reduction over the 4 waves -> partial row -> column sums.
//
// Fragment maps of v_mfma_f32_32x32x2_f32: lane l supplies A[row = l&31][k = l>>5] and
// B[k = l>>5][col = l&31]; acc reg r holds C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int N1 = 64, N2 = 32;
constexpr int XS = 33;                           // row stride of X / H2 / W2 tiles (floats)
constexpr int HS = 65;                           // row stride of H1 / Wx tiles
constexpr float kPadScore = -4294967296.0f;      // float32(-2**32 + 1), din_attention.py:31

template <int H>
struct Weights {            // workgroup-shared, staged once
    float Wx[32][HS];       // rows 0..H-1: W1b - W1c (k part), rows H..2H-1: W1d (q*k part), rest 0
    float Wq[H][N1];        // W1a + W1c
    float b1[N1];
    float W2[N1][XS];
    float b2[N2];
    float W3[N2];
    float b3[4];
};

struct WaveScratch {        // one per wave
    float Xs[64][XS];       // [k | q*k | 0]; later dX
    float H1s[64][HS];      // relu(H1); later dH1
    float H2s[64][XS];      // relu(H2); later dH2
    float v64[2][64];       // cq ; ds | dcq
};

template <int H>
__device__ __forceinline__ void stage_weights(Weights<H>& S, const float* __restrict__ f1w,
                                              const float* __restrict__ f1b, const float* __restrict__ f2w,
                                              const float* __restrict__ f2b, const float* __restrict__ f3w,
                                              const float* __restrict__ f3b) {
    for (unsigned e = threadIdx.x; e < 32 * N1; e += kThreads) {
        unsigned i = e / N1, j = e - i * N1;
        float v = 0.f;
        if (i < (unsigned)H) v = f1w[(1 * H + i) * N1 + j] - f1w[(2 * H + i) * N1 + j];
        else if (i < 2u * H) v = f1w[(3 * H + (i - H)) * N1 + j];
        S.Wx[i][j] = v;
    }
    for (unsigned e = threadIdx.x; e < H * N1; e += kThreads) {
        unsigned i = e / N1, j = e - i * N1;
        S.Wq[i][j] = f1w[(0 * H + i) * N1 + j] + f1w[(2 * H + i) * N1 + j];
    }
    for (unsigned e = threadIdx.x; e < N1 * N2; e += kThreads) S.W2[e / N2][e % N2] = f2w[e];
    if (threadIdx.x < N1) S.b1[threadIdx.x] = f1b[threadIdx.x];
    if (threadIdx.x < N2) {
        S.b2[threadIdx.x] = f2b[threadIdx.x];
        S.W3[threadIdx.x] = f3w[threadIdx.x];
    }
    if (threadIdx.x == 0) S.b3[0] = f3b[0];
}

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ unsigned acc_row(int r, unsigned hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// loads q (wave-uniform) and this lane's key row (zero beyond T)
template <int H>
__device__ __forceinline__ void load_example(unsigned lane, unsigned ex, unsigned T, const float* __restrict__ query,
                                             const float* __restrict__ keys, float (&q)[H], float (&k)[H]) {
    const float4* qr = reinterpret_cast<const float4*>(query + (size_t)ex * H);
#pragma unroll
    for (int i = 0; i < H; i += 4) {
        float4 v = qr[i / 4];
        q[i] = v.x; q[i + 1] = v.y; q[i + 2] = v.z; q[i + 3] = v.w;
    }
    if (lane < T) {
        const float4* kr = reinterpret_cast<const float4*>(keys + ((size_t)ex * T + lane) * H);
#pragma unroll
        for (int i = 0; i < H; i += 4) {
            float4 v = kr[i / 4];
            k[i] = v.x; k[i + 1] = v.y; k[i + 2] = v.z; k[i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < H; ++i) k[i] = 0.f;
    }
}

// the MLP of one example on the matrix cores: leaves relu(H1) in sc.H1s, relu(H2) in sc.H2s, X in
// sc.Xs and returns the raw score of row `lane`
template <int H>
__device__ __forceinline__ float mlp_forward(const Weights<H>& W, WaveScratch& sc, unsigned lane,
                                             const float (&q)[H], const float (&k)[H]) {
    const unsigned hi = lane >> 5, l32 = lane & 31;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        sc.Xs[lane][i] = k[i];
        sc.Xs[lane][H + i] = q[i] * k[i];
    }
#pragma unroll
    for (int i = 2 * H; i < 32; ++i) sc.Xs[lane][i] = 0.f;
    {   // cq[j] = b1[j] + sum_i q_i (W1a + W1c)[i][j]   — lane j computes output j
        float c = W.b1[lane];
#pragma unroll
        for (int i = 0; i < H; ++i) c = fmaf(q[i], W.Wq[i][lane], c);
        sc.v64[0][lane] = c;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- layer 1: H1[64 x 64] = X[64 x 2H] Wx[2H x 64] + cq ----
    f32x16 a1[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const float c = sc.v64[0][nt * 32 + l32];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) a1[mt][nt][r] = c;
    }
#pragma unroll 4
    for (int k0 = 0; k0 < 2 * H; k0 += 2) {
        const float x0 = sc.Xs[l32][k0 + hi], x1 = sc.Xs[32 + l32][k0 + hi];
        const float w0 = W.Wx[k0 + hi][l32], w1 = W.Wx[k0 + hi][32 + l32];
        a1[0][0] = mfma(x0, w0, a1[0][0]);
        a1[0][1] = mfma(x0, w1, a1[0][1]);
        a1[1][0] = mfma(x1, w0, a1[1][0]);
        a1[1][1] = mfma(x1, w1, a1[1][1]);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sc.H1s[mt * 32 + acc_row(r, hi)][nt * 32 + l32] = fmaxf(a1[mt][nt][r], 0.f);
    __builtin_amdgcn_wave_barrier();
    // ---- layer 2: H2[64 x 32] = H1[64 x 64] W2[64 x 32] + b2 ----
    f32x16 a2[2];
    {
        const float c = W.b2[l32];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) a2[mt][r] = c;
    }
#pragma unroll 8
    for (int k0 = 0; k0 < N1; k0 += 2) {
        const float h0 = sc.H1s[l32][k0 + hi], h1 = sc.H1s[32 + l32][k0 + hi];
        const float w = W.W2[k0 + hi][l32];
        a2[0] = mfma(h0, w, a2[0]);
        a2[1] = mfma(h1, w, a2[1]);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc.H2s[mt * 32 + acc_row(r, hi)][l32] = fmaxf(a2[mt][r], 0.f);
    __builtin_amdgcn_wave_barrier();
    // ---- layer 3 (lane = row) ----
    float s = W.b3[0];
#pragma unroll
    for (int n = 0; n < N2; ++n) s = fmaf(sc.H2s[lane][n], W.W3[n], s);
    return s;
}

template <int H>
__device__ __forceinline__ float attention_weight(float s, bool in_len, bool in_T, int is_softmax) {
    if (is_softmax) {
        float v = (in_len ? s : kPadScore) / sqrtf((float)H);      // mask, then scale (:32-34)
        float vm = in_T ? v : -INFINITY;
        float mx = wave_max(vm);
        float e = in_T ? expf(v - mx) : 0.f;
        float den = wave_sum(e);
        return e / den;
    }
    return in_len ? s : 0.f;                                       // s * mask (:37-38)
}

template <int H>
__global__ __launch_bounds__(kThreads) void din_attention_fwd_kernel(
    const float* __restrict__ query, const float* __restrict__ keys, const int32_t* __restrict__ keys_length,
    const float* __restrict__ f1w, const float* __restrict__ f1b, const float* __restrict__ f2w,
    const float* __restrict__ f2b, const float* __restrict__ f3w, const float* __restrict__ f3b, unsigned B,
    unsigned T, int is_softmax, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Weights<H>& W = *reinterpret_cast<Weights<H>*>(smem_raw);
    WaveScratch* scs = reinterpret_cast<WaveScratch*>(smem_raw + ((sizeof(Weights<H>) + 15) & ~(size_t)15));
    stage_weights<H>(W, f1w, f1b, f2w, f2b, f3w, f3b);
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveScratch& sc = scs[wave];
    for (unsigned ex = blockIdx.x * kWaves + wave; ex < B; ex += gridDim.x * kWaves) {
        float q[H], k[H];
        load_example<H>(lane, ex, T, query, keys, q, k);
        const float s = mlp_forward<H>(W, sc, lane, q, k);
        const int len = keys_length[ex];
        const float w = attention_weight<H>(s, lane < T && (int)lane < len, lane < T, is_softmax);
        float o[H];
#pragma unroll
        for (int i = 0; i < H; ++i) o[i] = w * k[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int i = 0; i < H; ++i) o[i] += __shfl_xor(o[i], off, 64);
        if (lane < H) {
            float v = o[0];
#pragma unroll
            for (int i = 1; i < H; ++i) v = lane == (unsigned)i ? o[i] : v;
            out[(size_t)ex * H + lane] = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// backward
// per-workgroup partial layout (floats):
//   dW1 [4H][64] | db1 [64] | dW2 [64][32] | db2 [32] | dW3 [32] | db3 [1]
// ---------------------------------------------------------------------------------------------
template <int H>
constexpr int din_partial_floats() { return 4 * H * N1 + N1 + N1 * N2 + N2 + N2 + 1; }

template <int H>
__global__ __launch_bounds__(kThreads) void din_attention_bwd_kernel(
    const float* __restrict__ query, const float* __restrict__ keys, const int32_t* __restrict__ keys_length,
    const float* __restrict__ f1w, const float* __restrict__ f1b, const float* __restrict__ f2w,
    const float* __restrict__ f2b, const float* __restrict__ f3w, const float* __restrict__ f3b,
    const float* __restrict__ g_out, unsigned ldg, const float* __restrict__ dq_extra, unsigned ld_extra, unsigned B,
    unsigned T, int is_softmax, float* __restrict__ dquery, float* __restrict__ dkeys, float* __restrict__ partials) {
    static_assert(2 * H <= 32, "k and q*k must fit one 32-wide MFMA tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Weights<H>& W = *reinterpret_cast<Weights<H>*>(smem_raw);
    WaveScratch* scs = reinterpret_cast<WaveScratch*>(smem_raw + ((sizeof(Weights<H>) + 15) & ~(size_t)15));
    stage_weights<H>(W, f1w, f1b, f2w, f2b, f3w, f3b);
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned hi = lane >> 5, l32 = lane & 31;
    WaveScratch& sc = scs[wave];

    // persistent accumulators: dWx [32 (k|qk) x 64] = 2 tiles, dW2 [64 x 32] = 2 tiles
    f32x16 accX[2], acc2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) accX[t][r] = acc2[t][r] = 0.f;
    float dWq[H];                       // lane j: sum_b q_i * colsum_t(dh1_j)
#pragma unroll
    for (int i = 0; i < H; ++i) dWq[i] = 0.f;
    float db1 = 0.f, db2 = 0.f, dW3 = 0.f, db3 = 0.f;

    for (unsigned ex = blockIdx.x * kWaves + wave; ex < B; ex += gridDim.x * kWaves) {
        float q[H], k[H];
        load_example<H>(lane, ex, T, query, keys, q, k);
        const float s = mlp_forward<H>(W, sc, lane, q, k);
        const int len = keys_length[ex];
        const bool in_T = lane < T, in_len = in_T && (int)lane < len;
        const float w = attention_weight<H>(s, in_len, in_T, is_softmax);
        // ---- attention output backward (lane = row) ----
        // (the query's other gradient, added at the very end: requested here, with the loads of g, not in front of the store)
        const float dq_add = (dq_extra && lane < (unsigned)H) ? dq_extra[(size_t)ex * ld_extra + lane] : 0.f;
        float g[H];
        {
            const float4* gr = reinterpret_cast<const float4*>(g_out + (size_t)ex * ldg);
#pragma unroll
            for (int i = 0; i < H; i += 4) {
                float4 v = gr[i / 4];
                g[i] = v.x; g[i + 1] = v.y; g[i + 2] = v.z; g[i + 3] = v.w;
            }
        }
        // (pins the dq_extra load up here, in the shadow of g's round trip: left alone, the compiler sinks it to its use at
        // the end of the iteration, where one wave per SIMD waits out a whole memory round trip per example — +8 us)
        float dq_pin = dq_add;
        asm volatile("" : "+v"(dq_pin));
        float dwt = 0.f;
#pragma unroll
        for (int i = 0; i < H; ++i) dwt = fmaf(g[i], k[i], dwt);          // d out / d w_t
        float ds;
        if (is_softmax) {
            float dot = wave_sum(w * dwt);
            ds = in_len ? w * (dwt - dot) / sqrtf((float)H) : 0.f;        // only masked-in scores get grad
        } else {
            ds = in_len ? dwt : 0.f;
        }
        sc.v64[1][lane] = ds;
        db3 += wave_sum(ds);
        __builtin_amdgcn_wave_barrier();
        // dW3[n] = sum_t ds_t h2[t][n]   (lane n), then dH2 in place over H2s (lane = row)
        if (lane < N2) {
            float a = 0.f;
#pragma unroll 8
            for (int t = 0; t < 64; ++t) a = fmaf(sc.v64[1][t], sc.H2s[t][lane], a);
            dW3 += a;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n = 0; n < N2; ++n) {
            const float h = sc.H2s[lane][n];
            sc.H2s[lane][n] = h > 0.f ? ds * W.W3[n] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < N2) {
            float cs = 0.f;
#pragma unroll 8
            for (int t = 0; t < 64; ++t) cs += sc.H2s[t][lane];
            db2 += cs;
        }
        // ---- dW2 += H1^T dH2   (A[i][t] = H1[t][i], B[t][n] = dH2[t][n]) ----
#pragma unroll 8
        for (int k0 = 0; k0 < 64; k0 += 2) {
            const float b = sc.H2s[k0 + hi][l32];
            acc2[0] = mfma(sc.H1s[k0 + hi][l32], b, acc2[0]);
            acc2[1] = mfma(sc.H1s[k0 + hi][32 + l32], b, acc2[1]);
        }
        // ---- dH1 = (dH2 W2^T) * [H1 > 0]   (A[t][n] = dH2[t][n], B[n][i] = W2[i][n]) ----
        {
            f32x16 d1[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) d1[mt][nt][r] = 0.f;
#pragma unroll 4
            for (int k0 = 0; k0 < N2; k0 += 2) {
                const float x0 = sc.H2s[l32][k0 + hi], x1 = sc.H2s[32 + l32][k0 + hi];
                const float w0 = W.W2[l32][k0 + hi], w1 = W.W2[32 + l32][k0 + hi];
                d1[0][0] = mfma(x0, w0, d1[0][0]);
                d1[0][1] = mfma(x0, w1, d1[0][1]);
                d1[1][0] = mfma(x1, w0, d1[1][0]);
                d1[1][1] = mfma(x1, w1, d1[1][1]);
            }
            __builtin_amdgcn_wave_barrier();          // dW2 has consumed H1s
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float* p = &sc.H1s[mt * 32 + acc_row(r, hi)][nt * 32 + l32];
                        *p = *p > 0.f ? d1[mt][nt][r] : 0.f;
                    }
        }
        __builtin_amdgcn_wave_barrier();
        // column sums of dH1 (lane = column j): db1, the q-only part of dW1, and d(cq)
        float dcq = 0.f;
#pragma unroll 8
        for (int t = 0; t < 64; ++t) dcq += sc.H1s[t][lane];
        db1 += dcq;
#pragma unroll
        for (int i = 0; i < H; ++i) dWq[i] = fmaf(q[i], dcq, dWq[i]);
        // ---- dWx += X^T dH1   (A[c][t] = X[t][c], B[t][j] = dH1[t][j]) ----
#pragma unroll 8
        for (int k0 = 0; k0 < 64; k0 += 2) {
            const float a = sc.Xs[k0 + hi][l32];
            accX[0] = mfma(a, sc.H1s[k0 + hi][l32], accX[0]);
            accX[1] = mfma(a, sc.H1s[k0 + hi][32 + l32], accX[1]);
        }
        // ---- dX = dH1 Wx^T   (A[t][j] = dH1[t][j], B[j][c] = Wx[c][j]) ----
        {
            f32x16 dx[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dx[mt][r] = 0.f;
#pragma unroll 8
            for (int k0 = 0; k0 < N1; k0 += 2) {
                const float b = W.Wx[l32][k0 + hi];
                dx[0] = mfma(sc.H1s[l32][k0 + hi], b, dx[0]);
                dx[1] = mfma(sc.H1s[32 + l32][k0 + hi], b, dx[1]);
            }
            __builtin_amdgcn_wave_barrier();          // dWx has consumed Xs
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sc.Xs[mt * 32 + acc_row(r, hi)][l32] = dx[mt][r];
        }
        __builtin_amdgcn_wave_barrier();
        // ---- back to lane = row: dk, dq ----
        float dq[H], dk[H];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const float dxk = sc.Xs[lane][i], dxd = sc.Xs[lane][H + i];
            dk[i] = fmaf(w, g[i], fmaf(dxd, q[i], dxk));
            dq[i] = fmaf(dxd, k[i], dcq * W.Wq[i][lane]);        // + this lane's (column j = lane) share of dcq Wq^T
        }
        if (in_T) {
            float4* dkr = reinterpret_cast<float4*>(dkeys + ((size_t)ex * T + lane) * H);
#pragma unroll
            for (int i = 0; i < H; i += 4) dkr[i / 4] = make_float4(dk[i], dk[i + 1], dk[i + 2], dk[i + 3]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int i = 0; i < H; ++i) dq[i] += __shfl_xor(dq[i], off, 64);
        if (lane < H) {
            float v = dq[0];
#pragma unroll
            for (int i = 1; i < H; ++i) v = lane == (unsigned)i ? dq[i] : v;
            v += dq_pin;                                                  // the query's other consumer's gradient (GradJoin)
            dquery[(size_t)ex * H + lane] = v;
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---- workgroup reduction of the weight-gradient partials (fixed wave order) ----
    constexpr int PF = din_partial_floats<H>();
    static_assert(PF * sizeof(float) <= kWaves * sizeof(WaveScratch), "partial row must fit the wave scratch");
    __syncthreads();
    float* red = reinterpret_cast<float*>(scs);     // [PF], reuses the wave scratch area
    for (unsigned wv = 0; wv < kWaves; ++wv) {
        if (wave == wv) {
            auto put = [&](unsigned idx, float v) { red[idx] = (wv == 0 ? 0.f : red[idx]) + v; };
            // dWx tile jt: rows i = acc_row (0..31: k rows 0..H-1, qk rows H..2H-1), col jt*32+l32
            // final dW1 = [dWq ; dWk ; dWq - dWk ; dWqk]   (blocks a, b, c, d of f1's kernel)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int i = acc_row(r, hi);
                    unsigned j = jt * 32 + l32;
                    if (i < H) {                       // k part: block b (+), block c (-)
                        put((1 * H + i) * N1 + j, accX[jt][r]);
                        put((2 * H + i) * N1 + j, -accX[jt][r]);
                    } else if (i < 2 * H) {            // q*k part: block d
                        put((3 * H + (i - H)) * N1 + j, accX[jt][r]);
                    }
                }
            __builtin_amdgcn_wave_barrier();
            // q-only part: lane j holds dWq[i] for column j: block a (+), block c (+)
#pragma unroll
            for (int i = 0; i < H; ++i) {
                put((0 * H + i) * N1 + lane, dWq[i]);
                red[(2 * H + i) * N1 + lane] += dWq[i];
            }
            const unsigned o_b1 = 4 * H * N1, o_w2 = o_b1 + N1, o_b2 = o_w2 + N1 * N2, o_w3 = o_b2 + N2,
                           o_b3 = o_w3 + N2;
            put(o_b1 + lane, db1);
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int i = it * 32 + acc_row(r, hi);
                    put(o_w2 + i * N2 + l32, acc2[it][r]);
                }
            if (lane < N2) {
                put(o_b2 + lane, db2);
                put(o_w3 + lane, dW3);
            }
            if (lane == 0) put(o_b3, db3);
        }
        __syncthreads();
    }
    float* prow = partials + (size_t)blockIdx.x * PF;
    for (unsigned e = threadIdx.x; e < (unsigned)PF; e += kThreads) prow[e] = red[e];
}

// column sums of [nrows][ncols] -> out[ncols]; 64 columns x 4 row slices per workgroup
__global__ __launch_bounds__(256) void din_sum_partials_kernel(const float* __restrict__ partials, unsigned nrows,
                                                               unsigned stride, unsigned ncols,
                                                               float* __restrict__ out) {
    __shared__ float sh[4][64];
    const unsigned cl = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const unsigned col = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (col < ncols) {
#pragma unroll 8
        for (unsigned r = slice; r < nrows; r += 4) acc += partials[(size_t)r * stride + col];
    }
    sh[slice][cl] = acc;
    __syncthreads();
    if (slice == 0 && col < ncols) out[col] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}


// =============================================================================================
// H == 16 (the reference's fixed hidden size, din.py:103): the round-6 kernels.  Same math, same partial-row layout, but the
// three-layer MLP is a chain of TRANSPOSED products so that the accumulator layout of one MFMA is the B-operand layout of the
// next — activations never go through LDS between layers:
//     H1^T [64 j x 32 t] = Wx^T X^T (+ cq)      A = weights (LDS, ds_read_b128), B = this lane's key row k_t / q*k_t (registers)
//     H2^T [32 n x 32 t] = W2^T relu(H1^T)      B = the accumulator registers of the line above: reg r of lane (t, hi) holds row
//                                               acc_row(r, hi), i.e. exactly the K pair (j, j + 4) of one 32x32x2 step
//     dH1^T = W2 dH2^T,  dX^T = Wx dH1^T        the same chain backwards
// Only the two weight-gradient products (dW2 += H1 dH2^T, dWx += X^T dH1: reduction over t) need t in the K position; H1 / dH2 /
// dH1 / X go through one LDS transpose each ([row][t'] tiles, columns permuted t' = (t & 1) * 16 + (t >> 1) so that the K order
// (t = 2u + hi) is contiguous per half wave: ds_read_b128 fragments, and the reduction stops at the tile's last valid row).
// A 32-row tile at a time (T <= 32: one tile — half the matrix work of the 64-row formulation), d(cq) falls out of the dWx
// operand reads as row sums, dk / dq / out need 8 values per lane (the rows of dX^T a half wave holds), the next example's rows
// are requested before the current example's matrix work.  Round 5: one wave per SIMD with ~860 LDS instructions per example
// outside the MFMA loops and 1.3 ds_read_b32 per MFMA inside them (106 us for B = 4096, T = 50; matrix pipe 39 % busy).
// =============================================================================================
namespace din16 {

// scripts/din_lab.hip builds this file with -DDIN16_TIMELINE: shader-clock totals per phase of wave 0 of workgroup 0
#ifdef DIN16_TIMELINE
__device__ unsigned long long din16_tl[32];
#define DIN16_TL_BEGIN() unsigned long long tl_prev_ = clock64()
#define DIN16_TL(i)                                              \
    do {                                                         \
        __builtin_amdgcn_sched_barrier(0);                       \
        const unsigned long long t_ = clock64();                 \
        if (blockIdx.x == 0 && threadIdx.x == 0) din16_tl[i] += t_ - tl_prev_; \
        tl_prev_ = t_;                                           \
        __builtin_amdgcn_sched_barrier(0);                       \
    } while (0)
#define DIN16_TL_ARG , unsigned long long& tl_prev_
#define DIN16_TL_PASS , tl_prev_
#else
#define DIN16_TL_BEGIN()
#define DIN16_TL(i)
#define DIN16_TL_ARG
#define DIN16_TL_PASS
#endif

constexpr int H = 16;
constexpr int SA = 20;                           // row stride of 16-float fragments (ds_read_b128: 8 lanes x 4 banks distinct)
constexpr int SB = 36;                           // row stride of 32-float tiles

struct Weights {
    float WxA[2][2][32][SA];     // [jt][hi][j][s]       = Wx[hi * 16 + s][jt * 32 + j]            layer 1, A operand
    float W2A[2][32][SB];        // [hi][n][jt * 16 + r] = W2[jt * 32 + acc_row(r, hi)][n]         layer 2, A operand
    float W2B[2][2][32][SA];     // [jt][hi][j][r]       = W2[jt * 32 + j][acc_row(r, hi)]         dH1, A operand
    float WxB[2][32][SB];        // [hi][c][jt * 16 + r] = Wx[c][jt * 32 + acc_row(r, hi)]         dX, A operand
    float Wq[H][N1];             // W1a + W1c
    float WqF[2][2][32][8];      // [jt][hi][l32][m] = Wq[i(m, hi)][jt * 32 + l32], i(m, hi) = 8 * (m >> 2) + 4 * hi + (m & 3)   (d cq -> d q)
    float b1[N1];
    float b2[N2];
    float W3[N2];
    float b3[4];
};
struct FwdScratch {              // one per wave
    float cq[N1];
    float qg[2 * H];             // q | g_out row of the current example
};
struct Scratch : FwdScratch {    // one per wave (backward)
    float P[64][SB];             // relu(H1) tile [j][t'], later dH1 [j][t']
    float Q[32][SB];             // dH2 [n][t']
    float R[32][SB];             // X^T [c][t']
};

__device__ __forceinline__ float wx_of(const float* __restrict__ f1w, unsigned c, unsigned j) {
    return c < (unsigned)H ? f1w[(H + c) * N1 + j] - f1w[(2 * H + c) * N1 + j] : f1w[(3 * H + (c - H)) * N1 + j];
}

__device__ __forceinline__ void stage_weights(Weights& S, const float* __restrict__ f1w, const float* __restrict__ f1b,
                                              const float* __restrict__ f2w, const float* __restrict__ f2b,
                                              const float* __restrict__ f3w, const float* __restrict__ f3b) {
    for (unsigned e = threadIdx.x; e < 2 * 2 * 32 * 16; e += kThreads) {
        const unsigned s = e & 15, j = (e >> 4) & 31, hi = (e >> 9) & 1, jt = e >> 10;
        S.WxA[jt][hi][j][s] = wx_of(f1w, hi * 16 + s, jt * 32 + j);
        S.W2B[jt][hi][j][s] = f2w[(jt * 32 + j) * N2 + acc_row((int)s, hi)];
    }
    for (unsigned e = threadIdx.x; e < 2 * 32 * 32; e += kThreads) {
        const unsigned x = e & 31, row = (e >> 5) & 31, hi = e >> 10;
        const unsigned jt = x >> 4, r = x & 15;
        S.W2A[hi][row][x] = f2w[(jt * 32 + acc_row((int)r, hi)) * N2 + row];
        S.WxB[hi][row][x] = wx_of(f1w, row, jt * 32 + acc_row((int)r, hi));
    }
    for (unsigned e = threadIdx.x; e < H * N1; e += kThreads) {
        const unsigned i = e / N1, j = e - i * N1;
        S.Wq[i][j] = f1w[i * N1 + j] + f1w[(2 * H + i) * N1 + j];
    }
    for (unsigned e = threadIdx.x; e < 2 * 2 * 32 * 8; e += kThreads) {
        const unsigned m = e & 7, l = (e >> 3) & 31, hi = (e >> 8) & 1, jt = e >> 9;
        const unsigned i = 8 * (m >> 2) + 4 * hi + (m & 3), j = jt * 32 + l;
        S.WqF[jt][hi][l][m] = f1w[i * N1 + j] + f1w[(2 * H + i) * N1 + j];
    }
    if (threadIdx.x < N1) S.b1[threadIdx.x] = f1b[threadIdx.x];
    if (threadIdx.x < N2) {
        S.b2[threadIdx.x] = f2b[threadIdx.x];
        S.W3[threadIdx.x] = f3w[threadIdx.x];
    }
    if (threadIdx.x == 0) S.b3[0] = f3b[0];
}

__device__ __forceinline__ float4 lds4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float f4_at(const float4& v, int x) { return x == 0 ? v.x : (x == 1 ? v.y : (x == 2 ? v.z : v.w)); }
// column of row t inside a transposed tile (see above)
__device__ __forceinline__ unsigned tcol(unsigned l32) { return (l32 & 1) * 16 + (l32 >> 1); }

struct Lane {                    // lane coordinates (per-lane weight constants are read from LDS where they are used: registers are
    unsigned lane, hi, l32;      // the scarce resource of the backward kernel)
};
// W3 / b2 at the rows acc_row(r, hi) this lane's H2^T registers hold: four contiguous runs of four
__device__ __forceinline__ void sel16(const float* __restrict__ v32, unsigned hi, float (&o)[16]) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const float4 c = *reinterpret_cast<const float4*>(v32 + 8 * g4 + 4 * hi);
        o[4 * g4] = c.x; o[4 * g4 + 1] = c.y; o[4 * g4 + 2] = c.z; o[4 * g4 + 3] = c.w;
    }
}
// the 8 hidden indices a half wave's dX^T registers cover: m = 4 * bl + x  ->  i = 8 * bl + 4 * hi + x
#define DIN16_SEL(arr, bl, x, hi) ((hi) ? (arr)[8 * (bl) + 4 + (x)] : (arr)[8 * (bl) + (x)])

__device__ __forceinline__ void init_lane(Lane& L, const Weights& W) {
    L.lane = threadIdx.x & 63; L.hi = L.lane >> 5; L.l32 = L.lane & 31;
}

// the lane's key row of tile `tile` (zero beyond T); both half waves hold the same rows
__device__ __forceinline__ void load_row(const Lane& L, unsigned ex, unsigned tile, unsigned T, const float* __restrict__ keys,
                                         float (&k)[H]) {
    const unsigned t = tile * 32 + L.l32;
    if (t < T) {
        const float4* kr = reinterpret_cast<const float4*>(keys + ((size_t)ex * T + t) * H);
#pragma unroll
        for (int i = 0; i < H; i += 4) {
            const float4 v = kr[i / 4];
            k[i] = v.x; k[i + 1] = v.y; k[i + 2] = v.z; k[i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < H; ++i) k[i] = 0.f;
    }
}
// 16 wave-uniform floats out of the wave's scratch (broadcast ds_read_b128)
__device__ __forceinline__ void lds_vec16(const float* p, float (&v)[H]) {
#pragma unroll
    for (int i = 0; i < H; i += 4) {
        const float4 x = lds4(p + i);
        v[i] = x.x; v[i + 1] = x.y; v[i + 2] = x.z; v[i + 3] = x.w;
    }
}

// per example: q (and the upstream gradient g) into the wave's scratch — one coalesced load, read back as broadcasts where they
// are used (32 registers fewer than holding them) — and cq[j] = b1[j] + sum_i q_i (W1a + W1c)[i][j] (lane = j)
// lane i < 16: q[i], 16 <= i < 32: g[i - 16] of an example (one coalesced load; 0 elsewhere / without a gradient row)
__device__ __forceinline__ float load_qg(const Lane& L, const float* __restrict__ qrow, const float* __restrict__ grow) {
    if (L.lane < (unsigned)H) return qrow[L.lane];
    if (L.lane < 2u * H && grow != nullptr) return grow[L.lane - H];
    return 0.f;
}
__device__ __forceinline__ void begin_example(const Lane& L, const Weights& W, FwdScratch& sc, float qg) {
    if (L.lane < 2u * H) sc.qg[L.lane] = qg;
    __builtin_amdgcn_wave_barrier();
    float c = W.b1[L.lane], q[H];
    lds_vec16(sc.qg, q);
#pragma unroll
    for (int i = 0; i < H; ++i) c = fmaf(q[i], W.Wq[i][L.lane], c);
    sc.cq[L.lane] = c;
    __builtin_amdgcn_wave_barrier();
}

// forward of one 32-row tile: a1[jt] = H1^T (pre-ReLU), a2 = H2^T (pre-ReLU), returns the raw score of row t = tile * 32 + l32
__device__ __forceinline__ float fwd_tile(const Lane& L, const Weights& W, const FwdScratch& sc, const float (&k)[H],
                                          f32x16 (&a1)[2], f32x16& a2, float (&xb)[H]) {
    {
        float q[H];
        lds_vec16(sc.qg, q);
#pragma unroll
        for (int s = 0; s < H; ++s) xb[s] = L.hi ? q[s] * k[s] : k[s];
    }
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 c = lds4(&sc.cq[jt * 32 + 8 * g4 + 4 * L.hi]);
            a1[jt][4 * g4] = c.x; a1[jt][4 * g4 + 1] = c.y; a1[jt][4 * g4 + 2] = c.z; a1[jt][4 * g4 + 3] = c.w;
        }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float4 w0 = lds4(&W.WxA[0][L.hi][L.l32][4 * v]), w1 = lds4(&W.WxA[1][L.hi][L.l32][4 * v]);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            a1[0] = mfma(f4_at(w0, x), xb[4 * v + x], a1[0]);
            a1[1] = mfma(f4_at(w1, x), xb[4 * v + x], a1[1]);
        }
    }
    f32x16 a2b;
    {
        float b2s[16];
        sel16(W.b2, L.hi, b2s);
#pragma unroll
        for (int r = 0; r < 16; ++r) { a2[r] = b2s[r]; a2b[r] = 0.f; }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float4 w0 = lds4(&W.W2A[L.hi][L.l32][4 * v]), w1 = lds4(&W.W2A[L.hi][L.l32][16 + 4 * v]);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            a2 = mfma(f4_at(w0, x), fmaxf(a1[0][4 * v + x], 0.f), a2);
            a2b = mfma(f4_at(w1, x), fmaxf(a1[1][4 * v + x], 0.f), a2b);
        }
    }
    a2 += a2b;
    float sp = 0.f, w3s[16];
    sel16(W.W3, L.hi, w3s);
#pragma unroll
    for (int r = 0; r < 16; ++r) sp = fmaf(fmaxf(a2[r], 0.f), w3s[r], sp);
    return W.b3[0] + (sp + __shfl_xor(sp, 32, 64));
}

// sum over the 32 lanes of a half wave (every lane of the half gets the total)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct AttnW { float w[2], ds[2]; };
// attention weights (and, for the backward, d loss / d score) of the lane's two rows from their raw scores
template <bool BWD>
__device__ __forceinline__ AttnW attention(const Lane& L, const float (&s)[2], unsigned T, int len, int is_softmax,
                                           const float (&dwt)[2]) {
    AttnW a;
    bool in_T[2], in_len[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const unsigned t = tt * 32 + L.l32;
        in_T[tt] = t < T;
        in_len[tt] = in_T[tt] && (int)t < len;
    }
    if (is_softmax) {
        const float rs = 1.0f / sqrtf((float)H);
        float v[2], e[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) v[tt] = (in_len[tt] ? s[tt] : kPadScore) / sqrtf((float)H);      // mask, then scale (:32-34)
        const float mx = wave_max(fmaxf(in_T[0] ? v[0] : -INFINITY, in_T[1] ? v[1] : -INFINITY));
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) e[tt] = in_T[tt] ? expf(v[tt] - mx) : 0.f;
        const float den = wave_sum(L.hi == 0 ? e[0] + e[1] : 0.f);                                        // (both half waves hold every row)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) a.w[tt] = e[tt] / den;
        if (BWD) {
            const float dot = wave_sum(L.hi == 0 ? fmaf(a.w[0], dwt[0], a.w[1] * dwt[1]) : 0.f);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) a.ds[tt] = in_len[tt] ? a.w[tt] * (dwt[tt] - dot) * rs : 0.f;     // only masked-in scores get grad
        }
    } else {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            a.w[tt] = in_len[tt] ? s[tt] : 0.f;                                                            // s * mask (:37-38)
            if (BWD) a.ds[tt] = in_len[tt] ? dwt[tt] : 0.f;
        }
    }
    return a;
}

// (the forward needs cq / qg of the scratch only: two workgroups per CU, i.e. two waves per SIMD — one wave's layer-3 / attention
// / reduction phases run under the other's MFMAs — if the kernel stays within 256 registers)
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void fwd_kernel(
    const float* __restrict__ query, const float* __restrict__ keys, const int32_t* __restrict__ keys_length,
    const float* __restrict__ f1w, const float* __restrict__ f1b, const float* __restrict__ f2w,
    const float* __restrict__ f2b, const float* __restrict__ f3w, const float* __restrict__ f3b, unsigned B,
    unsigned T, int is_softmax, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Weights& W = *reinterpret_cast<Weights*>(smem_raw);
    FwdScratch* scs = reinterpret_cast<FwdScratch*>(smem_raw + ((sizeof(Weights) + 15) & ~(size_t)15));
    stage_weights(W, f1w, f1b, f2w, f2b, f3w, f3b);
    __syncthreads();
    Lane L;
    init_lane(L, W);
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    FwdScratch& sc = scs[wave];
    const unsigned ntile = T > 32 ? 2 : 1;
    const unsigned stride = gridDim.x * kWaves;
    // (no software prefetch of the next example's rows here: the co-resident wave of the SIMD covers the load latency, and the
    // 32 registers it would take are what keeps the kernel at two waves per SIMD)
    for (unsigned ex = blockIdx.x * kWaves + wave; ex < B; ex += stride) {
        float k[2][H];
        load_row(L, ex, 0, T, keys, k[0]);
        load_row(L, ex, 1, T, keys, k[1]);
        const int len = keys_length[ex];
        begin_example(L, W, sc, load_qg(L, query + (size_t)ex * H, nullptr));
        float s[2] = {0.f, 0.f}, xb[H];
        f32x16 a1[2], a2;
        // (default branch: a tile without a row inside the history's length only produces weights that are masked to 0)
        if (is_softmax || len > 0) s[0] = fwd_tile(L, W, sc, k[0], a1, a2, xb);
        if (ntile > 1 && (is_softmax || len > 32)) s[1] = fwd_tile(L, W, sc, k[1], a1, a2, xb);
        const float dw0[2] = {0.f, 0.f};
        const AttnW a = attention<false>(L, s, T, len, is_softmax, dw0);
        float o[8];
#pragma unroll
        for (int bl = 0; bl < 2; ++bl)
#pragma unroll
            for (int x = 0; x < 4; ++x)
                o[4 * bl + x] = half_sum(fmaf(a.w[0], DIN16_SEL(k[0], bl, x, L.hi), a.w[1] * DIN16_SEL(k[1], bl, x, L.hi)));
        if (L.l32 == 0) {
            float4* orow = reinterpret_cast<float4*>(out + (size_t)ex * H + 4 * L.hi);
            orow[0] = make_float4(o[0], o[1], o[2], o[3]);
            orow[2] = make_float4(o[4], o[5], o[6], o[7]);
        }
        __builtin_amdgcn_wave_barrier();               // (q / cq of the next example are written after this one's reads)
    }
}

struct BwdAcc {                  // weight-gradient accumulators of a wave (registers, across its examples)
    f32x16 accX[2], acc2[2];     // dWx [32 c x 64 j], dW2 [64 j x 32 n]
    float dWq[H];                // lane j: sum_b q_i d cq_j
    float dW3[16], db2[16];      // partial over the lane's rows, at n = acc_row(r, hi)
    float db1, db3;
};

// backward of one 32-row tile (forward state a1 / a2 / xb of THIS tile in registers)
__device__ __forceinline__ void bwd_tile(const Lane& L, const Weights& W, Scratch& sc, BwdAcc& A, unsigned tile, unsigned T, unsigned Tv,
                                         unsigned ex, const float (&k)[H], const f32x16 (&a1)[2], const f32x16& a2,
                                         const float (&xb)[H], float w, float ds, float (&dqacc)[8], float* __restrict__ dkeys DIN16_TL_ARG) {
    const unsigned col = tcol(L.l32);
    // rows of this tile that can carry a non-zero gradient: inside T, and — default branch — inside the example's history
    // (Tv = min(T, length): a row beyond it has weight 0 and d score 0, hence dH2 = dH1 = 0)
    const unsigned valid = Tv <= tile * 32 ? 0u : (Tv - tile * 32 < 32 ? Tv - tile * 32 : 32u);
    const unsigned u_end = (valid + 1) / 2;                             // K steps (t = 2u + hi) that can carry non-zero rows
    // ---- layer 3 / 2 gradients in registers: dW3, dH2^T, db2 ----
    f32x16 d2;
    {
        float w3s[16];
        sel16(W.W3, L.hi, w3s);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float h = a2[r];
            A.dW3[r] = fmaf(ds, fmaxf(h, 0.f), A.dW3[r]);
            d2[r] = h > 0.f ? ds * w3s[r] : 0.f;
            A.db2[r] += d2[r];
        }
    }
    if (L.hi == 0) A.db3 += ds;
    // ---- transposes for dW2: relu(H1) -> P[j][t'], dH2 -> Q[n][t'] ----
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc.P[jt * 32 + acc_row(r, L.hi)][col] = fmaxf(a1[jt][r], 0.f);
#pragma unroll
    for (int r = 0; r < 16; ++r) sc.Q[acc_row(r, L.hi)][col] = d2[r];
    DIN16_TL(3);
    // ---- dH1^T = W2 dH2^T (chained: B = d2 registers) ----
    f32x16 d1[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) d1[jt][r] = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float4 w0 = lds4(&W.W2B[0][L.hi][L.l32][4 * v]), w1 = lds4(&W.W2B[1][L.hi][L.l32][4 * v]);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            d1[0] = mfma(f4_at(w0, x), d2[4 * v + x], d1[0]);
            d1[1] = mfma(f4_at(w1, x), d2[4 * v + x], d1[1]);
        }
    }
    __builtin_amdgcn_wave_barrier();
    DIN16_TL(4);
    // ---- dW2 += H1 dH2^T over the tile's rows: A[j][t] from P, B[t][n] from Q ----
    for (unsigned v = 0; 4 * v < u_end; ++v) {
        const float4 b = lds4(&sc.Q[L.l32][L.hi * 16 + 4 * v]);
        const float4 p0 = lds4(&sc.P[L.l32][L.hi * 16 + 4 * v]), p1 = lds4(&sc.P[32 + L.l32][L.hi * 16 + 4 * v]);
        // (K steps past the last valid row multiply zeros: whole groups of four are skipped, the rest is not worth a branch)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            A.acc2[0] = mfma(f4_at(p0, x), f4_at(b, x), A.acc2[0]);
            A.acc2[1] = mfma(f4_at(p1, x), f4_at(b, x), A.acc2[1]);
        }
    }
    DIN16_TL(5);
    // ---- ReLU mask of layer 1, then dH1 -> P[j][t'] and X^T -> R[c][t'] ----
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) d1[jt][r] = a1[jt][r] > 0.f ? d1[jt][r] : 0.f;
    __builtin_amdgcn_wave_barrier();                   // (the dW2 reads of P are issued before these writes)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc.P[jt * 32 + acc_row(r, L.hi)][col] = d1[jt][r];
#pragma unroll
    for (int s = 0; s < H; ++s) sc.R[L.hi * 16 + s][col] = xb[s];
    DIN16_TL(6);
    // ---- dX^T = Wx dH1^T (chained: B = d1 registers), two chains ----
    f32x16 dx, dxb;
#pragma unroll
    for (int r = 0; r < 16; ++r) dx[r] = dxb[r] = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float4 w0 = lds4(&W.WxB[L.hi][L.l32][4 * v]), w1 = lds4(&W.WxB[L.hi][L.l32][16 + 4 * v]);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            dx = mfma(f4_at(w0, x), d1[0][4 * v + x], dx);
            dxb = mfma(f4_at(w1, x), d1[1][4 * v + x], dxb);
        }
    }
    dx += dxb;
    __builtin_amdgcn_wave_barrier();
    DIN16_TL(7);
    // ---- dWx += X^T dH1 over the tile's rows: A[c][t] from R, B[t][j] from P; the row sums of dH1 are d(cq) ----
    float rs0 = 0.f, rs1 = 0.f;
    for (unsigned v = 0; 4 * v < u_end; ++v) {
        const float4 a = lds4(&sc.R[L.l32][L.hi * 16 + 4 * v]);
        const float4 p0 = lds4(&sc.P[L.l32][L.hi * 16 + 4 * v]), p1 = lds4(&sc.P[32 + L.l32][L.hi * 16 + 4 * v]);
        rs0 += (p0.x + p0.y) + (p0.z + p0.w);
        rs1 += (p1.x + p1.y) + (p1.z + p1.w);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            A.accX[0] = mfma(f4_at(a, x), f4_at(p0, x), A.accX[0]);
            A.accX[1] = mfma(f4_at(a, x), f4_at(p1, x), A.accX[1]);
        }
    }
    DIN16_TL(8);
    const float dcq0 = rs0 + __shfl_xor(rs0, 32, 64), dcq1 = rs1 + __shfl_xor(rs1, 32, 64);   // column j = l32 / 32 + l32
    const float dcq_own = L.hi ? dcq1 : dcq0;                                                    // column j = lane
    A.db1 += dcq_own;
    float q[H], g[H];
    lds_vec16(sc.qg, q);
    lds_vec16(sc.qg + H, g);
#pragma unroll
    for (int i = 0; i < H; ++i) A.dWq[i] = fmaf(q[i], dcq_own, A.dWq[i]);
    // ---- dk (this lane's 8 columns of its row), dq partials ----
    const unsigned t = tile * 32 + L.l32;
    float wqf[2][8];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int bl = 0; bl < 2; ++bl) {
            const float4 c = lds4(&W.WqF[jt][L.hi][L.l32][4 * bl]);
            wqf[jt][4 * bl] = c.x; wqf[jt][4 * bl + 1] = c.y; wqf[jt][4 * bl + 2] = c.z; wqf[jt][4 * bl + 3] = c.w;
        }
#pragma unroll
    for (int bl = 0; bl < 2; ++bl) {
        float dk[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const float dxk = dx[4 * bl + x], dxd = dx[8 + 4 * bl + x];
            const float qi = DIN16_SEL(q, bl, x, L.hi), ki = DIN16_SEL(k, bl, x, L.hi), gi = DIN16_SEL(g, bl, x, L.hi);
            dk[x] = fmaf(w, gi, fmaf(dxd, qi, dxk));
            float acc = fmaf(dxd, ki, dqacc[4 * bl + x]);
            acc = fmaf(dcq0, wqf[0][4 * bl + x], acc);                // this lane's columns j = l32, 32 + l32 of d(cq) Wq^T
            dqacc[4 * bl + x] = fmaf(dcq1, wqf[1][4 * bl + x], acc);
        }
        if (t < T)
            *reinterpret_cast<float4*>(dkeys + ((size_t)ex * T + t) * H + 8 * bl + 4 * L.hi) = make_float4(dk[0], dk[1], dk[2], dk[3]);
    }
    __builtin_amdgcn_wave_barrier();                   // (P / Q / R are rewritten by the next tile)
    DIN16_TL(9);
}

template <bool SOFTMAX>
__global__ __launch_bounds__(kThreads) void bwd_kernel(
    const float* __restrict__ query, const float* __restrict__ keys, const int32_t* __restrict__ keys_length,
    const float* __restrict__ f1w, const float* __restrict__ f1b, const float* __restrict__ f2w,
    const float* __restrict__ f2b, const float* __restrict__ f3w, const float* __restrict__ f3b,
    const float* __restrict__ g_out, unsigned ldg, const float* __restrict__ dq_extra, unsigned ld_extra, unsigned B,
    unsigned T, float* __restrict__ dquery, float* __restrict__ dkeys, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Weights& W = *reinterpret_cast<Weights*>(smem_raw);
    Scratch* scs = reinterpret_cast<Scratch*>(smem_raw + ((sizeof(Weights) + 15) & ~(size_t)15));
    stage_weights(W, f1w, f1b, f2w, f2b, f3w, f3b);
    __syncthreads();
    Lane L;
    init_lane(L, W);
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Scratch& sc = scs[wave];
    const unsigned ntile = T > 32 ? 2 : 1;
    const unsigned stride = gridDim.x * kWaves;
    BwdAcc A;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) A.accX[t][r] = A.acc2[t][r] = 0.f;
#pragma unroll
    for (int i = 0; i < H; ++i) A.dWq[i] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) A.dW3[r] = A.db2[r] = 0.f;
    A.db1 = A.db3 = 0.f;

    // the wave walks (example, tile) pairs; the key row of the NEXT pair is requested before the current pair's matrix work
    unsigned ex = blockIdx.x * kWaves + wave;
    float kc[H], qg = 0.f;
    if (ex < B) {
        load_row(L, ex, 0, T, keys, kc);
        qg = load_qg(L, query + (size_t)ex * H, g_out + (size_t)ex * ldg);
    }
    DIN16_TL_BEGIN();
    DIN16_TL(12);
    for (; ex < B; ex += stride) {
        begin_example(L, W, sc, qg);
        // (the next example's q / g row is requested now, a whole example of matrix work ahead of its use)
        if (ex + stride < B) qg = load_qg(L, query + (size_t)(ex + stride) * H, g_out + (size_t)(ex + stride) * ldg);
        DIN16_TL(0);
        // (the query's other gradient, added at the very end: requested here, not in front of the store)
        float dq_add[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) dq_add[m] = 0.f;
        if (dq_extra && L.l32 == 0) {
#pragma unroll
            for (int bl = 0; bl < 2; ++bl)
#pragma unroll
                for (int x = 0; x < 4; ++x) dq_add[4 * bl + x] = dq_extra[(size_t)ex * ld_extra + 8 * bl + 4 * L.hi + x];
        }
        const int len = keys_length[ex];
        float dqacc[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) dqacc[m] = 0.f;
        AttnW aw;
        if (SOFTMAX) {
            // the weights of all rows come first (a softmax over t): the scores of both tiles, then each tile's forward again
            float s[2] = {0.f, 0.f}, dwt[2] = {0.f, 0.f}, g[H];
            lds_vec16(sc.qg + H, g);
            for (unsigned tile = 0; tile < ntile; ++tile) {
                float kt[H], xb[H];
                f32x16 a1[2], a2;
                if (tile == 0) {
#pragma unroll
                    for (int i = 0; i < H; ++i) kt[i] = kc[i];
                } else {
                    load_row(L, ex, tile, T, keys, kt);
                }
                const float sv = fwd_tile(L, W, sc, kt, a1, a2, xb);
                float d = 0.f;
#pragma unroll
                for (int i = 0; i < H; ++i) d = fmaf(g[i], kt[i], d);
                if (tile == 0) { s[0] = sv; dwt[0] = d; } else { s[1] = sv; dwt[1] = d; }
            }
            aw = attention<true>(L, s, T, len, 1, dwt);
        }
        // default branch: a row at or beyond the history's length has weight 0 AND d score 0 — it contributes exactly nothing to
        // any gradient and its own dk is 0: tiles that hold only such rows are skipped (their dk rows are written as zeros).
        // Histories of <= 32 items cost one tile, empty ones none.  (Softmax: a padded row's weight is exp(-2^32 / 4 - max), not
        // structurally zero when the history is empty — din_attention.py:31-35 — so that branch walks every tile.)
        const unsigned Tv = SOFTMAX ? T : (len <= 0 ? 0u : ((unsigned)len < T ? (unsigned)len : T));
        const unsigned nt_ex = SOFTMAX ? ntile : (Tv + 31) / 32;
        if (nt_ex == 0 && ex + stride < B) load_row(L, ex + stride, 0, T, keys, kc);      // (nothing to overlap the request with)
        for (unsigned tile = 0; tile < nt_ex; ++tile) {
            float kn[H];
            const bool last = tile + 1 == nt_ex;
            const unsigned nex = last ? ex + stride : ex, ntl = last ? 0 : tile + 1;
            if (nex < B) load_row(L, nex, ntl, T, keys, kn);
            f32x16 a1[2], a2;
            float xb[H];
            DIN16_TL(1);
            const float sv = fwd_tile(L, W, sc, kc, a1, a2, xb);
            DIN16_TL(2);
            float w, ds;
            if (SOFTMAX) {
                w = tile ? aw.w[1] : aw.w[0];
                ds = tile ? aw.ds[1] : aw.ds[0];
            } else {
                // default branch: w_t = s_t * mask, d s_t = mask * <g, k_t> (d out / d w_t)
                const unsigned t = tile * 32 + L.l32;
                const bool in_len = t < T && (int)t < len;
                float g[H], d = 0.f;
                lds_vec16(sc.qg + H, g);
#pragma unroll
                for (int i = 0; i < H; ++i) d = fmaf(g[i], kc[i], d);
                w = in_len ? sv : 0.f;
                ds = in_len ? d : 0.f;
            }
            bwd_tile(L, W, sc, A, tile, T, Tv, ex, kc, a1, a2, xb, w, ds, dqacc, dkeys DIN16_TL_PASS);
            if (nex < B) {
#pragma unroll
                for (int i = 0; i < H; ++i) kc[i] = kn[i];
            }
        }
        for (unsigned tile = nt_ex; tile < ntile; ++tile) {            // skipped tiles: dk = 0
            const unsigned t = tile * 32 + L.l32;
            if (t < T) {
                float4* dkr = reinterpret_cast<float4*>(dkeys + ((size_t)ex * T + t) * H + 4 * L.hi);
                dkr[0] = f4_zero();
                dkr[2] = f4_zero();
            }
        }
        // ---- dq: the half wave's 8 columns, summed over its 32 lanes (rows t and columns j = l32, 32 + l32 of the fold) ----
#pragma unroll
        for (int m = 0; m < 8; ++m) dqacc[m] = half_sum(dqacc[m]);
        if (L.l32 == 0) {
            float4* dqr = reinterpret_cast<float4*>(dquery + (size_t)ex * H + 4 * L.hi);
            dqr[0] = make_float4(dqacc[0] + dq_add[0], dqacc[1] + dq_add[1], dqacc[2] + dq_add[2], dqacc[3] + dq_add[3]);
            dqr[2] = make_float4(dqacc[4] + dq_add[4], dqacc[5] + dq_add[5], dqacc[6] + dq_add[6], dqacc[7] + dq_add[7]);
        }
        DIN16_TL(10);
    }

    // ---- the lane-partial vectors: sums over the half wave's lanes (rows) ----
#pragma unroll
    for (int r = 0; r < 16; ++r) { A.dW3[r] = half_sum(A.dW3[r]); A.db2[r] = half_sum(A.db2[r]); }
    A.db3 = wave_sum(A.db3);
    // ---- workgroup reduction of the weight-gradient partials: every wave stores its accumulators into its OWN region of
    // the (now free) dynamic LDS — plain stores, all four waves at once — and the 256 threads then add the four copies in
    // wave order while they write the partial row, laid out as the caller's six buffers.  (The round-2 form, one wave after
    // the other read-modify-writing one shared row, was a chain of ~150 dependent LDS round trips per wave: 20 % of the
    // kernel, profiles/r06_din_lab.md.)
    constexpr int PF = din_partial_floats<H>();
    constexpr unsigned kRegion = 3 * H * N1 + N1 + N1 * N2 + N2 + N2 + 4;     // dWk | dWqk | dWq | db1 | dW2 | db2 | dW3 | db3
    constexpr unsigned r_wk = 0, r_wqk = H * N1, r_wq = 2 * H * N1, r_b1 = 3 * H * N1, r_w2 = r_b1 + N1, r_b2 = r_w2 + N1 * N2,
                       r_w3 = r_b2 + N2, r_b3 = r_w3 + N2;
    static_assert(kWaves * kRegion * sizeof(float) <= sizeof(Weights) + kWaves * sizeof(Scratch), "the four regions must fit the dynamic LDS");
    __syncthreads();                                   // (weights and scratch are dead from here on)
    float* regions = reinterpret_cast<float*>(smem_raw);
    {
        float* reg = regions + wave * kRegion;
        const unsigned hi = L.hi, l32 = L.l32, lane = L.lane;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // rows c = acc_row(r, hi) of dWx: r < 8 -> the k part (c < 16), else the q*k part (c - 16)
                const unsigned c = acc_row(r, hi), j = jt * 32 + l32;
                if (r < 8) reg[r_wk + c * N1 + j] = A.accX[jt][r];
                else reg[r_wqk + (c - H) * N1 + j] = A.accX[jt][r];
            }
#pragma unroll
        for (int i = 0; i < H; ++i) reg[r_wq + i * N1 + lane] = A.dWq[i];
        reg[r_b1 + lane] = A.db1;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) reg[r_w2 + (it * 32 + acc_row(r, hi)) * N2 + l32] = A.acc2[it][r];
        if (l32 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                reg[r_b2 + acc_row(r, hi)] = A.db2[r];
                reg[r_w3 + acc_row(r, hi)] = A.dW3[r];
            }
        }
        if (lane == 0) reg[r_b3] = A.db3;
    }
    __syncthreads();
    float* prow = partials + (size_t)blockIdx.x * PF;
    for (unsigned e = threadIdx.x; e < (unsigned)PF; e += kThreads) {
        // final dW1 = [dWq ; dWk ; dWq - dWk ; dWqk]   (blocks a, b, c, d of f1's kernel), then db1 | dW2 | db2 | dW3 | db3
        unsigned src, src2 = ~0u;
        if (e < 4u * H * N1) {
            const unsigned blk = e / (H * N1), rem = e - blk * (H * N1);
            src = (blk == 0 ? r_wq : (blk == 1 ? r_wk : (blk == 2 ? r_wq : r_wqk))) + rem;
            if (blk == 2) src2 = r_wk + rem;
        } else {
            src = r_b1 + (e - 4u * H * N1);            // db1 | dW2 | db2 | dW3 | db3 are contiguous in both layouts
        }
        float v = 0.f;
#pragma unroll
        for (unsigned wv = 0; wv < kWaves; ++wv) {
            const float* reg = regions + wv * kRegion;
            v += src2 == ~0u ? reg[src] : reg[src] - reg[src2];
        }
        prow[e] = v;
    }
    DIN16_TL(11);
}

inline size_t smem_bytes() { return ((sizeof(Weights) + 15) & ~(size_t)15) + (size_t)kWaves * sizeof(Scratch); }
inline size_t fwd_smem_bytes() { return ((sizeof(Weights) + 15) & ~(size_t)15) + (size_t)kWaves * sizeof(FwdScratch); }
inline int fwd_grid(int B) {      // two workgroups per CU
    const int need = cdiv(B, kWaves);
    return need < 1 ? 1 : (need > 512 ? 512 : need);
}

}  // namespace din16

inline int din_grid(int B) {
    int need = cdiv(B, kWaves);
    return need < 1 ? 1 : (need > 256 ? 256 : need);
}

template <int H>
size_t din_smem() { return ((sizeof(Weights<H>) + 15) & ~(size_t)15) + (size_t)kWaves * sizeof(WaveScratch); }

}  // namespace

RECALGO_EXPORT int recalgo_din_attention_fwd(const float* query, const float* keys, const int32_t* keys_length,
                                             const float* f1_w, const float* f1_b, const float* f2_w,
                                             const float* f2_b, const float* f3_w, const float* f3_b, int B, int T,
                                             int H, int is_softmax, float* out, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && T >= 1 && T <= 64 && (H == 4 || H == 8 || H == 16));
    if (B == 0) return 0;
    hipStream_t st = as_stream(stream);
    if (H == 16 && (reinterpret_cast<uintptr_t>(query) & 15) == 0 && (reinterpret_cast<uintptr_t>(keys) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        const size_t smem = din16::fwd_smem_bytes();
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&din16::fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(din16::fwd_kernel, dim3(din16::fwd_grid(B)), dim3(kThreads), smem, st, query, keys, keys_length, f1_w, f1_b, f2_w, f2_b,
                           f3_w, f3_b, (unsigned)B, (unsigned)T, is_softmax, out);
        RECALGO_RETURN_LAST();
    }
#define LAUNCH(HH)                                                                                          \
    do {                                                                                                    \
        size_t smem = din_smem<HH>();                                                                       \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&din_attention_fwd_kernel<HH>),    \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);          \
        if (e != hipSuccess) return (int)e;                                                                 \
        hipLaunchKernelGGL(din_attention_fwd_kernel<HH>, dim3(din_grid(B)), dim3(kThreads), smem, st, query, keys, \
                           keys_length, f1_w, f1_b, f2_w, f2_b, f3_w, f3_b, (unsigned)B, (unsigned)T, is_softmax,  \
                           out);                                                                            \
    } while (0)
    if (H == 4) LAUNCH(4); else if (H == 8) LAUNCH(8); else LAUNCH(16);
#undef LAUNCH
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int64_t recalgo_din_attention_bwd_workspace_bytes(int B, int T, int H) {
    if (B <= 0 || !(H == 4 || H == 8 || H == 16)) return 0;
    (void)T;
    int pf = H == 4 ? din_partial_floats<4>() : (H == 8 ? din_partial_floats<8>() : din_partial_floats<16>());
    return (int64_t)din_grid(B) * pf * (int64_t)sizeof(float);
}

RECALGO_EXPORT int recalgo_din_attention_bwd_partial_rows(int B) { return B > 0 ? din_grid(B) : 0; }
RECALGO_EXPORT int recalgo_din_attention_bwd_partial_floats(int H) {
    return H == 4 ? din_partial_floats<4>() : (H == 8 ? din_partial_floats<8>() : (H == 16 ? din_partial_floats<16>() : 0));
}

RECALGO_EXPORT int recalgo_din_attention_bwd(const float* query, const float* keys, const int32_t* keys_length,
                                             const float* f1_w, const float* f1_b, const float* f2_w,
                                             const float* f2_b, const float* f3_w, const float* f3_b,
                                             const float* g_out, int B, int T, int H, int is_softmax,
                                             float* dquery, float* dkeys, float* d_f1_w, float* d_f1_b,
                                             float* d_f2_w, float* d_f2_b, float* d_f3_w, float* d_f3_b,
                                             void* workspace, recalgo_stream_t stream) {
    return recalgo_din_attention_bwd_joined(query, keys, keys_length, f1_w, f1_b, f2_w, f2_b, f3_w, f3_b, g_out, H, nullptr, 0, B,
                                            T, H, is_softmax, dquery, dkeys, d_f1_w, d_f1_b, d_f2_w, d_f2_b, d_f3_w, d_f3_b,
                                            workspace, stream);
}

RECALGO_EXPORT int recalgo_din_attention_bwd_joined(const float* query, const float* keys, const int32_t* keys_length,
                                                    const float* f1_w, const float* f1_b, const float* f2_w,
                                                    const float* f2_b, const float* f3_w, const float* f3_b,
                                                    const float* g_out, int ldg, const float* dq_extra, int ld_extra, int B,
                                                    int T, int H, int is_softmax, float* dquery, float* dkeys, float* d_f1_w,
                                                    float* d_f1_b, float* d_f2_w, float* d_f2_b, float* d_f3_w, float* d_f3_b,
                                                    void* workspace, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B > 0 && T >= 1 && T <= 64 && (H == 4 || H == 8 || H == 16) && workspace != nullptr);
    RECALGO_REQUIRE(g_out != nullptr && ldg >= H && ldg % 4 == 0 && (reinterpret_cast<uintptr_t>(g_out) & 15) == 0);
    RECALGO_REQUIRE(dq_extra == nullptr || ld_extra >= H);
    hipStream_t st = as_stream(stream);
    float* partials = static_cast<float*>(workspace);
    const int grid = din_grid(B);
    int pf;
#define LAUNCH(HH)                                                                                            \
    do {                                                                                                      \
        pf = din_partial_floats<HH>();                                                                        \
        size_t smem = din_smem<HH>();                                                                     \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&din_attention_bwd_kernel<HH>),      \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);            \
        if (e != hipSuccess) return (int)e;                                                                   \
        hipLaunchKernelGGL(din_attention_bwd_kernel<HH>, dim3(grid), dim3(kThreads), smem, st, query, keys,   \
                           keys_length, f1_w, f1_b, f2_w, f2_b, f3_w, f3_b, g_out, (unsigned)ldg, dq_extra,       \
                           (unsigned)ld_extra, (unsigned)B, (unsigned)T, is_softmax, dquery, dkeys, partials);    \
    } while (0)
    const bool v16 = H == 16 && (reinterpret_cast<uintptr_t>(query) & 15) == 0 && (reinterpret_cast<uintptr_t>(keys) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(dquery) & 15) == 0 && (reinterpret_cast<uintptr_t>(dkeys) & 15) == 0;
    if (v16) {
        pf = din_partial_floats<16>();
        const size_t smem = din16::smem_bytes();
        auto kern = is_softmax ? &din16::bwd_kernel<true> : &din16::bwd_kernel<false>;
        hipError_t e16 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e16 != hipSuccess) return (int)e16;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), smem, st, query, keys, keys_length, f1_w, f1_b, f2_w, f2_b, f3_w,
                           f3_b, g_out, (unsigned)ldg, dq_extra, (unsigned)ld_extra, (unsigned)B, (unsigned)T, dquery, dkeys, partials);
    } else if (H == 4) LAUNCH(4); else if (H == 8) LAUNCH(8); else LAUNCH(16);
#undef LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (d_f1_w == nullptr) return 0;              // the caller sums the partial rows (a job of the step's deferred-sum launch)
    // the partial row is laid out exactly as [d_f1_w | d_f1_b | d_f2_w | d_f2_b | d_f3_w | d_f3_b]:
    // reduce it segment by segment into the caller's six buffers
    struct Seg { float* dst; int off, n; };
    const int o_b1 = 4 * H * N1, o_w2 = o_b1 + N1, o_b2 = o_w2 + N1 * N2, o_w3 = o_b2 + N2, o_b3 = o_w3 + N2;
    const Seg segs[6] = {{d_f1_w, 0, o_b1}, {d_f1_b, o_b1, N1}, {d_f2_w, o_w2, N1 * N2},
                         {d_f2_b, o_b2, N2}, {d_f3_w, o_w3, N2}, {d_f3_b, o_b3, 1}};
    bool contiguous = true;                       // the six outputs laid out like the partial row (flat gradient buffer)?
    for (int sgi = 1; sgi < 6; ++sgi) contiguous = contiguous && segs[sgi].dst == d_f1_w + segs[sgi].off;
    if (contiguous) {
        launch_colsum16(partials, (unsigned)grid, (unsigned)pf, d_f1_w, (unsigned)pf, static_cast<float*>(nullptr), st);
    } else {
        for (int sgi = 0; sgi < 6; ++sgi) {
            const Seg& sg = segs[sgi];
            hipLaunchKernelGGL(din_sum_partials_kernel, dim3(cdiv(sg.n, 64)), dim3(256), 0, st, partials + sg.off,
                               (unsigned)grid, (unsigned)pf, (unsigned)sg.n, sg.dst);
        }
    }
    RECALGO_RETURN_LAST();
}
