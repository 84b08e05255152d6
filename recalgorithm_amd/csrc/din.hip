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
    if (H == 4) LAUNCH(4); else if (H == 8) LAUNCH(8); else LAUNCH(16);
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
