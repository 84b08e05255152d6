// K9: DIN attention (algorithm/DIN/din_attention.py:4-43), gfx950.
//
//   x_t = [q, k_t, q - k_t, q * k_t]                                   (4H)
//   s_t = f3(relu(f2(relu(f1 x_t))))            dense 4H -> 64 -> 32 -> 1, with biases
//   softmax branch : s_t <- (t < len ? s_t : -2^32+1) / sqrt(H);  w = softmax_t(s)
//   default branch : w_t = s_t * [t < len]
//   out = sum_t w_t k_t
//
// FLOP-bound (fp32 vector).  Mapping: one wave per example, lane t owns history position t
// (T <= 64) and runs the tiny MLP on its own row entirely in registers; the weights are staged
// once per (persistent) workgroup in LDS and read as wave-uniform broadcasts.  f1 is factored so
// that its q-only part is computed once per example:
//   f1 x = q (W1a + W1c) + k (W1b - W1c) + (q*k) W1d
// (half the layer-1 FLOPs of the reference's concat + dense; same math, different fp32 order).
// The backward recomputes the forward per row, back-propagates per row on the VALU, and forms the
// weight gradients — sums over all (b, t) rows of outer products — on the fp32 MFMA pipe, which
// runs concurrently with the VALU work of the other waves.
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int N1 = 64, N2 = 32;
constexpr int kXS = 33, kS64 = 68, kS32 = 36;            // padded LDS row strides (floats)
constexpr int kScratch = 64 * kS64 + 64 * kS32;          // per-wave backward scratch (floats)
constexpr float kPadScore = -4294967296.0f;      // float32(-2**32 + 1), din_attention.py:31

template <int H>
struct Smem {
    float Wq[H][N1];       // W1a + W1c
    float Wk[H][N1];       // W1b - W1c
    float Wd[H][N1];       // W1d
    float b1[N1];
    float W2[N1][N2];
    float b2[N2];
    float W3[N2];
    float b3[4];
    float cq[kWaves][N1];  // per-example q-only part of layer 1
};

template <int H>
__device__ __forceinline__ void stage_weights(Smem<H>& S, const float* __restrict__ f1w,
                                              const float* __restrict__ f1b, const float* __restrict__ f2w,
                                              const float* __restrict__ f2b, const float* __restrict__ f3w,
                                              const float* __restrict__ f3b) {
    for (unsigned e = threadIdx.x; e < H * N1; e += kThreads) {
        unsigned i = e / N1, j = e - i * N1;
        float a = f1w[(0 * H + i) * N1 + j], b = f1w[(1 * H + i) * N1 + j];
        float c = f1w[(2 * H + i) * N1 + j], d = f1w[(3 * H + i) * N1 + j];
        S.Wq[i][j] = a + c;
        S.Wk[i][j] = b - c;
        S.Wd[i][j] = d;
    }
    for (unsigned e = threadIdx.x; e < N1 * N2; e += kThreads) S.W2[e / N2][e % N2] = f2w[e];
    if (threadIdx.x < N1) S.b1[threadIdx.x] = f1b[threadIdx.x];
    if (threadIdx.x < N2) {
        S.b2[threadIdx.x] = f2b[threadIdx.x];
        S.W3[threadIdx.x] = f3w[threadIdx.x];
    }
    if (threadIdx.x == 0) S.b3[0] = f3b[0];
}

// per-row forward: fills h1, h2 (post-relu) and returns the raw score
template <int H>
__device__ __forceinline__ float row_forward(const Smem<H>& S, unsigned wave, const float (&k)[H],
                                             const float (&qk)[H], float (&h1)[N1], float (&h2)[N2]) {
#pragma unroll
    for (int j = 0; j < N1; j += 4) {
        float4 c = *reinterpret_cast<const float4*>(&S.cq[wave][j]);
        h1[j] = c.x; h1[j + 1] = c.y; h1[j + 2] = c.z; h1[j + 3] = c.w;
    }
#pragma unroll
    for (int i = 0; i < H; ++i) {
#pragma unroll
        for (int j = 0; j < N1; j += 4) {
            float4 wk = *reinterpret_cast<const float4*>(&S.Wk[i][j]);
            float4 wd = *reinterpret_cast<const float4*>(&S.Wd[i][j]);
            h1[j + 0] = fmaf(k[i], wk.x, fmaf(qk[i], wd.x, h1[j + 0]));
            h1[j + 1] = fmaf(k[i], wk.y, fmaf(qk[i], wd.y, h1[j + 1]));
            h1[j + 2] = fmaf(k[i], wk.z, fmaf(qk[i], wd.z, h1[j + 2]));
            h1[j + 3] = fmaf(k[i], wk.w, fmaf(qk[i], wd.w, h1[j + 3]));
        }
    }
#pragma unroll
    for (int j = 0; j < N1; ++j) h1[j] = fmaxf(h1[j], 0.f);
#pragma unroll
    for (int j = 0; j < N2; j += 4) {
        float4 c = *reinterpret_cast<const float4*>(&S.b2[j]);
        h2[j] = c.x; h2[j + 1] = c.y; h2[j + 2] = c.z; h2[j + 3] = c.w;
    }
#pragma unroll
    for (int i = 0; i < N1; ++i) {
#pragma unroll
        for (int j = 0; j < N2; j += 4) {
            float4 w = *reinterpret_cast<const float4*>(&S.W2[i][j]);
            h2[j + 0] = fmaf(h1[i], w.x, h2[j + 0]);
            h2[j + 1] = fmaf(h1[i], w.y, h2[j + 1]);
            h2[j + 2] = fmaf(h1[i], w.z, h2[j + 2]);
            h2[j + 3] = fmaf(h1[i], w.w, h2[j + 3]);
        }
    }
    float s = S.b3[0];
#pragma unroll
    for (int j = 0; j < N2; ++j) {
        h2[j] = fmaxf(h2[j], 0.f);
        s = fmaf(h2[j], S.W3[j], s);
    }
    return s;
}

// loads q (wave-uniform) and this lane's key row, computes the per-example q-only layer-1 part
template <int H>
__device__ __forceinline__ void load_example(Smem<H>& S, unsigned wave, unsigned lane, unsigned ex, unsigned T,
                                             const float* __restrict__ query, const float* __restrict__ keys,
                                             float (&q)[H], float (&k)[H], float (&qk)[H]) {
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
#pragma unroll
    for (int i = 0; i < H; ++i) qk[i] = q[i] * k[i];
    // cq[j] = b1[j] + sum_i q_i (W1a + W1c)[i][j]   — lane j computes output j
    float c = S.b1[lane];
#pragma unroll
    for (int i = 0; i < H; ++i) c = fmaf(q[i], S.Wq[i][lane], c);
    S.cq[wave][lane] = c;
    __builtin_amdgcn_wave_barrier();
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
    __shared__ Smem<H> S;
    stage_weights<H>(S, f1w, f1b, f2w, f2b, f3w, f3b);
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (unsigned ex = blockIdx.x * kWaves + wave; ex < B; ex += gridDim.x * kWaves) {
        float q[H], k[H], qk[H], h1[N1], h2[N2];
        load_example<H>(S, wave, lane, ex, T, query, keys, q, k, qk);
        float s = row_forward<H>(S, wave, k, qk, h1, h2);
        const int len = keys_length[ex];
        float w = attention_weight<H>(s, lane < T && (int)lane < len, lane < T, is_softmax);
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
    const float* __restrict__ g_out, unsigned B, unsigned T, int is_softmax, float* __restrict__ dquery,
    float* __restrict__ dkeys, float* __restrict__ partials) {
    static_assert(2 * H <= 32, "k and q*k must fit one 32-wide MFMA tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Smem<H>& S = *reinterpret_cast<Smem<H>*>(smem_raw);
    // per-wave scratch (kScratch floats): X|dH1, then H1|dH2, then H2|ds.  Row strides are padded
    // (33 / 68 / 36 floats) so that the row-per-lane ds_write_b32 / b128 stores are conflict free
    float* scratch_all = reinterpret_cast<float*>(smem_raw + ((sizeof(Smem<H>) + 15) & ~(size_t)15));
    stage_weights<H>(S, f1w, f1b, f2w, f2b, f3w, f3b);
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned hi = lane >> 5, l32 = lane & 31;
    float* sc = scratch_all + wave * kScratch;

    // MFMA accumulators: dWx [32 (k|qk, zero padded) x 64] = 2 tiles, dW2 [64 x 32] = 2 tiles
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
        float q[H], k[H], qk[H], h1[N1], h2[N2];
        load_example<H>(S, wave, lane, ex, T, query, keys, q, k, qk);
        const float s = row_forward<H>(S, wave, k, qk, h1, h2);
        const int len = keys_length[ex];
        const bool in_T = lane < T, in_len = in_T && (int)lane < len;
        const float w = attention_weight<H>(s, in_len, in_T, is_softmax);
        // ---- attention output backward ----
        float g[H];
        {
            const float4* gr = reinterpret_cast<const float4*>(g_out + (size_t)ex * H);
#pragma unroll
            for (int i = 0; i < H; i += 4) {
                float4 v = gr[i / 4];
                g[i] = v.x; g[i + 1] = v.y; g[i + 2] = v.z; g[i + 3] = v.w;
            }
        }
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
        float dk[H];
#pragma unroll
        for (int i = 0; i < H; ++i) dk[i] = w * g[i];
        // ---- MLP backward (per row) ----
        float dh2[N2];
#pragma unroll
        for (int j = 0; j < N2; ++j) dh2[j] = h2[j] > 0.f ? ds * S.W3[j] : 0.f;
        float dh1[N1];
#pragma unroll
        for (int i = 0; i < N1; ++i) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < N2; j += 4) {
                float4 wv = *reinterpret_cast<const float4*>(&S.W2[i][j]);
                a = fmaf(dh2[j], wv.x, fmaf(dh2[j + 1], wv.y, fmaf(dh2[j + 2], wv.z, fmaf(dh2[j + 3], wv.w, a))));
            }
            dh1[i] = h1[i] > 0.f ? a : 0.f;
        }
        float dq[H];
#pragma unroll
        for (int i = 0; i < H; ++i) {
            float aq = 0.f, ak = 0.f, ad = 0.f;
#pragma unroll
            for (int j = 0; j < N1; j += 4) {
                float4 wq = *reinterpret_cast<const float4*>(&S.Wq[i][j]);
                float4 wk = *reinterpret_cast<const float4*>(&S.Wk[i][j]);
                float4 wd = *reinterpret_cast<const float4*>(&S.Wd[i][j]);
                aq = fmaf(dh1[j], wq.x, fmaf(dh1[j + 1], wq.y, fmaf(dh1[j + 2], wq.z, fmaf(dh1[j + 3], wq.w, aq))));
                ak = fmaf(dh1[j], wk.x, fmaf(dh1[j + 1], wk.y, fmaf(dh1[j + 2], wk.z, fmaf(dh1[j + 3], wk.w, ak))));
                ad = fmaf(dh1[j], wd.x, fmaf(dh1[j + 1], wd.y, fmaf(dh1[j + 2], wd.z, fmaf(dh1[j + 3], wd.w, ad))));
            }
            dq[i] = fmaf(ad, k[i], aq);
            dk[i] += fmaf(ad, q[i], ak);
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
            dquery[(size_t)ex * H + lane] = v;
        }

        // ---- weight gradients: outer products summed over the 64 rows, on the MFMA pipe ----
        // phase A: X = [k | q*k | 0] (32 cols) and dH1 (64 cols)
        {
            float* X = sc;               // [64][kXS]
            float* Dh = sc + 64 * kXS;   // [64][kS64]  (64*33 is a multiple of 4: float4 aligned)
#pragma unroll
            for (int i = 0; i < H; ++i) {
                X[lane * kXS + i] = k[i];
                X[lane * kXS + H + i] = qk[i];
            }
#pragma unroll
            for (int i = 2 * H; i < 32; ++i) X[lane * kXS + i] = 0.f;
#pragma unroll
            for (int j = 0; j < N1; j += 4)
                *reinterpret_cast<float4*>(&Dh[lane * kS64 + j]) = make_float4(dh1[j], dh1[j + 1], dh1[j + 2], dh1[j + 3]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll 4
            for (int st = 0; st < 32; ++st) {
                float a = X[(2 * st + hi) * kXS + l32];
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
                    float bfr = Dh[(2 * st + hi) * kS64 + jt * 32 + l32];
                    accX[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bfr, accX[jt], 0, 0, 0);
                }
            }
            float cs = 0.f;            // column sum of dH1 for column `lane`
#pragma unroll 8
            for (int r = 0; r < 64; ++r) cs += Dh[r * kS64 + lane];
            db1 += cs;
#pragma unroll
            for (int i = 0; i < H; ++i) dWq[i] = fmaf(q[i], cs, dWq[i]);
            __builtin_amdgcn_wave_barrier();
        }
        // phase B: H1 (64 cols) and dH2 (32 cols)
        {
            float* Hs = sc;              // [64][kS64]
            float* Dh = sc + 64 * kS64;  // [64][kS32]
#pragma unroll
            for (int j = 0; j < N1; j += 4)
                *reinterpret_cast<float4*>(&Hs[lane * kS64 + j]) = make_float4(h1[j], h1[j + 1], h1[j + 2], h1[j + 3]);
#pragma unroll
            for (int j = 0; j < N2; j += 4)
                *reinterpret_cast<float4*>(&Dh[lane * kS32 + j]) = make_float4(dh2[j], dh2[j + 1], dh2[j + 2], dh2[j + 3]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll 4
            for (int st = 0; st < 32; ++st) {
                float bfr = Dh[(2 * st + hi) * kS32 + l32];
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    float a = Hs[(2 * st + hi) * kS64 + it * 32 + l32];
                    acc2[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bfr, acc2[it], 0, 0, 0);
                }
            }
            if (lane < N2) {
                float cs = 0.f;
#pragma unroll 8
                for (int r = 0; r < 64; ++r) cs += Dh[r * kS32 + lane];
                db2 += cs;
            }
            __builtin_amdgcn_wave_barrier();
        }
        // phase C: dW3[j] = sum_rows ds * h2[j], db3 = sum_rows ds
        {
            float* Hs = sc;              // [64][kS32] h2
            float* Ds = sc + 64 * kS32;  // [64]
#pragma unroll
            for (int j = 0; j < N2; j += 4)
                *reinterpret_cast<float4*>(&Hs[lane * kS32 + j]) = make_float4(h2[j], h2[j + 1], h2[j + 2], h2[j + 3]);
            Ds[lane] = ds;
            __builtin_amdgcn_wave_barrier();
            if (lane < N2) {
                float a = 0.f;
#pragma unroll 8
                for (int r = 0; r < 64; ++r) a = fmaf(Ds[r], Hs[r * kS32 + lane], a);
                dW3 += a;
            }
            db3 += wave_sum(ds);
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ---- workgroup reduction of the weight-gradient partials (fixed wave order) ----
    constexpr int PF = din_partial_floats<H>();
    __syncthreads();
    float* red = scratch_all;                       // [PF], reuses the wave scratch area
    for (unsigned wv = 0; wv < kWaves; ++wv) {
        if (wave == wv) {
            auto put = [&](unsigned idx, float v) { red[idx] = (wv == 0 ? 0.f : red[idx]) + v; };
            // dWx tile jt: rows i = (r&3)+8*(r>>2)+4*hi (0..31: k rows 0..H-1, qk rows H..2H-1), col jt*32+l32
            // final dW1 = [dWq ; dWk ; dWq - dWk ; dWqk]   (blocks a, b, c, d of f1's kernel)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
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
                    int i = it * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
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
size_t din_bwd_smem() { return ((sizeof(Smem<H>) + 15) & ~(size_t)15) + (size_t)kWaves * kScratch * sizeof(float); }

}  // namespace

RECALGO_EXPORT int recalgo_din_attention_fwd(const float* query, const float* keys, const int32_t* keys_length,
                                             const float* f1_w, const float* f1_b, const float* f2_w,
                                             const float* f2_b, const float* f3_w, const float* f3_b, int B, int T,
                                             int H, int is_softmax, float* out, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && T >= 1 && T <= 64 && (H == 4 || H == 8 || H == 16));
    if (B == 0) return 0;
    hipStream_t st = as_stream(stream);
#define LAUNCH(HH)                                                                                          \
    hipLaunchKernelGGL(din_attention_fwd_kernel<HH>, dim3(din_grid(B) * 2 > cdiv(B, kWaves) ? cdiv(B, kWaves) : din_grid(B) * 2), \
                       dim3(kThreads), 0, st, query, keys, keys_length, f1_w, f1_b, f2_w, f2_b, f3_w, f3_b,  \
                       (unsigned)B, (unsigned)T, is_softmax, out)
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

RECALGO_EXPORT int recalgo_din_attention_bwd(const float* query, const float* keys, const int32_t* keys_length,
                                             const float* f1_w, const float* f1_b, const float* f2_w,
                                             const float* f2_b, const float* f3_w, const float* f3_b,
                                             const float* g_out, int B, int T, int H, int is_softmax,
                                             float* dquery, float* dkeys, float* d_f1_w, float* d_f1_b,
                                             float* d_f2_w, float* d_f2_b, float* d_f3_w, float* d_f3_b,
                                             void* workspace, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B > 0 && T >= 1 && T <= 64 && (H == 4 || H == 8 || H == 16) && workspace != nullptr);
    hipStream_t st = as_stream(stream);
    float* partials = static_cast<float*>(workspace);
    const int grid = din_grid(B);
    int pf;
#define LAUNCH(HH)                                                                                            \
    do {                                                                                                      \
        pf = din_partial_floats<HH>();                                                                        \
        size_t smem = din_bwd_smem<HH>();                                                                     \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&din_attention_bwd_kernel<HH>),      \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);            \
        if (e != hipSuccess) return (int)e;                                                                   \
        hipLaunchKernelGGL(din_attention_bwd_kernel<HH>, dim3(grid), dim3(kThreads), smem, st, query, keys,   \
                           keys_length, f1_w, f1_b, f2_w, f2_b, f3_w, f3_b, g_out, (unsigned)B, (unsigned)T,  \
                           is_softmax, dquery, dkeys, partials);                                              \
    } while (0)
    if (H == 4) LAUNCH(4); else if (H == 8) LAUNCH(8); else LAUNCH(16);
#undef LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    // the partial row is laid out exactly as [d_f1_w | d_f1_b | d_f2_w | d_f2_b | d_f3_w | d_f3_b]:
    // reduce it segment by segment into the caller's six buffers
    struct Seg { float* dst; int off, n; };
    const int o_b1 = 4 * H * N1, o_w2 = o_b1 + N1, o_b2 = o_w2 + N1 * N2, o_w3 = o_b2 + N2, o_b3 = o_w3 + N2;
    const Seg segs[6] = {{d_f1_w, 0, o_b1}, {d_f1_b, o_b1, N1}, {d_f2_w, o_w2, N1 * N2},
                         {d_f2_b, o_b2, N2}, {d_f3_w, o_w3, N2}, {d_f3_b, o_b3, 1}};
    for (int sgi = 0; sgi < 6; ++sgi) {
        const Seg& sg = segs[sgi];
        hipLaunchKernelGGL(din_sum_partials_kernel, dim3(cdiv(sg.n, 64)), dim3(256), 0, st, partials + sg.off,
                           (unsigned)grid, (unsigned)pf, (unsigned)sg.n, sg.dst);
    }
    RECALGO_RETURN_LAST();
}
