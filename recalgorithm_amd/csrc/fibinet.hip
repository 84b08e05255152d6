// K7 + K8: FiBiNET SENET re-weighting and bilinear pair interaction, gfx950.
//
//   senet(input, embedding_dim, reduction_ratio)            algorithm/FiBiNET/senet.py:4-36
//     z = mean_k E ; a = relu(relu(z w1) w2) ; V = E * a[..., None]        (no biases)
//   bilinear_interaction_layer(input, embedding_dim, type, name)
//                                            algorithm/FiBiNET/bilinear_interaction_layer.py:5-42
//     p_(i,j) = (e_i W_*) * e_j  for (i, j) in combinations(range(F-1), 2)     (quirk B-3: the last
//     field never participates);  W_* = W (all) | W_i (each) | W_pair (interaction)
//   the model concatenates the interaction of E and of V on the last axis (fibinet.py:177-187).
//
// Both layers are per-example independent and HBM-bound: the bilinear output is
// (F-1)(F-2)/2 * 2K floats per example (38.4 KB at F=26, K=16) against 1.6 KB of input, so the
// kernel is a coalesced streaming write with the 25 e_i W products (K^2 FMAs each) done once per
// example in LDS.  One wave owns one example; a persistent workgroup of 4 waves stages the
// batch-constant pieces (pair table, W) in LDS once.  The bilinear kernels take up to two
// (input, weight) sets and write them interleaved [pair][set][K] — i.e. the concat of
// fibinet.py:186 is fused, and every 128-byte line of the output is written by one wave
// instruction.
//
// Weight gradients (sums over the batch) are formed by a second, batch-split kernel from the
// per-example row gradients d(e_i W_*) that the backward stores in the workspace, then summed in
// a fixed order (deterministic).
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kMaxBlocks = 1024;

enum { kAll = 0, kEach = 1, kInteraction = 2 };

__device__ __forceinline__ unsigned tri_index(unsigned i, unsigned j, unsigned n) {
    return i * (2 * n - i - 1) / 2 + (j - i - 1);
}
__device__ __forceinline__ void lds_add(float* p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) {
    return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}

// =============================================================================================
// SENET
// =============================================================================================
// per-wave LDS: X[F*K] | z[F] | h[Rd] | a[F] | (bwd) dap[F] | dhp[Rd] | Gv[F*K]
__device__ __forceinline__ unsigned senet_wave_floats(unsigned F, unsigned K, unsigned Rd) {
    return ((2 * F * K + 3 * F + 2 * Rd) + 3) & ~3u;
}

// global -> LDS row copy with the loads of 4 iterations in flight (a plain load/store loop waits
// for the HBM latency once per iteration)
__device__ __forceinline__ void copy_row(float* __restrict__ dst, const float* __restrict__ src, unsigned n,
                                         unsigned lane) {
    for (unsigned i0 = 0; i0 < n; i0 += 256) {
        float t[4];
#pragma unroll
        for (unsigned u = 0; u < 4; ++u) {
            const unsigned i = i0 + u * 64 + lane;
            t[u] = i < n ? src[i] : 0.f;
        }
#pragma unroll
        for (unsigned u = 0; u < 4; ++u) {
            const unsigned i = i0 + u * 64 + lane;
            if (i < n) dst[i] = t[u];
        }
    }
}

// z, h, a of one example from X (already in LDS); all lanes must call
__device__ __forceinline__ void senet_squeeze_excite(const float* X, float* z, float* h, float* a,
                                                     const float* __restrict__ w1,
                                                     const float* __restrict__ w2, unsigned F, unsigned K,
                                                     unsigned Rd, unsigned lane) {
    const float invK = 1.0f / (float)K;
    for (unsigned f = lane; f < F; f += 64) {
        float s = 0.f;
        for (unsigned k = 0; k < K; ++k) s += X[f * K + k];
        z[f] = s * invK;
    }
    __builtin_amdgcn_wave_barrier();
    for (unsigned r = lane; r < Rd; r += 64) {
        float s = 0.f;
        for (unsigned f = 0; f < F; ++f) s = fmaf(z[f], w1[f * Rd + r], s);
        h[r] = fmaxf(s, 0.f);
    }
    __builtin_amdgcn_wave_barrier();
    for (unsigned f = lane; f < F; f += 64) {
        float s = 0.f;
        for (unsigned r = 0; r < Rd; ++r) s = fmaf(h[r], w2[r * F + f], s);
        a[f] = fmaxf(s, 0.f);
    }
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(kThreads) void senet_fwd_kernel(
    const float* __restrict__ emb, const float* w1, const float* w2, unsigned B,
    unsigned F, unsigned K, unsigned Rd, float* __restrict__ v_out, float* __restrict__ a_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const unsigned FK = F * K;
    float* X = smem + wib * senet_wave_floats(F, K, Rd);
    float* z = X + FK;
    float* h = z + F;
    float* a = h + Rd;
    float* w1s = smem + kWaves * senet_wave_floats(F, K, Rd);      // [F][Rd] | [Rd][F], staged once
    float* w2s = w1s + F * Rd;
    for (unsigned i = threadIdx.x; i < F * Rd; i += kThreads) {
        w1s[i] = w1[i];
        w2s[i] = w2[i];
    }
    __syncthreads();
    w1 = w1s;
    w2 = w2s;
    for (unsigned b = blockIdx.x * kWaves + wib; b < B; b += gridDim.x * kWaves) {
        copy_row(X, emb + (size_t)b * FK, FK, lane);
        __builtin_amdgcn_wave_barrier();
        senet_squeeze_excite(X, z, h, a, w1, w2, F, K, Rd, lane);
        float* vr = v_out + (size_t)b * FK;
        for (unsigned i = lane; i < FK; i += 64) vr[i] = X[i] * a[i / K];
        if (a_out)
            for (unsigned f = lane; f < F; f += 64) a_out[(size_t)b * F + f] = a[f];
        __builtin_amdgcn_wave_barrier();
    }
}

// backward; per-wave gradient accumulators for (w1, w2) live in LDS across the wave's examples
__global__ __launch_bounds__(kThreads) void senet_bwd_kernel(
    const float* __restrict__ emb, const float* w1, const float* w2,
    const float* __restrict__ g_v, unsigned B, unsigned F, unsigned K, unsigned Rd,
    float* __restrict__ d_emb, int accumulate, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const unsigned FK = F * K, WR = F * Rd;
    const unsigned wf = senet_wave_floats(F, K, Rd);
    float* X = smem + wib * wf;
    float* z = X + FK;
    float* h = z + F;
    float* a = h + Rd;
    float* dap = a + F;
    float* dhp = dap + F;
    float* Gv = dhp + Rd;                                    // upstream gradient row [F*K]
    float* acc = smem + kWaves * wf + wib * 2 * WR;          // [dw1 (F,Rd) | dw2 (Rd,F)]
    float* w1s = smem + kWaves * wf + kWaves * 2 * WR;       // [F][Rd] | [Rd][F], staged once
    float* w2s = w1s + WR;
    for (unsigned i = threadIdx.x; i < WR; i += kThreads) {
        w1s[i] = w1[i];
        w2s[i] = w2[i];
    }
    for (unsigned i = lane; i < 2 * WR; i += 64) acc[i] = 0.f;
    __syncthreads();
    w1 = w1s;
    w2 = w2s;
    const float invK = 1.0f / (float)K;
    for (unsigned b = blockIdx.x * kWaves + wib; b < B; b += gridDim.x * kWaves) {
        copy_row(X, emb + (size_t)b * FK, FK, lane);
        copy_row(Gv, g_v + (size_t)b * FK, FK, lane);
        const float* gr = Gv;
        __builtin_amdgcn_wave_barrier();
        senet_squeeze_excite(X, z, h, a, w1, w2, F, K, Rd, lane);
        // da_f = sum_k gV[f,k] E[f,k], through relu
        for (unsigned f = lane; f < F; f += 64) {
            float s = 0.f;
            for (unsigned k = 0; k < K; ++k) s = fmaf(gr[f * K + k], X[f * K + k], s);
            dap[f] = a[f] > 0.f ? s : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        for (unsigned r = lane; r < Rd; r += 64) {
            float s = 0.f;
            for (unsigned f = 0; f < F; ++f) s = fmaf(dap[f], w2[r * F + f], s);
            dhp[r] = h[r] > 0.f ? s : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        // d_emb = gV * a + dz / K
        float* dr = d_emb + (size_t)b * FK;
        for (unsigned i = lane; i < FK; i += 64) {
            unsigned f = i / K;
            float dz = 0.f;
            for (unsigned r = 0; r < Rd; ++r) dz = fmaf(dhp[r], w1[f * Rd + r], dz);
            float v = fmaf(gr[i], a[f], dz * invK);
            dr[i] = accumulate ? dr[i] + v : v;
        }
        // weight gradient contributions of this example
        for (unsigned i = lane; i < WR; i += 64) {
            unsigned f = i / Rd, r = i - f * Rd;
            acc[i] = fmaf(z[f], dhp[r], acc[i]);                   // dw1[f, r]
            unsigned r2 = i / F, f2 = i - r2 * F;
            acc[WR + i] = fmaf(h[r2], dap[f2], acc[WR + i]);       // dw2[r, f]
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    const float* acc0 = smem + kWaves * wf;
    for (unsigned i = threadIdx.x; i < 2 * WR; i += kThreads) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) s += acc0[w * 2 * WR + i];
        partials[(size_t)blockIdx.x * 2 * WR + i] = s;
    }
}

// =============================================================================================
// bilinear interaction
// =============================================================================================
struct BiSets {
    const float* x[2];   // [B, F, K]
    const float* w[2];   // all: [K,K]; each: [F-1,K,K]; interaction: [F(F-1)/2, K, K]
    float* dx[2];        // backward: [B, F, K]
};

template <int K>
__device__ __forceinline__ void build_pair_table(unsigned short* ptab, unsigned n) {
    for (unsigned i = threadIdx.x; i + 1 < n; i += blockDim.x)
        for (unsigned j = i + 1; j < n; ++j) ptab[tri_index(i, j, n)] = (unsigned short)(i | (j << 8));
}

// LDS layout helpers (floats); the pair table (uint16[P]) comes first, padded to 16 bytes
__host__ __device__ inline unsigned ptab_floats(unsigned P) { return ((P * 2 + 15) / 16) * 4; }
constexpr unsigned kWS(unsigned K) { return K + 4; }          // row stride of a staged W: rows 16-byte aligned (ds_read_b128), and
                                                              // 16 lanes walking 16 rows of one column still hit 16 banks (20 k mod 64)

template <int K, int NV>
__global__ __launch_bounds__(kThreads) void bilinear_fwd_kernel(BiSets a, unsigned B, unsigned F, int type,
                                                                float* __restrict__ out, unsigned out_stride,
                                                                unsigned out_col) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned n = F - 1, P = n * (n - 1) / 2, FK = F * K, nK = n * K;
    unsigned short* ptab = reinterpret_cast<unsigned short*>(smem);
    float* Wl = smem + ptab_floats(P);                                   // [NV][K][K+1] (type all)
    float* wave0 = Wl + (type == kAll ? NV * K * kWS(K) : 0);
    const unsigned lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    float* X = wave0 + wib * NV * (FK + nK);                             // [NV][FK]
    float* vW = X + NV * FK;                                             // [NV][nK]

    build_pair_table<K>(ptab, n);
    if (type == kAll)
        for (unsigned e = threadIdx.x; e < NV * K * K; e += kThreads) {
            unsigned v = e / (K * K), r = e % (K * K);
            Wl[v * K * kWS(K) + (r / K) * kWS(K) + (r % K)] = a.w[v][r];
        }
    __syncthreads();

    constexpr unsigned C4 = NV * K / 4;          // float4 per output pair row
    constexpr unsigned PPP = 64 / C4;            // pairs per wave pass
    const unsigned pl = lane / C4, c4 = lane % C4;
    const unsigned v_l = (c4 * 4) / K, k0 = (c4 * 4) % K;

    for (unsigned b = blockIdx.x * kWaves + wib; b < B; b += gridDim.x * kWaves) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4* xr = reinterpret_cast<const float4*>(a.x[v] + (size_t)b * FK);
            for (unsigned i = lane; i < FK / 4; i += 64) reinterpret_cast<float4*>(X + v * FK)[i] = xr[i];
        }
        __builtin_amdgcn_wave_barrier();
        if (type != kInteraction) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
                for (unsigned idx = lane; idx < nK; idx += 64) {
                    const unsigned i = idx / K, kk = idx % K;
                    const float* xi = X + v * FK + i * K;
                    float acc = 0.f;
                    if (type == kAll) {
                        const float* W = Wl + v * K * kWS(K) + kk;
#pragma unroll
                        for (int k = 0; k < K; ++k) acc = fmaf(xi[k], W[k * kWS(K)], acc);
                    } else {
                        const float* W = a.w[v] + (size_t)i * K * K + kk;
#pragma unroll
                        for (int k = 0; k < K; ++k) acc = fmaf(xi[k], W[k * K], acc);
                    }
                    vW[v * nK + idx] = acc;
                }
            __builtin_amdgcn_wave_barrier();
        }
        float* ob = out + (size_t)b * P * out_stride + out_col + c4 * 4;
        for (unsigned p0 = 0; p0 < P; p0 += PPP) {
            const unsigned pair = p0 + pl;
            if (pair < P) {
                const unsigned ij = ptab[pair], i = ij & 255u, j = ij >> 8;
                const float4 xj = *reinterpret_cast<const float4*>(X + v_l * FK + j * K + k0);
                float4 t;
                if (type != kInteraction) {
                    t = *reinterpret_cast<const float4*>(vW + v_l * nK + i * K + k0);
                } else {
                    const float* xi = X + v_l * FK + i * K;
                    const float* W = a.w[v_l] + (size_t)pair * K * K + k0;
                    t = f4_zero();
#pragma unroll
                    for (int k = 0; k < K; ++k) t = f4_fma(*reinterpret_cast<const float4*>(W + k * K), xi[k], t);
                }
                *reinterpret_cast<float4*>(ob + (size_t)pair * out_stride) = f4_mul(t, xj);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// backward wrt the inputs; stores the row gradients dvW[v][b][i][:] (all / each) for the weight
// gradient kernel.
// COOP = false: one WAVE owns one example (the forward's shape).  With the gradient tile staged (48 KB of LDS per example
// at F = 26, K = 16, two sets) that is 3 waves per CU, each walking load -> stage -> two sums -> store alone: 138 us for
// 157 MB.  COOP = true: the whole WORKGROUP (256 threads) owns one example — the same 48 KB now carry 4 waves, three
// workgroups per CU, and every element is still produced by the same chain of operations in the same order
// (bit-identical results).  Interaction-type weights keep the wave form (its reductions are wave shuffles).
constexpr int kCoopThreads = 256;
template <int K, int NV, bool COOP>
__global__ __launch_bounds__(COOP ? kCoopThreads : kThreads, COOP ? 3 : 1) void bilinear_bwd_kernel(BiSets a, unsigned B, unsigned F, int type,
                                                                const float* __restrict__ g, unsigned g_stride,
                                                                unsigned g_col, float* __restrict__ dvw_ws,
                                                                int stage_g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned n = F - 1, P = n * (n - 1) / 2, FK = F * K, nK = n * K;
    unsigned short* ptab = reinterpret_cast<unsigned short*>(smem);
    float* Wl = smem + ptab_floats(P);
    float* wave0 = Wl + (type == kAll ? NV * K * kWS(K) : 0);
    const unsigned lane = COOP ? threadIdx.x : (threadIdx.x & 63);   // index among the NT threads that share an example
    const unsigned wib = COOP ? 0u : threadIdx.x >> 6;
    const unsigned NT = COOP ? blockDim.x : 64u;
    const unsigned nwaves = COOP ? 1u : blockDim.x >> 6;             // examples in flight per workgroup (LDS budget decides)
    auto sync = [&]() {
        if (COOP) __syncthreads();
        else __builtin_amdgcn_wave_barrier();
    };
    const unsigned per_wave = NV * 2 * (FK + nK) + (stage_g ? P * NV * K : 0);
    float* X = wave0 + wib * per_wave;               // [NV][FK]
    float* vW = X + NV * FK;                         // [NV][nK]
    float* dX = vW + NV * nK;                        // [NV][FK]
    float* dvW = dX + NV * FK;                       // [NV][nK]
    float* gL = stage_g ? dvW + NV * nK : nullptr;   // [P][NV*K] gradient tile of the current example

    build_pair_table<K>(ptab, n);
    if (type == kAll)
        for (unsigned e = threadIdx.x; e < NV * K * K; e += blockDim.x) {
            unsigned v = e / (K * K), r = e % (K * K);
            Wl[v * K * kWS(K) + (r / K) * kWS(K) + (r % K)] = a.w[v][r];
        }
    __syncthreads();

    constexpr unsigned C4 = NV * K / 4;
    const unsigned PPP = NT / C4;                    // pair rows per pass of the NT threads
    constexpr unsigned LPV = K / 4;                  // lanes per (pair, set)
    const unsigned pl = lane / C4, c4 = lane % C4;
    const unsigned v_l = (c4 * 4) / K, k0 = (c4 * 4) % K;

    // COOP, the example's gradient tile and input rows a few requests per thread (P <= 320 pair rows at K = 16 x 2 sets):
    // the NEXT example's requests are issued as soon as the tile of this one is parked in LDS — their HBM round trip runs
    // under the two sums instead of in front of them
    constexpr unsigned kUpre = 10;
    constexpr bool pipelined = COOP;                 // (the launcher picks COOP only for such shapes, all / each types, tile staged)
    float4 gpre[kUpre], xpre[NV];
    auto request = [&](unsigned bb) {
        const float* gbb = g + (size_t)bb * P * g_stride + g_col;
#pragma unroll
        for (unsigned u = 0; u < kUpre; ++u) {
            const unsigned pair = u * PPP + pl;
            gpre[u] = pair < P ? *reinterpret_cast<const float4*>(gbb + (size_t)pair * g_stride + c4 * 4) : f4_zero();
        }
#pragma unroll
        for (int v = 0; v < NV; ++v)
            xpre[v] = lane < FK / 4 ? reinterpret_cast<const float4*>(a.x[v] + (size_t)bb * FK)[lane] : f4_zero();
    };
    if (pipelined && blockIdx.x < B) request(blockIdx.x);

    for (unsigned b = blockIdx.x * nwaves + wib; b < B; b += gridDim.x * nwaves) {
        if (pipelined) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
                if (lane < FK / 4) reinterpret_cast<float4*>(X + v * FK)[lane] = xpre[v];
        } else {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float4* xr = reinterpret_cast<const float4*>(a.x[v] + (size_t)b * FK);
                for (unsigned i = lane; i < FK / 4; i += NT) reinterpret_cast<float4*>(X + v * FK)[i] = xr[i];
            }
        }
        sync();
        const float* gb = g + (size_t)b * P * g_stride + g_col;
        if (COOP || type != kInteraction) {
            // recompute vW
            if constexpr (COOP) {
                // four outputs per thread: x_i once (K / 4 ds_read_b128), a float4 of W per k; each output is the same k-ascending chain
                constexpr unsigned K4 = K / 4;
                for (unsigned idx = lane; idx < NV * n * K4; idx += NT) {
                    const unsigned v = idx / (n * K4), rem = idx % (n * K4), i = rem / K4, q = rem % K4;
                    float xi[K];
#pragma unroll
                    for (unsigned u = 0; u < K4; ++u) {
                        const float4 t = reinterpret_cast<const float4*>(X + v * FK + i * K)[u];
                        xi[4 * u] = t.x; xi[4 * u + 1] = t.y; xi[4 * u + 2] = t.z; xi[4 * u + 3] = t.w;
                    }
                    const float4* W4 = type == kAll ? reinterpret_cast<const float4*>(Wl + v * K * kWS(K)) + q
                                                    : reinterpret_cast<const float4*>(a.w[v] + (size_t)i * K * K) + q;
                    const unsigned ws4 = type == kAll ? kWS(K) / 4 : K4;
                    float4 acc = f4_zero();
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const float4 w = W4[k * ws4];
                        acc.x = fmaf(xi[k], w.x, acc.x); acc.y = fmaf(xi[k], w.y, acc.y);
                        acc.z = fmaf(xi[k], w.z, acc.z); acc.w = fmaf(xi[k], w.w, acc.w);
                    }
                    *reinterpret_cast<float4*>(vW + v * nK + i * K + 4 * q) = acc;
                }
            } else
#pragma unroll
            for (int v = 0; v < NV; ++v)
                for (unsigned idx = lane; idx < nK; idx += NT) {
                    const unsigned i = idx / K, kk = idx % K;
                    const float* xi = X + v * FK + i * K;
                    float acc = 0.f;
                    if (type == kAll) {
                        const float* W = Wl + v * K * kWS(K) + kk;
#pragma unroll
                        for (int k = 0; k < K; ++k) acc = fmaf(xi[k], W[k * kWS(K)], acc);
                    } else {
                        const float* W = a.w[v] + (size_t)i * K * K + kk;
#pragma unroll
                        for (int k = 0; k < K; ++k) acc = fmaf(xi[k], W[k * K], acc);
                    }
                    vW[v * nK + idx] = acc;
                }
            sync();
            // dvW[i][k'] = sum_{j>i} g[(i,j)][k'] x_j[k'] ;  dX[j][k'] = sum_{i<j} g[(i,j)][k'] vW[i][k'].
            // Every g element is needed twice (once per sum) along two different walks of the pair
            // triangle.  Fast path: the example's whole gradient tile [P][NV*K] (38.4 KB at F=26, K=16)
            // is staged in LDS with whole-row float4 loads (the forward's write pattern), and both sums
            // are then lane-owned loops over conflict-free ds_read_b32 (lane = column c of one row;
            // LDS float atomics were measured 2x slower than even the direct-from-global walk below).
            constexpr unsigned NVK = NV * K;
            if (COOP || gL) {
                float4* gL4 = reinterpret_cast<float4*>(gL);
                // kU whole rows per lane are requested before the first is parked in LDS: a
                // load -> wait -> ds_write loop would pay the full HBM latency once per row
                constexpr unsigned kU = 16;
                if constexpr (pipelined) {
#pragma unroll
                    for (unsigned u = 0; u < kUpre; ++u) {
                        const unsigned pair = u * PPP + pl;
                        if (pair < P) gL4[pair * C4 + c4] = gpre[u];
                    }
                } else
                for (unsigned p0 = 0; p0 < P; p0 += PPP * kU) {
                    float4 t[kU];
#pragma unroll
                    for (unsigned u = 0; u < kU; ++u) {
                        const unsigned pair = p0 + u * PPP + pl;
                        t[u] = pair < P ? *reinterpret_cast<const float4*>(gb + (size_t)pair * g_stride + c4 * 4)
                                        : f4_zero();
                    }
#pragma unroll
                    for (unsigned u = 0; u < kU; ++u) {
                        const unsigned pair = p0 + u * PPP + pl;
                        if (pair < P) gL4[pair * C4 + c4] = t[u];
                    }
                }
                sync();
                if (pipelined) {
                    const unsigned nb = b + gridDim.x * nwaves;
                    if (nb < B) request(nb);
                }
                if constexpr (COOP) {
                    // a thread owns FOUR columns of one row: one ds_read_b128 of the gradient tile and one of X / vW feed four
                    // chains (each still j- resp. i-ascending: the same sums as the scalar walk below, bit for bit)
                    const float4* gL4c = reinterpret_cast<const float4*>(gL);
                    for (unsigned idx = lane; idx < F * C4; idx += NT) {
                        const unsigned r = idx / C4, q = idx % C4, v = (q * 4) / K, kk = (q * 4) % K;
                        const float4* Xv = reinterpret_cast<const float4*>(X + v * FK + kk);
                        const float4* vWv = reinterpret_cast<const float4*>(vW + v * nK + kk);
                        float4 s1 = f4_zero(), s2 = f4_zero();
                        if (r < n) {
                            const float4* gp = gL4c + tri_index(r, r + 1, n) * C4 + q;
#pragma unroll 8
                            for (unsigned j = r + 1; j < n; ++j, gp += C4) {
                                const float4 gv = gp[0], xv = Xv[j * (K / 4)];
                                s1.x = fmaf(gv.x, xv.x, s1.x); s1.y = fmaf(gv.y, xv.y, s1.y);
                                s1.z = fmaf(gv.z, xv.z, s1.z); s1.w = fmaf(gv.w, xv.w, s1.w);
                            }
                            unsigned t = r - 1;                                              // tri_index(0, r, n)
#pragma unroll 8
                            for (unsigned i = 0; i < r; ++i) {
                                const float4 gv = gL4c[t * C4 + q], wv = vWv[i * (K / 4)];
                                s2.x = fmaf(gv.x, wv.x, s2.x); s2.y = fmaf(gv.y, wv.y, s2.y);
                                s2.z = fmaf(gv.z, wv.z, s2.z); s2.w = fmaf(gv.w, wv.w, s2.w);
                                t += n - i - 2;
                            }
                            *reinterpret_cast<float4*>(dvW + v * nK + r * K + kk) = s1;
                            *reinterpret_cast<float4*>(dvw_ws + ((size_t)v * B + b) * nK + r * K + kk) = s1;
                        }
                        *reinterpret_cast<float4*>(dX + v * FK + r * K + kk) = s2;
                    }
                } else
                for (unsigned idx = lane; idx < F * NVK; idx += NT) {
                    const unsigned r = idx / NVK, c = idx % NVK, v = c / K, kk = c % K;
                    const float* Xv = X + v * FK + kk;
                    const float* vWv = vW + v * nK + kk;
                    float s1 = 0.f, s2 = 0.f;
                    if (r < n) {
                        const float* gp = gL + tri_index(r, r + 1, n) * NVK + c;            // pairs (r, r+1..)
#pragma unroll 8
                        for (unsigned j = r + 1; j < n; ++j, gp += NVK) s1 = fmaf(gp[0], Xv[j * K], s1);
                        // pair (i, r) sits at tri(i, r) = tri(i-1, r) + (n - i - 1): walk it incrementally
                        unsigned t = r - 1;                                                  // tri_index(0, r, n)
#pragma unroll 8
                        for (unsigned i = 0; i < r; ++i) {
                            s2 = fmaf(gL[t * NVK + c], vWv[i * K], s2);
                            t += n - i - 2;
                        }
                        dvW[v * nK + r * K + kk] = s1;
                        dvw_ws[((size_t)v * B + b) * nK + r * K + kk] = s1;
                    }
                    dX[v * FK + r * K + kk] = s2;
                }
            } else {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const float* gv = gb + v * K;
                    const float* Xv = X + v * FK;
                    const float* vWv = vW + v * nK;
                    for (unsigned idx = lane; idx < FK; idx += NT) {
                        const unsigned r = idx / K, kk = idx % K;
                        float s1 = 0.f, s2 = 0.f;
                        if (r < n) {
                            const float* gp = gv + (size_t)tri_index(r, r + 1, n) * g_stride + kk;
                            for (unsigned j = r + 1; j < n; ++j, gp += g_stride) s1 = fmaf(gp[0], Xv[j * K + kk], s1);
                            for (unsigned i = 0; i < r; ++i)
                                s2 = fmaf(gv[(size_t)tri_index(i, r, n) * g_stride + kk], vWv[i * K + kk], s2);
                            dvW[v * nK + idx] = s1;
                            dvw_ws[((size_t)v * B + b) * nK + idx] = s1;
                        }
                        dX[v * FK + idx] = s2;
                    }
                }
            }
            sync();
            // dX[i][k] += sum_k' dvW[i][k'] W[k][k']
            if constexpr (COOP) {
                constexpr unsigned K4 = K / 4;
                for (unsigned idx = lane; idx < NV * n * K4; idx += NT) {
                    const unsigned v = idx / (n * K4), rem = idx % (n * K4), i = rem / K4, q = rem % K4;
                    float dq[K];
#pragma unroll
                    for (unsigned u = 0; u < K4; ++u) {
                        const float4 t = reinterpret_cast<const float4*>(dvW + v * nK + i * K)[u];
                        dq[4 * u] = t.x; dq[4 * u + 1] = t.y; dq[4 * u + 2] = t.z; dq[4 * u + 3] = t.w;
                    }
                    const float* Wb = type == kAll ? Wl + v * K * kWS(K) : a.w[v] + (size_t)i * K * K;
                    const unsigned ws = type == kAll ? kWS(K) : (unsigned)K;
                    float o[4];
#pragma unroll
                    for (unsigned c = 0; c < 4; ++c) {
                        const float4* Wr = reinterpret_cast<const float4*>(Wb + (4 * q + c) * ws);
                        float acc = 0.f;
#pragma unroll
                        for (unsigned u = 0; u < K4; ++u) {
                            const float4 w = Wr[u];
                            acc = fmaf(dq[4 * u], w.x, acc); acc = fmaf(dq[4 * u + 1], w.y, acc);
                            acc = fmaf(dq[4 * u + 2], w.z, acc); acc = fmaf(dq[4 * u + 3], w.w, acc);
                        }
                        o[c] = acc;
                    }
                    float4* d = reinterpret_cast<float4*>(dX + v * FK + i * K + 4 * q);
                    float4 t = *d;
                    t.x += o[0]; t.y += o[1]; t.z += o[2]; t.w += o[3];
                    *d = t;
                }
            } else
#pragma unroll
            for (int v = 0; v < NV; ++v)
                for (unsigned idx = lane; idx < nK; idx += NT) {
                    const unsigned i = idx / K, k = idx % K;
                    const float* dq = dvW + v * nK + i * K;
                    float acc = 0.f;
                    if (type == kAll) {
                        const float* W = Wl + v * K * kWS(K) + k * kWS(K);
#pragma unroll
                        for (int kk = 0; kk < K; ++kk) acc = fmaf(dq[kk], W[kk], acc);
                    } else {
                        const float* W = a.w[v] + (size_t)i * K * K + k * K;
#pragma unroll
                        for (int kk = 0; kk < K; ++kk) acc = fmaf(dq[kk], W[kk], acc);
                    }
                    dX[v * FK + idx] += acc;
                }
            sync();
        } else {
            for (unsigned i = lane; i < NV * FK; i += NT) dX[i] = 0.f;
            sync();
            for (unsigned p0 = 0; p0 < P; p0 += PPP) {
                const unsigned pair = p0 + pl;
                const bool ok = pair < P;
                unsigned i = 0, j = 0;
                float4 q = f4_zero(), gt = f4_zero();
                const float* W = a.w[v_l] + (size_t)(ok ? pair : 0) * K * K;
                if (ok) {
                    const unsigned ij = ptab[pair];
                    i = ij & 255u; j = ij >> 8;
                    const float4 g4 = *reinterpret_cast<const float4*>(gb + (size_t)pair * g_stride + c4 * 4);
                    const float4 xj = *reinterpret_cast<const float4*>(X + v_l * FK + j * K + k0);
                    const float* xi = X + v_l * FK + i * K;
                    float4 t = f4_zero();
#pragma unroll
                    for (int k = 0; k < K; ++k) t = f4_fma(*reinterpret_cast<const float4*>(W + k * K + k0), xi[k], t);
                    q = f4_mul(g4, xj);                    // d(x_i W_pair)[k0..k0+3]
                    gt = f4_mul(g4, t);                    // d x_j [k0..k0+3]
                    float* dj = dX + v_l * FK + j * K + k0;
                    lds_add(dj + 0, gt.x); lds_add(dj + 1, gt.y); lds_add(dj + 2, gt.z); lds_add(dj + 3, gt.w);
                }
                // dx_i[k] = sum_k' q[k'] W[k][k'] : partial dot on this lane's 4 k', reduced over LPV lanes
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    float s = ok ? f4_dot(q, *reinterpret_cast<const float4*>(W + k * K + k0)) : 0.f;
#pragma unroll
                    for (unsigned o = LPV / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                    if (ok && (c4 % LPV) == 0) lds_add(dX + v_l * FK + i * K + k, s);
                }
            }
            sync();
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float4* dr = reinterpret_cast<float4*>(a.dx[v] + (size_t)b * FK);
            for (unsigned i = lane; i < FK / 4; i += NT) dr[i] = reinterpret_cast<const float4*>(dX + v * FK)[i];
        }
        sync();
    }
}

// weight gradient: dW[m][k][k'] = sum over the items of matrix m of x_src[k] * q[k']
//   all        : items (b, i), i < n          q = dvW[b][i]       (M = 1)
//   each       : items (b, i = m)             q = dvW[b][m]       (M = n)
//   interaction: items (b, pair = m)          q = g[b][m] * x_j   (M = P),  src = i(pair)
// grid (M, S): split s handles examples [s*per, (s+1)*per); writes partials[s][m][K*K].
constexpr int kCH = 64;
struct WgradSets {
    const float* x[2];      // [B, F, K] per set
    const float* dvw[2];    // [B, n, K] per set (all / each)
};
template <int K>
__global__ __launch_bounds__(kThreads) void bilinear_wgrad_kernel(
    WgradSets sets, const float* __restrict__ g, unsigned g_stride, unsigned g_col0, unsigned B, unsigned F, int type,
    unsigned per, unsigned M, float* __restrict__ partials) {
    __shared__ float Xs[kCH][K + 1];
    __shared__ float Qs[kCH][K + 1];
    const unsigned n = F - 1, FK = F * K, nK = n * K;
    const unsigned m = blockIdx.x, s = blockIdx.y, set = blockIdx.z, nsets = gridDim.z;   // partial row s: [set][m][K*K]
    const float* __restrict__ x = sets.x[set];
    const float* __restrict__ dvw = sets.dvw[set];
    const unsigned g_col = g_col0 + set * K;
    const unsigned b_begin = s * per, b_end = min(B, b_begin + per);
    unsigned pi = 0, pj = 0;
    if (type == kInteraction) {                    // invert the triangular index of pair m
        unsigned i = 0, base = 0;
        while (base + (n - 1 - i) <= m) { base += n - 1 - i; ++i; }
        pi = i; pj = i + 1 + (m - base);
    }
    const unsigned rows_per_ex = type == kAll ? n : 1;
    const unsigned items = (b_end > b_begin ? b_end - b_begin : 0) * rows_per_ex;
    constexpr unsigned OPT = (K * K + kThreads - 1) / kThreads;     // outputs per thread
    float acc[OPT];
#pragma unroll
    for (unsigned o = 0; o < OPT; ++o) acc[o] = 0.f;
    const unsigned P = n * (n - 1) / 2;
    for (unsigned c0 = 0; c0 < items; c0 += kCH) {
        __syncthreads();
        for (unsigned e = threadIdx.x; e < kCH * K; e += kThreads) {
            const unsigned c = e / K, k = e % K, it = c0 + c;
            float xv = 0.f, qv = 0.f;
            if (it < items) {
                const unsigned b = b_begin + it / rows_per_ex;
                const unsigned r = type == kAll ? it % rows_per_ex : m;
                if (type == kInteraction) {
                    xv = x[(size_t)b * FK + pi * K + k];
                    qv = g[((size_t)b * P + m) * g_stride + g_col + k] * x[(size_t)b * FK + pj * K + k];
                } else {
                    xv = x[(size_t)b * FK + r * K + k];
                    qv = dvw[(size_t)b * nK + r * K + k];
                }
            }
            Xs[c][k] = xv;
            Qs[c][k] = qv;
        }
        __syncthreads();
#pragma unroll
        for (unsigned o = 0; o < OPT; ++o) {
            const unsigned idx = threadIdx.x + o * kThreads;
            if (idx < K * K) {
                const unsigned k = idx / K, kk = idx % K;
                float t = acc[o];
#pragma unroll 8
                for (unsigned c = 0; c < kCH; ++c) t = fmaf(Xs[c][k], Qs[c][kk], t);
                acc[o] = t;
            }
        }
    }
#pragma unroll
    for (unsigned o = 0; o < OPT; ++o) {
        const unsigned idx = threadIdx.x + o * kThreads;
        if (idx < K * K) partials[(((size_t)s * nsets + set) * M + m) * K * K + idx] = acc[o];
    }
}

// ---- host helpers ---------------------------------------------------------------------------
inline bool k_ok(int K) { return K == 4 || K == 8 || K == 16 || K == 32 || K == 64; }
inline int mats_of(int type, int F) {
    const int n = F - 1;
    return type == kAll ? 1 : (type == kEach ? n : n * (n - 1) / 2);
}
inline int wgrad_splits(int B, int M) {
    // M x S x sets workgroups.  The 'all' type has ONE matrix: 256 splits left a workgroup 16 examples = 7 dependent
    // load -> LDS -> accumulate rounds (17.8 us per set); 1024 splits are two rounds
    int S = cdiv(1024, M);
    if (S > 1024) S = 1024;
    if (S > B) S = B;
    return S < 1 ? 1 : S;
}
inline int grid_for(int B) {
    int gblocks = cdiv(B, kWaves);
    return gblocks > kMaxBlocks ? kMaxBlocks : gblocks;
}
inline size_t bi_smem(int F, int K, int nv, int type, bool bwd) {
    const unsigned n = F - 1, P = n * (n - 1) / 2;
    size_t fl = (size_t)ptab_floats(P) + (type == kAll ? (size_t)nv * K * kWS(K) : 0);
    fl += (size_t)kWaves * nv * (bwd ? 2 : 1) * ((size_t)F * K + (size_t)n * K);
    return fl * sizeof(float);
}
struct BiWs {
    size_t dvw, partials, total;
};
inline BiWs bi_ws(int B, int F, int K, int nv, int type) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    BiWs w;
    const size_t n = F - 1;
    w.dvw = 0;
    size_t off = type == kInteraction ? 0 : al((size_t)nv * B * n * K * sizeof(float));
    w.partials = off;
    const int M = mats_of(type, F);
    off += al((size_t)wgrad_splits(B, M) * nv * M * K * K * sizeof(float));
    w.total = off;
    return w;
}

#define ENSURE_SMEM(kern, bytes)                                                                       \
    do {                                                                                               \
        if ((bytes) > 160 * 1024) return (int)hipErrorInvalidValue;                                    \
        if ((bytes) > 64 * 1024) {                                                                     \
            hipError_t e__ = hipFuncSetAttribute(reinterpret_cast<const void*>(&kern),                 \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            if (e__ != hipSuccess) return (int)e__;                                                    \
        }                                                                                              \
    } while (0)

template <int K, int NV>
int launch_bi_fwd(const BiSets& a, int B, int F, int type, float* out, int out_stride, int out_col, hipStream_t st) {
    const size_t smem = bi_smem(F, K, NV, type, false);
    ENSURE_SMEM((bilinear_fwd_kernel<K, NV>), smem);
    hipLaunchKernelGGL((bilinear_fwd_kernel<K, NV>), dim3(grid_for(B)), dim3(kThreads), smem, st, a, (unsigned)B,
                       (unsigned)F, type, out, (unsigned)out_stride, (unsigned)out_col);
    return (int)hipGetLastError();
}
template <int K, int NV>
int launch_bi_bwd(const BiSets& a, int B, int F, int type, const float* g, int g_stride, int g_col, float* dvw,
                  hipStream_t st) {
    // LDS budget decides the waves per workgroup and whether the per-example gradient tile is staged
    const unsigned n = F - 1, P = n * (n - 1) / 2;
    const size_t shared = ((size_t)ptab_floats(P) + (type == kAll ? (size_t)NV * K * kWS(K) : 0)) * sizeof(float);
    const size_t base = (size_t)NV * 2 * ((size_t)F * K + (size_t)n * K) * sizeof(float);
    const size_t tile = (size_t)P * NV * K * sizeof(float);
    const size_t budget = 160 * 1024;
    int stage = 0, wpb = 0, wpb_budget_all = 1;
    if (type != kInteraction && shared + base + tile <= budget) {
        stage = 1;
        wpb = (int)((budget - shared) / (base + tile));
        wpb_budget_all = (int)(budget / (shared + base + tile));
    } else if (shared + base <= budget) {
        wpb = (int)((budget - shared) / base);
    }
    if (wpb < 1) return (int)hipErrorInvalidValue;
    if (wpb > kWaves) wpb = kWaves;
    const size_t smem = shared + (size_t)wpb * (base + (stage ? tile : 0));
    const unsigned C4 = NV * K / 4;
    if (stage && K <= 32 && P <= (kCoopThreads / C4) * 10 && (unsigned)F * K / 4 <= (unsigned)kCoopThreads) {
        // a workgroup per example (COOP): as many workgroups per CU as the LDS holds, persistent over the batch
        const size_t smem1 = shared + base + tile;
        ENSURE_SMEM((bilinear_bwd_kernel<K, NV, true>), smem1);
        int grid = 256 * wpb_budget_all;
        if (grid > B) grid = B;
        hipLaunchKernelGGL((bilinear_bwd_kernel<K, NV, true>), dim3(grid), dim3(kCoopThreads), smem1, st, a, (unsigned)B,
                           (unsigned)F, type, g, (unsigned)g_stride, (unsigned)g_col, dvw, 1);
        return (int)hipGetLastError();
    }
    ENSURE_SMEM((bilinear_bwd_kernel<K, NV, false>), smem);
    int grid = cdiv(B, wpb);
    if (grid > kMaxBlocks) grid = kMaxBlocks;
    hipLaunchKernelGGL((bilinear_bwd_kernel<K, NV, false>), dim3(grid), dim3(64 * wpb), smem, st, a, (unsigned)B,
                       (unsigned)F, type, g, (unsigned)g_stride, (unsigned)g_col, dvw, stage);
    return (int)hipGetLastError();
}
template <int K>
int launch_bi_wgrad(const WgradSets& sets, int nv, const float* g, int g_stride, int g_col, int B, int F, int type, int S,
                    int M, float* partials, hipStream_t st) {
    const unsigned per = (unsigned)cdiv(B, S);
    hipLaunchKernelGGL((bilinear_wgrad_kernel<K>), dim3(M, S, nv), dim3(kThreads), 0, st, sets, g, (unsigned)g_stride,
                       (unsigned)g_col, (unsigned)B, (unsigned)F, type, per, (unsigned)M, partials);
    return (int)hipGetLastError();
}

#define DISPATCH_K(K_, CALL)                     \
    switch (K_) {                                \
        case 4: { constexpr int KK = 4; CALL; } break;   \
        case 8: { constexpr int KK = 8; CALL; } break;   \
        case 16: { constexpr int KK = 16; CALL; } break; \
        case 32: { constexpr int KK = 32; CALL; } break; \
        default: { constexpr int KK = 64; CALL; } break; \
    }

}  // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
RECALGO_EXPORT int recalgo_senet_fwd(const float* emb, const float* w1, const float* w2, int B, int F, int K,
                                     int reduction_dim, float* v_out, float* a_out, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && F > 0 && K > 0 && reduction_dim > 0 && reduction_dim < K);
    if (B == 0) return 0;
    const size_t smem = ((size_t)kWaves * ((2 * (size_t)F * K + 3 * F + 2 * reduction_dim + 3) & ~(size_t)3) +
                         2 * (size_t)F * reduction_dim) * sizeof(float);
    ENSURE_SMEM(senet_fwd_kernel, smem);
    hipLaunchKernelGGL(senet_fwd_kernel, dim3(grid_for(B)), dim3(kThreads), smem, as_stream(stream), emb, w1, w2,
                       (unsigned)B, (unsigned)F, (unsigned)K, (unsigned)reduction_dim, v_out, a_out);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int64_t recalgo_senet_bwd_workspace_bytes(int B, int F, int K, int reduction_dim) {
    if (B <= 0 || F <= 0 || reduction_dim <= 0) return 0;
    return (int64_t)grid_for(B) * 2 * F * reduction_dim * (int64_t)sizeof(float);
}

RECALGO_EXPORT int recalgo_senet_bwd(const float* emb, const float* w1, const float* w2, const float* g_v, int B,
                                     int F, int K, int reduction_dim, float* d_emb, int accumulate, float* dw1,
                                     float* dw2, void* workspace, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B > 0 && F > 0 && K > 0 && reduction_dim > 0 && reduction_dim < K && workspace != nullptr);
    const unsigned WR = (unsigned)F * reduction_dim;
    const size_t smem = ((size_t)kWaves * ((2 * (size_t)F * K + 3 * F + 2 * reduction_dim + 3) & ~(size_t)3) +
                         (size_t)kWaves * 2 * WR + 2 * (size_t)WR) * sizeof(float);
    ENSURE_SMEM(senet_bwd_kernel, smem);
    const int grid = grid_for(B);        // one partial row per workgroup; up to 1024 workgroups = 16 waves per CU in flight (256 left
                                         // one wave per SIMD walking its four examples' latency chains alone: 44 us)
    float* partials = static_cast<float*>(workspace);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(senet_bwd_kernel, dim3(grid), dim3(kThreads), smem, st, emb, w1, w2, g_v, (unsigned)B,
                       (unsigned)F, (unsigned)K, (unsigned)reduction_dim, d_emb, accumulate, partials);
    launch_colsum16(partials, (unsigned)grid, 2 * WR, dw1, WR, dw2, st);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_bilinear_fwd(const float* x0, const float* w0, const float* x1, const float* w1, int B,
                                        int F, int K, int type, float* out, int out_stride, int out_col,
                                        recalgo_stream_t stream) {
    const int nv = x1 ? 2 : 1;
    RECALGO_REQUIRE(B >= 0 && F >= 3 && F <= 128 && k_ok(K) && type >= kAll && type <= kInteraction);
    RECALGO_REQUIRE(x0 && w0 && (x1 == nullptr) == (w1 == nullptr));
    RECALGO_REQUIRE(out_stride % 4 == 0 && out_col % 4 == 0 && out_stride >= out_col + nv * K);
    if (B == 0) return 0;
    BiSets a{{x0, x1}, {w0, w1}, {nullptr, nullptr}};
    hipStream_t st = as_stream(stream);
    int rc;
    if (nv == 2) { DISPATCH_K(K, rc = (launch_bi_fwd<KK, 2>(a, B, F, type, out, out_stride, out_col, st))); }
    else { DISPATCH_K(K, rc = (launch_bi_fwd<KK, 1>(a, B, F, type, out, out_stride, out_col, st))); }
    return rc;
}

RECALGO_EXPORT int64_t recalgo_bilinear_bwd_workspace_bytes(int B, int F, int K, int n_sets, int type) {
    if (B <= 0 || F < 3 || !k_ok(K) || n_sets < 1 || n_sets > 2 || type < kAll || type > kInteraction) return 0;
    return (int64_t)bi_ws(B, F, K, n_sets, type).total;
}

RECALGO_EXPORT int recalgo_bilinear_bwd(const float* x0, const float* w0, const float* x1, const float* w1,
                                        const float* g, int g_stride, int g_col, int B, int F, int K, int type,
                                        float* dx0, float* dw0, float* dx1, float* dw1, void* workspace,
                                        recalgo_stream_t stream) {
    const int nv = x1 ? 2 : 1;
    RECALGO_REQUIRE(B > 0 && F >= 3 && F <= 128 && k_ok(K) && type >= kAll && type <= kInteraction);
    RECALGO_REQUIRE(x0 && w0 && dx0 && dw0 && g && workspace);
    RECALGO_REQUIRE((x1 == nullptr) == (w1 == nullptr) && (x1 == nullptr) == (dx1 == nullptr) &&
                    (x1 == nullptr) == (dw1 == nullptr));
    RECALGO_REQUIRE(g_stride % 4 == 0 && g_col % 4 == 0 && g_stride >= g_col + nv * K);
    const BiWs ws = bi_ws(B, F, K, nv, type);
    char* base = static_cast<char*>(workspace);
    float* dvw = reinterpret_cast<float*>(base + ws.dvw);
    float* partials = reinterpret_cast<float*>(base + ws.partials);
    BiSets a{{x0, x1}, {w0, w1}, {dx0, dx1}};
    hipStream_t st = as_stream(stream);
    int rc;
    if (nv == 2) { DISPATCH_K(K, rc = (launch_bi_bwd<KK, 2>(a, B, F, type, g, g_stride, g_col, dvw, st))); }
    else { DISPATCH_K(K, rc = (launch_bi_bwd<KK, 1>(a, B, F, type, g, g_stride, g_col, dvw, st))); }
    if (rc) return rc;
    const int M = mats_of(type, F);
    const int S = wgrad_splits(B, M);
    const size_t nK = (size_t)(F - 1) * K;
    const unsigned wn = (unsigned)M * K * K;
    // both sets in ONE grid (z = set) and one column-sum launch over the partial rows [S][set][M][K*K]
    const WgradSets sets{{x0, x1}, {dvw, dvw ? dvw + (size_t)B * nK : nullptr}};
    DISPATCH_K(K, rc = (launch_bi_wgrad<KK>(sets, nv, g, g_stride, g_col, B, F, type, S, M, partials, st)));
    if (rc) return rc;
    launch_colsum16(partials, (unsigned)S, (unsigned)nv * wn, dw0, wn, dw1, st);
    RECALGO_RETURN_LAST();
}
