// K6: PNN product layer (algorithm/PNN/pnn.py:133-181), gfx950.
//
// The reference builds lp with a Python loop of D (=1024) sub-graphs:
//   IPNN (pnn.py:146-158)  lp_i = || sum_f theta[i,f] e_f ||^2
//   OPNN (pnn.py:160-173)  lp_i = sum_{a,c} (s s^T)[a,c] * sym(W_i)[a,c],  s = sum_f e_f,
//                          sym(W) = triu(W) + triu(W)^T - diag(W)
// Both are quadratic forms in per-example second-order statistics, so the layer factors into
//   phi[b, t]   per-example features over the upper triangle t = (r <= r') of a Gram matrix
//               IPNN: <e_r, e_r'> over R = F rows;   OPNN: s_r * s_r' over R = K "rows" of width 1
//   omega[t, i] batch-constant weights  c_t * theta[i,r] theta[i,r']  |  c_t * W_i[r,r']
//               (c_t = 1 on the diagonal, 2 off it)
//   lp = phi @ omega                     one plain library GEMM (hipBLASLt), fused by the host
//                                        layer with lz = E @ linear_w, the bias and the ReLU.
// The hand-written kernels here are the per-example feature builders (HBM-bound: 1.6 KB in,
// 1.4 KB out per example at F=26, K=16; the tile lives in LDS, one wave per example) and the
// weight builders, forward and backward.  The D-way contraction is GEMM-shaped and is left to the
// library, instead of D separate reductions.
//
// Upper-triangle index: t(r, r') = r*R - r(r-1)/2 + (r' - r), r <= r', T = R(R+1)/2.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kMaxBlocks = 2048;

enum { kIPNN = 0, kOPNN = 1 };

__host__ __device__ inline unsigned tri_t(unsigned r, unsigned c, unsigned R) { return r * R - r * (r - 1) / 2 + (c - r); }

__device__ __forceinline__ void build_tri_table(unsigned short* tab, unsigned R) {
    for (unsigned r = threadIdx.x; r < R; r += kThreads)
        for (unsigned c = r; c < R; ++c) tab[tri_t(r, c, R)] = (unsigned short)(r | (c << 8));
}
__host__ __device__ inline unsigned tab_floats(unsigned T) { return ((T * 2 + 15) / 16) * 4; }

// IPNN features: phi[b, t(f,f')] = <e_f, e_f'>
__global__ __launch_bounds__(kThreads) void ipnn_features_fwd_kernel(const float* __restrict__ emb, unsigned B,
                                                                     unsigned F, unsigned K,
                                                                     float* __restrict__ phi, unsigned ld) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned T = F * (F + 1) / 2, FK = F * K, KS = K + 1;
    unsigned short* tab = reinterpret_cast<unsigned short*>(smem);
    const unsigned lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    float* X = smem + tab_floats(T) + wib * F * KS;          // [F][K+1] (padded: conflict-free row dots)
    build_tri_table(tab, F);
    __syncthreads();
    for (unsigned b = blockIdx.x * kWaves + wib; b < B; b += gridDim.x * kWaves) {
        const float* er = emb + (size_t)b * FK;
        for (unsigned i0 = 0; i0 < FK; i0 += 256) {           // 4 loads in flight per lane, then the LDS stores
            float t[4];
#pragma unroll
            for (unsigned u = 0; u < 4; ++u) {
                const unsigned i = i0 + u * 64 + lane;
                t[u] = i < FK ? er[i] : 0.f;
            }
#pragma unroll
            for (unsigned u = 0; u < 4; ++u) {
                const unsigned i = i0 + u * 64 + lane;
                if (i < FK) X[(i / K) * KS + (i % K)] = t[u];
            }
        }
        __builtin_amdgcn_wave_barrier();
        float* pr = phi + (size_t)b * ld;
        for (unsigned t = lane; t < ld; t += 64) {            // columns [T, ld): zero padding of the row
            const unsigned rc = tab[t < T ? t : 0], r = rc & 255u, c = rc >> 8;
            const float* xr = X + r * KS;
            const float* xc = X + c * KS;
            float acc = 0.f;
#pragma unroll 8
            for (unsigned k = 0; k < K; ++k) acc = fmaf(xr[k], xc[k], acc);
            pr[t] = t < T ? acc : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// d_emb[b,f,:] (=|+=) sum_f' dG[f,f'] e_f',  dG[f,f'] = dphi[t(f,f')] (f != f') | 2 dphi[t(f,f)]
__global__ __launch_bounds__(kThreads) void ipnn_features_bwd_kernel(const float* __restrict__ emb,
                                                                     const float* __restrict__ dphi, unsigned B,
                                                                     unsigned F, unsigned K, unsigned ld,
                                                                     float* __restrict__ d_emb, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned T = F * (F + 1) / 2, FK = F * K;
    const unsigned lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    float* X = smem + wib * (FK + T);                        // [F][K]
    float* dP = X + FK;                                      // [T]
    for (unsigned b = blockIdx.x * kWaves + wib; b < B; b += gridDim.x * kWaves) {
        const float* er = emb + (size_t)b * FK;
        const float* pr = dphi + (size_t)b * ld;
        for (unsigned i0 = 0; i0 < FK + T; i0 += 256) {       // X and dP are adjacent in LDS: one batched copy
            float t[4];
#pragma unroll
            for (unsigned u = 0; u < 4; ++u) {
                const unsigned i = i0 + u * 64 + lane;
                t[u] = i < FK ? er[i] : (i < FK + T ? pr[i - FK] : 0.f);
            }
#pragma unroll
            for (unsigned u = 0; u < 4; ++u) {
                const unsigned i = i0 + u * 64 + lane;
                if (i < FK + T) X[i] = t[u];
            }
        }
        __builtin_amdgcn_wave_barrier();
        float* dr = d_emb + (size_t)b * FK;
        for (unsigned i = lane; i < FK; i += 64) {
            const unsigned f = i / K, k = i % K;
            float acc = 0.f;
#pragma unroll 8
            for (unsigned c = 0; c < F; ++c) {
                const unsigned t = c >= f ? tri_t(f, c, F) : tri_t(c, f, F);
                const float w = c == f ? 2.f * dP[t] : dP[t];
                acc = fmaf(w, X[c * K + k], acc);
            }
            dr[i] = accumulate ? dr[i] + acc : acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// The same for K % 4 == 0 (every configuration of the scripts).  The kernel above spends its time on the triangle index
// (ten integer instructions per fma) and on three dependent rounds of scalar loads: 28 us for 19 MB.  Here the gradient
// row is spread ONCE per example into the symmetric [F][F] matrix dG (diagonal doubled — the value the walk above forms
// per step), all of an example's loads are requested together, and a lane owns four columns of one field: per step one
// broadcast ds_read_b32 of dG[f][c] and one ds_read_b128 of e_c feed four chains — each the same c-ascending chain of fmas
// as above (bit-identical results).
__global__ __launch_bounds__(kThreads) void ipnn_features_bwd4_kernel(const float* __restrict__ emb,
                                                                      const float* __restrict__ dphi, unsigned B,
                                                                      unsigned F, unsigned K, unsigned ld,
                                                                      float* __restrict__ d_emb, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned T = F * (F + 1) / 2, FK = F * K, K4 = K / 4, FK4 = FK / 4;
    unsigned short* tab = reinterpret_cast<unsigned short*>(smem);
    const unsigned lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const unsigned per_wave = FK + ((F * F + 3) & ~3u);
    float* X = smem + tab_floats(T) + wib * per_wave;        // [F][K]
    float* dG = X + FK;                                      // [F][F]
    build_tri_table(tab, F);
    __syncthreads();
    constexpr unsigned kXU = 4, kPU = 8;                     // float4 of e / floats of dphi requested per lane and round
    for (unsigned b = blockIdx.x * kWaves + wib; b < B; b += gridDim.x * kWaves) {
        const float4* er = reinterpret_cast<const float4*>(emb + (size_t)b * FK);
        const float* pr = dphi + (size_t)b * ld;
        for (unsigned i0 = 0, t0 = 0; i0 < FK4 || t0 < T; i0 += 64 * kXU, t0 += 64 * kPU) {
            float4 xv[kXU];
            float pv[kPU];
#pragma unroll
            for (unsigned u = 0; u < kXU; ++u) {
                const unsigned i = i0 + u * 64 + lane;
                xv[u] = i < FK4 ? er[i] : f4_zero();
            }
#pragma unroll
            for (unsigned u = 0; u < kPU; ++u) {
                const unsigned t = t0 + u * 64 + lane;
                pv[u] = t < T ? pr[t] : 0.f;
            }
#pragma unroll
            for (unsigned u = 0; u < kXU; ++u) {
                const unsigned i = i0 + u * 64 + lane;
                if (i < FK4) reinterpret_cast<float4*>(X)[i] = xv[u];
            }
#pragma unroll
            for (unsigned u = 0; u < kPU; ++u) {
                const unsigned t = t0 + u * 64 + lane;
                if (t < T) {
                    const unsigned rc = tab[t], r = rc & 255u, c = rc >> 8;
                    if (r == c) {
                        dG[r * F + r] = 2.f * pv[u];
                    } else {
                        dG[r * F + c] = pv[u];
                        dG[c * F + r] = pv[u];
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        float4* dr = reinterpret_cast<float4*>(d_emb + (size_t)b * FK);
        for (unsigned i = lane; i < FK4; i += 64) {
            const unsigned f = i / K4, q = i % K4;
            const float* gr = dG + f * F;
            const float4* xc = reinterpret_cast<const float4*>(X) + q;
            float4 acc = f4_zero();
#pragma unroll 8
            for (unsigned c = 0; c < F; ++c) {
                const float w = gr[c];
                const float4 x = xc[c * K4];
                acc.x = fmaf(w, x.x, acc.x); acc.y = fmaf(w, x.y, acc.y);
                acc.z = fmaf(w, x.z, acc.z); acc.w = fmaf(w, x.w, acc.w);
            }
            if (accumulate) {
                const float4 o = dr[i];
                acc.x = o.x + acc.x; acc.y = o.y + acc.y; acc.z = o.z + acc.z; acc.w = o.w + acc.w;
            }
            dr[i] = acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// OPNN features: s = sum_f e_f ; phi[b, t(a,c)] = s_a s_c
__global__ __launch_bounds__(kThreads) void opnn_features_fwd_kernel(const float* __restrict__ emb, unsigned B,
                                                                     unsigned F, unsigned K,
                                                                     float* __restrict__ phi, unsigned ld,
                                                                     float* __restrict__ s_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned T = K * (K + 1) / 2, FK = F * K;
    unsigned short* tab = reinterpret_cast<unsigned short*>(smem);
    const unsigned lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    float* S = smem + tab_floats(T) + wib * K;
    build_tri_table(tab, K);
    __syncthreads();
    for (unsigned b = blockIdx.x * kWaves + wib; b < B; b += gridDim.x * kWaves) {
        const float* er = emb + (size_t)b * FK;
        for (unsigned k = lane; k < K; k += 64) {
            float acc = 0.f;
            for (unsigned f = 0; f < F; ++f) acc += er[f * K + k];        // tf.reduce_sum(axis=1), field order
            S[k] = acc;
            if (s_out) s_out[(size_t)b * K + k] = acc;
        }
        __builtin_amdgcn_wave_barrier();
        float* pr = phi + (size_t)b * ld;
        for (unsigned t = lane; t < ld; t += 64) {
            const unsigned rc = tab[t < T ? t : 0];
            pr[t] = t < T ? S[rc & 255u] * S[rc >> 8] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ds_a = sum_c dphi_sym[a,c] s_c ; d_emb[b,f,:] (=|+=) ds for every f
__global__ __launch_bounds__(kThreads) void opnn_features_bwd_kernel(const float* __restrict__ emb,
                                                                     const float* __restrict__ dphi, unsigned B,
                                                                     unsigned F, unsigned K, unsigned ld,
                                                                     float* __restrict__ d_emb, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned T = K * (K + 1) / 2, FK = F * K;
    const unsigned lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    float* S = smem + wib * (2 * K + T);
    float* dS = S + K;
    float* dP = dS + K;
    for (unsigned b = blockIdx.x * kWaves + wib; b < B; b += gridDim.x * kWaves) {
        const float* er = emb + (size_t)b * FK;
        const float* pr = dphi + (size_t)b * ld;
        for (unsigned k = lane; k < K; k += 64) {
            float acc = 0.f;
            for (unsigned f = 0; f < F; ++f) acc += er[f * K + k];
            S[k] = acc;
        }
        for (unsigned t = lane; t < T; t += 64) dP[t] = pr[t];
        __builtin_amdgcn_wave_barrier();
        for (unsigned a = lane; a < K; a += 64) {
            float acc = 0.f;
            for (unsigned c = 0; c < K; ++c) {
                const unsigned t = c >= a ? tri_t(a, c, K) : tri_t(c, a, K);
                acc = fmaf(c == a ? 2.f * dP[t] : dP[t], S[c], acc);
            }
            dS[a] = acc;
        }
        __builtin_amdgcn_wave_barrier();
        float* dr = d_emb + (size_t)b * FK;
        for (unsigned i = lane; i < FK; i += 64) {
            const float v = dS[i % K];
            dr[i] = accumulate ? dr[i] + v : v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// omega[t, i]:  IPNN c_t * theta[i,r] * theta[i,c]   (theta = product_w [D, F])
//               OPNN c_t * W[i, r, c]                (W = product_w [D, K, K], upper triangle only)
__global__ __launch_bounds__(256) void pnn_weights_fwd_kernel(const float* __restrict__ pw, unsigned D, unsigned R,
                                                              int method, float* __restrict__ omega) {
    const unsigned T = R * (R + 1) / 2;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)T * D) return;
    const unsigned i = (unsigned)(idx % D), t = (unsigned)(idx / D);
    unsigned r = 0, base = 0;                                // invert t -> (r, c)
    while (base + (R - r) <= t) { base += R - r; ++r; }
    const unsigned c = r + (t - base);
    const float ct = r == c ? 1.f : 2.f;
    float v;
    if (method == kIPNN) v = ct * pw[(size_t)i * R + r] * pw[(size_t)i * R + c];
    else v = ct * pw[((size_t)i * R + r) * R + c];
    omega[idx] = v;
}

// IPNN: d theta[i,f] = 2 * sum_f' domega[t(f,f'), i] * theta[i,f']
// OPNN: dW[i,a,c] = c_t * domega[t(a,c), i] (a <= c), 0 below the diagonal (quirk B-10)
__global__ __launch_bounds__(256) void pnn_weights_bwd_kernel(const float* __restrict__ pw,
                                                              const float* __restrict__ domega, unsigned D,
                                                              unsigned R, int method, float* __restrict__ dpw) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (method == kIPNN) {
        if (idx >= (size_t)D * R) return;
        // consecutive threads walk i: the R loads of d(omega) per thread are coalesced rows of [T, D] (with f fastest a
        // wave's load touched 64 rows: 12 us for 1.4 MB); theta [D, R] is small and L2-resident
        const unsigned f = (unsigned)(idx / D), i = (unsigned)(idx % D);
        float acc = 0.f;
#pragma unroll 4
        for (unsigned c = 0; c < R; ++c) {
            const unsigned t = c >= f ? tri_t(f, c, R) : tri_t(c, f, R);
            acc = fmaf(domega[(size_t)t * D + i], pw[(size_t)i * R + c], acc);
        }
        dpw[(size_t)i * R + f] = 2.f * acc;
    } else {
        if (idx >= (size_t)D * R * R) return;
        const unsigned c = (unsigned)(idx % R), a = (unsigned)((idx / R) % R), i = (unsigned)(idx / ((size_t)R * R));
        dpw[idx] = a > c ? 0.f : (a == c ? 1.f : 2.f) * domega[(size_t)tri_t(a, c, R) * D + i];
    }
}

inline int grid_for(int B) {
    int gblocks = cdiv(B, kWaves);
    return gblocks > kMaxBlocks ? kMaxBlocks : gblocks;
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

}  // namespace

RECALGO_EXPORT int recalgo_pnn_feature_count(int F, int K, int method) {
    if (method == kIPNN) return F > 0 ? F * (F + 1) / 2 : 0;
    if (method == kOPNN) return K > 0 ? K * (K + 1) / 2 : 0;
    return 0;
}

RECALGO_EXPORT int recalgo_pnn_features_fwd(const float* emb, int B, int F, int K, int method, float* phi, int ld_phi,
                                            recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && F > 0 && K > 0 && (method == kIPNN || method == kOPNN));
    RECALGO_REQUIRE((method == kIPNN ? F : K) <= 255);
    RECALGO_REQUIRE(ld_phi >= recalgo_pnn_feature_count(F, K, method));
    if (B == 0) return 0;
    hipStream_t st = as_stream(stream);
    if (method == kIPNN) {
        const unsigned T = (unsigned)F * (F + 1) / 2;
        const size_t smem = ((size_t)tab_floats(T) + (size_t)kWaves * F * (K + 1)) * sizeof(float);
        ENSURE_SMEM(ipnn_features_fwd_kernel, smem);
        hipLaunchKernelGGL(ipnn_features_fwd_kernel, dim3(grid_for(B)), dim3(kThreads), smem, st, emb, (unsigned)B,
                           (unsigned)F, (unsigned)K, phi, (unsigned)ld_phi);
    } else {
        const unsigned T = (unsigned)K * (K + 1) / 2;
        const size_t smem = ((size_t)tab_floats(T) + (size_t)kWaves * K) * sizeof(float);
        ENSURE_SMEM(opnn_features_fwd_kernel, smem);
        hipLaunchKernelGGL(opnn_features_fwd_kernel, dim3(grid_for(B)), dim3(kThreads), smem, st, emb, (unsigned)B,
                           (unsigned)F, (unsigned)K, phi, (unsigned)ld_phi, static_cast<float*>(nullptr));
    }
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_pnn_features_bwd(const float* emb, const float* dphi, int ld_dphi, int B, int F, int K, int method,
                                            float* d_emb, int accumulate, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && F > 0 && K > 0 && (method == kIPNN || method == kOPNN));
    RECALGO_REQUIRE((method == kIPNN ? F : K) <= 255);
    RECALGO_REQUIRE(ld_dphi >= recalgo_pnn_feature_count(F, K, method));
    if (B == 0) return 0;
    hipStream_t st = as_stream(stream);
    if (method == kIPNN) {
        const unsigned T = (unsigned)F * (F + 1) / 2;
        const size_t smem4 = ((size_t)tab_floats(T) + (size_t)kWaves * ((size_t)F * K + (((size_t)F * F + 3) & ~(size_t)3))) * sizeof(float);
        if (K % 4 == 0 && smem4 <= 160 * 1024 && (reinterpret_cast<uintptr_t>(emb) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(d_emb) & 15) == 0) {
            ENSURE_SMEM(ipnn_features_bwd4_kernel, smem4);
            hipLaunchKernelGGL(ipnn_features_bwd4_kernel, dim3(grid_for(B)), dim3(kThreads), smem4, st, emb, dphi,
                               (unsigned)B, (unsigned)F, (unsigned)K, (unsigned)ld_dphi, d_emb, accumulate);
            RECALGO_RETURN_LAST();
        }
        const size_t smem = (size_t)kWaves * ((size_t)F * K + T) * sizeof(float);
        ENSURE_SMEM(ipnn_features_bwd_kernel, smem);
        hipLaunchKernelGGL(ipnn_features_bwd_kernel, dim3(grid_for(B)), dim3(kThreads), smem, st, emb, dphi,
                           (unsigned)B, (unsigned)F, (unsigned)K, (unsigned)ld_dphi, d_emb, accumulate);
    } else {
        const unsigned T = (unsigned)K * (K + 1) / 2;
        const size_t smem = (size_t)kWaves * (2 * (size_t)K + T) * sizeof(float);
        ENSURE_SMEM(opnn_features_bwd_kernel, smem);
        hipLaunchKernelGGL(opnn_features_bwd_kernel, dim3(grid_for(B)), dim3(kThreads), smem, st, emb, dphi,
                           (unsigned)B, (unsigned)F, (unsigned)K, (unsigned)ld_dphi, d_emb, accumulate);
    }
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_pnn_weights_fwd(const float* product_w, int D, int F, int K, int method, float* omega,
                                           recalgo_stream_t stream) {
    RECALGO_REQUIRE(D > 0 && F > 0 && K > 0 && (method == kIPNN || method == kOPNN));
    const unsigned R = method == kIPNN ? F : K;
    const size_t n = (size_t)R * (R + 1) / 2 * D;
    hipLaunchKernelGGL(pnn_weights_fwd_kernel, dim3(cdiv((int64_t)n, 256)), dim3(256), 0, as_stream(stream),
                       product_w, (unsigned)D, R, method, omega);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_pnn_weights_bwd(const float* product_w, const float* domega, int D, int F, int K,
                                           int method, float* d_product_w, recalgo_stream_t stream) {
    RECALGO_REQUIRE(D > 0 && F > 0 && K > 0 && (method == kIPNN || method == kOPNN));
    const unsigned R = method == kIPNN ? F : K;
    const size_t n = method == kIPNN ? (size_t)D * R : (size_t)D * R * R;
    hipLaunchKernelGGL(pnn_weights_bwd_kernel, dim3(cdiv((int64_t)n, 256)), dim3(256), 0, as_stream(stream),
                       product_w, domega, (unsigned)D, R, method, d_product_w);
    RECALGO_RETURN_LAST();
}
