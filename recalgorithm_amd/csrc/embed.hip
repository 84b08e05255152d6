// Embedding gathers (K1, K1m, K1s) and the fused DeepFM sparse path (K1+K2+K3), gfx950.
//
// All of these are HBM-bound: a 16-float row is 64 B, fetched as 4 x float4 by 4 adjacent
// lanes, so one wave instruction moves 16 independent rows.  The output side is written
// fully coalesced (float4 per lane, consecutive lanes -> consecutive addresses).
#include "common.h"

namespace {

constexpr int kThreads = 256;

// ---------------------------------------------------------------------------------------
// K1 forward: one float4 of one (b, f) row per thread.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void gather_fwd_kernel(
    const int64_t* __restrict__ ids, const float4* __restrict__ arena,
    const int64_t* __restrict__ row_base, unsigned total4, unsigned F, unsigned K4,
    float* __restrict__ out, unsigned out_stride, unsigned out_col) {
    unsigned i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= total4) return;
    unsigned row = i / K4;          // flattened (b, f)
    unsigned q = i - row * K4;
    unsigned b = row / F;
    unsigned f = row - b * F;
    int64_t id = ids[row];
    float4 v = f4_zero();
    if (id >= 0) v = arena[(row_base[f] + id) * K4 + q];
    *reinterpret_cast<float4*>(out + (size_t)b * out_stride + out_col + (f * K4 + q) * 4) = v;
}

__global__ __launch_bounds__(kThreads) void gather_bwd_kernel(
    const int64_t* __restrict__ ids, const float* __restrict__ g,
    const int64_t* __restrict__ row_base, unsigned total4, unsigned F, unsigned K4,
    unsigned g_stride, unsigned g_col, float* __restrict__ grad_arena) {
    unsigned i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= total4) return;
    unsigned row = i / K4;
    unsigned q = i - row * K4;
    unsigned b = row / F;
    unsigned f = row - b * F;
    int64_t id = ids[row];
    if (id < 0) return;
    float4 v = *reinterpret_cast<const float4*>(g + (size_t)b * g_stride + g_col + (f * K4 + q) * 4);
    float* dst = grad_arena + ((row_base[f] + id) * K4 + q) * 4;
    atomic_add_f32(dst + 0, v.x);
    atomic_add_f32(dst + 1, v.y);
    atomic_add_f32(dst + 2, v.z);
    atomic_add_f32(dst + 3, v.w);
}

// ---------------------------------------------------------------------------------------
// K1m: mean-combined bags.  One thread per (bag, float4 column chunk); the bag is walked
// sequentially so the fp32 sum order is the bag order (TF SparseSegmentMean).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void bag_mean_fwd_kernel(
    const int64_t* __restrict__ values, const int64_t* __restrict__ offsets,
    const float4* __restrict__ table, unsigned total4, unsigned K4, float* __restrict__ out,
    unsigned out_stride, unsigned out_col) {
    unsigned i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= total4) return;
    unsigned b = i / K4;
    unsigned q = i - b * K4;
    int64_t beg = offsets[b], end = offsets[b + 1];
    float4 acc = f4_zero();
    int cnt = 0;
    for (int64_t j = beg; j < end; ++j) {
        int64_t id = values[j];
        if (id >= 0) {
            acc = f4_add(acc, table[id * K4 + q]);
            ++cnt;
        }
    }
    if (cnt > 0) {
        float c = (float)cnt;
        acc = make_float4(acc.x / c, acc.y / c, acc.z / c, acc.w / c);
    }
    *reinterpret_cast<float4*>(out + (size_t)b * out_stride + out_col + q * 4) = acc;
}

__global__ __launch_bounds__(kThreads) void bag_mean_bwd_kernel(
    const int64_t* __restrict__ values, const int64_t* __restrict__ offsets,
    const float* __restrict__ g, unsigned total4, unsigned K4, unsigned g_stride, unsigned g_col,
    float* __restrict__ grad_table) {
    unsigned i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= total4) return;
    unsigned b = i / K4;
    unsigned q = i - b * K4;
    int64_t beg = offsets[b], end = offsets[b + 1];
    int cnt = 0;
    for (int64_t j = beg; j < end; ++j) cnt += values[j] >= 0;
    if (cnt == 0) return;
    float4 v = *reinterpret_cast<const float4*>(g + (size_t)b * g_stride + g_col + q * 4);
    float c = (float)cnt;
    v = make_float4(v.x / c, v.y / c, v.z / c, v.w / c);
    for (int64_t j = beg; j < end; ++j) {
        int64_t id = values[j];
        if (id < 0) continue;
        float* dst = grad_table + (id * K4 + q) * 4;
        atomic_add_f32(dst + 0, v.x);
        atomic_add_f32(dst + 1, v.y);
        atomic_add_f32(dst + 2, v.z);
        atomic_add_f32(dst + 3, v.w);
    }
}

// ---------------------------------------------------------------------------------------
// K1s: zero padded sequence gather (B, T, K).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void seq_gather_fwd_kernel(
    const int64_t* __restrict__ values, const int64_t* __restrict__ offsets,
    const float4* __restrict__ table, unsigned total4, unsigned T, unsigned K4,
    float4* __restrict__ out, int32_t* __restrict__ seq_len) {
    unsigned i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= total4) return;
    unsigned row = i / K4;  // (b, t)
    unsigned q = i - row * K4;
    unsigned b = row / T;
    unsigned t = row - b * T;
    int64_t beg = offsets[b];
    int64_t len = offsets[b + 1] - beg;
    if (t == 0 && q == 0) seq_len[b] = (int32_t)(len < (int64_t)T ? len : (int64_t)T);
    float4 v = f4_zero();
    if ((int64_t)t < len) {
        int64_t id = values[beg + t];
        if (id >= 0) v = table[id * K4 + q];
    }
    out[i] = v;
}

__global__ __launch_bounds__(kThreads) void seq_gather_bwd_kernel(
    const int64_t* __restrict__ values, const int64_t* __restrict__ offsets,
    const float4* __restrict__ g, unsigned total4, unsigned T, unsigned K4,
    float* __restrict__ grad_table) {
    unsigned i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= total4) return;
    unsigned row = i / K4;
    unsigned q = i - row * K4;
    unsigned b = row / T;
    unsigned t = row - b * T;
    int64_t beg = offsets[b];
    int64_t len = offsets[b + 1] - beg;
    if ((int64_t)t >= len) return;
    int64_t id = values[beg + t];
    if (id < 0) return;
    float4 v = g[i];
    float* dst = grad_table + (id * K4 + q) * 4;
    atomic_add_f32(dst + 0, v.x);
    atomic_add_f32(dst + 1, v.y);
    atomic_add_f32(dst + 2, v.z);
    atomic_add_f32(dst + 3, v.w);
}

// ---------------------------------------------------------------------------------------
// DeepFM sparse path, fused.  One workgroup owns EB examples.  Phase 1 gathers the EB*F rows
// (float4 per lane), streams them to `emb` and parks them in an LDS tile; phase 2 gives every
// example a 16-lane group that walks the F fields in LDS (sum and sum of squares per k), then
// shuffle-reduces over k.  Example stride in LDS is padded by 16 floats so that the two
// examples sharing a 32-lane ds_read_b32 group land on disjoint banks.
// ---------------------------------------------------------------------------------------
template <int EB>
__global__ __launch_bounds__(kThreads) void deepfm_sparse_fwd_kernel(
    const int64_t* __restrict__ ids, const float4* __restrict__ arena,
    const float* __restrict__ w1, const float* __restrict__ bias,
    const int64_t* __restrict__ row_base, unsigned B, unsigned F, unsigned K4,
    float4* __restrict__ emb, float* __restrict__ fm1, float* __restrict__ fm2) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned K = K4 * 4;
    const unsigned FK = F * K;
    const unsigned ex_stride = FK + 16;          // floats
    float* tile = smem;                          // [EB][ex_stride]
    float* w1s = smem + EB * ex_stride;          // [EB][F]
    const unsigned b0 = blockIdx.x * EB;
    const unsigned nex = min((unsigned)EB, B - b0);
    const unsigned per_ex4 = F * K4;
    const unsigned total4 = nex * per_ex4;

    for (unsigned i = threadIdx.x; i < total4; i += kThreads) {
        unsigned e = i / per_ex4;
        unsigned r = i - e * per_ex4;            // f*K4 + q
        unsigned f = r / K4;
        unsigned q = r - f * K4;
        size_t grow = (size_t)(b0 + e) * F + f;
        int64_t id = ids[grow];
        float4 v = f4_zero();
        int64_t arow = 0;
        if (id >= 0) {
            arow = row_base[f] + id;
            v = arena[arow * K4 + q];
        }
        emb[grow * K4 + q] = v;
        *reinterpret_cast<float4*>(tile + e * ex_stride + r * 4) = v;
        if (q == 0) w1s[e * F + f] = (id >= 0) ? w1[arow] : 0.f;
    }
    __syncthreads();

    // phase 2: 16 lanes per example
    const unsigned e = threadIdx.x >> 4;
    const unsigned l16 = threadIdx.x & 15;
    if (e < EB) {
        float acc2 = 0.f, acc1 = 0.f;
        if (e < nex) {
            const float* te = tile + e * ex_stride;
            for (unsigned k = l16; k < K; k += 16) {
                float s = 0.f, sq = 0.f;
                for (unsigned f = 0; f < F; ++f) {
                    float x = te[f * K + k];
                    s += x;
                    sq = fmaf(x, x, sq);
                }
                acc2 += 0.5f * (s * s - sq);
            }
            for (unsigned f = l16; f < F; f += 16) acc1 += w1s[e * F + f];
        }
        acc2 = group_sum<16>(acc2);
        acc1 = group_sum<16>(acc1);
        if (e < nex && l16 == 0) {
            fm2[b0 + e] = acc2;
            fm1[b0 + e] = acc1 + bias[0];
        }
    }
}

template <int EB>
__global__ __launch_bounds__(kThreads) void deepfm_sparse_bwd_kernel(
    const int64_t* __restrict__ ids, const float4* __restrict__ emb,
    const float4* __restrict__ g_emb, const float* __restrict__ g_fm1,
    const float* __restrict__ g_fm2, const int64_t* __restrict__ row_base, unsigned B,
    unsigned F, unsigned K4, float* __restrict__ grad_arena, float* __restrict__ grad_w1) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned K = K4 * 4;
    const unsigned FK = F * K;
    const unsigned ex_stride = FK + 16;
    float* tile = smem;                   // [EB][ex_stride]
    float* S = smem + EB * ex_stride;     // [EB][K]
    const unsigned b0 = blockIdx.x * EB;
    const unsigned nex = min((unsigned)EB, B - b0);
    const unsigned per_ex4 = F * K4;
    const unsigned total4 = nex * per_ex4;

    for (unsigned i = threadIdx.x; i < total4; i += kThreads) {
        unsigned e = i / per_ex4;
        unsigned r = i - e * per_ex4;
        float4 v = emb[(size_t)(b0 + e) * per_ex4 + r];
        *reinterpret_cast<float4*>(tile + e * ex_stride + r * 4) = v;
    }
    __syncthreads();
    {
        const unsigned e = threadIdx.x >> 4;
        const unsigned l16 = threadIdx.x & 15;
        if (e < nex) {
            const float* te = tile + e * ex_stride;
            for (unsigned k = l16; k < K; k += 16) {
                float s = 0.f;
                for (unsigned f = 0; f < F; ++f) s += te[f * K + k];
                S[e * K + k] = s;
            }
        }
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < total4; i += kThreads) {
        unsigned e = i / per_ex4;
        unsigned r = i - e * per_ex4;
        unsigned f = r / K4;
        unsigned q = r - f * K4;
        size_t grow = (size_t)(b0 + e) * F + f;
        int64_t id = ids[grow];
        if (id < 0) continue;
        int64_t arow = row_base[f] + id;
        float4 ge = g_emb[grow * K4 + q];
        float g2 = g_fm2[b0 + e];
        float4 ev = *reinterpret_cast<const float4*>(tile + e * ex_stride + r * 4);
        float4 sv = *reinterpret_cast<const float4*>(S + e * K + q * 4);
        float* dst = grad_arena + (arow * K4 + q) * 4;
        atomic_add_f32(dst + 0, fmaf(g2, sv.x - ev.x, ge.x));
        atomic_add_f32(dst + 1, fmaf(g2, sv.y - ev.y, ge.y));
        atomic_add_f32(dst + 2, fmaf(g2, sv.z - ev.z, ge.z));
        atomic_add_f32(dst + 3, fmaf(g2, sv.w - ev.w, ge.w));
        if (q == 0) atomic_add_f32(grad_w1 + arow, g_fm1[b0 + e]);
    }
}

}  // namespace

// =========================================================================================
// C-ABI
// =========================================================================================
RECALGO_EXPORT int recalgo_embedding_gather_fwd(const int64_t* ids, const float* arena,
                                                const int64_t* row_base, int B, int F, int K,
                                                float* out, int out_stride, int out_col,
                                                recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && F > 0 && K > 0 && K % 4 == 0 && out_stride % 4 == 0 && out_col % 4 == 0);
    int64_t total4 = (int64_t)B * F * (K / 4);
    RECALGO_REQUIRE(total4 < (1ll << 31));
    if (total4 == 0) return 0;
    hipLaunchKernelGGL(gather_fwd_kernel, dim3(cdiv(total4, kThreads)), dim3(kThreads), 0,
                       as_stream(stream), ids, reinterpret_cast<const float4*>(arena), row_base,
                       (unsigned)total4, (unsigned)F, (unsigned)(K / 4), out, (unsigned)out_stride,
                       (unsigned)out_col);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_embedding_gather_bwd(const int64_t* ids, const float* g,
                                                const int64_t* row_base, int B, int F, int K,
                                                int g_stride, int g_col, float* grad_arena,
                                                recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && F > 0 && K > 0 && K % 4 == 0 && g_stride % 4 == 0 && g_col % 4 == 0);
    int64_t total4 = (int64_t)B * F * (K / 4);
    RECALGO_REQUIRE(total4 < (1ll << 31));
    if (total4 == 0) return 0;
    hipLaunchKernelGGL(gather_bwd_kernel, dim3(cdiv(total4, kThreads)), dim3(kThreads), 0,
                       as_stream(stream), ids, g, row_base, (unsigned)total4, (unsigned)F,
                       (unsigned)(K / 4), (unsigned)g_stride, (unsigned)g_col, grad_arena);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_embedding_bag_mean_fwd(const int64_t* values, const int64_t* offsets,
                                                  const float* table, int B, int K, float* out,
                                                  int out_stride, int out_col,
                                                  recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && K > 0 && K % 4 == 0 && out_stride % 4 == 0 && out_col % 4 == 0);
    int64_t total4 = (int64_t)B * (K / 4);
    RECALGO_REQUIRE(total4 < (1ll << 31));
    if (total4 == 0) return 0;
    hipLaunchKernelGGL(bag_mean_fwd_kernel, dim3(cdiv(total4, kThreads)), dim3(kThreads), 0,
                       as_stream(stream), values, offsets, reinterpret_cast<const float4*>(table),
                       (unsigned)total4, (unsigned)(K / 4), out, (unsigned)out_stride,
                       (unsigned)out_col);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_embedding_bag_mean_bwd(const int64_t* values, const int64_t* offsets,
                                                  const float* g, int B, int K, int g_stride,
                                                  int g_col, float* grad_table,
                                                  recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && K > 0 && K % 4 == 0 && g_stride % 4 == 0 && g_col % 4 == 0);
    int64_t total4 = (int64_t)B * (K / 4);
    RECALGO_REQUIRE(total4 < (1ll << 31));
    if (total4 == 0) return 0;
    hipLaunchKernelGGL(bag_mean_bwd_kernel, dim3(cdiv(total4, kThreads)), dim3(kThreads), 0,
                       as_stream(stream), values, offsets, g, (unsigned)total4, (unsigned)(K / 4),
                       (unsigned)g_stride, (unsigned)g_col, grad_table);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_sequence_gather_fwd(const int64_t* values, const int64_t* offsets,
                                               const float* table, int B, int T, int K, float* out,
                                               int32_t* seq_len, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && T > 0 && K > 0 && K % 4 == 0);
    int64_t total4 = (int64_t)B * T * (K / 4);
    RECALGO_REQUIRE(total4 < (1ll << 31));
    if (total4 == 0) return 0;
    hipLaunchKernelGGL(seq_gather_fwd_kernel, dim3(cdiv(total4, kThreads)), dim3(kThreads), 0,
                       as_stream(stream), values, offsets, reinterpret_cast<const float4*>(table),
                       (unsigned)total4, (unsigned)T, (unsigned)(K / 4),
                       reinterpret_cast<float4*>(out), seq_len);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_sequence_gather_bwd(const int64_t* values, const int64_t* offsets,
                                               const float* g, int B, int T, int K,
                                               float* grad_table, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && T > 0 && K > 0 && K % 4 == 0);
    int64_t total4 = (int64_t)B * T * (K / 4);
    RECALGO_REQUIRE(total4 < (1ll << 31));
    if (total4 == 0) return 0;
    hipLaunchKernelGGL(seq_gather_bwd_kernel, dim3(cdiv(total4, kThreads)), dim3(kThreads), 0,
                       as_stream(stream), values, offsets, reinterpret_cast<const float4*>(g),
                       (unsigned)total4, (unsigned)T, (unsigned)(K / 4), grad_table);
    RECALGO_RETURN_LAST();
}

namespace {
constexpr int kDeepfmEB = 4;
inline size_t deepfm_smem(int F, int K, int extra_per_ex) {
    return (size_t)kDeepfmEB * (F * K + 16 + extra_per_ex) * sizeof(float);
}
}  // namespace

RECALGO_EXPORT int recalgo_deepfm_sparse_fwd(const int64_t* ids, const float* arena,
                                             const float* w1, const float* bias,
                                             const int64_t* row_base, int B, int F, int K,
                                             float* emb, float* fm1, float* fm2,
                                             recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && F > 0 && K > 0 && K % 4 == 0 && K <= 64);
    if (B == 0) return 0;
    size_t smem = deepfm_smem(F, K, F);
    RECALGO_REQUIRE(smem <= 64 * 1024);
    hipLaunchKernelGGL(deepfm_sparse_fwd_kernel<kDeepfmEB>, dim3(cdiv(B, kDeepfmEB)),
                       dim3(kThreads), smem, as_stream(stream), ids,
                       reinterpret_cast<const float4*>(arena), w1, bias, row_base, (unsigned)B,
                       (unsigned)F, (unsigned)(K / 4), reinterpret_cast<float4*>(emb), fm1, fm2);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_deepfm_sparse_bwd(const int64_t* ids, const float* emb,
                                             const float* g_emb, const float* g_fm1,
                                             const float* g_fm2, const int64_t* row_base, int B,
                                             int F, int K, float* grad_arena, float* grad_w1,
                                             recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && F > 0 && K > 0 && K % 4 == 0 && K <= 64);
    if (B == 0) return 0;
    size_t smem = deepfm_smem(F, K, K);
    RECALGO_REQUIRE(smem <= 64 * 1024);
    hipLaunchKernelGGL(deepfm_sparse_bwd_kernel<kDeepfmEB>, dim3(cdiv(B, kDeepfmEB)),
                       dim3(kThreads), smem, as_stream(stream), ids,
                       reinterpret_cast<const float4*>(emb),
                       reinterpret_cast<const float4*>(g_emb), g_fm1, g_fm2, row_base, (unsigned)B,
                       (unsigned)F, (unsigned)(K / 4), grad_arena, grad_w1);
    RECALGO_RETURN_LAST();
}
