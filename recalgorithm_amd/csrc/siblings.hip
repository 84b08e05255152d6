// Interaction kernels of the sibling models that reuse the hot-path kernels, gfx950:
//   NFM  bi-interaction pooling  (algorithm/NFM/nfm.py:155-167): the FM second-order term WITHOUT the reduction over k
//   AFM  attention pooling over the pair Hadamard products (algorithm/AFM/afm.py:184-188): softmax over the pairs +
//        weighted sum (the pair products themselves are the FiBiNET bilinear kernel with W = I, the attention MLP the
//        dense kernels)
//   FFM  field-aware pair dots (algorithm/FFM/ffm.py:146-160) over the gathered [B, F, F-1, K] sub-table rows
// All three are HBM-bound streams over a few KB per example: one wave (or one thread per output element) per
// example, no LDS staging needed at these sizes (F <= 32, K <= 64).
#include "common.h"

namespace {

// ---- NFM -------------------------------------------------------------------------------------------------------
// out[b, k] = 0.5 * ((sum_f e[b,f,k])^2 - sum_f e[b,f,k]^2); sums in field order (tf.add_n, nfm.py:163-165)
__global__ __launch_bounds__(256) void bi_interaction_fwd_kernel(const float* __restrict__ emb, unsigned total, unsigned F,
                                                                 unsigned K, float* __restrict__ out) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;           // (b, k)
    if (i >= total) return;
    const unsigned b = i / K, k = i - b * K;
    const float* e = emb + (size_t)b * F * K + k;
    float s = 0.f, q = 0.f;
    for (unsigned f = 0; f < F; ++f) {
        const float x = e[f * K];
        s += x;
        q = fmaf(x, x, q);
    }
    out[i] = 0.5f * (s * s - q);
}
// d e[b,f,k] = g[b,k] * (S[b,k] - e[b,f,k])
__global__ __launch_bounds__(256) void bi_interaction_bwd_kernel(const float* __restrict__ emb, const float* __restrict__ g,
                                                                 unsigned total, unsigned F, unsigned K,
                                                                 float* __restrict__ d_emb) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const unsigned b = i / K, k = i - b * K;
    const float* e = emb + (size_t)b * F * K + k;
    float s = 0.f;
    for (unsigned f = 0; f < F; ++f) s += e[f * K];
    const float gv = g[i];
    float* d = d_emb + (size_t)b * F * K + k;
    for (unsigned f = 0; f < F; ++f) d[f * K] = gv * (s - e[f * K]);
}

// ---- AFM -------------------------------------------------------------------------------------------------------
// one wave per example: score = softmax_p(att[b, :]); out[b, k] = sum_p score[p] * pairs[b, p, k]
__global__ __launch_bounds__(256) void attn_pool_fwd_kernel(const float* __restrict__ pairs, const float* __restrict__ att,
                                                            unsigned B, unsigned P, unsigned K, unsigned KL,
                                                            float* __restrict__ out, float* __restrict__ score) {
    const unsigned b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const float* a = att + (size_t)b * P;
    float mx = -3.402823466e38f;
    for (unsigned p = lane; p < P; p += 64) mx = fmaxf(mx, a[p]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (unsigned p = lane; p < P; p += 64) sum += expf(a[p] - mx);
    sum = wave_sum(sum);
    float* sc = score + (size_t)b * P;
    for (unsigned p = lane; p < P; p += 64) sc[p] = expf(a[p] - mx) / sum;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    // lanes = (pair group, k): KL = K rounded up to a power of two, 64 / KL pair groups
    const unsigned G = 64 / KL, grp = lane / KL, k = lane % KL;
    float acc = 0.f;
    if (k < K)
        for (unsigned p = grp; p < P; p += G) acc = fmaf(expf(a[p] - mx) / sum, pairs[((size_t)b * P + p) * K + k], acc);
    for (unsigned o = KL; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (grp == 0 && k < K) out[(size_t)b * K + k] = acc;
}
// d att[p] = score[p] * (dalpha[p] - sum_q score[q] dalpha[q]), dalpha[p] = <g, pairs[p,:]>;  d pairs[p,k] = score[p] * g[k]
// One wave per example, lanes = (pair group, k) as in the forward: a wave instruction reads / writes 64 / KL whole
// pair rows (contiguous).  Pass 1 reads the pairs once and parks dalpha in LDS; pass 2 only writes.
__global__ __launch_bounds__(256) void attn_pool_bwd_kernel(const float* __restrict__ pairs, const float* __restrict__ score,
                                                            const float* __restrict__ g, unsigned B, unsigned P, unsigned K,
                                                            unsigned KL, float* __restrict__ d_pairs,
                                                            float* __restrict__ d_att) {
    extern __shared__ float smem[];                                  // [4 waves][P] dalpha
    const unsigned wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned b = blockIdx.x * 4 + wib;
    if (b >= B) return;
    float* da = smem + (size_t)wib * P;
    const float* sc = score + (size_t)b * P;
    const unsigned G = 64 / KL, grp = lane / KL, k = lane % KL;
    const float gk = k < K ? g[(size_t)b * K + k] : 0.f;
    float dot = 0.f;
    for (unsigned p0 = 0; p0 < P; p0 += G) {
        const unsigned p = p0 + grp;
        float v = (p < P && k < K) ? gk * pairs[((size_t)b * P + p) * K + k] : 0.f;
        for (unsigned o = 1; o < KL; o <<= 1) v += __shfl_xor(v, o, 64);      // sum over k inside the pair group
        if (p < P && k == 0) {
            da[p] = v;
            dot = fmaf(sc[p], v, dot);
        }
    }
    dot = wave_sum(dot);
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    for (unsigned p0 = 0; p0 < P; p0 += G) {
        const unsigned p = p0 + grp;
        if (p < P) {
            const float s = sc[p];
            if (k < K) d_pairs[((size_t)b * P + p) * K + k] = s * gk;
            if (k == 0) d_att[(size_t)b * P + p] = s * (da[p] - dot);
        }
    }
}

// ---- FFM -------------------------------------------------------------------------------------------------------
// x [B, F, F-1, K]: row (i, s) is field i looked up in its sub-table s.  Pair i < j uses x[i][j-1] and x[j][i]
// (ffm.py:150-157); out[b] = sum_{i<j} <x[i][j-1], x[j][i]>, pairs accumulated in the reference's loop order per lane.
__global__ __launch_bounds__(256) void ffm_pairs_fwd_kernel(const float* __restrict__ x, unsigned B, unsigned F, unsigned K,
                                                            float* __restrict__ out) {
    const unsigned b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const float* xb = x + (size_t)b * F * (F - 1) * K;
    const unsigned P = F * (F - 1) / 2;
    float acc = 0.f;
    for (unsigned t = lane; t < P * K; t += 64) {
        const unsigned p = t / K, k = t - p * K;
        unsigned i = 0, rem = p;                         // p -> (i, j), row-major strict upper triangle
        while (rem >= F - 1 - i) { rem -= F - 1 - i; ++i; }
        const unsigned j = i + 1 + rem;
        acc = fmaf(xb[(i * (F - 1) + (j - 1)) * K + k], xb[(j * (F - 1) + i) * K + k], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) out[b] = acc;
}
// every element (a, s) belongs to exactly one pair: partner (s + 1, a) when s >= a, (s, a - 1) when s < a
__global__ __launch_bounds__(256) void ffm_pairs_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            unsigned total, unsigned F, unsigned K, float* __restrict__ dx) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const unsigned per = F * (F - 1) * K;
    const unsigned b = i / per, r = i - b * per;
    const unsigned k = r % K, as = r / K, a = as / (F - 1), s = as - a * (F - 1);
    const unsigned pa = s >= a ? s + 1 : s, ps = s >= a ? a : a - 1;
    dx[i] = g[b] * x[(size_t)b * per + (pa * (F - 1) + ps) * K + k];
}

}  // namespace

RECALGO_EXPORT int recalgo_bi_interaction_fwd(const float* emb, int B, int F, int K, float* out, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && F >= 1 && K >= 1);
    if (B == 0) return 0;
    const int64_t total = (int64_t)B * K;
    RECALGO_REQUIRE(total < (1ll << 31));
    hipLaunchKernelGGL(bi_interaction_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), emb, (unsigned)total,
                       (unsigned)F, (unsigned)K, out);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_bi_interaction_bwd(const float* emb, const float* g, int B, int F, int K, float* d_emb,
                                              recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && F >= 1 && K >= 1);
    if (B == 0) return 0;
    const int64_t total = (int64_t)B * K;
    RECALGO_REQUIRE(total < (1ll << 31));
    hipLaunchKernelGGL(bi_interaction_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), emb, g, (unsigned)total,
                       (unsigned)F, (unsigned)K, d_emb);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_attention_pool_fwd(const float* pairs, const float* att, int B, int P, int K, float* out,
                                              float* score, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && P >= 1 && K >= 1 && K <= 64);
    if (B == 0) return 0;
    unsigned KL = 1;
    while ((int)KL < K) KL <<= 1;
    hipLaunchKernelGGL(attn_pool_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, as_stream(stream), pairs, att, (unsigned)B,
                       (unsigned)P, (unsigned)K, KL, out, score);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_attention_pool_bwd(const float* pairs, const float* score, const float* g, int B, int P, int K,
                                              float* d_pairs, float* d_att, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && P >= 1 && K >= 1 && K <= 64);
    if (B == 0) return 0;
    unsigned KL = 1;
    while ((int)KL < K) KL <<= 1;
    const size_t smem = (size_t)4 * P * sizeof(float);
    RECALGO_REQUIRE(smem <= 64 * 1024);
    hipLaunchKernelGGL(attn_pool_bwd_kernel, dim3(cdiv(B, 4)), dim3(256), smem, as_stream(stream), pairs, score, g, (unsigned)B,
                       (unsigned)P, (unsigned)K, KL, d_pairs, d_att);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_ffm_pairs_fwd(const float* x, int B, int F, int K, float* out, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && F >= 2 && K >= 1);
    if (B == 0) return 0;
    hipLaunchKernelGGL(ffm_pairs_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, as_stream(stream), x, (unsigned)B, (unsigned)F,
                       (unsigned)K, out);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_ffm_pairs_bwd(const float* x, const float* g, int B, int F, int K, float* dx,
                                         recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && F >= 2 && K >= 1);
    if (B == 0) return 0;
    const int64_t total = (int64_t)B * F * (F - 1) * K;
    RECALGO_REQUIRE(total < (1ll << 31));
    hipLaunchKernelGGL(ffm_pairs_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), x, g, (unsigned)total,
                       (unsigned)F, (unsigned)K, dx);
    RECALGO_RETURN_LAST();
}
