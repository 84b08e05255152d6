// Embedding gathers (K1, K1m, K1s) and the fused DeepFM sparse path (K1+K2+K3), gfx950.
//
// Forward: HBM-bound.  A 16-float row is 64 B, fetched as 4 x float4 by 4 adjacent lanes, so one
// wave instruction moves 16 independent rows; outputs are written fully coalesced.  Rows whose
// width is not a multiple of 4 (the reference's dim-2 `device` column) take the scalar (VEC=1)
// instantiation of the same kernels.
//
// Backward (row-gradient scatter): CTR ids are Zipf distributed, so a naive atomicAdd per
// occurrence serialises thousands of L2 atomics on the cache line of each hot row (measured:
// 156 us for 106 k rows at B=4096).  Every backward kernel therefore first combines duplicate
// rows inside the workgroup in an LDS hash table (ds_add_f32), and only then issues ONE global
// atomic per distinct row per workgroup.  Workgroups are organised per (field, example chunk) so
// that duplicates meet in the same table.
#include <cstdlib>

#include "deferred.h"

namespace {

constexpr int kThreads = 256;

template <int VEC> struct VecT;
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<1> { using type = float; };

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type vzero();
template <> __device__ __forceinline__ float4 vzero<4>() { return f4_zero(); }
template <> __device__ __forceinline__ float vzero<1>() { return 0.f; }

// ---------------------------------------------------------------------------------------------
// workgroup-local row-gradient aggregator
// ---------------------------------------------------------------------------------------------
// The table has 2x as many slots as the workgroup has examples (a power of two, runtime: the tile
// size is chosen per kernel, see scatter_tile()).
constexpr unsigned long long kEmpty = ~0ull;

struct Agg {
    unsigned long long* keys;   // [slots]
    float* acc;                 // [slots][W]   (W = row width, + 1 for the fused w1 gradient)
    unsigned W;
    unsigned slots;
    int* new_rows;              // [slots] rows this workgroup touched first (live-row bookkeeping)
    unsigned* new_cnt;          // [2]: number of entries in new_rows, base index reserved in the arena's list
};
__device__ __forceinline__ Agg agg_carve(unsigned char* smem, unsigned W, unsigned slots) {
    unsigned char* after = smem + slots * (sizeof(unsigned long long) + (size_t)W * sizeof(float));
    return Agg{reinterpret_cast<unsigned long long*>(smem),
               reinterpret_cast<float*>(smem + slots * sizeof(unsigned long long)), W, slots,
               reinterpret_cast<int*>(after), reinterpret_cast<unsigned*>(after + slots * sizeof(int))};
}

__device__ __forceinline__ void agg_init(const Agg& a) {
    for (unsigned i = threadIdx.x; i < a.slots; i += blockDim.x) a.keys[i] = kEmpty;
    for (unsigned i = threadIdx.x; i < a.slots * a.W; i += blockDim.x) a.acc[i] = 0.f;
    if (threadIdx.x < 2) a.new_cnt[threadIdx.x] = 0;
}

// returns the slot of `row`, or a.slots if the probe sequence is exhausted
__device__ __forceinline__ unsigned agg_slot(const Agg& a, unsigned long long row) {
    unsigned h = (unsigned)((row * 0x9E3779B97F4A7C15ull) >> 40) & (a.slots - 1);
#pragma unroll 1
    for (int probe = 0; probe < 16; ++probe) {
        unsigned long long k = a.keys[h];
        if (k == row) return h;
        if (k == kEmpty) {
            unsigned long long old = atomicCAS(&a.keys[h], kEmpty, row);
            if (old == kEmpty || old == row) return h;
        }
        h = (h + 1) & (a.slots - 1);
    }
    return a.slots;
}

__device__ __forceinline__ void lds_add(float* p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Live-row bookkeeping of the arena the gradient is scattered into (include/recalgo.h recalgo_live_t): the
// first workgroup to flush a row sets its liveness byte (test-and-set on the aligned word) and appends the row to
// the arena's live list — the flush has the distinct row in hand, so the separate recalgo_mark_live_rows pass
// over all B*F ids (14 us in the DCN step, as much as the scatter itself) is not needed.
struct Live {
    unsigned* words;          // liveness bytes as aligned 32-bit words; nullptr: no bookkeeping
    int* list;
    int* count;
    long long row_offset;     // arena row of row 0 of the table the kernel scatters into
};
__device__ __forceinline__ void live_mark(const Live& L, unsigned long long row) {
    if (L.words == nullptr) return;
    row += (unsigned long long)L.row_offset;
    const unsigned bit = 1u << (8 * (unsigned)(row & 3));
    unsigned* w = L.words + (row >> 2);
    if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) return;      // common case: already live
    const unsigned old = atomicOr(w, bit);
    if (!(old & bit)) L.list[atomicAdd(L.count, 1)] = (int)row;
}

template <int VEC>
__device__ __forceinline__ void agg_add(const Agg& a, unsigned long long row, unsigned chunk,
                                        typename VecT<VEC>::type v, float* __restrict__ gdst_row, const Live& live) {
    unsigned s = agg_slot(a, row);
    if (s >= a.slots && chunk == 0) live_mark(live, row);     // probe sequence exhausted: this row bypasses the flush
    if constexpr (VEC == 4) {
        if (s < a.slots) {
            float* p = a.acc + s * a.W + chunk * 4;
            lds_add(p + 0, v.x); lds_add(p + 1, v.y); lds_add(p + 2, v.z); lds_add(p + 3, v.w);
        } else {
            float* p = gdst_row + chunk * 4;
            atomic_add_f32(p + 0, v.x); atomic_add_f32(p + 1, v.y);
            atomic_add_f32(p + 2, v.z); atomic_add_f32(p + 3, v.w);
        }
    } else {
        if (s < a.slots) lds_add(a.acc + s * a.W + chunk, v);
        else atomic_add_f32(gdst_row + chunk, v);
    }
}

// Live-row bookkeeping for the distinct rows of this workgroup: test-and-set the liveness byte of every row; rows
// touched for the first time are collected in LDS and appended to the arena's list with ONE global atomic per
// workgroup (a global atomic per new row serialises ~1e5 adds on one counter while the model is still meeting new
// rows: gather_bwd 18 -> 60-260 us in the first steps).
__device__ __forceinline__ void agg_mark_live(const Agg& a, const Live& L) {
    if (L.words == nullptr) return;                       // kernel argument: uniform
    __syncthreads();
    if (threadIdx.x < 2) a.new_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (unsigned s = threadIdx.x; s < a.slots; s += blockDim.x) {
        const unsigned long long key = a.keys[s];
        if (key == kEmpty) continue;
        const unsigned long long row = key + (unsigned long long)L.row_offset;
        const unsigned bit = 1u << (8 * (unsigned)(row & 3));
        unsigned* w = L.words + (row >> 2);
        if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) continue;   // common case: already live
        const unsigned old = atomicOr(w, bit);
        if (!(old & bit)) a.new_rows[atomicAdd(&a.new_cnt[0], 1u)] = (int)row;
    }
    __syncthreads();
    const unsigned n = a.new_cnt[0];
    if (n == 0) return;
    if (threadIdx.x == 0) a.new_cnt[1] = (unsigned)atomicAdd(L.count, (int)n);
    __syncthreads();
    const unsigned base = a.new_cnt[1];
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) L.list[base + i] = a.new_rows[i];
}

// one global atomic per (distinct row, float) of this workgroup; K floats per row go to
// grad + row*K, the optional (K+1)-th float to grad_w1 + row.
__device__ __forceinline__ void agg_flush(const Agg& a, unsigned K, float* __restrict__ grad,
                                          float* __restrict__ grad_w1, const Live& live, const Live& live_w1) {
    const unsigned lanes = K <= 16 ? 16 : (K <= 32 ? 32 : 64);   // lanes per slot
    const unsigned per_pass = blockDim.x / lanes;
    const unsigned l = threadIdx.x % lanes, grp = threadIdx.x / lanes;
    for (unsigned s = grp; s < a.slots; s += per_pass) {
        unsigned long long row = a.keys[s];
        if (row == kEmpty) continue;
        for (unsigned k = l; k < K; k += lanes) {
            float v = a.acc[s * a.W + k];
            if (v != 0.f) atomic_add_f32(grad + row * K + k, v);
        }
        if (grad_w1 && l == 0) {
            float v = a.acc[s * a.W + K];
            if (v != 0.f) atomic_add_f32(grad_w1 + row, v);
        }
    }
    agg_mark_live(a, live);
    if (grad_w1) agg_mark_live(a, live_w1);
}

// ---------------------------------------------------------------------------------------------
// K1 forward: one VEC-wide chunk of one (b, f) row per thread.
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kThreads) void gather_fwd_kernel(
    const int64_t* __restrict__ ids, const float* __restrict__ arena,
    const int64_t* __restrict__ row_base, unsigned total, unsigned F, unsigned KV,
    float* __restrict__ out, unsigned out_stride, unsigned out_col, recalgo_deferred::ReadView D) {
    using V = typename VecT<VEC>::type;
    unsigned i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    unsigned row = i / KV;          // flattened (b, f)
    unsigned q = i - row * KV;
    unsigned b = row / F;
    unsigned f = row - b * F;
    int64_t id = ids[row];
    V v = vzero<VEC>();
    if (id >= 0) {
        const int64_t ar = row_base[f] + id;
        v = recalgo_deferred::current_piece(D, reinterpret_cast<const V*>(arena)[ar * KV + q], ar, q, KV);   // (deferred Adam)
    }
    *reinterpret_cast<V*>(out + (size_t)b * out_stride + out_col + (f * KV + q) * VEC) = v;
}

// K1 backward: grid (F, chunks); workgroup = one field x ex examples.
// `ex` (examples per workgroup, a kernel argument) trades duplicate combining against parallelism: with
// 256-example tiles a B = 4096, F = 26 launch is 1.6 waves per SIMD and the waves sit parked 70 % of the
// time (profiles/r01q_dcn_pmc_sq.md); 64-example tiles give 6.5 waves per SIMD.  Measured in the step
// (600 steps, two repeats, +-0.1 %): gather_bwd 64 vs 256 -> DCN step 0.317 vs 0.324 ms; the sequence and
// DeepFM variants are faster with 256 (more duplicates per tile; with 64 the DIN step is 3 % slower even
// though the isolated, L2-warm launch is faster: the extra global atomics land on cold lines in the step).

// kFieldsPerWG adjacent fields per workgroup: the gradient pieces of two K = 16 fields of one example are ONE 128-byte
// line.  With one field per workgroup every line of g was fetched by two workgroups (20 MB for 7.7 MB algorithmic,
// profiles/r01r_dcn_pmc_fullrun.md); the rows of the two fields live in different tables, so they share the LDS
// aggregator without meeting.
constexpr unsigned kFieldsPerWG = 2;

template <int VEC>
__global__ __launch_bounds__(kThreads) void gather_bwd_kernel(
    const int64_t* __restrict__ ids, const float* __restrict__ g,
    const int64_t* __restrict__ row_base, unsigned B, unsigned F, unsigned KV, unsigned g_stride,
    unsigned g_col, float* __restrict__ grad_arena, unsigned ex, Live live) {
    using V = typename VecT<VEC>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const unsigned K = KV * VEC;
    const Agg a = agg_carve(smem_raw, K, 2 * ex * kFieldsPerWG);
    agg_init(a);
    __syncthreads();
    const unsigned f0 = blockIdx.x * kFieldsPerWG;
    const unsigned nf = min(kFieldsPerWG, F - f0);
    const unsigned b0 = blockIdx.y * ex;
    const unsigned nex = min(ex, B - b0);
    const unsigned per_ex = nf * KV;                       // contiguous VEC-chunks of one example in this tile
    int64_t rb[kFieldsPerWG];
#pragma unroll
    for (unsigned j = 0; j < kFieldsPerWG; ++j) rb[j] = row_base[min(f0 + j, F - 1)];
    for (unsigned i = threadIdx.x; i < nex * per_ex; i += kThreads) {
        const unsigned e = i / per_ex, r = i - e * per_ex;
        const unsigned ff = r / KV, q = r - ff * KV;
        const unsigned b = b0 + e, f = f0 + ff;
        int64_t id = ids[(size_t)b * F + f];
        if (id < 0) continue;
        unsigned long long row = (unsigned long long)((ff ? rb[1] : rb[0]) + id);
        V v = *reinterpret_cast<const V*>(g + (size_t)b * g_stride + g_col + (f * KV + q) * VEC);
        agg_add<VEC>(a, row, q, v, grad_arena + row * K, live);
    }
    __syncthreads();
    agg_flush(a, K, grad_arena, nullptr, live, live);
}

// ---------------------------------------------------------------------------------------------
// K1m: mean-combined bags.  Forward: one thread per (bag, chunk), the bag is walked sequentially
// so the fp32 sum order is the bag order (TF SparseSegmentMean).
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kThreads) void bag_mean_fwd_kernel(
    const int64_t* __restrict__ values, const int64_t* __restrict__ offsets,
    const float* __restrict__ table, unsigned total, unsigned KV, float* __restrict__ out,
    unsigned out_stride, unsigned out_col, recalgo_deferred::ReadView D) {
    using V = typename VecT<VEC>::type;
    unsigned i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    unsigned b = i / KV;
    unsigned q = i - b * KV;
    int64_t beg = offsets[b], end = offsets[b + 1];
    float acc[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = 0.f;
    int cnt = 0;
    for (int64_t j = beg; j < end; ++j) {
        int64_t id = values[j];
        if (id >= 0) {
            V r = recalgo_deferred::current_piece(D, reinterpret_cast<const V*>(table)[id * KV + q], id, q, KV);
            const float* rp = reinterpret_cast<const float*>(&r);
#pragma unroll
            for (int c = 0; c < VEC; ++c) acc[c] += rp[c];
            ++cnt;
        }
    }
    float* o = out + (size_t)b * out_stride + out_col + q * VEC;
    float cf = (float)(cnt > 0 ? cnt : 1);
#pragma unroll
    for (int c = 0; c < VEC; ++c) o[c] = acc[c] / cf;
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void bag_mean_bwd_kernel(
    const int64_t* __restrict__ values, const int64_t* __restrict__ offsets,
    const float* __restrict__ g, unsigned B, unsigned KV, unsigned g_stride, unsigned g_col,
    float* __restrict__ grad_table, unsigned ex, Live live) {
    using V = typename VecT<VEC>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const unsigned K = KV * VEC;
    const Agg a = agg_carve(smem_raw, K, 2 * ex);
    agg_init(a);
    __syncthreads();
    const unsigned b0 = blockIdx.x * ex;
    const unsigned nex = min(ex, B - b0);
    for (unsigned i = threadIdx.x; i < nex * KV; i += kThreads) {
        unsigned e = i / KV, q = i - e * KV;
        unsigned b = b0 + e;
        int64_t beg = offsets[b], end = offsets[b + 1];
        int cnt = 0;
        for (int64_t j = beg; j < end; ++j) cnt += values[j] >= 0;
        if (cnt == 0) continue;
        V v = *reinterpret_cast<const V*>(g + (size_t)b * g_stride + g_col + q * VEC);
        float c = (float)cnt;
        float* vp = reinterpret_cast<float*>(&v);
#pragma unroll
        for (int cc = 0; cc < VEC; ++cc) vp[cc] = vp[cc] / c;
        for (int64_t j = beg; j < end; ++j) {
            int64_t id = values[j];
            if (id < 0) continue;
            agg_add<VEC>(a, (unsigned long long)id, q, v, grad_table + id * K, live);
        }
    }
    __syncthreads();
    agg_flush(a, K, grad_table, nullptr, live, live);
}

// ---------------------------------------------------------------------------------------------
// K1s: zero padded sequence gather (B, T, K).
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kThreads) void seq_gather_fwd_kernel(
    const int64_t* __restrict__ values, const int64_t* __restrict__ offsets,
    const float* __restrict__ table, unsigned total, unsigned T, unsigned KV,
    float* __restrict__ out, int32_t* __restrict__ seq_len, recalgo_deferred::ReadView D) {
    using V = typename VecT<VEC>::type;
    unsigned i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    unsigned row = i / KV;  // (b, t)
    unsigned q = i - row * KV;
    unsigned b = row / T;
    unsigned t = row - b * T;
    int64_t beg = offsets[b];
    int64_t len = offsets[b + 1] - beg;
    if (t == 0 && q == 0) seq_len[b] = (int32_t)(len < (int64_t)T ? len : (int64_t)T);
    V v = vzero<VEC>();
    if ((int64_t)t < len) {
        int64_t id = values[beg + t];
        if (id >= 0) v = recalgo_deferred::current_piece(D, reinterpret_cast<const V*>(table)[id * KV + q], id, q, KV);
    }
    reinterpret_cast<V*>(out)[i] = v;
}

// Several plain lookups in ONE launch (the lookups a model issues together, behind one `prepare` launch: sparse.batch_lookups):
// job j owns the workgroups [first_block[j], first_block[j + 1]); kind 0 = the id-matrix gather above, kind 1 = the sequence
// gather above — the same per-thread work, no deferred view (the rows were caught up by the launch in front).
constexpr int kMultiLookups = 4;
struct LookupJob {
    const int64_t* ids;        // gather: ids [B, F]; sequence: values
    const int64_t* aux;        // gather: row_base [F]; sequence: offsets [B + 1]
    const float* table;        // gather: the arena; sequence: the table's first row
    float* out;
    int32_t* seq_len;          // sequence only
    unsigned total, FT, KV, out_stride, out_col, kind, first_block;
};
struct LookupJobs { LookupJob job[kMultiLookups]; int n; };

template <int VEC>
__global__ __launch_bounds__(kThreads) void lookup_multi_fwd_kernel(LookupJobs J) {
    using V = typename VecT<VEC>::type;
    int j = 0;
#pragma unroll
    for (int k = 1; k < kMultiLookups; ++k)
        if (k < J.n && blockIdx.x >= J.job[k].first_block) j = k;
    LookupJob A = J.job[0];                                   // (select by value: a dynamic index into the kernel arguments
#pragma unroll                                                //  would go through scratch)
    for (int k = 1; k < kMultiLookups; ++k)
        if (j == k) A = J.job[k];
    const unsigned i = (blockIdx.x - A.first_block) * kThreads + threadIdx.x;
    if (i >= A.total) return;
    const unsigned row = i / A.KV, q = i - row * A.KV;
    const unsigned b = row / A.FT, f = row - b * A.FT;
    V v = vzero<VEC>();
    if (A.kind == 0) {
        const int64_t id = A.ids[row];
        if (id >= 0) v = reinterpret_cast<const V*>(A.table)[(A.aux[f] + id) * A.KV + q];
        *reinterpret_cast<V*>(A.out + (size_t)b * A.out_stride + A.out_col + (f * A.KV + q) * VEC) = v;
    } else {
        const int64_t beg = A.aux[b];
        const int64_t len = A.aux[b + 1] - beg;
        if (f == 0 && q == 0) A.seq_len[b] = (int32_t)(len < (int64_t)A.FT ? len : (int64_t)A.FT);
        if ((int64_t)f < len) {
            const int64_t id = A.ids[beg + f];
            if (id >= 0) v = reinterpret_cast<const V*>(A.table)[id * A.KV + q];
        }
        reinterpret_cast<V*>(A.out)[i] = v;
    }
}

// workgroup = ex consecutive (b, t) positions
template <int VEC>
__global__ __launch_bounds__(kThreads) void seq_gather_bwd_kernel(
    const int64_t* __restrict__ values, const int64_t* __restrict__ offsets,
    const float* __restrict__ g, unsigned BT, unsigned T, unsigned KV,
    float* __restrict__ grad_table, unsigned ex, Live live) {
    using V = typename VecT<VEC>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const unsigned K = KV * VEC;
    const Agg a = agg_carve(smem_raw, K, 2 * ex);
    agg_init(a);
    __syncthreads();
    const unsigned r0 = blockIdx.x * ex;
    const unsigned nr = min(ex, BT - r0);
    for (unsigned i = threadIdx.x; i < nr * KV; i += kThreads) {
        unsigned e = i / KV, q = i - e * KV;
        unsigned row = r0 + e;
        unsigned b = row / T, t = row - b * T;
        int64_t beg = offsets[b];
        int64_t len = offsets[b + 1] - beg;
        if ((int64_t)t >= len) continue;
        int64_t id = values[beg + t];
        if (id < 0) continue;
        V v = reinterpret_cast<const V*>(g)[(size_t)row * KV + q];
        agg_add<VEC>(a, (unsigned long long)id, q, v, grad_table + id * K, live);
    }
    __syncthreads();
    agg_flush(a, K, grad_table, nullptr, live, live);
}

// ---------------------------------------------------------------------------------------------
// DeepFM sparse path, fused.  One workgroup owns EB examples.  Phase 1 gathers the EB*F rows
// (float4 per lane), streams them to `emb` and parks them in an LDS tile; phase 2 gives every
// example a 16-lane group that walks the F fields in LDS (sum and sum of squares per k), then
// shuffle-reduces over k.  Example stride in LDS is padded by 16 floats so that the two
// examples sharing a 32-lane ds_read_b32 group land on disjoint banks.  The per-example field
// sum S[b, :] is also written out: the backward needs it (d fm2 / d e_f = g2 * (S - e_f)).
// ---------------------------------------------------------------------------------------------
template <int EB>
__global__ __launch_bounds__(kThreads) void deepfm_sparse_fwd_kernel(
    const int64_t* __restrict__ ids, const float4* __restrict__ arena,
    const float* __restrict__ w1, const float* __restrict__ bias,
    const int64_t* __restrict__ row_base, unsigned B, unsigned F, unsigned K4,
    float4* __restrict__ emb, float* __restrict__ fm1, float* __restrict__ fm2,
    float* __restrict__ fsum, recalgo_deferred::ReadView D, recalgo_deferred::ReadView D1) {
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
            v = recalgo_deferred::current_piece(D, arena[arow * K4 + q], arow, q, K4);
        }
        emb[grow * K4 + q] = v;
        *reinterpret_cast<float4*>(tile + e * ex_stride + r * 4) = v;
        if (q == 0) w1s[e * F + f] = (id >= 0) ? recalgo_deferred::current_piece(D1, w1[arow], arow, 0u, 1u) : 0.f;
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
                fsum[(size_t)(b0 + e) * K + k] = s;
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

// backward: grid (F, chunks) like gather_bwd; the LDS rows carry K + 1 floats (w1 gradient).
__global__ __launch_bounds__(kThreads) void deepfm_sparse_bwd_kernel(
    const int64_t* __restrict__ ids, const float4* __restrict__ emb,
    const float4* __restrict__ fsum, const float4* __restrict__ g_emb,
    const float* __restrict__ g_fm1, const float* __restrict__ g_fm2,
    const int64_t* __restrict__ row_base, unsigned B, unsigned F, unsigned K4,
    float* __restrict__ grad_arena, float* __restrict__ grad_w1, unsigned ex, Live live, Live live_w1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const unsigned K = K4 * 4;
    const Agg a = agg_carve(smem_raw, K + 1, 2 * ex);
    agg_init(a);
    __syncthreads();
    const unsigned f = blockIdx.x;
    const unsigned b0 = blockIdx.y * ex;
    const unsigned nex = min(ex, B - b0);
    const int64_t rb = row_base[f];
    for (unsigned i = threadIdx.x; i < nex * K4; i += kThreads) {
        unsigned e = i / K4, q = i - e * K4;
        unsigned b = b0 + e;
        int64_t id = ids[(size_t)b * F + f];
        if (id < 0) continue;
        unsigned long long row = (unsigned long long)(rb + id);
        size_t gi = ((size_t)b * F + f) * K4 + q;
        float4 ge = g_emb[gi], ev = emb[gi], sv = fsum[(size_t)b * K4 + q];
        float g2 = g_fm2[b];
        float4 v = make_float4(fmaf(g2, sv.x - ev.x, ge.x), fmaf(g2, sv.y - ev.y, ge.y),
                               fmaf(g2, sv.z - ev.z, ge.z), fmaf(g2, sv.w - ev.w, ge.w));
        unsigned s = agg_slot(a, row);
        if (s < a.slots) {
            float* p = a.acc + s * a.W + q * 4;
            lds_add(p + 0, v.x); lds_add(p + 1, v.y); lds_add(p + 2, v.z); lds_add(p + 3, v.w);
            if (q == 0) lds_add(a.acc + s * a.W + K, g_fm1[b]);
        } else {
            float* p = grad_arena + row * K + q * 4;
            atomic_add_f32(p + 0, v.x); atomic_add_f32(p + 1, v.y);
            atomic_add_f32(p + 2, v.z); atomic_add_f32(p + 3, v.w);
            if (q == 0) {
                atomic_add_f32(grad_w1 + row, g_fm1[b]);
                live_mark(live, row);
                live_mark(live_w1, row);
            }
        }
    }
    __syncthreads();
    agg_flush(a, K, grad_arena, grad_w1, live, live_w1);
}

// ---------------------------------------------------------------------------------------------
// Deterministic scatter (recalgo_scatter_rows_sorted; round 2's parity mode, not on the training path since the owner-computes
// scatter of sparse.hip became the default): the (row, item) pairs arrive sorted by row (stable: items of one
// row keep their original order); the thread that holds the first item of a row sums the row's items in that order
// and adds the total to the gradient row with a plain read-modify-write (one writer per row per launch) — no float
// atomics, so two runs are bit-identical.  A parity / resume mode: a hot row is summed by one lane group sequentially.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void scatter_sorted_kernel(const int64_t* __restrict__ sorted_rows,
                                                                  const int64_t* __restrict__ perm,
                                                                  const float* __restrict__ vals, int64_t M, unsigned K,
                                                                  float* __restrict__ grad) {
    const int64_t t = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    const int64_t i = t / K;
    const unsigned k = (unsigned)(t - i * K);
    if (i >= M) return;
    const int64_t row = sorted_rows[i];
    if (row < 0 || (i > 0 && sorted_rows[i - 1] == row)) return;
    float acc = 0.f;
    for (int64_t j = i; j < M && sorted_rows[j] == row; ++j) acc += vals[(size_t)perm[j] * K + k];
    grad[(size_t)row * K + k] += acc;
}

inline size_t agg_smem(int W, unsigned ex) {
    return 2 * ex * (sizeof(unsigned long long) + (size_t)W * sizeof(float) + sizeof(int)) + 16;
}

// examples per workgroup of a scatter kernel
inline unsigned scatter_tile(unsigned dflt) { return dflt; }

// dynamic LDS above 64 KiB must be opted into per kernel
#define ENSURE_SMEM(kern, bytes)                                                                       \
    do {                                                                                               \
        if ((bytes) > 64 * 1024) {                                                                     \
            hipError_t e__ = hipFuncSetAttribute(reinterpret_cast<const void*>(&kern),                 \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            if (e__ != hipSuccess) return (int)e__;                                                    \
        }                                                                                              \
    } while (0)

inline Live to_live(const recalgo_live_t* l) {
    if (l == nullptr || l->row_live == nullptr) return Live{nullptr, nullptr, nullptr, 0};
    return Live{reinterpret_cast<unsigned*>(l->row_live), l->live_list, l->live_count, (long long)l->row_offset};
}
inline bool live_ok(const recalgo_live_t* l) {
    return l == nullptr || l->row_live == nullptr ||
           (l->live_list != nullptr && l->live_count != nullptr && (reinterpret_cast<uintptr_t>(l->row_live) & 3) == 0);
}

inline int vec_of(int K, int stride, int col) { return (K % 4 == 0 && stride % 4 == 0 && col % 4 == 0) ? 4 : 1; }

}  // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
namespace {
// the read view of an arena's deferred-Adam state for a forward lookup (nullptr / no state: plain lookup)
inline recalgo_deferred::ReadView read_view(const recalgo_deferred_adam_t* d, const int64_t* step_dev, int step_offset,
                                            int64_t row_offset) {
    if (!d || !d->last_step || !step_dev)
        return recalgo_deferred::ReadView{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0.f, 0.f, 0};
    return recalgo_deferred::ReadView{d->m, d->v, d->last_step, d->lr_ring, reinterpret_cast<const long long*>(step_dev), step_offset,
                                      d->beta1, d->beta2, d->eps, (long long)row_offset};
}
}  // namespace

RECALGO_EXPORT int recalgo_embedding_gather_fwd(const int64_t* ids, const float* arena,
                                                const int64_t* row_base, int B, int F, int K,
                                                float* out, int out_stride, int out_col,
                                                recalgo_stream_t stream) {
    return recalgo_embedding_gather_fwd_deferred(ids, arena, row_base, B, F, K, out, out_stride, out_col, nullptr, nullptr, 0, stream);
}

RECALGO_EXPORT int recalgo_embedding_gather_fwd_deferred(const int64_t* ids, const float* arena, const int64_t* row_base, int B,
                                                         int F, int K, float* out, int out_stride, int out_col,
                                                         const recalgo_deferred_adam_t* deferred, const int64_t* step_dev,
                                                         int step_offset, recalgo_stream_t stream) {
    const recalgo_deferred::ReadView D = read_view(deferred, step_dev, step_offset, 0);
    RECALGO_REQUIRE(B >= 0 && F > 0 && K > 0 && out_stride >= out_col + F * K);
    const int vec = vec_of(K, out_stride, out_col);
    int64_t total = (int64_t)B * F * (K / vec);
    RECALGO_REQUIRE(total < (1ll << 31));
    if (total == 0) return 0;
    if (vec == 4)
        hipLaunchKernelGGL(gather_fwd_kernel<4>, dim3(cdiv(total, kThreads)), dim3(kThreads), 0, as_stream(stream),
                           ids, arena, row_base, (unsigned)total, (unsigned)F, (unsigned)(K / 4), out,
                           (unsigned)out_stride, (unsigned)out_col, D);
    else
        hipLaunchKernelGGL(gather_fwd_kernel<1>, dim3(cdiv(total, kThreads)), dim3(kThreads), 0, as_stream(stream),
                           ids, arena, row_base, (unsigned)total, (unsigned)F, (unsigned)K, out,
                           (unsigned)out_stride, (unsigned)out_col, D);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_embedding_gather_bwd(const int64_t* ids, const float* g,
                                                const int64_t* row_base, int B, int F, int K,
                                                int g_stride, int g_col, float* grad_arena,
                                                const recalgo_live_t* live, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && F > 0 && K > 0 && K <= 64 && g_stride >= g_col + F * K && live_ok(live));
    if (B == 0) return 0;
    const int vec = vec_of(K, g_stride, g_col);
    const unsigned ex = scatter_tile(32);          // x kFieldsPerWG fields = 64 (example, field) items per workgroup
    dim3 grid(cdiv(F, kFieldsPerWG), cdiv(B, ex));
    const size_t smem = agg_smem(K, ex * kFieldsPerWG);
    if (vec == 4) {
        ENSURE_SMEM(gather_bwd_kernel<4>, smem);
        hipLaunchKernelGGL(gather_bwd_kernel<4>, grid, dim3(kThreads), smem, as_stream(stream), ids, g,
                           row_base, (unsigned)B, (unsigned)F, (unsigned)(K / 4), (unsigned)g_stride,
                           (unsigned)g_col, grad_arena, ex, to_live(live));
    } else {
        ENSURE_SMEM(gather_bwd_kernel<1>, smem);
        hipLaunchKernelGGL(gather_bwd_kernel<1>, grid, dim3(kThreads), smem, as_stream(stream), ids, g,
                           row_base, (unsigned)B, (unsigned)F, (unsigned)K, (unsigned)g_stride,
                           (unsigned)g_col, grad_arena, ex, to_live(live));
    }
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_embedding_bag_mean_fwd(const int64_t* values, const int64_t* offsets,
                                                  const float* table, int B, int K, float* out,
                                                  int out_stride, int out_col,
                                                  recalgo_stream_t stream) {
    return recalgo_embedding_bag_mean_fwd_deferred(values, offsets, table, B, K, out, out_stride, out_col, nullptr, 0, nullptr, 0, stream);
}

RECALGO_EXPORT int recalgo_embedding_bag_mean_fwd_deferred(const int64_t* values, const int64_t* offsets, const float* table, int B,
                                                           int K, float* out, int out_stride, int out_col,
                                                           const recalgo_deferred_adam_t* deferred, int64_t table_row_base,
                                                           const int64_t* step_dev, int step_offset, recalgo_stream_t stream) {
    const recalgo_deferred::ReadView D = read_view(deferred, step_dev, step_offset, table_row_base);
    RECALGO_REQUIRE(B >= 0 && K > 0 && out_stride >= out_col + K);
    const int vec = vec_of(K, out_stride, out_col);
    int64_t total = (int64_t)B * (K / vec);
    RECALGO_REQUIRE(total < (1ll << 31));
    if (total == 0) return 0;
    if (vec == 4)
        hipLaunchKernelGGL(bag_mean_fwd_kernel<4>, dim3(cdiv(total, kThreads)), dim3(kThreads), 0, as_stream(stream),
                           values, offsets, table, (unsigned)total, (unsigned)(K / 4), out, (unsigned)out_stride,
                           (unsigned)out_col, D);
    else
        hipLaunchKernelGGL(bag_mean_fwd_kernel<1>, dim3(cdiv(total, kThreads)), dim3(kThreads), 0, as_stream(stream),
                           values, offsets, table, (unsigned)total, (unsigned)K, out, (unsigned)out_stride,
                           (unsigned)out_col, D);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_embedding_bag_mean_bwd(const int64_t* values, const int64_t* offsets,
                                                  const float* g, int B, int K, int g_stride,
                                                  int g_col, float* grad_table, const recalgo_live_t* live,
                                                  recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && K > 0 && K <= 64 && g_stride >= g_col + K && live_ok(live));
    if (B == 0) return 0;
    const int vec = vec_of(K, g_stride, g_col);
    const unsigned ex = scatter_tile(256);
    if (vec == 4) {
        ENSURE_SMEM(bag_mean_bwd_kernel<4>, agg_smem(K, ex));
        hipLaunchKernelGGL(bag_mean_bwd_kernel<4>, dim3(cdiv(B, ex)), dim3(kThreads), agg_smem(K, ex),
                           as_stream(stream), values, offsets, g, (unsigned)B, (unsigned)(K / 4),
                           (unsigned)g_stride, (unsigned)g_col, grad_table, ex, to_live(live));
    } else {
        ENSURE_SMEM(bag_mean_bwd_kernel<1>, agg_smem(K, ex));
        hipLaunchKernelGGL(bag_mean_bwd_kernel<1>, dim3(cdiv(B, ex)), dim3(kThreads), agg_smem(K, ex),
                           as_stream(stream), values, offsets, g, (unsigned)B, (unsigned)K,
                           (unsigned)g_stride, (unsigned)g_col, grad_table, ex, to_live(live));
    }
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_sequence_gather_fwd(const int64_t* values, const int64_t* offsets,
                                               const float* table, int B, int T, int K, float* out,
                                               int32_t* seq_len, recalgo_stream_t stream) {
    return recalgo_sequence_gather_fwd_deferred(values, offsets, table, B, T, K, out, seq_len, nullptr, 0, nullptr, 0, stream);
}

RECALGO_EXPORT int recalgo_sequence_gather_fwd_deferred(const int64_t* values, const int64_t* offsets, const float* table, int B,
                                                        int T, int K, float* out, int32_t* seq_len,
                                                        const recalgo_deferred_adam_t* deferred, int64_t table_row_base,
                                                        const int64_t* step_dev, int step_offset, recalgo_stream_t stream) {
    const recalgo_deferred::ReadView D = read_view(deferred, step_dev, step_offset, table_row_base);
    RECALGO_REQUIRE(B >= 0 && T > 0 && K > 0);
    const int vec = K % 4 == 0 ? 4 : 1;
    int64_t total = (int64_t)B * T * (K / vec);
    RECALGO_REQUIRE(total < (1ll << 31));
    if (total == 0) return 0;
    if (vec == 4)
        hipLaunchKernelGGL(seq_gather_fwd_kernel<4>, dim3(cdiv(total, kThreads)), dim3(kThreads), 0,
                           as_stream(stream), values, offsets, table, (unsigned)total, (unsigned)T,
                           (unsigned)(K / 4), out, seq_len, D);
    else
        hipLaunchKernelGGL(seq_gather_fwd_kernel<1>, dim3(cdiv(total, kThreads)), dim3(kThreads), 0,
                           as_stream(stream), values, offsets, table, (unsigned)total, (unsigned)T, (unsigned)K,
                           out, seq_len, D);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_lookup_multi_fwd(const recalgo_lookup_job_t* jobs, int n_jobs, recalgo_stream_t stream) {
    RECALGO_REQUIRE(jobs != nullptr && n_jobs >= 1 && n_jobs <= kMultiLookups);
    LookupJobs J;
    J.n = n_jobs;
    unsigned blocks = 0;
    int vec = 4;
    for (int i = 0; i < n_jobs; ++i) {
        const recalgo_lookup_job_t& c = jobs[i];
        RECALGO_REQUIRE(c.B >= 0 && c.F_or_T > 0 && c.K > 0 && c.ids && c.aux && c.table && c.out);
        if (c.kind == 0) {
            RECALGO_REQUIRE(c.out_stride >= c.out_col + c.F_or_T * c.K);
            if (vec_of(c.K, c.out_stride, c.out_col) != 4) vec = 1;
        } else {
            RECALGO_REQUIRE(c.kind == 1 && c.seq_len != nullptr);
            if (c.K % 4 != 0) vec = 1;
        }
    }
    for (int i = 0; i < kMultiLookups; ++i) {
        LookupJob& A = J.job[i];
        if (i >= n_jobs) { A = LookupJob{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, 1, 0, 0, 0, 0xffffffffu}; continue; }
        const recalgo_lookup_job_t& c = jobs[i];
        const int64_t total = (int64_t)c.B * c.F_or_T * (c.K / vec);
        RECALGO_REQUIRE(total < (1ll << 31));
        A = LookupJob{c.ids, c.aux, c.table, c.out, c.seq_len, (unsigned)total, (unsigned)c.F_or_T, (unsigned)(c.K / vec),
                      (unsigned)c.out_stride, (unsigned)c.out_col, (unsigned)c.kind, blocks};
        blocks += (unsigned)cdiv(total, kThreads);
    }
    if (blocks == 0) return 0;
    if (vec == 4) hipLaunchKernelGGL(lookup_multi_fwd_kernel<4>, dim3(blocks), dim3(kThreads), 0, as_stream(stream), J);
    else hipLaunchKernelGGL(lookup_multi_fwd_kernel<1>, dim3(blocks), dim3(kThreads), 0, as_stream(stream), J);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_sequence_gather_bwd(const int64_t* values, const int64_t* offsets,
                                               const float* g, int B, int T, int K,
                                               float* grad_table, const recalgo_live_t* live, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && T > 0 && K > 0 && K <= 64 && live_ok(live));
    int64_t BT = (int64_t)B * T;
    RECALGO_REQUIRE(BT < (1ll << 31));
    if (BT == 0) return 0;
    const unsigned ex = scatter_tile(256);     // history ids repeat across the batch: big tiles combine more
    if (K % 4 == 0) {
        ENSURE_SMEM(seq_gather_bwd_kernel<4>, agg_smem(K, ex));
        hipLaunchKernelGGL(seq_gather_bwd_kernel<4>, dim3(cdiv(BT, ex)), dim3(kThreads), agg_smem(K, ex),
                           as_stream(stream), values, offsets, g, (unsigned)BT, (unsigned)T, (unsigned)(K / 4),
                           grad_table, ex, to_live(live));
    } else {
        ENSURE_SMEM(seq_gather_bwd_kernel<1>, agg_smem(K, ex));
        hipLaunchKernelGGL(seq_gather_bwd_kernel<1>, dim3(cdiv(BT, ex)), dim3(kThreads), agg_smem(K, ex),
                           as_stream(stream), values, offsets, g, (unsigned)BT, (unsigned)T, (unsigned)K,
                           grad_table, ex, to_live(live));
    }
    RECALGO_RETURN_LAST();
}

namespace {
constexpr int kDeepfmEB = 4;
}  // namespace

RECALGO_EXPORT int recalgo_deepfm_sparse_fwd(const int64_t* ids, const float* arena,
                                             const float* w1, const float* bias,
                                             const int64_t* row_base, int B, int F, int K,
                                             float* emb, float* fm1, float* fm2, float* field_sum,
                                             recalgo_stream_t stream) {
    return recalgo_deepfm_sparse_fwd_deferred(ids, arena, w1, bias, row_base, B, F, K, emb, fm1, fm2, field_sum, nullptr, nullptr,
                                              nullptr, 0, stream);
}

RECALGO_EXPORT int recalgo_deepfm_sparse_fwd_deferred(const int64_t* ids, const float* arena, const float* w1, const float* bias,
                                                      const int64_t* row_base, int B, int F, int K, float* emb, float* fm1,
                                                      float* fm2, float* field_sum, const recalgo_deferred_adam_t* deferred,
                                                      const recalgo_deferred_adam_t* deferred_w1, const int64_t* step_dev,
                                                      int step_offset, recalgo_stream_t stream) {
    const recalgo_deferred::ReadView D = read_view(deferred, step_dev, step_offset, 0);
    const recalgo_deferred::ReadView D1 = read_view(deferred_w1, step_dev, step_offset, 0);
    RECALGO_REQUIRE(B >= 0 && F > 0 && K > 0 && K % 4 == 0 && K <= 64 && field_sum != nullptr);
    if (B == 0) return 0;
    size_t smem = (size_t)kDeepfmEB * (F * K + 16 + F) * sizeof(float);
    RECALGO_REQUIRE(smem <= 64 * 1024);
    hipLaunchKernelGGL(deepfm_sparse_fwd_kernel<kDeepfmEB>, dim3(cdiv(B, kDeepfmEB)),
                       dim3(kThreads), smem, as_stream(stream), ids,
                       reinterpret_cast<const float4*>(arena), w1, bias, row_base, (unsigned)B,
                       (unsigned)F, (unsigned)(K / 4), reinterpret_cast<float4*>(emb), fm1, fm2, field_sum, D, D1);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_deepfm_sparse_bwd(const int64_t* ids, const float* emb,
                                             const float* field_sum, const float* g_emb,
                                             const float* g_fm1, const float* g_fm2,
                                             const int64_t* row_base, int B, int F, int K,
                                             float* grad_arena, float* grad_w1, const recalgo_live_t* live,
                                             const recalgo_live_t* live_w1, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B >= 0 && F > 0 && K > 0 && K % 4 == 0 && K <= 64 && live_ok(live) && live_ok(live_w1));
    if (B == 0) return 0;
    const unsigned ex = scatter_tile(256);
    ENSURE_SMEM(deepfm_sparse_bwd_kernel, agg_smem(K + 1, ex));
    hipLaunchKernelGGL(deepfm_sparse_bwd_kernel, dim3(F, cdiv(B, ex)), dim3(kThreads), agg_smem(K + 1, ex),
                       as_stream(stream), ids, reinterpret_cast<const float4*>(emb),
                       reinterpret_cast<const float4*>(field_sum), reinterpret_cast<const float4*>(g_emb),
                       g_fm1, g_fm2, row_base, (unsigned)B, (unsigned)F, (unsigned)(K / 4), grad_arena, grad_w1, ex, to_live(live),
                       to_live(live_w1));
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_scatter_rows_sorted(const int64_t* sorted_rows, const int64_t* perm, const float* vals,
                                               int64_t M, int K, float* grad, recalgo_stream_t stream) {
    RECALGO_REQUIRE(M >= 0 && K >= 1 && (M == 0 || (sorted_rows && perm && vals && grad)));
    if (M == 0) return 0;
    const int64_t total = M * K;
    RECALGO_REQUIRE(cdiv(total, kThreads) > 0);
    hipLaunchKernelGGL(scatter_sorted_kernel, dim3(cdiv(total, kThreads)), dim3(kThreads), 0, as_stream(stream), sorted_rows,
                       perm, vals, M, (unsigned)K, grad);
    RECALGO_RETURN_LAST();
}
