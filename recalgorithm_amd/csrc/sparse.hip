// Row-gradient scatter WITHOUT float atomics, fused with the sparse optimizer, gfx950.
//
// The backward of every embedding lookup is  dTable[row] += g[request]  over the requests (b, f) of the batch
// (Appendix D "Gather"); CTR ids are Zipf distributed, so thousands of requests hit the same row.  Rounds 1-2 combined
// duplicates in an LDS hash table per workgroup and then issued one device-scope float atomic per (distinct row, float,
// workgroup) — ~1.1 M fabric atomics per DCN step, the weakest kernel of every model (0.08 of HBM in the step), order
// non-deterministic, and the summed gradient arena was then re-read (and zeroed) by the optimizer launch.
//
// Here the scatter is OWNER-COMPUTES:
//   1. `prepare` (one launch per lookup, before its forward gather): counts the requests per bucket
//      (bucket = hash(row), kNB .. 4*kNB buckets) and — deferred Adam only — brings every requested row's (w, m, v) up to
//      date (see "deferred exact Adam" below) so that the unchanged forward kernels read current weights;
//   2. `place` (one launch per arena and step, after the backward pass): exclusive scan of the bucket counts, every
//      request is written as a (row << 32 | request index) key into its bucket's range; extra workgroups of the same
//      launch run the deferred-Adam sweep;
//   3. `apply` (one workgroup per bucket): sorts the bucket's keys in LDS (bitonic; oversize buckets: LDS-sorted runs +
//      merge passes in global memory), so that all requests of a row are adjacent and in request order; a group of
//      K/4 lanes owns a row: it adds the row's gradient rows IN REQUEST ORDER (bit-reproducible), and — the row being
//      exclusively its own — finishes the job in registers: TF1 Adam (dense semantics, exact), LazyAdam, or a plain
//      `grad[row] += sum` store for callers that want the gradient arena.  No float atomics anywhere, no gradient arena
//      round trip, no live-row list.
//
// Deferred exact Adam.  tf.train.AdamOptimizer applies a DENSE update to embedding variables: m, v of every row decay
// and w moves every step, gradient or not (deepfm.py:246-250).  Rounds 1-2 walked every row a gradient
// had ever reached each step (1.25 GB per DCN step once the tables are warm: 0.25 -> 0.47 ms).  The g = 0 update of a row
// is a pure function of its own (w, m, v) and of lr_t(step):  m *= b1; v *= b2; w -= lr_t * m / (sqrt(v) + eps).  So it can
// be postponed: `last_step[row]` records the step the row's state is valid for, and whoever needs the row next (the
// `prepare` launch of a lookup that requests it, the round-robin sweep, a flush before EVAL / PREDICT / checkpoint)
// replays the missed steps in registers with the SAME fp32 operations in the SAME order — bit-identical to the dense
// pass (tests/test_gpu_sparse.py), at the cost of the batch's rows.  lr_t of recent steps comes from a small ring
// written by the optimizer launch; the sweep (1/P of the arena per step, contiguous rows) bounds every row's lag to
// P + 1 steps so that the ring and the replay loops stay short.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxSources = RECALGO_SCATTER_MAX_SOURCES;
static_assert(kMaxSources == 4, "load_g / place select among exactly four sources");
constexpr unsigned kLrRing = RECALGO_LR_RING;           // power of two
constexpr unsigned kSortCap = 2048;                     // keys sorted in LDS per run (16 KB)
constexpr unsigned kLongSeg = 48;                       // requests per row above which the whole workgroup sums it
constexpr unsigned kMaxLong = 64;                       // long rows remembered per bucket (more: summed by one group)
constexpr unsigned long long kPadKey = ~0ull;

struct SrcDev {
    const int64_t* ids;
    const int64_t* offsets;
    const int64_t* row_base;
    long long base;
    unsigned n_ex, F, first, n;        // n = n_ex * F requests; `first` = index of request 0 in the plan
    const float* g;
    long long g_stride;
    unsigned g_col, g_fmul;
};

// arena row of local request i of source S, -1: no row (OOV id, beyond the sequence's length)
__device__ __forceinline__ long long src_row(const SrcDev& S, unsigned i) {
    long long id;
    unsigned f;
    if (S.offsets) {
        const unsigned e = i / S.F;
        f = i - e * S.F;
        const long long beg = S.offsets[e], len = S.offsets[e + 1] - beg;
        if ((long long)f >= len) return -1;
        id = S.ids[beg + f];
    } else {
        id = S.ids[i];
        f = S.row_base ? i % S.F : 0;
    }
    if (id < 0) return -1;
    return id + S.base + (S.row_base ? S.row_base[f] : 0);
}

__device__ __forceinline__ unsigned bucket_of(unsigned row, unsigned nb_log2) {
    return (row * 0x9E3779B1u) >> (32 - nb_log2);
}

// lr_t of TF1 Adam, evaluated exactly as recalgo_adam_tf1_step does
__device__ __forceinline__ float lr_t_of(float lr, float b1, float b2, long long t) {
    const double td = (double)t;
    return (float)((double)lr * sqrt(1.0 - pow((double)b2, td)) / (1.0 - pow((double)b1, td)));
}

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float lr_t, float b1, float b2, float eps) {
    m = fmaf(b1, m, (1.f - b1) * g);
    v = fmaf(b2, v, (1.f - b2) * g * g);
    p -= lr_t * m / (sqrtf(v) + eps);
}

// ---- row state access: VEC = 4 (K % 4 == 0: lane q of a group holds floats 4q .. 4q+3) or 1 ---------------------
template <int VEC> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<1> { using T = float; };
template <int VEC> __device__ __forceinline__ typename Vec<VEC>::T vz();
template <> __device__ __forceinline__ float4 vz<4>() { return f4_zero(); }
template <> __device__ __forceinline__ float vz<1>() { return 0.f; }
__device__ __forceinline__ void vadd(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ void vadd(float& a, const float b) { a += b; }
__device__ __forceinline__ void vadam(float4& p, const float4 g, float4& m, float4& v, float lr_t, float b1, float b2, float eps) {
    adam1(p.x, g.x, m.x, v.x, lr_t, b1, b2, eps);
    adam1(p.y, g.y, m.y, v.y, lr_t, b1, b2, eps);
    adam1(p.z, g.z, m.z, v.z, lr_t, b1, b2, eps);
    adam1(p.w, g.w, m.w, v.w, lr_t, b1, b2, eps);
}
__device__ __forceinline__ void vadam(float& p, const float g, float& m, float& v, float lr_t, float b1, float b2, float eps) {
    adam1(p, g, m, v, lr_t, b1, b2, eps);
}

struct Deferred {                  // deferred-Adam state of one arena
    float* w; float* m; float* v;
    int* last_step;                // [rows]: 0 = never touched (m = v = 0), s > 0 = (w, m, v) valid for step s, < 0 = claimed
    const float* lr_ring;          // [kLrRing]: lr_t(j) at j & (kLrRing - 1)
    float b1, b2, eps;
};

// replay the g = 0 updates of steps s+1 .. target on one lane's piece of a row
template <int VEC>
__device__ __forceinline__ void replay(typename Vec<VEC>::T& w, typename Vec<VEC>::T& m, typename Vec<VEC>::T& v, int s,
                                       int target, const Deferred& D) {
    for (int j = s + 1; j <= target; ++j) vadam(w, vz<VEC>(), m, v, D.lr_ring[(unsigned)j & (kLrRing - 1)], D.b1, D.b2, D.eps);
}

// bring row `row` (state valid for step s) to `target`; the L lanes of a group call this together (q = lane in group)
template <int VEC>
__device__ __forceinline__ void catch_up_row(const Deferred& D, long long row, int s, int target, unsigned q, unsigned KV) {
    using V = typename Vec<VEC>::T;
    if (q < KV) {
        const size_t o = (size_t)row * KV + q;
        V w = reinterpret_cast<V*>(D.w)[o], m = reinterpret_cast<V*>(D.m)[o], v = reinterpret_cast<V*>(D.v)[o];
        replay<VEC>(w, m, v, s, target, D);
        reinterpret_cast<V*>(D.w)[o] = w;
        reinterpret_cast<V*>(D.m)[o] = m;
        reinterpret_cast<V*>(D.v)[o] = v;
    }
    if (q == 0) D.last_step[row] = target;
}

// 256-thread exclusive scan of one value per thread; sh: 8 unsigned of LDS; returns the exclusive prefix, total in `total`
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned* sh, unsigned& total) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(inc, o, 64);
        if (lane >= (unsigned)o) inc += t;
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    unsigned before = 0, tot = 0;
#pragma unroll
    for (unsigned w = 0; w < kThreads / 64; ++w) {
        const unsigned t = sh[w];
        if (w < wave) before += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return before + inc - v;
}

// ---------------------------------------------------------------------------------------------
// 1. prepare: bucket counts of one lookup's requests (+ deferred-Adam catch-up of the requested rows)
// ---------------------------------------------------------------------------------------------
struct PrepareArgs {
    SrcDev S;
    int* cnt;                      // [nb] bucket counts of the plan being assembled (nullptr: catch-up only)
    unsigned nb_log2;
    Deferred D;                    // D.last_step == nullptr: no catch-up
    const long long* step;         // catch-up target = step[0] + step_off
    int step_off;
    unsigned KV, L;                // row = KV pieces of VEC floats, owned by L >= KV lanes (L a power of two <= 64)
};

template <int VEC>
__global__ __launch_bounds__(kThreads) void sparse_prepare_kernel(PrepareArgs A) {
    extern __shared__ unsigned lds_u[];
    const unsigned nb = 1u << A.nb_log2;
    unsigned* hist = lds_u;                                   // [nb]
    long long* stale_row = reinterpret_cast<long long*>(lds_u + nb);   // [kThreads]
    int* stale_s = reinterpret_cast<int*>(stale_row + kThreads);        // [kThreads]
    unsigned* n_stale = reinterpret_cast<unsigned*>(stale_s + kThreads);
    if (A.cnt)
        for (unsigned b = threadIdx.x; b < nb; b += kThreads) hist[b] = 0;
    if (threadIdx.x == 0) *n_stale = 0;
    __syncthreads();
    const unsigned i = blockIdx.x * kThreads + threadIdx.x;
    const long long row = i < A.S.n ? src_row(A.S, i) : -1;
    const int target = A.D.last_step ? (int)(A.step[0] + A.step_off) : 0;
    if (row >= 0) {
        if (A.cnt) atomicAdd(&hist[bucket_of((unsigned)row, A.nb_log2)], 1u);
        if (A.D.last_step) {
            // hot rows are current (their last_step is the previous step): only stale rows cost an atomic, and exactly one
            // of the requests of a stale row wins the claim
            const int s = __hip_atomic_load(&A.D.last_step[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s > 0 && s < target && atomicCAS(&A.D.last_step[row], s, -s) == s) {
                const unsigned k = atomicAdd(n_stale, 1u);
                stale_row[k] = row;
                stale_s[k] = s;
            }
        }
    }
    __syncthreads();
    if (A.cnt)
        for (unsigned b = threadIdx.x; b < nb; b += kThreads) {
            const unsigned h = hist[b];
            if (h) atomicAdd(&A.cnt[b], (int)h);
        }
    const unsigned ns = *n_stale;
    const unsigned q = threadIdx.x & (A.L - 1), grp = threadIdx.x / A.L, ngrp = kThreads / A.L;
    for (unsigned k = grp; k < ns; k += ngrp) catch_up_row<VEC>(A.D, stale_row[k], stale_s[k], target, q, A.KV);
}

// ---------------------------------------------------------------------------------------------
// 2. place: keys into bucket ranges (+ the deferred-Adam sweep in extra workgroups)
// ---------------------------------------------------------------------------------------------
struct PlaceArgs {
    SrcDev src[kMaxSources];
    int n_src;
    unsigned n_total, req_blocks;
    int* cnt; int* cursor; int* offs;          // [nb], [nb], [nb + 1]
    unsigned long long* keys;                  // [n_total]
    unsigned nb_log2;
    // sweep (deferred Adam): rows [c * chunk, (c + 1) * chunk), c = target % period, are brought to `target`
    Deferred D;
    const long long* step;
    int step_off;
    unsigned KV, L;
    long long rows, chunk;
    int period;
};

template <int VEC>
__global__ __launch_bounds__(kThreads) void sparse_place_kernel(PlaceArgs A) {
    if (blockIdx.x >= A.req_blocks) {                         // ---- sweep workgroups -------------------------------
        const int target = (int)(A.step[0] + A.step_off);
        if (target <= 0) return;
        const long long c0 = (long long)(target % A.period) * A.chunk;
        const long long idx = (long long)(blockIdx.x - A.req_blocks) * kThreads + threadIdx.x;
        const long long row = c0 + idx / A.L;
        const unsigned q = (unsigned)(idx & (A.L - 1));
        if (row >= A.rows || row >= c0 + A.chunk) return;
        const int s = A.D.last_step[row];
        if (s > 0 && s < target) catch_up_row<VEC>(A.D, row, s, target, q, A.KV);
        return;
    }
    extern __shared__ unsigned lds_u[];
    const unsigned nb = 1u << A.nb_log2, bpt = nb / kThreads;   // nb is a multiple of kThreads
    unsigned* offs = lds_u;                                   // [nb]
    unsigned* hist = offs + nb;                               // [nb]
    unsigned* base = hist + nb;                               // [nb]
    unsigned* sh = base + nb;                                 // [8]
    {
        unsigned sum = 0;
        for (unsigned k = 0; k < bpt; ++k) sum += (unsigned)A.cnt[threadIdx.x * bpt + k];
        unsigned total;
        unsigned run = block_excl_scan(sum, sh, total);
        for (unsigned k = 0; k < bpt; ++k) {
            const unsigned b = threadIdx.x * bpt + k;
            offs[b] = run;
            hist[b] = 0;
            if (blockIdx.x == 0) A.offs[b] = (int)run;
            run += (unsigned)A.cnt[b];
        }
        if (blockIdx.x == 0 && threadIdx.x == kThreads - 1) A.offs[nb] = (int)total;
    }
    __syncthreads();
    const unsigned i = blockIdx.x * kThreads + threadIdx.x;
    long long row = -1;
    if (i < A.n_total) {
        int s = 0;
#pragma unroll
        for (int k = 1; k < kMaxSources; ++k)
            if (k < A.n_src && i >= A.src[k].first) s = k;
        SrcDev S = A.src[0];
#pragma unroll
        for (int k = 1; k < kMaxSources; ++k)
            if (s == k) S = A.src[k];
        row = src_row(S, i - S.first);
    }
    const unsigned b = row >= 0 ? bucket_of((unsigned)row, A.nb_log2) : 0;
    if (row >= 0) atomicAdd(&hist[b], 1u);
    __syncthreads();
    for (unsigned k = 0; k < bpt; ++k) {
        const unsigned bin = threadIdx.x * bpt + k;
        const unsigned h = hist[bin];
        base[bin] = h ? (unsigned)atomicAdd(&A.cursor[bin], (int)h) : 0u;
        hist[bin] = 0;
    }
    __syncthreads();
    if (row >= 0) {
        const unsigned r = atomicAdd(&hist[b], 1u);           // order inside a bucket is arbitrary: `apply` sorts
        A.keys[offs[b] + base[b] + r] = ((unsigned long long)row << 32) | i;
    }
}

// ---------------------------------------------------------------------------------------------
// 3. apply
// ---------------------------------------------------------------------------------------------
struct GSrc {                      // what `apply` needs of a source: where the gradient row of a request is
    const float* g;
    long long g_stride;
    unsigned g_col, g_fmul, F, first;
};
struct ApplyArgs {
    GSrc src[kMaxSources];
    int n_src;
    int* cnt; int* cursor; const int* offs;
    unsigned long long* keys; unsigned long long* keys_alt;   // keys_alt: scratch of the same size (oversize buckets)
    int mode;                      // RECALGO_SCATTER_GRAD / _ADAM / _LAZY_ADAM
    float* w; float* m; float* v; float* grad;                // grad: GRAD target; ADAM modes: rows zeroed when non-null
    int* last_step;                // ADAM (deferred-exact) only
    float* lr_ring;
    const long long* step;         // t = step[0] + step_off
    int step_off;
    float lr, b1, b2, eps;
    unsigned K, KV, L;
    unsigned* live_words; int* live_list; int* live_count;    // GRAD mode: live-row bookkeeping of the old optimizer path
};

__device__ __forceinline__ unsigned key_row(unsigned long long k) { return (unsigned)(k >> 32); }

// bitonic sort of m (power of two) keys in LDS by all threads of the workgroup
__device__ __forceinline__ void lds_bitonic(unsigned long long* keys, unsigned m) {
    for (unsigned k = 2; k <= m; k <<= 1)
        for (unsigned j = k >> 1; j > 0; j >>= 1) {
            for (unsigned t = threadIdx.x; t < (m >> 1); t += kThreads) {
                const unsigned lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const unsigned long long a = keys[lo], c = keys[hi];
                const bool up = (lo & k) == 0;
                if ((a > c) == up) { keys[lo] = c; keys[hi] = a; }
            }
            __syncthreads();
        }
}

// gradient piece q of the request a key refers to
template <int VEC>
__device__ __forceinline__ typename Vec<VEC>::T load_g(const ApplyArgs& A, const GSrc* lsrc, unsigned long long key, unsigned q) {
    using V = typename Vec<VEC>::T;
    const unsigned ref = (unsigned)key;
    // (the descriptors live in LDS: a data-dependent index into the kernel-argument array would go through scratch;
    //  unused sources have first = 0xffffffff)
    const unsigned si = (ref >= A.src[1].first) + (ref >= A.src[2].first) + (ref >= A.src[3].first);
    const GSrc S = lsrc[si];
    const float* g = S.g;
    const long long stride = S.g_stride;
    const unsigned col = S.g_col, fmul = S.g_fmul, F = S.F, first = S.first;
    const unsigned i = ref - first, e = i / F, f = i - e * F;
    return *reinterpret_cast<const V*>(g + (size_t)e * stride + col + (size_t)f * fmul + q * VEC);
}

// what the owner of a row does with the row's summed gradient
template <int VEC>
__device__ __forceinline__ void finish_row(const ApplyArgs& A, unsigned row, typename Vec<VEC>::T acc, unsigned q, int t, float lr_t) {
    using V = typename Vec<VEC>::T;
    const size_t o = (size_t)row * A.KV + q;
    if (A.mode == RECALGO_SCATTER_GRAD) {
        if (q < A.KV) {
            V cur = reinterpret_cast<V*>(A.grad)[o];
            vadd(cur, acc);
            reinterpret_cast<V*>(A.grad)[o] = cur;
        }
        if (q == 0 && A.live_words) {                         // first touch: the row joins the arena's live list
            const unsigned bit = 1u << (8 * (row & 3));
            unsigned* wd = A.live_words + (row >> 2);
            if (!(__hip_atomic_load(wd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) {
                const unsigned old = atomicOr(wd, bit);
                if (!(old & bit)) A.live_list[atomicAdd(A.live_count, 1)] = (int)row;
            }
        }
        return;
    }
    if (q < A.KV) {
        V w = reinterpret_cast<V*>(A.w)[o], m = reinterpret_cast<V*>(A.m)[o], v = reinterpret_cast<V*>(A.v)[o];
        if (A.mode == RECALGO_SCATTER_ADAM) {
            const int s = A.last_step[row];
            if (s > 0 && s < t - 1) {                         // (normally done by `prepare`; kept for lookups without one)
                Deferred D{A.w, A.m, A.v, A.last_step, A.lr_ring, A.b1, A.b2, A.eps};
                replay<VEC>(w, m, v, s, t - 1, D);
            }
        }
        vadam(w, acc, m, v, lr_t, A.b1, A.b2, A.eps);
        reinterpret_cast<V*>(A.w)[o] = w;
        reinterpret_cast<V*>(A.m)[o] = m;
        reinterpret_cast<V*>(A.v)[o] = v;
        if (A.grad) reinterpret_cast<V*>(A.grad)[o] = vz<VEC>();
    }
    if (q == 0 && A.mode == RECALGO_SCATTER_ADAM) A.last_step[row] = t;
}

// first index in [lo, n) whose row differs from `row` (keys sorted)
__device__ __forceinline__ unsigned seg_end(const unsigned long long* keys, unsigned lo, unsigned n, unsigned row) {
#pragma unroll 1
    for (unsigned k = 1; k <= 4; ++k) {                       // short rows: linear peek
        if (lo + k >= n || key_row(keys[lo + k]) != row) return lo + k;
    }
    unsigned a = lo + 4, b = n;                               // keys[a] has `row`; answer in (a, n]
    while (b - a > 1) {
        const unsigned mid = a + ((b - a) >> 1);
        if (key_row(keys[mid]) == row) a = mid; else b = mid;
    }
    return b;
}

template <int VEC>
__device__ __forceinline__ void process_sorted(const ApplyArgs& A, const GSrc* lsrc, const unsigned long long* keys, unsigned n, unsigned* long_list,
                               unsigned* n_long, float* red, int t, float lr_t) {
    using V = typename Vec<VEC>::T;
    const unsigned L = A.L, q = threadIdx.x & (L - 1), grp = threadIdx.x / L, ngrp = kThreads / L;
    // every group scans a contiguous block of entries for row heads and owns the rows that START in its block
    const unsigned per = (n + ngrp - 1) / ngrp;
    unsigned i = grp * per;
    const unsigned stop = min(n, i + per);
    while (i < stop) {
        const unsigned row = key_row(keys[i]);
        if (i > 0 && key_row(keys[i - 1]) == row) { ++i; continue; }          // not a head (only at the block start)
        const unsigned end = seg_end(keys, i, n, row);
        if (end - i > kLongSeg) {
            unsigned slot = kMaxLong;
            if (q == 0) slot = atomicAdd(n_long, 1u);
            slot = __shfl(slot, (int)((threadIdx.x & 63) & ~(L - 1)), 64);
            if (slot < kMaxLong) {
                if (q == 0) { long_list[2 * slot] = i; long_list[2 * slot + 1] = end; }
                i = end;
                continue;
            }
        }
        V acc = vz<VEC>();
        if (q < A.KV) {
            unsigned j = i;
            for (; j + 4 <= end; j += 4) {                    // four row loads in flight, added in request order
                const V g0 = load_g<VEC>(A, lsrc, keys[j], q), g1 = load_g<VEC>(A, lsrc, keys[j + 1], q);
                const V g2 = load_g<VEC>(A, lsrc, keys[j + 2], q), g3 = load_g<VEC>(A, lsrc, keys[j + 3], q);
                vadd(acc, g0); vadd(acc, g1); vadd(acc, g2); vadd(acc, g3);
            }
            for (; j < end; ++j) vadd(acc, load_g<VEC>(A, lsrc, keys[j], q));
        }
        finish_row<VEC>(A, row, acc, q, t, lr_t);
        i = end;
    }
    __syncthreads();
    // long rows: all groups of the workgroup sum strided slices, fixed-order combination through LDS
    const unsigned nl = min(*n_long, kMaxLong);
    for (unsigned k = 0; k < nl; ++k) {
        const unsigned lo = long_list[2 * k], hi = long_list[2 * k + 1];
        V acc = vz<VEC>();
        if (q < A.KV)
            for (unsigned j = lo + grp; j < hi; j += ngrp) vadd(acc, load_g<VEC>(A, lsrc, keys[j], q));
        if (q < A.KV) reinterpret_cast<V*>(red)[grp * A.KV + q] = acc;
        __syncthreads();
        if (grp == 0) {
            V tot = vz<VEC>();
            if (q < A.KV)
                for (unsigned g2 = 0; g2 < ngrp; ++g2) vadd(tot, reinterpret_cast<const V*>(red)[g2 * A.KV + q]);
            finish_row<VEC>(A, key_row(keys[lo]), tot, q, t, lr_t);
        }
        __syncthreads();
    }
}

// oversize bucket: sort runs of kSortCap keys in LDS, then merge passes between `a` and `b` in global memory; returns
// the buffer holding the sorted keys
__device__ __forceinline__ unsigned long long* global_merge_sort(unsigned long long* a, unsigned long long* b, unsigned n,
                                                 unsigned long long* lds_keys) {
    for (unsigned r0 = 0; r0 < n; r0 += kSortCap) {
        const unsigned cnt = min(kSortCap, n - r0);
        for (unsigned k = threadIdx.x; k < kSortCap; k += kThreads) lds_keys[k] = k < cnt ? a[r0 + k] : kPadKey;
        __syncthreads();
        lds_bitonic(lds_keys, kSortCap);
        for (unsigned k = threadIdx.x; k < cnt; k += kThreads) a[r0 + k] = lds_keys[k];
        __syncthreads();
    }
    unsigned long long* src = a;
    unsigned long long* dst = b;
    for (unsigned width = kSortCap; width < n; width <<= 1) {
        for (unsigned lo = 0; lo < n; lo += 2 * width) {
            const unsigned mid = min(n, lo + width), hi = min(n, lo + 2 * width);
            const unsigned na = mid - lo, nbb = hi - mid, tot = hi - lo;
            const unsigned per = (tot + kThreads - 1) / kThreads;
            const unsigned d0 = min(tot, threadIdx.x * per), d1 = min(tot, d0 + per);
            if (d0 < d1) {
                // merge path: ia + ib = d0 with A[ia-1] < B[ib] and B[ib-1] < A[ia] (keys are unique)
                unsigned x = d0 > nbb ? d0 - nbb : 0, y = min(d0, na);
                while (x < y) {
                    const unsigned ia = (x + y) >> 1, ib = d0 - ia;
                    if (src[lo + ia] < src[mid + ib - 1]) x = ia + 1; else y = ia;
                }
                unsigned ia = x, ib = d0 - x;
                for (unsigned o = d0; o < d1; ++o) {
                    const bool takeA = ib >= nbb || (ia < na && src[lo + ia] < src[mid + ib]);
                    dst[lo + o] = takeA ? src[lo + ia++] : src[mid + ib++];
                }
            }
        }
        __syncthreads();                                      // workgroup-scope visibility of the global stores
        unsigned long long* tsw = src; src = dst; dst = tsw;
    }
    return src;
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void sparse_apply_kernel(ApplyArgs A) {
    __shared__ unsigned long long lds_keys[kSortCap];
    __shared__ unsigned long_list[2 * kMaxLong];
    __shared__ unsigned n_long;
    __shared__ float red[kThreads * 4];
    __shared__ float s_lr_t;
    __shared__ GSrc lsrc[kMaxSources];
    const unsigned b = blockIdx.x;
    const int t = (int)(A.step[0] + A.step_off);
    if (threadIdx.x == 0) {
        lsrc[0] = A.src[0]; lsrc[1] = A.src[1]; lsrc[2] = A.src[2]; lsrc[3] = A.src[3];
        n_long = 0;
        float lr_t = 0.f;
        if (A.mode != RECALGO_SCATTER_GRAD) {
            lr_t = lr_t_of(A.lr, A.b1, A.b2, t);
            if (b == 0 && A.lr_ring) A.lr_ring[(unsigned)t & (kLrRing - 1)] = lr_t;
        }
        s_lr_t = lr_t;
    }
    const unsigned beg = (unsigned)A.offs[b], n = (unsigned)A.offs[b + 1] - beg;
    __syncthreads();
    const float lr_t = s_lr_t;
    if (n) {
        if (n <= kSortCap) {
            unsigned m = 2;
            while (m < n) m <<= 1;
            for (unsigned k = threadIdx.x; k < m; k += kThreads) lds_keys[k] = k < n ? A.keys[beg + k] : kPadKey;
            __syncthreads();
            lds_bitonic(lds_keys, m);
            process_sorted<VEC>(A, lsrc, lds_keys, n, long_list, &n_long, red, t, lr_t);
        } else {
            const unsigned long long* sorted = global_merge_sort(A.keys + beg, A.keys_alt + beg, n, lds_keys);
            process_sorted<VEC>(A, lsrc, sorted, n, long_list, &n_long, red, t, lr_t);
        }
    }
    if (threadIdx.x == 0) {                                   // the plan's counters are clean for the next step
        A.cnt[b] = 0;
        A.cursor[b] = 0;
    }
}

// ---------------------------------------------------------------------------------------------
// standalone sweep: rows [row0, row1) brought to step[0] + step_off (flush before EVAL / PREDICT / checkpoint)
// ---------------------------------------------------------------------------------------------
struct SweepArgs {
    Deferred D;
    const long long* step;
    int step_off;
    unsigned KV, L;
    long long row0, row1;
};
template <int VEC>
__global__ __launch_bounds__(kThreads) void sparse_sweep_kernel(SweepArgs A) {
    const int target = (int)(A.step[0] + A.step_off);
    const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
    const long long row = A.row0 + idx / A.L;
    const unsigned q = (unsigned)(idx & (A.L - 1));
    if (row >= A.row1) return;
    const int s = A.D.last_step[row];
    if (s > 0 && s < target) catch_up_row<VEC>(A.D, row, s, target, q, A.KV);
}

// ---- host helpers -----------------------------------------------------------------------------
struct Geometry { int vec; unsigned KV, L; };
inline bool geometry(int K, const recalgo_scatter_source_t* src, int n_src, Geometry* G) {
    if (K < 1 || K > 256) return false;
    bool v4 = (K & 3) == 0 && K / 4 <= 64;
    for (int i = 0; v4 && src && i < n_src; ++i) {
        const recalgo_scatter_source_t& s = src[i];
        if (s.g && ((reinterpret_cast<uintptr_t>(s.g) & 15) || (s.g_stride & 3) || (s.g_col & 3) || (s.g_fmul & 3))) v4 = false;
    }
    G->vec = v4 ? 4 : 1;
    G->KV = v4 ? (unsigned)K / 4 : (unsigned)K;
    if (G->KV > 64) return false;
    unsigned L = 1;
    while (L < G->KV) L <<= 1;
    G->L = L;
    return true;
}

inline bool to_dev(const recalgo_scatter_source_t* src, int n_src, SrcDev* out, unsigned* n_total, bool need_g) {
    unsigned first = 0;
    for (int i = 0; i < kMaxSources; ++i) out[i] = SrcDev{nullptr, nullptr, nullptr, 0, 0, 1, 0xffffffffu, 0, nullptr, 0, 0, 0};
    for (int i = 0; i < n_src; ++i) {
        const recalgo_scatter_source_t& s = src[i];
        if (!s.ids || s.n_ex < 0 || s.F < 1 || (need_g && !s.g)) return false;
        const int64_t n = (int64_t)s.n_ex * s.F;
        if ((int64_t)first + n >= (1ll << 31)) return false;
        out[i] = SrcDev{s.ids, s.offsets, s.row_base, (long long)s.base, (unsigned)s.n_ex, (unsigned)s.F, first, (unsigned)n,
                        s.g, (long long)s.g_stride, (unsigned)s.g_col, (unsigned)s.g_fmul};
        first += (unsigned)n;
    }
    *n_total = first;
    return true;
}

inline Deferred deferred_of(const recalgo_deferred_adam_t* d) {
    if (!d || !d->last_step) return Deferred{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f};
    return Deferred{d->w, d->m, d->v, d->last_step, d->lr_ring, d->beta1, d->beta2, d->eps};
}
inline bool nb_ok(int nb_log2) { return nb_log2 >= 8 && nb_log2 <= 13; }

}  // namespace

RECALGO_EXPORT int recalgo_scatter_plan_buckets_log2(int64_t n_requests) {
    // ~100 requests per bucket (one workgroup each in `apply`), 1024 .. 8192 buckets
    int l = 10;
    while (l < 13 && (n_requests >> l) > 128) ++l;
    return l;
}

RECALGO_EXPORT int64_t recalgo_scatter_plan_workspace_bytes(int64_t n_requests, int nb_log2) {
    if (n_requests < 0 || !nb_ok(nb_log2)) return 0;
    const int64_t nb = 1ll << nb_log2;
    return (3 * nb + 8) * (int64_t)sizeof(int) + 2 * ((n_requests + 1) & ~1ll) * (int64_t)sizeof(unsigned long long) + 64;
}

namespace {
struct Ws { int* cnt; int* cursor; int* offs; unsigned long long* keys; unsigned long long* keys_alt; };
inline Ws carve(void* ws, int64_t n_requests, int nb_log2) {
    const int64_t nb = 1ll << nb_log2;
    char* p = static_cast<char*>(ws);
    Ws w;
    w.cnt = reinterpret_cast<int*>(p);
    w.cursor = w.cnt + nb;
    w.offs = w.cursor + nb;
    uintptr_t k = (reinterpret_cast<uintptr_t>(w.offs + nb + 8) + 15) & ~(uintptr_t)15;
    w.keys = reinterpret_cast<unsigned long long*>(k);
    w.keys_alt = w.keys + ((n_requests + 1) & ~1ll);
    return w;
}
}  // namespace

RECALGO_EXPORT int recalgo_scatter_prepare(const recalgo_scatter_source_t* source, int K, void* plan_workspace,
                                           int64_t plan_requests, int nb_log2, const recalgo_deferred_adam_t* deferred,
                                           const int64_t* step_dev, int step_offset, recalgo_stream_t stream) {
    RECALGO_REQUIRE(source != nullptr && nb_ok(nb_log2));
    RECALGO_REQUIRE(plan_workspace != nullptr || (deferred != nullptr && deferred->last_step != nullptr));
    SrcDev S[kMaxSources];
    unsigned n = 0;
    RECALGO_REQUIRE(to_dev(source, 1, S, &n, false));
    if (n == 0) return 0;
    Geometry G;
    RECALGO_REQUIRE(geometry(K, nullptr, 0, &G));
    PrepareArgs A;
    A.S = S[0];
    A.cnt = plan_workspace ? carve(plan_workspace, plan_requests, nb_log2).cnt : nullptr;
    A.nb_log2 = (unsigned)nb_log2;
    A.D = deferred_of(deferred);
    RECALGO_REQUIRE(A.D.last_step == nullptr || (step_dev != nullptr && A.D.lr_ring != nullptr));
    A.step = reinterpret_cast<const long long*>(step_dev);
    A.step_off = step_offset;
    A.KV = G.KV; A.L = G.L;
    const size_t smem = ((size_t)1 << nb_log2) * sizeof(unsigned) + kThreads * (sizeof(long long) + sizeof(int)) + 16;
    const dim3 grid(cdiv(n, kThreads));
    if (G.vec == 4)
        hipLaunchKernelGGL(sparse_prepare_kernel<4>, grid, dim3(kThreads), smem, as_stream(stream), A);
    else
        hipLaunchKernelGGL(sparse_prepare_kernel<1>, grid, dim3(kThreads), smem, as_stream(stream), A);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_scatter_apply(const recalgo_scatter_source_t* sources, int n_sources, int K, void* plan_workspace,
                                         int64_t plan_requests, int nb_log2, int mode, float* w, float* m, float* v,
                                         float* grad, const recalgo_deferred_adam_t* deferred, int64_t rows,
                                         int sweep_period, const recalgo_live_t* live, const int64_t* step_dev,
                                         int step_offset, float lr, float beta1, float beta2, float eps,
                                         recalgo_stream_t stream) {
    RECALGO_REQUIRE(sources != nullptr && n_sources >= 1 && n_sources <= kMaxSources && plan_workspace != nullptr);
    RECALGO_REQUIRE(nb_ok(nb_log2) && rows >= 0);
    RECALGO_REQUIRE(mode == RECALGO_SCATTER_GRAD || mode == RECALGO_SCATTER_ADAM || mode == RECALGO_SCATTER_LAZY_ADAM);
    RECALGO_REQUIRE(mode != RECALGO_SCATTER_GRAD ? (w && m && v && step_dev) : grad != nullptr);
    RECALGO_REQUIRE(mode != RECALGO_SCATTER_ADAM || (deferred && deferred->last_step && deferred->lr_ring && sweep_period >= 1 &&
                                                     sweep_period <= (int)kLrRing - 8));
    Geometry G;
    RECALGO_REQUIRE(geometry(K, sources, n_sources, &G));
    PlaceArgs P;
    unsigned n_total = 0;
    RECALGO_REQUIRE(to_dev(sources, n_sources, P.src, &n_total, true));
    RECALGO_REQUIRE((int64_t)n_total <= plan_requests);
    const Ws ws = carve(plan_workspace, plan_requests, nb_log2);
    hipStream_t st = as_stream(stream);
    P.n_src = n_sources;
    P.n_total = n_total;
    P.req_blocks = (unsigned)cdiv(n_total, kThreads);
    if (P.req_blocks == 0) P.req_blocks = 1;                  // the scan of (all-zero) counts still publishes offs[]
    P.cnt = ws.cnt; P.cursor = ws.cursor; P.offs = ws.offs; P.keys = ws.keys;
    P.nb_log2 = (unsigned)nb_log2;
    P.D = mode == RECALGO_SCATTER_ADAM ? deferred_of(deferred) : deferred_of(nullptr);
    P.step = reinterpret_cast<const long long*>(step_dev);
    P.step_off = step_offset - 1;                             // the sweep (like `prepare`) targets the step BEFORE this one
    P.KV = G.KV; P.L = G.L;
    P.rows = rows;
    P.period = sweep_period < 1 ? 1 : sweep_period;
    P.chunk = (rows + P.period - 1) / P.period;
    unsigned sweep_blocks = 0;
    if (P.D.last_step) sweep_blocks = (unsigned)cdiv(P.chunk * G.L, kThreads);
    const size_t smem = (3 * ((size_t)1 << nb_log2) + 8) * sizeof(unsigned);
    if (smem > 64 * 1024) {
        hipError_t e = G.vec == 4 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&sparse_place_kernel<4>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)
                                  : hipFuncSetAttribute(reinterpret_cast<const void*>(&sparse_place_kernel<1>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    if (G.vec == 4)
        hipLaunchKernelGGL(sparse_place_kernel<4>, dim3(P.req_blocks + sweep_blocks), dim3(kThreads), smem, st, P);
    else
        hipLaunchKernelGGL(sparse_place_kernel<1>, dim3(P.req_blocks + sweep_blocks), dim3(kThreads), smem, st, P);
    ApplyArgs A;
    for (int i = 0; i < kMaxSources; ++i)
        A.src[i] = GSrc{P.src[i].g, P.src[i].g_stride, P.src[i].g_col, P.src[i].g_fmul, P.src[i].F, P.src[i].first};
    A.n_src = n_sources;
    A.cnt = ws.cnt; A.cursor = ws.cursor; A.offs = ws.offs; A.keys = ws.keys; A.keys_alt = ws.keys_alt;
    A.mode = mode;
    A.w = w; A.m = m; A.v = v; A.grad = grad;
    A.last_step = mode == RECALGO_SCATTER_ADAM ? deferred->last_step : nullptr;
    A.lr_ring = mode == RECALGO_SCATTER_ADAM ? deferred->lr_ring : nullptr;
    A.step = reinterpret_cast<const long long*>(step_dev);
    A.step_off = step_offset;
    A.lr = lr; A.b1 = beta1; A.b2 = beta2; A.eps = eps;
    A.K = (unsigned)K; A.KV = G.KV; A.L = G.L;
    A.live_words = nullptr; A.live_list = nullptr; A.live_count = nullptr;
    if (mode == RECALGO_SCATTER_GRAD && live && live->row_live) {
        RECALGO_REQUIRE((reinterpret_cast<uintptr_t>(live->row_live) & 3) == 0 && live->row_offset == 0);
        A.live_words = reinterpret_cast<unsigned*>(live->row_live);
        A.live_list = live->live_list;
        A.live_count = live->live_count;
    }
    if (A.step == nullptr) { RECALGO_REQUIRE(mode == RECALGO_SCATTER_GRAD); A.step = reinterpret_cast<const long long*>(ws.offs); A.step_off = 0; }
    const dim3 grid(1u << nb_log2);
    if (G.vec == 4)
        hipLaunchKernelGGL(sparse_apply_kernel<4>, grid, dim3(kThreads), 0, st, A);
    else
        hipLaunchKernelGGL(sparse_apply_kernel<1>, grid, dim3(kThreads), 0, st, A);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_adam_deferred_sweep(const recalgo_deferred_adam_t* deferred, int K, int64_t row_begin, int64_t row_end,
                                               const int64_t* step_dev, int step_offset, recalgo_stream_t stream) {
    RECALGO_REQUIRE(deferred && deferred->last_step && deferred->lr_ring && step_dev && row_begin >= 0 && row_end >= row_begin);
    if (row_end == row_begin) return 0;
    Geometry G;
    RECALGO_REQUIRE(geometry(K, nullptr, 0, &G));
    SweepArgs A;
    A.D = deferred_of(deferred);
    A.step = reinterpret_cast<const long long*>(step_dev);
    A.step_off = step_offset;
    A.KV = G.KV; A.L = G.L;
    A.row0 = row_begin; A.row1 = row_end;
    const int64_t threads = (row_end - row_begin) * G.L;
    const dim3 grid((unsigned)((threads + kThreads - 1) / kThreads));
    if (G.vec == 4)
        hipLaunchKernelGGL(sparse_sweep_kernel<4>, grid, dim3(kThreads), 0, as_stream(stream), A);
    else
        hipLaunchKernelGGL(sparse_sweep_kernel<1>, grid, dim3(kThreads), 0, as_stream(stream), A);
    RECALGO_RETURN_LAST();
}
