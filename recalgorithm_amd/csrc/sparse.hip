// Row-gradient scatter WITHOUT atomics, fused with the sparse optimizer, gfx950.
//
// The backward of every embedding lookup is  dTable[row] += g[request]  over the requests (b, f) of the batch
// (Appendix D "Gather"); CTR ids are Zipf distributed and real tables include two-valued fields, so one row can own
// two thirds of a batch's requests of its field.  Rounds 1-2 combined duplicates in an LDS hash table per workgroup and
// then issued one device-scope float atomic per (distinct row, float, workgroup) — ~1.1 M fabric atomics per DCN step,
// the weakest kernel of every model (0.08 of HBM in the step), order non-deterministic, and the summed gradient arena was
// then re-read (and zeroed) by the optimizer launch.
//
// Here the scatter is OWNER-COMPUTES, built on a deterministic STABLE MULTISPLIT of the requests by bucket = hash(row)
// (no float atomics and no sort; the only atomics on global memory are the deferred-Adam claims: one integer CAS per lagging
// row and one counter add per tile that found any):
//   1. `prepare` (one launch per lookup, before its forward gather): the plan's slot space is cut into tiles of 256 (an id
//      matrix field-major: a tile is 256 consecutive examples of ONE field).  Workgroup w finds the distinct rows of its
//      tile (LDS hash + 256-bit member masks) and writes, one entry per DISTINCT row, its bucket histogram as row w of a
//      count matrix C[W][nb] (plain stores).  Deferred Adam only: rows whose state lags are claimed (one CAS per stale
//      row) and listed; `catchup`, a second, evenly spread launch, replays their missed updates so that the unchanged
//      forward kernels read current weights;
//   2. `scan` (one launch per arena and step, after the backward pass): exclusive prefix of every column of C over the
//      tiles, and the bucket totals.  Extra workgroups of the same launch run the deferred-Adam sweep;
//   3. `place`: entry i of tile w goes to  offs[b] + Cp[w][b] + (number of earlier entries of w in bucket b):
//      every bucket receives its entries IN SLOT ORDER, independent of scheduling.  The duplicates of a row inside a tile
//      are summed in request order into ONE partial gradient row (a hot row of a field reaches its bucket as B / 256
//      entries, not thousands);
//   4. `apply` (one workgroup per bucket, heavy buckets dispatched first): a STABLE group-by-row of the bucket (small
//      buckets: rank by comparison; large ones: LDS hash of the distinct rows + stable counting scatter), after which all
//      entries of a row are adjacent and still in order.  A group of K/4 lanes owns a row: it adds the row's entries IN
//      ORDER (bit-reproducible; rows with many entries are summed by the whole workgroup in a fixed strided order), and —
//      the row being exclusively its own — finishes the job in registers: TF1 Adam (dense semantics, exact), LazyAdam, or
//      a plain `grad[row] += sum` store for callers that want the gradient arena.  No gradient arena round trip, no
//      live-row list.  A COMPANION arena (one float per row, looked up with the same requests: DeepFM's first-order
//      weights) reuses the placed entries: `place` also sums its scalars, a second `apply` launch walks them.
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
#include <cstddef>
#include <cstdlib>

#include "deferred.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxSources = RECALGO_SCATTER_MAX_SOURCES;
using recalgo_deferred::kLrRing;
constexpr unsigned kLdsKeys = 2048;                     // grouped keys kept in LDS (16 KB); larger buckets go through global memory
constexpr unsigned kMaxSeg = 512;                       // rows (segments) of a bucket listed in LDS
constexpr unsigned kSlots = 512;                        // LDS hash of the distinct rows of a large bucket
constexpr unsigned kLongSeg = 48;                       // requests per row above which the whole workgroup sums it
constexpr unsigned kMaxLong = 64;                       // long rows remembered per bucket (more: summed by one group)
constexpr unsigned long long kPadKey = ~0ull;
constexpr unsigned kEmptyRow = 0xffffffffu;

struct SrcDev {
    const int64_t* ids;
    const int64_t* offsets;
    const int64_t* row_base;
    long long base;
    unsigned n_ex, F, first, n;        // n = SLOTS of the source in the plan's slot space; `first` = its first slot
    unsigned e256, pad_;               // dense sources: examples per field rounded up to whole tiles
    const float* g;
    long long g_stride;
    unsigned g_col, g_fmul;
    // companion (a second arena of ONE float per row that is looked up with exactly these requests: DeepFM's first-order
    // weights): where the request's scalar gradient is; nullptr: none
    const float* g1;
    long long g1_stride;
    unsigned g1_col, g1_fmul;
};
constexpr unsigned kPartialBit = 0x80000000u;   // key.ref: the gradient row is a tile's partial sum, not a request's row

// Copy `bytes` of the kernel's (single, by-value) argument struct, starting at byte `offset`, into LDS — one dword per
// thread, read straight from the kernarg segment.  (Indexing a by-value kernel-argument array with a data-dependent
// index makes the compiler copy the array to scratch first.)
__device__ __forceinline__ void copy_kernarg_words(unsigned* dst, size_t offset, size_t bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    using ka_ptr = const unsigned __attribute__((address_space(4)))*;
    ka_ptr ka = (ka_ptr)__builtin_amdgcn_kernarg_segment_ptr() + offset / 4;
    for (unsigned w = threadIdx.x; w < bytes / 4; w += kThreads) dst[w] = ka[w];
#else
    (void)dst; (void)offset; (void)bytes;
#endif
}

// The plan's SLOT space.  Workgroup ("tile") w of prepare / place owns slots [256 w, 256 w + 256).  An id matrix
// [n_ex, F] is laid out FIELD-MAJOR: slot f * e256 + e  (e256 = n_ex rounded up to 256), so that a tile is 256 consecutive
// examples of ONE field — where the duplicates of a batch are (all requests of a hot row of that field meet in B / 256
// tiles).  Ragged sources stay row-major (e * F + f).  -> arena row (-1: empty slot, OOV id, beyond the sequence's
// length) and the request's index e * F + f inside its source (what the gradient row is addressed by).
__device__ __forceinline__ long long slot_row(const SrcDev& S, unsigned li, unsigned* ref_local) {
    long long id;
    unsigned e, f;
    if (S.offsets) {
        e = li / S.F;
        f = li - e * S.F;
        if (e >= S.n_ex) return -1;
        const long long beg = S.offsets[e], len = S.offsets[e + 1] - beg;
        if ((long long)f >= len) return -1;
        id = S.ids[beg + f];
    } else {
        f = li / S.e256;
        e = li - f * S.e256;
        if (e >= S.n_ex) return -1;
        id = S.ids[(size_t)e * S.F + f];
    }
    *ref_local = e * S.F + f;
    if (id < 0) return -1;
    return id + S.base + (S.row_base ? S.row_base[f] : 0);
}

// slot of `key` in an LDS open-addressing table of kSlots words (insert = true: claims an empty slot); kSlots if full
__device__ __forceinline__ unsigned hash_slot(unsigned* hkey, unsigned key, bool insert) {
    unsigned h = (key * 0x85EBCA6Bu) >> (32 - 9);             // kSlots = 512
#pragma unroll 1
    for (unsigned probe = 0; probe < kSlots; ++probe) {
        const unsigned k = hkey[h];
        if (k == key) return h;
        if (k == kEmptyRow) {
            if (!insert) return kSlots;
            const unsigned old = atomicCAS(&hkey[h], kEmptyRow, key);
            if (old == kEmptyRow || old == key) return h;
        }
        h = (h + 1) & (kSlots - 1);
    }
    return kSlots;
}

// Equal keys among the 256 threads of a workgroup, in thread order — without the 256 x 256 comparisons (measured: two such
// loops were half of `place`): every thread sets its bit in the 256-bit member mask of its key's hash slot; the number of
// equal keys, how many of them come before this thread and the first of them (the LEADER) are popcounts of that mask.
// mask: LDS [kSlots][8]; all threads call (barriers inside); key 0xffffffff = none (slot = kSlots).
struct EqInfo { unsigned same, before, leader; };
__device__ __forceinline__ void eq_masks_clear(unsigned* mask) {
    uint4* m4 = reinterpret_cast<uint4*>(mask);
    for (unsigned k = threadIdx.x; k < kSlots * 2; k += kThreads) m4[k] = make_uint4(0, 0, 0, 0);
}
__device__ __forceinline__ EqInfo eq_from_mask(const unsigned* mask, unsigned slot) {
    EqInfo e{0, 0, 0};
    if (slot >= kSlots) return e;
    const uint4 a = reinterpret_cast<const uint4*>(mask)[slot * 2], b = reinterpret_cast<const uint4*>(mask)[slot * 2 + 1];
    const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const unsigned mw = threadIdx.x >> 5, lowm = (1u << (threadIdx.x & 31)) - 1u;
    unsigned leader = 0xffffffffu;
#pragma unroll
    for (int k = 7; k >= 0; --k) {
        e.same += __popc(w[k]);
        e.before += (unsigned)k < mw ? __popc(w[k]) : ((unsigned)k == mw ? __popc(w[k] & lowm) : 0u);
        if (w[k]) leader = 32u * k + (unsigned)__ffs((int)w[k]) - 1u;
    }
    e.leader = leader;
    return e;
}
// table hkey [kSlots] + mask [kSlots][8] (both cleared here); returns the thread's slot through *slot_out
__device__ __forceinline__ EqInfo tile_equal(unsigned key, unsigned* hkey, unsigned* mask, unsigned* slot_out) {
    __syncthreads();                                          // (the tables may still be read from an earlier call)
    for (unsigned k = threadIdx.x; k < kSlots; k += kThreads) hkey[k] = kEmptyRow;
    eq_masks_clear(mask);
    __syncthreads();
    unsigned slot = kSlots;
    if (key != 0xffffffffu) {
        slot = hash_slot(hkey, key, true);                    // (<= 256 distinct keys in 512 slots: always finds one)
        atomicOr(&mask[slot * 8 + (threadIdx.x >> 5)], 1u << (threadIdx.x & 31));
    }
    __syncthreads();
    *slot_out = slot;
    return eq_from_mask(mask, slot);
}

__device__ __forceinline__ unsigned bucket_of(unsigned row, unsigned nb_log2) {
    return (row * 0x9E3779B1u) >> (32 - nb_log2);
}

// lr_t of TF1 Adam, evaluated exactly as recalgo_adam_tf1_step does
__device__ __forceinline__ float lr_t_of(float lr, float b1, float b2, long long t) {
    const double td = (double)t;
    return (float)((double)lr * sqrt(1.0 - pow((double)b2, td)) / (1.0 - pow((double)b1, td)));
}

using recalgo_deferred::vadam;

// ---- row state access: VEC = 4 (K % 4 == 0: lane q of a group holds floats 4q .. 4q+3) or 1 ---------------------
template <int VEC> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<1> { using T = float; };
template <int VEC> __device__ __forceinline__ typename Vec<VEC>::T vz();
template <> __device__ __forceinline__ float4 vz<4>() { return f4_zero(); }
template <> __device__ __forceinline__ float vz<1>() { return 0.f; }
__device__ __forceinline__ void vadd(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ void vadd(float& a, const float b) { a += b; }
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
    recalgo_deferred::replay(w, m, v, s, target, D.lr_ring, D.b1, D.b2, D.eps);
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

// the same with ONE float per lane (K lanes of a group of L1 >= K own the row): a lane's replay is a dependent chain of
// ~25 instructions per missed step, so four times as many, four times shorter chains finish sooner than float4 lanes
__device__ __forceinline__ void catch_up_row_scalar(const Deferred& D, long long row, int s, int target, unsigned lane, unsigned K) {
    if (lane < K) {
        const size_t o = (size_t)row * K + lane;
        float w = D.w[o], m = D.m[o], v = D.v[o];
        recalgo_deferred::replay1(w, m, v, s, target, D.lr_ring, D.b1, D.b2, D.eps);
        D.w[o] = w;
        D.m[o] = m;
        D.v[o] = v;
    }
    if (lane == 0) D.last_step[row] = target;
}

// The sweep's unit of work: one group of L1 lanes walks R CONSECUTIVE rows — last_step first (four loads in flight), the
// replay only for rows that lag.  R = 1 for tables of the benchmark's size (most parallel slack for the replay chains);
// for 100 M-row tables, where nearly every row of a chunk is untouched, R = 32 cuts the launch from 200 k workgroups
// (dispatch-bound: ~90 us per step) to 6 k.
__device__ __forceinline__ void sweep_rows(const Deferred& D, long long row0, long long end, unsigned R, int target, unsigned lane,
                                           unsigned K) {
    for (unsigned r0 = 0; r0 < R; r0 += 4) {
        int sv[4];
#pragma unroll
        for (unsigned u = 0; u < 4; ++u) {
            const long long row = row0 + r0 + u;
            sv[u] = (r0 + u < R && row < end) ? D.last_step[row] : 0;
        }
#pragma unroll
        for (unsigned u = 0; u < 4; ++u)
            if (sv[u] > 0 && sv[u] < target) catch_up_row_scalar(D, row0 + r0 + u, sv[u], target, lane, K);
    }
}
inline unsigned sweep_rows_per_group(long long rows_in_launch) {
    const long long r = rows_in_launch / 65536;
    return (unsigned)(r < 1 ? 1 : (r > 32 ? 32 : r));
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
// 1. prepare: row w of the count matrix for one lookup's workgroups (+ deferred-Adam catch-up of the requested rows)
// ---------------------------------------------------------------------------------------------
struct PrepareArgs {
    SrcDev S;                      // S.first = the source's first request in the plan (a multiple of kThreads)
    unsigned short* C;             // [W][nb] request counts per (workgroup, bucket); nullptr: catch-up only
    unsigned nb_log2;
    Deferred D;                    // D.last_step == nullptr: no catch-up
    const long long* step;         // catch-up target = step[0] + step_off
    int step_off;
    unsigned KV, L;                // row = KV pieces of VEC floats, owned by L >= KV lanes (L a power of two <= 64)
    unsigned L1;                   // lanes of a one-float-per-lane group: the power of two >= K (<= 256)
    int* stale_rows; int* stale_s; // the rows this launch claimed for the catch-up kernel, and the step they are valid for
    unsigned* stale_n;             // this lookup's entry count (cleared by the step's `scan` launch)
};

template <int VEC>
__global__ __launch_bounds__(kThreads) void sparse_prepare_kernel(PrepareArgs A) {
    extern __shared__ unsigned lds_u[];
    const unsigned nb = 1u << A.nb_log2;
    unsigned* hkey = lds_u;                                   // [kSlots]     the tile's distinct rows ...
    unsigned* mask = hkey + kSlots;                           // [kSlots][8]  ... and which threads request them
    unsigned* hist = mask + kSlots * 8;                       // [nb]
    long long* stale_row = reinterpret_cast<long long*>(hist + nb);    // [kThreads]
    int* stale_s = reinterpret_cast<int*>(stale_row + kThreads);        // [kThreads]
    unsigned* n_stale = reinterpret_cast<unsigned*>(stale_s + kThreads);
    if (A.C)
        for (unsigned b = threadIdx.x; b < nb; b += kThreads) hist[b] = 0;
    if (threadIdx.x == 0) *n_stale = 0;
    const unsigned li = blockIdx.x * kThreads + threadIdx.x;
    unsigned refl = 0;
    const long long row = li < A.S.n ? slot_row(A.S, li, &refl) : -1;
    const int target = A.D.last_step ? (int)(A.step[0] + A.step_off) : 0;
    unsigned slot;
    const EqInfo eq = tile_equal(row >= 0 ? (unsigned)row : 0xffffffffu, hkey, mask, &slot);
    // one entry per DISTINCT row of the tile (its first request, the leader) — `place` sums the tile's duplicates
    if (row >= 0 && eq.before == 0) {
        if (A.C) atomicAdd(&hist[bucket_of((unsigned)row, A.nb_log2)], 1u);     // (LDS)
        if (A.D.last_step) {
            // hot rows are current (their last_step is the previous step): only stale rows cost an atomic, and exactly one
            // of the tiles that request a stale row wins the claim
            const int s = __hip_atomic_load(&A.D.last_step[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s > 0 && s < target && atomicCAS(&A.D.last_step[row], s, -s) == s) {
                const unsigned k = atomicAdd(n_stale, 1u);
                stale_row[k] = row;
                stale_s[k] = s;
            }
        }
    }
    __syncthreads();
    if (A.C) {
        unsigned short* crow = A.C + ((size_t)(A.S.first / kThreads) + blockIdx.x) * nb;
        for (unsigned b = threadIdx.x; b < nb; b += kThreads) crow[b] = (unsigned short)hist[b];
    }
    // the claimed rows go to the plan's global list: whichever tile meets a row first claims it, so the early tiles of a
    // field hold most of the claims — the catch-up itself is a separate, evenly spread launch (sparse_catchup_kernel)
    const unsigned ns = *n_stale;
    __syncthreads();                                          // (every wave has read the count before it is overwritten)
    if (ns) {
        if (threadIdx.x == 0) *n_stale = A.S.first + atomicAdd(A.stale_n, ns);     // (this lookup's part of the list)
        __syncthreads();
        const unsigned base = *n_stale;
        for (unsigned k = threadIdx.x; k < ns; k += kThreads) {
            A.stale_rows[base + k] = (int)stale_row[k];
            A.stale_s[base + k] = stale_s[k];
        }
    }
}

// catch-up of the rows `prepare` claimed: one float per task, tasks (row, element) spread over the whole grid
struct CatchupArgs {
    Deferred D;
    Deferred D1;                   // the companion arena (one float per row, its own last_step); last_step == nullptr: none
    const int* stale_rows; const int* stale_s;               // this lookup's part of the list
    const unsigned* stale_n;
    const long long* step;
    int step_off;
    unsigned K;
};
__global__ __launch_bounds__(kThreads) void sparse_catchup_kernel(CatchupArgs A) {
    const unsigned ns = A.stale_n[0];
    const int target = (int)(A.step[0] + A.step_off);
    const unsigned per_row = A.K + (A.D1.last_step ? 1u : 0u);
    const unsigned ntask = ns * per_row;
    for (unsigned task = blockIdx.x * kThreads + threadIdx.x; task < ntask; task += gridDim.x * kThreads) {
        const unsigned k = task / per_row, e = task - k * per_row;
        // ONE code path for both arenas (a divergent branch would run the two replay loops of a wave one after the other):
        // lane e < K owns float e of the row, lane e == K the row's float in the companion arena
        const int row = A.stale_rows[k];
        const bool comp = e == A.K;
        int s = A.stale_s[k];
        if (comp) s = A.D1.last_step[row];
        const bool live = !comp || (s > 0 && s < target);
        if (!live) s = target;
        float* pw = comp ? A.D1.w + row : A.D.w + (size_t)row * A.K + e;
        float* pm = comp ? A.D1.m + row : A.D.m + (size_t)row * A.K + e;
        float* pv = comp ? A.D1.v + row : A.D.v + (size_t)row * A.K + e;
        float w = *pw, m = *pm, v = *pv;
        recalgo_deferred::replay1(w, m, v, s, target, comp ? A.D1.lr_ring : A.D.lr_ring, A.D.b1, A.D.b2, A.D.eps);
        if (live) { *pw = w; *pm = m; *pv = v; }
        if (e == 0) A.D.last_step[row] = target;
        if (comp && live) A.D1.last_step[row] = target;
    }
}

// ---------------------------------------------------------------------------------------------
// 2. scan: Cp[w][b] = sum_{w' < w} C[w'][b],  total[b] = sum_w C[w][b]
//    workgroup = 16 columns x 16 row lanes; a lane owns a contiguous range of the W rows
// ---------------------------------------------------------------------------------------------
struct ScanArgs {
    const unsigned short* C; unsigned* Cp; unsigned* total; unsigned* stale_n;
    unsigned W, nb, scan_blocks;
    // sweep (deferred Adam), in the extra workgroups of this launch: rows [c * chunk, (c + 1) * chunk), c = target % period,
    // are brought to `target` — beside the scan's 64 workgroups the sweep has the chip to itself, and it touches neither
    // the plan nor (before `apply`) any row another kernel of this launch sequence is working on
    Deferred D;
    const long long* step;
    int step_off;
    unsigned KV, L, L1;
    long long rows, chunk;
    int period;
    unsigned R;                    // consecutive rows per group of L1 lanes (sweep_rows)
    // the companion arena's share of the sweep (one float per row): workgroups from comp_first on
    Deferred D1;
    long long rows1, chunk1;
    unsigned comp_first, R1;
};

template <int VEC>
__global__ __launch_bounds__(kThreads) void sparse_scan_kernel(ScanArgs A) {
    if (blockIdx.x >= A.comp_first) {                         // ---- sweep of the companion arena -------------------
        const int target = (int)(A.step[0] + A.step_off);
        if (target <= 0) return;
        const long long c0 = (long long)(target % A.period) * A.chunk1;
        const long long row0 = c0 + ((long long)(blockIdx.x - A.comp_first) * kThreads + threadIdx.x) * A.R1;
        sweep_rows(A.D1, row0, min(A.rows1, c0 + A.chunk1), A.R1, target, 0, 1);
        return;
    }
    if (blockIdx.x >= A.scan_blocks) {                        // ---- sweep workgroups -------------------------------
        const int target = (int)(A.step[0] + A.step_off);
        if (target <= 0) return;
        const long long c0 = (long long)(target % A.period) * A.chunk;
        const unsigned idx = (blockIdx.x - A.scan_blocks) * kThreads + threadIdx.x;
        const long long row0 = c0 + (long long)(idx / A.L1) * A.R;
        sweep_rows(A.D, row0, min(A.rows, c0 + A.chunk), A.R, target, idx & (A.L1 - 1), A.KV * VEC);
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x < 16 && A.stale_n) A.stale_n[threadIdx.x] = 0;   // the step's catch-up lists are consumed
    const unsigned short* __restrict__ C = A.C;
    unsigned* __restrict__ Cp = A.Cp;
    const unsigned W = A.W, nb = A.nb;
    __shared__ unsigned part[16][17];
    const unsigned c = threadIdx.x & 15, r = threadIdx.x >> 4;
    const unsigned col = blockIdx.x * 16 + c;
    const unsigned Q = (W + 15) / 16;
    const unsigned r0 = min(W, r * Q), r1 = min(W, r0 + Q);
    unsigned s = 0;
#pragma unroll 8
    for (unsigned row = r0; row < r1; ++row) s += C[(size_t)row * nb + col];
    part[r][c] = s;
    __syncthreads();
    unsigned run = 0;
#pragma unroll
    for (unsigned k = 0; k < 16; ++k)
        if (k < r) run += part[k][c];
    if (r == 15) A.total[col] = run + s;
#pragma unroll 8
    for (unsigned row = r0; row < r1; ++row) {
        const unsigned v = C[(size_t)row * nb + col];
        Cp[(size_t)row * nb + col] = run;
        run += v;
    }
}

// ---------------------------------------------------------------------------------------------
// 3. place: keys into bucket ranges in slot order, tile duplicates pre-combined
// ---------------------------------------------------------------------------------------------
struct PlaceArgs {
    SrcDev src[kMaxSources];
    int n_src;
    unsigned n_total, req_blocks;
    const unsigned* total; const unsigned* Cp; unsigned* offs;      // [nb], [W][nb], [nb + 1]
    unsigned* order;                           // [nb]: the buckets in the order `apply` takes them (the heavy ones first)
    unsigned long long* keys;                  // [n_total]
    float* partials;                           // [n_total][K]: the summed gradient rows of a tile's duplicated rows
    float* partials1;                          // [n_total]: the same for the companion's scalar gradients (nullptr: no companion)
    unsigned nb_log2;
    unsigned KV, L;
    int stage_ok;                              // the tile's duplicated gradient rows fit the LDS staging area ([256][K] floats)
};

constexpr unsigned kTileLong = 24;             // duplicates of a row in a tile above which the whole workgroup sums them

// gradient piece q of the request `ref` (= its source's first slot + e * F + f)
template <int VEC>
__device__ __forceinline__ typename Vec<VEC>::T load_g_src(const SrcDev* lsrc, unsigned ref, unsigned q) {
    using V = typename Vec<VEC>::T;
    unsigned si = 0;                                          // (unused sources have first = 0xffffffff)
#pragma unroll
    for (int k = 1; k < kMaxSources; ++k) si += ref >= lsrc[k].first;
    const unsigned first = lsrc[si].first, F = lsrc[si].F, col = lsrc[si].g_col, fmul = lsrc[si].g_fmul;
    const float* g = lsrc[si].g;
    const long long stride = lsrc[si].g_stride;
    const unsigned i = ref - first, e = i / F, f = i - e * F;
    return *reinterpret_cast<const V*>(g + (size_t)e * stride + col + (size_t)f * fmul + q * VEC);
}

// the companion's scalar gradient of the request `ref` (0 for a source without one)
__device__ __forceinline__ float load_g1_src(const SrcDev* lsrc, unsigned ref) {
    unsigned si = 0;
#pragma unroll
    for (int k = 1; k < kMaxSources; ++k) si += ref >= lsrc[k].first;
    const float* g1 = lsrc[si].g1;
    if (!g1) return 0.f;
    const unsigned i = ref - lsrc[si].first, F = lsrc[si].F, e = i / F, f = i - e * F;
    return g1[(size_t)e * lsrc[si].g1_stride + lsrc[si].g1_col + (size_t)f * lsrc[si].g1_fmul];
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void sparse_place_kernel(PlaceArgs A) {
    using V = typename Vec<VEC>::T;
    extern __shared__ unsigned lds_u[];
    const unsigned nb = 1u << A.nb_log2, bpt = nb / kThreads;   // nb is a multiple of kThreads
    unsigned* rows = lds_u;                                   // [kThreads] the tile's rows
    unsigned* bk = rows + kThreads;                           // [kThreads] bucket of the tile's leaders (else none)
    unsigned* samec = bk + kThreads;                          // [kThreads] requests of the thread's row in the tile
    unsigned* mbase = samec + kThreads;                       // [kThreads] (leaders of duplicated rows) first entry in mlist
    unsigned* mlist = mbase + kThreads;                       // [kThreads] member threads of the duplicated rows, in order
    unsigned* grefs = mlist + kThreads;                       // [kThreads] gradient reference of every thread's request
    unsigned* jobs = grefs + kThreads;                        // [kThreads] leaders of the duplicated rows
    float* red = reinterpret_cast<float*>(jobs + kThreads);   // [kThreads * 4]
    unsigned* offs = reinterpret_cast<unsigned*>(red + kThreads * 4);   // [nb]
    unsigned* sh = offs + nb;                                 // [8]; sh[6] = number of jobs, sh[7] = number of long jobs
    // the source descriptors go to LDS: a data-dependent index into the kernel-argument array would go through scratch
    unsigned* ljobs = sh + 8;                                 // [16] leaders of the rows with > kTileLong duplicates (<= 10)
    unsigned* hkey = ljobs + 16;                              // [kSlots]     equal-key bookkeeping (tile_equal)
    unsigned* mask = hkey + kSlots;                           // [kSlots][8]
    SrcDev* lsrc = reinterpret_cast<SrcDev*>(mask + kSlots * 8);     // [kMaxSources]
    copy_kernarg_words(reinterpret_cast<unsigned*>(lsrc), offsetof(PlaceArgs, src), sizeof(SrcDev) * kMaxSources);
    if (threadIdx.x == 0) { sh[6] = 0; sh[7] = 0; }
    __syncthreads();
    // this thread's slot: its row first (ids / row bases from memory), the scan of the bucket totals runs in its shadow
    const unsigned i = blockIdx.x * kThreads + threadIdx.x;
    long long row = -1;
    unsigned gref = 0;
    if (i < A.n_total) {
        unsigned si = 0;                                      // (unused sources have first = 0xffffffff)
#pragma unroll
        for (int k = 1; k < kMaxSources; ++k) si += i >= lsrc[k].first;
        const SrcDev S = lsrc[si];
        unsigned refl = 0;
        if (i - S.first < S.n) row = slot_row(S, i - S.first, &refl);     // (padding between two sources: no request)
        gref = S.first + refl;
    }
    {
        unsigned sum = 0;
        for (unsigned k = 0; k < bpt; ++k) sum += A.total[threadIdx.x * bpt + k];
        unsigned total;
        unsigned run = block_excl_scan(sum, sh, total);
        for (unsigned k = 0; k < bpt; ++k) {
            const unsigned b = threadIdx.x * bpt + k;
            offs[b] = run;
            if (blockIdx.x == 0) A.offs[b] = run;
            run += A.total[b];
        }
        if (blockIdx.x == 0 && threadIdx.x == kThreads - 1) A.offs[nb] = total;
        if (blockIdx.x == 0) {
            // `apply` runs one workgroup per bucket, more of them than fit the chip at once for large plans: the buckets that
            // hold a hot row (many entries: a long tail of the launch when they start late) are dispatched first
            const unsigned heavy_min = 2u * (total >> A.nb_log2) + 64u;
            unsigned nh = 0;
            for (unsigned k = 0; k < bpt; ++k) nh += A.total[threadIdx.x * bpt + k] >= heavy_min;
            unsigned n_heavy;
            unsigned hrun = block_excl_scan(nh, sh, n_heavy);
            for (unsigned k = 0; k < bpt; ++k) {
                const unsigned b = threadIdx.x * bpt + k;
                if (A.total[b] >= heavy_min) A.order[hrun++] = b;
                else A.order[n_heavy + b - hrun] = b;       // (b - hrun = the light buckets before b)
            }
        }
    }
    rows[threadIdx.x] = row >= 0 ? (unsigned)row : 0xffffffffu;
    grefs[threadIdx.x] = gref;
    unsigned slot;
    const EqInfo d = tile_equal(row >= 0 ? (unsigned)row : 0xffffffffu, hkey, mask, &slot);
    const bool leader = row >= 0 && d.before == 0, dupl = leader && d.same > 1;
    const unsigned b = leader ? bucket_of((unsigned)row, A.nb_log2) : 0xffffffffu;
    bk[threadIdx.x] = b;
    samec[threadIdx.x] = d.same;
    {
        unsigned total;
        mbase[threadIdx.x] = block_excl_scan(dupl ? d.same : 0u, sh, total);
    }
    if (dupl) jobs[atomicAdd(&sh[6], 1u)] = threadIdx.x;
    __syncthreads();
    if (row >= 0 && d.same > 1) mlist[mbase[d.leader] + d.before] = threadIdx.x;
    // stable: the number of EARLIER leaders of this tile that go to the same bucket
    unsigned slot_b;
    const unsigned r = tile_equal(b, hkey, mask, &slot_b).before;
    if (leader) {
        // a duplicated row's entry refers to the tile's partial sum (written below), a single request to its own row
        A.keys[offs[b] + A.Cp[(size_t)blockIdx.x * nb + b] + r] = ((unsigned long long)row << 32) | (dupl ? (kPartialBit | i) : gref);
    }
    // (no barrier here: the staging loads below only need what was written before the last one, and overlap the Cp load)
    // ---- the duplicated rows of the tile: gradient rows added in request order ---------------------------------------
    const unsigned L = A.L, q = threadIdx.x & (L - 1), grp = threadIdx.x / L, ngrp = kThreads / L;
    const unsigned njobs = sh[6];
    if (njobs == 0) return;                                   // (uniform)
    if (A.stage_ok) {
        // ONE round trip to memory: every duplicated request's gradient row goes to LDS (a group of L lanes per row, all
        // loads of a thread in flight together), then each duplicated row is summed from LDS, in request order
        V* stage = reinterpret_cast<V*>(lsrc + kMaxSources);  // [kThreads][KV]
        float* stage1 = reinterpret_cast<float*>(stage + (size_t)kThreads * A.KV);      // [kThreads] (companion only)
        if (A.partials1) {
            const unsigned tm = threadIdx.x;
            stage1[tm] = (samec[tm] > 1 && rows[tm] != 0xffffffffu) ? load_g1_src(lsrc, grefs[tm]) : 0.f;
        }
        V gq[4];
        for (unsigned r0 = 0; r0 < L; r0 += 4) {              // a group owns the member threads grp, grp + ngrp, ...
#pragma unroll
            for (unsigned u = 0; u < 4; ++u) {
                const unsigned tm = grp + (r0 + u) * ngrp;
                gq[u] = vz<VEC>();
                if (r0 + u < L && q < A.KV && samec[tm] > 1 && rows[tm] != 0xffffffffu) gq[u] = load_g_src<VEC>(lsrc, grefs[tm], q);
            }
#pragma unroll
            for (unsigned u = 0; u < 4; ++u) {
                const unsigned tm = grp + (r0 + u) * ngrp;
                if (r0 + u < L && q < A.KV) stage[tm * A.KV + q] = gq[u];
            }
        }
        __syncthreads();
        for (unsigned k = grp; k < njobs; k += ngrp) {
            const unsigned ld = jobs[k], c = samec[ld], bs = mbase[ld];
            if (q < A.KV) {
                V acc = vz<VEC>();
                for (unsigned m = 0; m < c; ++m) vadd(acc, stage[mlist[bs + m] * A.KV + q]);
                reinterpret_cast<V*>(A.partials)[((size_t)blockIdx.x * kThreads + ld) * A.KV + q] = acc;
            }
            if (A.partials1 && q == 0) {                      // the companion's scalars of the same requests, same order
                float a1 = 0.f;
                for (unsigned m = 0; m < c; ++m) a1 += stage1[mlist[bs + m]];
                A.partials1[(size_t)blockIdx.x * kThreads + ld] = a1;
            }
        }
        return;
    }
    // wide rows (the staging tile would not fit): gradient rows straight from memory
    __syncthreads();
    for (unsigned k = grp; k < njobs; k += ngrp) {
        const unsigned ld = jobs[k], c = samec[ld], bs = mbase[ld];
        if (c > kTileLong) {                                  // (at most 256 / 25 = 10 of them per tile)
            if (q == 0) ljobs[atomicAdd(&sh[7], 1u)] = ld;
            continue;
        }
        V acc = vz<VEC>();
        if (q < A.KV) {
            unsigned m = 0;
            for (; m + 4 <= c; m += 4) {
                const V g0 = load_g_src<VEC>(lsrc, grefs[mlist[bs + m]], q), g1 = load_g_src<VEC>(lsrc, grefs[mlist[bs + m + 1]], q);
                const V g2 = load_g_src<VEC>(lsrc, grefs[mlist[bs + m + 2]], q), g3 = load_g_src<VEC>(lsrc, grefs[mlist[bs + m + 3]], q);
                vadd(acc, g0); vadd(acc, g1); vadd(acc, g2); vadd(acc, g3);
            }
            for (; m < c; ++m) vadd(acc, load_g_src<VEC>(lsrc, grefs[mlist[bs + m]], q));
            reinterpret_cast<V*>(A.partials)[((size_t)blockIdx.x * kThreads + ld) * A.KV + q] = acc;
        }
    }
    __syncthreads();
    const unsigned nlong = sh[7];
    for (unsigned k = 0; k < nlong; ++k) {                    // hot rows: all groups sum strided slices, fixed-order combination
        const unsigned ld = ljobs[k], c = samec[ld], bs = mbase[ld];
        V acc = vz<VEC>();
        if (q < A.KV) {
            for (unsigned m = grp; m < c; m += ngrp) vadd(acc, load_g_src<VEC>(lsrc, grefs[mlist[bs + m]], q));
            reinterpret_cast<V*>(red)[grp * A.KV + q] = acc;
        }
        __syncthreads();
        if (grp == 0 && q < A.KV) {
            V tot = vz<VEC>();
            for (unsigned g2 = 0; g2 < ngrp; ++g2) vadd(tot, reinterpret_cast<const V*>(red)[g2 * A.KV + q]);
            reinterpret_cast<V*>(A.partials)[((size_t)blockIdx.x * kThreads + ld) * A.KV + q] = tot;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// 4. apply
// ---------------------------------------------------------------------------------------------
struct GSrc {                      // what `apply` needs of a source: where the gradient row of a request is
    const float* g;
    long long g_stride;
    unsigned g_col, g_fmul, F, first;
};
struct ApplyArgs {
    GSrc src[kMaxSources];
    int n_src;
    const unsigned* offs;
    const unsigned* order;         // [nb]: workgroup i takes bucket order[i]
    const unsigned long long* keys; unsigned long long* keys_alt;   // keys_alt: scratch of the same size (large buckets)
    const float* partials;         // [slots][K]: gradient rows of the entries whose ref has kPartialBit set
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

// bitonic sort of m (power of two) keys in LDS by all threads of the workgroup (fallback of `apply` only)
__device__ __forceinline__ void lds_bitonic(unsigned long long* keys, unsigned m) {
    const unsigned half = m >> 1;
    for (unsigned k = 2; k <= m; k <<= 1)
        for (unsigned j = k >> 1; j > 0; j >>= 1) {
            for (unsigned t0 = threadIdx.x; t0 < half; t0 += 4 * kThreads) {
                unsigned lo[4];
                unsigned long long a[4], c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned t = t0 + u * kThreads;
                    lo[u] = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    if (t < half) { a[u] = keys[lo[u]]; c[u] = keys[lo[u] | j]; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned t = t0 + u * kThreads;
                    const bool up = (lo[u] & k) == 0;
                    if (t < half && (a[u] > c[u]) == up) { keys[lo[u]] = c[u]; keys[lo[u] | j] = a[u]; }
                }
            }
            __syncthreads();
        }
}

// gradient piece q of the request a key refers to
template <int VEC>
__device__ __forceinline__ typename Vec<VEC>::T load_g(const ApplyArgs& A, const GSrc* lsrc, unsigned long long key, unsigned q) {
    using V = typename Vec<VEC>::T;
    const unsigned ref = (unsigned)key;
    if (ref & kPartialBit)                                    // the summed duplicates of a tile (written by `place`)
        return reinterpret_cast<const V*>(A.partials)[(size_t)(ref & ~kPartialBit) * A.KV + q];
    unsigned si = 0;                                          // (unused sources have first = 0xffffffff)
#pragma unroll
    for (int k = 1; k < kMaxSources; ++k) si += ref >= lsrc[k].first;
    const GSrc S = lsrc[si];
    if (!S.g) return vz<VEC>();                               // (companion pass: a lookup that had no companion adds nothing)
    const unsigned i = ref - S.first, e = i / S.F, f = i - e * S.F;
    return *reinterpret_cast<const V*>(S.g + (size_t)e * S.g_stride + S.g_col + (size_t)f * S.g_fmul + q * VEC);
}

template <int VEC> struct RowState { typename Vec<VEC>::T w, m, v; int s; };

// the owner's loads of a row's state, issued BEFORE the gradient rows are summed (one memory round trip, not two)
template <int VEC>
__device__ __forceinline__ RowState<VEC> load_state(const ApplyArgs& A, unsigned row, unsigned q) {
    using V = typename Vec<VEC>::T;
    RowState<VEC> st{vz<VEC>(), vz<VEC>(), vz<VEC>(), 0};
    if (q >= A.KV) return st;
    const size_t o = (size_t)row * A.KV + q;
    if (A.mode == RECALGO_SCATTER_GRAD) {
        st.w = reinterpret_cast<const V*>(A.grad)[o];
    } else {
        st.w = reinterpret_cast<const V*>(A.w)[o];
        st.m = reinterpret_cast<const V*>(A.m)[o];
        st.v = reinterpret_cast<const V*>(A.v)[o];
        if (A.mode == RECALGO_SCATTER_ADAM) st.s = A.last_step[row];
    }
    return st;
}

// what the owner of a row does with the row's summed gradient
template <int VEC>
__device__ __forceinline__ void finish_row(const ApplyArgs& A, unsigned row, RowState<VEC> st, typename Vec<VEC>::T acc, unsigned q,
                                           int t, float lr_t) {
    using V = typename Vec<VEC>::T;
    const size_t o = (size_t)row * A.KV + q;
    if (A.mode == RECALGO_SCATTER_GRAD) {
        if (q < A.KV) {
            vadd(st.w, acc);
            reinterpret_cast<V*>(A.grad)[o] = st.w;
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
        if (A.mode == RECALGO_SCATTER_ADAM && st.s > 0 && st.s < t - 1) {     // (normally done by `prepare`; kept for lookups
            Deferred D{A.w, A.m, A.v, A.last_step, A.lr_ring, A.b1, A.b2, A.eps};   //  that were registered without one)
            replay<VEC>(st.w, st.m, st.v, st.s, t - 1, D);
        }
        vadam(st.w, acc, st.m, st.v, lr_t, A.b1, A.b2, A.eps);
        reinterpret_cast<V*>(A.w)[o] = st.w;
        reinterpret_cast<V*>(A.m)[o] = st.m;
        reinterpret_cast<V*>(A.v)[o] = st.v;
        if (A.grad) reinterpret_cast<V*>(A.grad)[o] = vz<VEC>();
    }
    if (q == 0 && A.mode == RECALGO_SCATTER_ADAM) A.last_step[row] = t;
}

// one row (segment [lo, hi) of the grouped keys) by one group of L lanes: gradient rows added in request order
template <int VEC>
__device__ __forceinline__ void short_row(const ApplyArgs& A, const GSrc* lsrc, const unsigned long long* keys, unsigned lo,
                                          unsigned hi, unsigned q, int t, float lr_t) {
    using V = typename Vec<VEC>::T;
    const unsigned row = key_row(keys[lo]);
    const RowState<VEC> st = load_state<VEC>(A, row, q);
    V acc = vz<VEC>();
    if (q < A.KV) {
        unsigned j = lo;
        for (; j + 4 <= hi; j += 4) {                         // four row loads in flight, added in request order
            const V g0 = load_g<VEC>(A, lsrc, keys[j], q), g1 = load_g<VEC>(A, lsrc, keys[j + 1], q);
            const V g2 = load_g<VEC>(A, lsrc, keys[j + 2], q), g3 = load_g<VEC>(A, lsrc, keys[j + 3], q);
            vadd(acc, g0); vadd(acc, g1); vadd(acc, g2); vadd(acc, g3);
        }
        for (; j < hi; ++j) vadd(acc, load_g<VEC>(A, lsrc, keys[j], q));
    }
    finish_row<VEC>(A, row, st, acc, q, t, lr_t);
}

// rows with many requests: all groups of the workgroup sum strided slices, fixed-order combination through LDS
template <int VEC>
__device__ __forceinline__ void long_rows(const ApplyArgs& A, const GSrc* lsrc, const unsigned long long* keys,
                                          const unsigned* long_list, unsigned nl, float* red, int t, float lr_t) {
    using V = typename Vec<VEC>::T;
    const unsigned L = A.L, q = threadIdx.x & (L - 1), grp = threadIdx.x / L, ngrp = kThreads / L;
    for (unsigned k = 0; k < nl; ++k) {
        const unsigned lo = long_list[2 * k], hi = long_list[2 * k + 1];
        const unsigned row = key_row(keys[lo]);
        RowState<VEC> st{vz<VEC>(), vz<VEC>(), vz<VEC>(), 0};
        if (grp == 0) st = load_state<VEC>(A, row, q);
        V acc = vz<VEC>();
        if (q < A.KV) {
            constexpr int kU = 4;                             // row loads in flight (register budget: 4 workgroups / CU)
            unsigned j = lo + grp;
            for (; j + (kU - 1) * ngrp < hi; j += kU * ngrp) {
                V gq[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) gq[u] = load_g<VEC>(A, lsrc, keys[j + u * ngrp], q);
#pragma unroll
                for (int u = 0; u < kU; ++u) vadd(acc, gq[u]);
            }
            V gq[kU];
            unsigned cnt = 0;
#pragma unroll
            for (int u = 0; u < kU; ++u)
                if (j + u * ngrp < hi) { gq[u] = load_g<VEC>(A, lsrc, keys[j + u * ngrp], q); cnt = u + 1; }
#pragma unroll
            for (int u = 0; u < kU; ++u)
                if ((unsigned)u < cnt) vadd(acc, gq[u]);
            reinterpret_cast<V*>(red)[grp * A.KV + q] = acc;
        }
        __syncthreads();
        if (grp == 0) {
            V tot = vz<VEC>();
            if (q < A.KV)
                for (unsigned g2 = 0; g2 < ngrp; ++g2) vadd(tot, reinterpret_cast<const V*>(red)[g2 * A.KV + q]);
            finish_row<VEC>(A, row, st, tot, q, t, lr_t);
        }
        __syncthreads();
    }
}

// segments listed in LDS (seg_lo / seg_n): groups take them round robin; long rows are deferred to long_rows()
template <int VEC>
__device__ __forceinline__ void process_segments(const ApplyArgs& A, const GSrc* lsrc, const unsigned long long* keys,
                                                 const unsigned* seg_lo, const unsigned* seg_n, unsigned nseg,
                                                 unsigned* long_list, unsigned* n_long, float* red, int t, float lr_t) {
    const unsigned L = A.L, q = threadIdx.x & (L - 1), grp = threadIdx.x / L, ngrp = kThreads / L;
    for (unsigned h = grp; h < nseg; h += ngrp) {
        const unsigned lo = seg_lo[h], len = seg_n[h];
        if (len > kLongSeg) {
            unsigned slot = kMaxLong;
            if (q == 0) slot = atomicAdd(n_long, 1u);
            slot = __shfl(slot, (int)((threadIdx.x & 63) & ~(L - 1)), 64);
            if (slot < kMaxLong) {
                if (q == 0) { long_list[2 * slot] = lo; long_list[2 * slot + 1] = lo + len; }
                continue;
            }
        }
        short_row<VEC>(A, lsrc, keys, lo, lo + len, q, t, lr_t);
    }
    __syncthreads();
    long_rows<VEC>(A, lsrc, keys, long_list, min(*n_long, kMaxLong), red, t, lr_t);
}

// first index in [lo, n) whose row differs from `row` (keys grouped)
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

// fallback form of process_segments for SORTED keys whose rows were not listed: every group scans a contiguous block of
// entries for row heads and owns the rows that START in its block
template <int VEC>
__device__ __forceinline__ void process_sorted_scan(const ApplyArgs& A, const GSrc* lsrc, const unsigned long long* keys, unsigned n,
                                                    unsigned* long_list, unsigned* n_long, float* red, int t, float lr_t) {
    const unsigned L = A.L, q = threadIdx.x & (L - 1), grp = threadIdx.x / L, ngrp = kThreads / L;
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
        short_row<VEC>(A, lsrc, keys, i, end, q, t, lr_t);
        i = end;
    }
    __syncthreads();
    long_rows<VEC>(A, lsrc, keys, long_list, min(*n_long, kMaxLong), red, t, lr_t);
}

// fallback: sort runs of kLdsKeys keys in LDS, then merge passes between `a` and `b` in global memory; returns the buffer
// holding the sorted keys
__device__ __forceinline__ unsigned long long* global_merge_sort(unsigned long long* a, unsigned long long* b, unsigned n,
                                                                 unsigned long long* lds_keys) {
    for (unsigned r0 = 0; r0 < n; r0 += kLdsKeys) {
        const unsigned cnt = min(kLdsKeys, n - r0);
        for (unsigned k = threadIdx.x; k < kLdsKeys; k += kThreads) lds_keys[k] = k < cnt ? a[r0 + k] : kPadKey;
        __syncthreads();
        lds_bitonic(lds_keys, kLdsKeys);
        for (unsigned k = threadIdx.x; k < cnt; k += kThreads) a[r0 + k] = lds_keys[k];
        __syncthreads();
    }
    unsigned long long* src = a;
    unsigned long long* dst = b;
    for (unsigned width = kLdsKeys; width < n; width <<= 1) {
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
__global__ __launch_bounds__(kThreads, 4) void sparse_apply_kernel(ApplyArgs A) {     // 1024 workgroups resident at once
    __shared__ unsigned long long lds_keys[kLdsKeys];         // grouped keys of buckets up to kLdsKeys requests
    __shared__ unsigned hrow[kSlots], hcnt[kSlots], hbase[kSlots], hrun[kSlots];
    __shared__ unsigned seg_lo[kMaxSeg], seg_n[kMaxSeg];
    __shared__ unsigned cs[kThreads];                         // rows (small bucket) / hash slots (a chunk of a large one)
    __shared__ unsigned long_list[2 * kMaxLong];
    __shared__ unsigned n_long, n_seg, overflow, sh[8];
    __shared__ float red[kThreads * 4];
    __shared__ float s_lr_t;
    __shared__ GSrc lsrc[kMaxSources];
    const unsigned b = A.order[blockIdx.x];
    const int t = (int)(A.step[0] + A.step_off);
    if (threadIdx.x == 0) {
        n_long = 0; n_seg = 0; overflow = 0;
        float lr_t = 0.f;
        if (A.mode != RECALGO_SCATTER_GRAD) {
            lr_t = lr_t_of(A.lr, A.b1, A.b2, t);
            if (blockIdx.x == 0 && A.lr_ring) A.lr_ring[(unsigned)t & (kLrRing - 1)] = lr_t;
        }
        s_lr_t = lr_t;
    }
    copy_kernarg_words(reinterpret_cast<unsigned*>(lsrc), offsetof(ApplyArgs, src), sizeof(GSrc) * kMaxSources);
    const unsigned beg = A.offs[b], n = A.offs[b + 1] - beg;
    const unsigned long long* in = A.keys + beg;              // the bucket's keys, in request order
    __syncthreads();
    const float lr_t = s_lr_t;
    if (n == 0) return;
    if (n <= kThreads) {
        // ---- small bucket: stable rank by comparison (one request per thread) -------------------------------
        const unsigned long long key = threadIdx.x < n ? in[threadIdx.x] : kPadKey;
        const unsigned row = key_row(key);                    // (padding: 0xffffffff, greater than every row)
        cs[threadIdx.x] = row;
        __syncthreads();
        if (threadIdx.x < n) {
            unsigned less = 0, same_before = 0, same = 0;
            const uint4* r4 = reinterpret_cast<const uint4*>(cs);
#pragma unroll 8
            for (unsigned j4 = 0; j4 < (n + 3) / 4; ++j4) {
                const uint4 v = r4[j4];
                const unsigned j = 4 * j4;
                less += (v.x < row) + (v.y < row) + (v.z < row) + (v.w < row);
                same += (v.x == row) + (v.y == row) + (v.z == row) + (v.w == row);
                same_before += (v.x == row && j < threadIdx.x) + (v.y == row && j + 1 < threadIdx.x) +
                               (v.z == row && j + 2 < threadIdx.x) + (v.w == row && j + 3 < threadIdx.x);
            }
            lds_keys[less + same_before] = key;
            if (same_before == 0) {                           // the row's first request lists the row
                const unsigned k = atomicAdd(&n_seg, 1u);
                seg_lo[k] = less;
                seg_n[k] = same;
            }
        }
        __syncthreads();
        process_segments<VEC>(A, lsrc, lds_keys, seg_lo, seg_n, n_seg, long_list, &n_long, red, t, lr_t);
        return;
    }
    // ---- large bucket: the distinct rows in an LDS hash, then a stable counting scatter ------------------------------
    unsigned long long* out = n <= kLdsKeys ? lds_keys : A.keys_alt + beg;
    for (unsigned k = threadIdx.x; k < kSlots; k += kThreads) { hrow[k] = kEmptyRow; hcnt[k] = 0; hrun[k] = 0; }
    __syncthreads();
    constexpr unsigned kStage = 8;                            // chunks of 256 keys staged in registers per round trip
    for (unsigned s0 = 0; s0 < n; s0 += kStage * kThreads) {  // pass 1: requests per distinct row
        unsigned long long kreg[kStage];
#pragma unroll
        for (unsigned u = 0; u < kStage; ++u) {
            const unsigned i = s0 + u * kThreads + threadIdx.x;
            kreg[u] = i < n ? in[i] : kPadKey;
        }
#pragma unroll
        for (unsigned u = 0; u < kStage; ++u) {
            const unsigned i = s0 + u * kThreads + threadIdx.x;
            if (i < n) {
                const unsigned row = key_row(kreg[u]);
                const unsigned slot = hash_slot(hrow, row, true);
                if (slot >= kSlots) {
                    overflow = 1;
                } else {
                    // the lanes of this wave that hold the first active lane's row (a hot row: most of them) add once
                    const unsigned first = __builtin_amdgcn_readfirstlane(row);
                    const unsigned long long same = __ballot(row == first);
                    if (row == first) {
                        if ((threadIdx.x & 63) == (unsigned)__ffsll((long long)same) - 1) atomicAdd(&hcnt[slot], (unsigned)__popcll(same));
                    } else {
                        atomicAdd(&hcnt[slot], 1u);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (overflow) {
        // more distinct rows than the hash holds (cannot happen at ~100 requests per bucket; kept correct): full sort
        const unsigned long long* sorted;
        if (n <= kLdsKeys) {
            unsigned m = 2;
            while (m < n) m <<= 1;
            for (unsigned k = threadIdx.x; k < m; k += kThreads) lds_keys[k] = k < n ? in[k] : kPadKey;
            __syncthreads();
            lds_bitonic(lds_keys, m);
            sorted = lds_keys;
        } else {
            // (sorts the bucket's range of `keys` in place: it is not read again)
            sorted = global_merge_sort(const_cast<unsigned long long*>(in), A.keys_alt + beg, n, lds_keys);
        }
        process_sorted_scan<VEC>(A, lsrc, sorted, n, long_list, &n_long, red, t, lr_t);
        return;
    }
    {   // first slot of every row's segment: exclusive scan of the counts in slot order (two slots per thread)
        const unsigned c0 = hcnt[2 * threadIdx.x], c1 = hcnt[2 * threadIdx.x + 1];
        unsigned total;
        const unsigned run = block_excl_scan(c0 + c1, sh, total);
        hbase[2 * threadIdx.x] = run;
        hbase[2 * threadIdx.x + 1] = run + c0;
        if (c0) { const unsigned k = atomicAdd(&n_seg, 1u); seg_lo[k] = run; seg_n[k] = c0; }
        if (c1) { const unsigned k = atomicAdd(&n_seg, 1u); seg_lo[k] = run + c0; seg_n[k] = c1; }
    }
    __syncthreads();
    for (unsigned s0 = 0; s0 < n; s0 += kStage * kThreads) {  // pass 2: stable scatter, 256 requests at a time
        unsigned long long kreg[kStage];
#pragma unroll
        for (unsigned u = 0; u < kStage; ++u) {
            const unsigned i = s0 + u * kThreads + threadIdx.x;
            kreg[u] = i < n ? in[i] : kPadKey;
        }
#pragma unroll
        for (unsigned u = 0; u < kStage; ++u) {
            if (s0 + u * kThreads >= n) break;                // (uniform over the workgroup)
            const unsigned i = s0 + u * kThreads + threadIdx.x;
            const unsigned long long cur = kreg[u];
            const unsigned slot = i < n ? hash_slot(hrow, key_row(cur), false) : 0xffffffffu;
            cs[threadIdx.x] = slot;
            __syncthreads();
            unsigned same_before = 0, same = 0;
            if (i < n) {
                const uint4* c4 = reinterpret_cast<const uint4*>(cs);
#pragma unroll 16
                for (unsigned j4 = 0; j4 < kThreads / 4; ++j4) {
                    const uint4 v = c4[j4];
                    const unsigned j = 4 * j4;
                    same += (v.x == slot) + (v.y == slot) + (v.z == slot) + (v.w == slot);
                    same_before += (v.x == slot && j < threadIdx.x) + (v.y == slot && j + 1 < threadIdx.x) +
                                   (v.z == slot && j + 2 < threadIdx.x) + (v.w == slot && j + 3 < threadIdx.x);
                }
                out[hbase[slot] + hrun[slot] + same_before] = cur;
            }
            __syncthreads();
            if (i < n && same_before + 1 == same) hrun[slot] += same;      // the row's last request of this chunk
            __syncthreads();
        }
    }
    process_segments<VEC>(A, lsrc, out, seg_lo, seg_n, n_seg, long_list, &n_long, red, t, lr_t);
}

// ---------------------------------------------------------------------------------------------
// standalone sweep: rows [row0, row1) brought to step[0] + step_off (flush before EVAL / PREDICT / checkpoint)
// ---------------------------------------------------------------------------------------------
struct SweepArgs {
    Deferred D;
    const long long* step;
    int step_off;
    unsigned KV, L, L1, R;
    long long row0, row1;
};
template <int VEC>
__global__ __launch_bounds__(kThreads) void sparse_sweep_kernel(SweepArgs A) {
    const int target = (int)(A.step[0] + A.step_off);
    if (target <= 0) return;
    const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
    const long long row0 = A.row0 + (idx / A.L1) * A.R;
    sweep_rows(A.D, row0, A.row1, A.R, target, (unsigned)(idx & (A.L1 - 1)), A.KV * VEC);
}

// ---- host helpers -----------------------------------------------------------------------------
struct Geometry { int vec; unsigned KV, L, L1; };
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
    unsigned L1 = 1;
    while (L1 < (unsigned)K) L1 <<= 1;
    G->L1 = L1;                                             // (K <= 256: a row never needs more than a workgroup)
    return true;
}

}  // namespace
RECALGO_EXPORT int64_t recalgo_scatter_source_slots(int n_ex, int F, int ragged) {
    if (n_ex <= 0 || F <= 0) return 0;
    const int64_t pad = [](int64_t x) { return (x + kThreads - 1) / kThreads * kThreads; }((int64_t)(ragged ? (int64_t)n_ex * F : n_ex));
    return ragged ? pad : pad * F;                          // id matrices: F fields x (examples rounded up to whole tiles)
}
namespace {
inline bool to_dev(const recalgo_scatter_source_t* src, int n_src, SrcDev* out, unsigned* n_total, bool need_g,
                   const recalgo_scatter_source_t* comp = nullptr) {
    unsigned first = 0;
    *n_total = 0;
    for (int i = 0; i < kMaxSources; ++i)
        out[i] = SrcDev{nullptr, nullptr, nullptr, 0, 0, 1, 0xffffffffu, 0, kThreads, 0, nullptr, 0, 0, 0, nullptr, 0, 0, 0};
    for (int i = 0; i < n_src; ++i) {
        const recalgo_scatter_source_t& s = src[i];
        if (!s.ids || s.n_ex < 0 || s.F < 1 || (need_g && !s.g)) return false;
        const int64_t n = recalgo_scatter_source_slots(s.n_ex, s.F, s.offsets != nullptr);
        if ((int64_t)first + n >= (1ll << 31)) return false;
        const unsigned e256 = ((unsigned)s.n_ex + kThreads - 1) / kThreads * kThreads;
        out[i] = SrcDev{s.ids, s.offsets, s.row_base, (long long)s.base, (unsigned)s.n_ex, (unsigned)s.F, first, (unsigned)n,
                        e256 ? e256 : kThreads, 0, s.g, (long long)s.g_stride, (unsigned)s.g_col, (unsigned)s.g_fmul,
                        comp ? comp[i].g : nullptr, comp ? (long long)comp[i].g_stride : 0, comp ? (unsigned)comp[i].g_col : 0u,
                        comp ? (unsigned)comp[i].g_fmul : 0u};
        // every source is a whole number of tiles: row w of the count matrix is written by the `prepare` launch of exactly
        // one source
        first += (unsigned)n;
        *n_total = first;
    }
    return true;
}

inline Deferred deferred_of(const recalgo_deferred_adam_t* d) {
    if (!d || !d->last_step) return Deferred{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f};
    return Deferred{d->w, d->m, d->v, d->last_step, d->lr_ring, d->beta1, d->beta2, d->eps};
}
inline bool nb_ok(int nb_log2) { return nb_log2 >= 8 && nb_log2 <= 13; }

struct Ws { unsigned* total; unsigned* offs; unsigned* order; unsigned* Cp; unsigned short* C; unsigned long long* keys; unsigned long long* keys_alt; float* partials;
            float* partials1; unsigned* stale_n; int* stale_rows; int* stale_s; };
inline Ws carve(void* ws, int64_t cap, int nb_log2) {
    const int64_t nb = 1ll << nb_log2, W = cap / kThreads;
    char* p = static_cast<char*>(ws);
    Ws w;
    w.stale_n = reinterpret_cast<unsigned*>(p);             // [16]: zero-filled once by the caller, kept clean by the kernels
    w.total = w.stale_n + 16;                               // [nb]
    w.offs = w.total + nb;                                  // [nb + 8]
    w.order = w.offs + nb + 8;                              // [nb]
    w.Cp = w.order + nb;                                    // [W][nb]
    w.C = reinterpret_cast<unsigned short*>(w.Cp + W * nb); // [W][nb]
    uintptr_t k = (reinterpret_cast<uintptr_t>(w.C + W * nb) + 15) & ~(uintptr_t)15;
    w.keys = reinterpret_cast<unsigned long long*>(k);
    w.keys_alt = w.keys + cap;
    w.stale_rows = reinterpret_cast<int*>(w.keys_alt + cap);   // [cap]
    w.stale_s = w.stale_rows + cap;                             // [cap]
    w.partials1 = reinterpret_cast<float*>(w.stale_s + cap);   // [cap]
    w.partials = w.partials1 + cap;                             // [cap][K]
    return w;
}

}  // namespace

RECALGO_EXPORT int recalgo_scatter_plan_buckets_log2(int64_t n_requests) {
    // ~100-160 requests per bucket (one workgroup each in `apply`), 1024 .. 8192 buckets; the count matrix is
    // [requests / 256][buckets] 16-bit words, so the bucket count grows slower than the requests
    int l = 10;
    while (l < 13 && (n_requests >> l) > 160) ++l;
    return l;
}

RECALGO_EXPORT int64_t recalgo_scatter_plan_workspace_bytes(int64_t n_slots, int nb_log2, int K) {
    if (n_slots < 0 || n_slots % kThreads != 0 || !nb_ok(nb_log2) || K < 1) return 0;
    const int64_t nb = 1ll << nb_log2;
    const int64_t cap = n_slots > 0 ? n_slots : kThreads, W = cap / kThreads;
    return (3 * nb + 8 + 16) * (int64_t)sizeof(unsigned) + W * nb * (int64_t)(sizeof(unsigned) + sizeof(unsigned short)) +
           2 * cap * (int64_t)sizeof(unsigned long long) + 2 * cap * (int64_t)sizeof(int) + cap * (int64_t)(K + 1) * (int64_t)sizeof(float) + 64;
}

RECALGO_EXPORT int recalgo_scatter_prepare(const recalgo_scatter_source_t* source, int K, void* plan_workspace,
                                           int64_t plan_requests, int nb_log2, int64_t first_request, int lookup_index,
                                           const recalgo_deferred_adam_t* deferred,
                                           const recalgo_deferred_adam_t* companion_deferred, const int64_t* step_dev,
                                           int step_offset, recalgo_stream_t stream) {
    RECALGO_REQUIRE(source != nullptr && nb_ok(nb_log2) && lookup_index >= 0 && lookup_index < kMaxSources);
    RECALGO_REQUIRE(first_request >= 0 && first_request % kThreads == 0 && first_request < (1ll << 31));
    RECALGO_REQUIRE(plan_requests >= 0 && plan_requests % kThreads == 0);
    RECALGO_REQUIRE(plan_workspace != nullptr);
    SrcDev S[kMaxSources];
    unsigned n = 0;
    RECALGO_REQUIRE(to_dev(source, 1, S, &n, false));
    if (n == 0) return 0;
    Geometry G;
    RECALGO_REQUIRE(geometry(K, nullptr, 0, &G));
    PrepareArgs A;
    A.S = S[0];
    A.S.first = (unsigned)first_request;
    A.C = nullptr;
    A.stale_rows = nullptr; A.stale_s = nullptr; A.stale_n = nullptr;
    if (plan_workspace) {
        RECALGO_REQUIRE(first_request + (int64_t)n <= plan_requests);
        const Ws ws = carve(plan_workspace, plan_requests, nb_log2);
        A.C = ws.C;
        A.stale_rows = ws.stale_rows; A.stale_s = ws.stale_s; A.stale_n = ws.stale_n + lookup_index;
    }
    A.nb_log2 = (unsigned)nb_log2;
    A.D = deferred_of(deferred);
    RECALGO_REQUIRE(A.D.last_step == nullptr || (step_dev != nullptr && A.D.lr_ring != nullptr && plan_workspace != nullptr));
    A.step = reinterpret_cast<const long long*>(step_dev);
    A.step_off = step_offset;
    A.KV = G.KV; A.L = G.L; A.L1 = G.L1;
    const size_t smem = (((size_t)1 << nb_log2) + kSlots * 9) * sizeof(unsigned) + kThreads * (sizeof(long long) + sizeof(int)) + 16;
    const dim3 grid(cdiv(n, kThreads));
    if (G.vec == 4)
        hipLaunchKernelGGL(sparse_prepare_kernel<4>, grid, dim3(kThreads), smem, as_stream(stream), A);
    else
        hipLaunchKernelGGL(sparse_prepare_kernel<1>, grid, dim3(kThreads), smem, as_stream(stream), A);
    if (A.D.last_step) {
        CatchupArgs Cu;
        Cu.D = A.D; Cu.D1 = deferred_of(companion_deferred); Cu.stale_rows = A.stale_rows + first_request; Cu.stale_s = A.stale_s + first_request; Cu.stale_n = A.stale_n;
        Cu.step = A.step; Cu.step_off = A.step_off; Cu.K = (unsigned)K;
        // at most one claim per distinct row of the lookup; the grid covers n / 4 rows x K floats in one pass
        const int64_t want = ((int64_t)n / 3 * (K + (Cu.D1.last_step ? 1 : 0)) + kThreads - 1) / kThreads;
        const unsigned blocks = (unsigned)(want < 64 ? 64 : (want > 4096 ? 4096 : want));   // (workgroups beyond the list exit at once)
        hipLaunchKernelGGL(sparse_catchup_kernel, dim3(blocks), dim3(kThreads), 0, as_stream(stream), Cu);
    }
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_scatter_apply(const recalgo_scatter_source_t* sources, int n_sources,
                                         const recalgo_scatter_companion_t* companion, int K, void* plan_workspace,
                                         int64_t plan_requests, int nb_log2, int mode, float* w, float* m, float* v,
                                         float* grad, const recalgo_deferred_adam_t* deferred, int64_t rows,
                                         int sweep_period, const recalgo_live_t* live, const int64_t* step_dev,
                                         int step_offset, float lr, float beta1, float beta2, float eps,
                                         recalgo_stream_t stream) {
    RECALGO_REQUIRE(sources != nullptr && n_sources >= 1 && n_sources <= kMaxSources && plan_workspace != nullptr);
    RECALGO_REQUIRE(nb_ok(nb_log2) && rows >= 0 && rows < (1ll << 31));
    RECALGO_REQUIRE(mode == RECALGO_SCATTER_GRAD || mode == RECALGO_SCATTER_ADAM || mode == RECALGO_SCATTER_LAZY_ADAM);
    RECALGO_REQUIRE(mode != RECALGO_SCATTER_GRAD ? (w && m && v && step_dev) : grad != nullptr);
    RECALGO_REQUIRE(mode != RECALGO_SCATTER_ADAM || (deferred && deferred->last_step && deferred->lr_ring && sweep_period >= 1 &&
                                                     sweep_period <= (int)kLrRing - 8));
    if (companion) {
        RECALGO_REQUIRE(companion->sources != nullptr && companion->rows >= 0 && companion->rows < (1ll << 31));
        RECALGO_REQUIRE((size_t)K * kThreads * sizeof(float) <= 32 * 1024);
        RECALGO_REQUIRE(mode != RECALGO_SCATTER_GRAD ? (companion->w && companion->m && companion->v) : companion->grad != nullptr);
        RECALGO_REQUIRE(mode != RECALGO_SCATTER_ADAM || (companion->deferred && companion->deferred->last_step &&
                                                         companion->deferred->lr_ring));
    }
    Geometry G;
    RECALGO_REQUIRE(geometry(K, sources, n_sources, &G));
    PlaceArgs P;
    unsigned n_total = 0;
    RECALGO_REQUIRE(to_dev(sources, n_sources, P.src, &n_total, true, companion ? companion->sources : nullptr));
    RECALGO_REQUIRE(plan_requests % kThreads == 0 && (int64_t)n_total <= plan_requests);
    const Ws ws = carve(plan_workspace, plan_requests, nb_log2);
    hipStream_t st = as_stream(stream);
    const unsigned nb = 1u << nb_log2;
    const unsigned W = (unsigned)cdiv(n_total, kThreads);     // rows of the count matrix in use (written by `prepare`)
    {
        ScanArgs S;
        S.C = ws.C; S.Cp = ws.Cp; S.total = ws.total; S.stale_n = ws.stale_n; S.W = W; S.nb = nb; S.scan_blocks = nb / 16;
        S.D = mode == RECALGO_SCATTER_ADAM ? deferred_of(deferred) : deferred_of(nullptr);
        S.step = reinterpret_cast<const long long*>(step_dev);
        S.step_off = step_offset - 1;                         // the sweep (like `prepare`) targets the step BEFORE this one
        S.KV = G.KV; S.L = G.L; S.L1 = G.L1;
        S.rows = rows;
        S.period = sweep_period < 1 ? 1 : sweep_period;
        S.chunk = (rows + S.period - 1) / S.period;
        S.R = sweep_rows_per_group(S.chunk);
        const unsigned sweep_blocks = S.D.last_step ? (unsigned)cdiv(cdiv(S.chunk, (long long)S.R) * G.L1, kThreads) : 0u;
        S.D1 = (companion && mode == RECALGO_SCATTER_ADAM) ? deferred_of(companion->deferred) : deferred_of(nullptr);
        S.rows1 = companion ? companion->rows : 0;
        S.chunk1 = (S.rows1 + S.period - 1) / S.period;
        S.comp_first = S.scan_blocks + sweep_blocks;
        S.R1 = sweep_rows_per_group(S.chunk1);
        const unsigned comp_blocks = S.D1.last_step ? (unsigned)cdiv(cdiv(S.chunk1, (long long)S.R1), kThreads) : 0u;
        if (G.vec == 4)
            hipLaunchKernelGGL(sparse_scan_kernel<4>, dim3(S.comp_first + comp_blocks), dim3(kThreads), 0, st, S);
        else
            hipLaunchKernelGGL(sparse_scan_kernel<1>, dim3(S.comp_first + comp_blocks), dim3(kThreads), 0, st, S);
    }
    P.n_src = n_sources;
    P.n_total = n_total;
    P.req_blocks = W ? W : 1;                                 // (the scan of all-zero totals still publishes offs[])
    P.total = ws.total; P.Cp = ws.Cp; P.offs = ws.offs; P.order = ws.order; P.keys = ws.keys; P.partials = ws.partials;
    P.partials1 = companion ? ws.partials1 : nullptr;
    P.nb_log2 = (unsigned)nb_log2;
    P.KV = G.KV; P.L = G.L;
    P.stage_ok = (size_t)K * kThreads * sizeof(float) <= 32 * 1024;      // (K <= 32: every model of the reference)
    const size_t smem = ((size_t)nb + 7 * kThreads + 8 + 16 + kSlots * 9) * sizeof(unsigned) + kThreads * 4 * sizeof(float) + kMaxSources * sizeof(SrcDev) +
                        (P.stage_ok ? (size_t)(K + 1) * kThreads * sizeof(float) : 0);
    if (smem > 64 * 1024) {
        hipError_t e = G.vec == 4 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&sparse_place_kernel<4>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)
                                  : hipFuncSetAttribute(reinterpret_cast<const void*>(&sparse_place_kernel<1>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    if (G.vec == 4)
        hipLaunchKernelGGL(sparse_place_kernel<4>, dim3(P.req_blocks), dim3(kThreads), smem, st, P);
    else
        hipLaunchKernelGGL(sparse_place_kernel<1>, dim3(P.req_blocks), dim3(kThreads), smem, st, P);
    ApplyArgs A;
    for (int i = 0; i < kMaxSources; ++i)
        A.src[i] = GSrc{P.src[i].g, P.src[i].g_stride, P.src[i].g_col, P.src[i].g_fmul, P.src[i].F, P.src[i].first};
    A.n_src = n_sources;
    A.offs = ws.offs; A.order = ws.order; A.keys = ws.keys; A.keys_alt = ws.keys_alt; A.partials = ws.partials;
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
    if (A.step == nullptr) {                                  // GRAD mode without a step counter: t is not used
        A.step = reinterpret_cast<const long long*>(ws.total);
        A.step_off = 0;
    }
    if (G.vec == 4)
        hipLaunchKernelGGL(sparse_apply_kernel<4>, dim3(nb), dim3(kThreads), 0, st, A);
    else
        hipLaunchKernelGGL(sparse_apply_kernel<1>, dim3(nb), dim3(kThreads), 0, st, A);
    if (companion) {
        // the second arena: the same placed entries, bucket by bucket; only the gradient (a scalar per request, the tile
        // partial sums `place` wrote beside the main ones) and the arena differ
        for (int i = 0; i < kMaxSources; ++i) {
            A.src[i].g = P.src[i].g1; A.src[i].g_stride = P.src[i].g1_stride;
            A.src[i].g_col = P.src[i].g1_col; A.src[i].g_fmul = P.src[i].g1_fmul;
        }
        A.partials = ws.partials1;
        A.w = companion->w; A.m = companion->m; A.v = companion->v; A.grad = companion->grad;
        A.last_step = mode == RECALGO_SCATTER_ADAM ? companion->deferred->last_step : nullptr;
        A.lr_ring = mode == RECALGO_SCATTER_ADAM ? companion->deferred->lr_ring : nullptr;
        A.K = 1; A.KV = 1; A.L = 1;
        A.live_words = nullptr; A.live_list = nullptr; A.live_count = nullptr;
        hipLaunchKernelGGL(sparse_apply_kernel<1>, dim3(nb), dim3(kThreads), 0, st, A);
    }
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
    A.KV = G.KV; A.L = G.L; A.L1 = G.L1;
    A.row0 = row_begin; A.row1 = row_end;
    A.R = sweep_rows_per_group(row_end - row_begin);
    const int64_t threads = cdiv(row_end - row_begin, (int64_t)A.R) * G.L1;
    const dim3 grid((unsigned)((threads + kThreads - 1) / kThreads));
    if (G.vec == 4)
        hipLaunchKernelGGL(sparse_sweep_kernel<4>, grid, dim3(kThreads), 0, as_stream(stream), A);
    else
        hipLaunchKernelGGL(sparse_sweep_kernel<1>, grid, dim3(kThreads), 0, as_stream(stream), A);
    RECALGO_RETURN_LAST();
}
