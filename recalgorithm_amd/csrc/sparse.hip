// Row-gradient scatter WITHOUT float atomics, fused with the sparse optimizer, gfx950.
//
// The backward of every embedding lookup is  dTable[row] += g[request]  over the requests (b, f) of the batch
// (Appendix D "Gather"); CTR ids are Zipf distributed and real tables include two-valued fields, so one row can own
// two thirds of a batch's requests of its field.  Rounds 1-2 combined duplicates in an LDS hash table per workgroup and
// then issued one device-scope float atomic per (distinct row, float, workgroup) — ~1.1 M fabric atomics per DCN step,
// order non-deterministic, and the summed gradient arena was then re-read (and zeroed) by the optimizer launch.  Round 3
// made the scatter OWNER-COMPUTES (a stable multisplit through a (tile x bucket) count matrix): deterministic, but five
// dependent launches (prepare, catch-up, scan, place, apply: 68 us of the DCN step for 3 us of memory traffic) and a
// workspace quadratic in the plan.  This round: THREE launches, a workspace linear in the plan, no count matrix.
//
// The plan's SLOT space is cut into tiles of 256 (an id matrix field-major: a tile is 256 consecutive examples of ONE
// field — where the duplicates of a batch are); bucket = hash(row) (1024 .. 8192 buckets, ~100 entries each).
//   1. `prepare` (one launch per lookup, before its forward gather), five kinds of workgroups in one grid:
//      COUNT tiles find the distinct rows of a tile (LDS hash + 256-bit member masks) and add ONE entry per distinct row
//      to the bucket totals (integer atomics, one per distinct row of a tile);
//      CATCH-UP workgroups (deferred Adam) take 64 requests each in request order — every workgroup sees the same mix
//      of fields, so the work is even —, claim the rows whose state lags (one integer CAS per lagging request) and
//      replay their missed g = 0 updates, one float per lane, so that the unchanged forward kernels read current weights;
//      the same for the COMPANION arena (one float per row); and the step's SWEEP of both arenas (1 / P of the rows).
//   2. `place` (after the backward pass): a tile's entries get their positions in their buckets — the tile's first entry
//      of a bucket draws a range from the bucket's cursor (one returning integer atomic per (tile, bucket)) — and the
//      duplicates of a row inside a tile are summed IN REQUEST ORDER into one partial gradient row (a hot row of a field
//      reaches its bucket as B / 256 entries, not thousands).  An entry's key is (row, slot of the tile's first request of
//      the row): unique, so the ORDER of a bucket is a property of its keys, not of the order the tiles arrived in.
//   3. `apply` (one workgroup per bucket, heavy buckets dispatched first): ranks the bucket's keys (<= 256 entries: by
//      comparison; more: LDS bitonic / global merge sort), after which the entries of a row are adjacent and in slot
//      order.  A group of K/4 lanes owns a row: it adds the row's entries in that order (bit-reproducible, whatever the
//      scheduling; rows with many entries are summed by the whole workgroup in a fixed strided order) and — the row
//      being exclusively its own — finishes in registers: TF1 Adam (dense semantics, exact), LazyAdam, or a plain
//      `grad[row] += sum`.  The same lane group finishes the row of a COMPANION arena (one float per row looked up with the
//      same requests: DeepFM's first-order weights).  A source may carry the FM second-order EPILOGUE: its request gradient
//      is  g_emb + g_fm2 * (S - e)  formed on load (deepfm.py:184-200; Appendix D "FM2") — no materialised [B, F, K] tensor.
//      The bucket's workgroup also clears the bucket's total and cursor for the next step.
// The only atomics on global memory are integer: one add per tile-distinct row (count), one returning add per (tile,
// bucket) (place), one CAS per lagging request (catch-up / sweep).
//
// Deferred exact Adam.  tf.train.AdamOptimizer applies a DENSE update to embedding variables: m, v of every row decay
// and w moves every step, gradient or not (deepfm.py:246-250).  The g = 0 update of a row is a pure
// function of its own (w, m, v) and of lr_t(step), so it is postponed: `last_step[row]` records the step the row's state
// is valid for, and whoever needs the row next (the catch-up of a lookup that requests it, the round-robin sweep, a flush
// before EVAL / PREDICT / checkpoint) replays the missed steps in registers with the SAME fp32 operations in the SAME
// order — bit-identical to the dense pass (tests/test_gpu_sparse.py), at the cost of the batch's rows.  lr_t of recent
// steps comes from a small ring written by the optimizer launch; the sweep bounds every row's lag to P + 1 steps.
#include <cstddef>
#include <cstdlib>

#include "deferred.h"

// tuning knobs of `apply` (build-time): workgroups resident per CU (the register budget follows) and gradient rows in flight
#ifndef RECALGO_APPLY_WGS
#define RECALGO_APPLY_WGS 4
#endif
#ifndef RECALGO_APPLY_LDS_KEYS
#define RECALGO_APPLY_LDS_KEYS 2048
#endif
#ifndef RECALGO_APPLY_KU
#define RECALGO_APPLY_KU 8
#endif

namespace {

constexpr int kThreads = 256;
constexpr int kMaxSources = RECALGO_SCATTER_MAX_SOURCES;
using recalgo_deferred::kLrRing;
constexpr unsigned kLdsKeys = RECALGO_APPLY_LDS_KEYS;                    // sorted keys kept in LDS (16 KB); larger buckets go through global memory
constexpr unsigned kMaxSeg = kThreads;                  // rows (segments) of a small bucket listed in LDS
constexpr unsigned kSlots = 512;                        // LDS hash of the distinct keys of a tile
constexpr unsigned kLongSeg = 48;                       // entries per row above which the whole workgroup sums it
constexpr unsigned kMaxLong = 64;                       // long rows remembered per bucket (more: summed by one group)
constexpr unsigned kCatchReq = 128;                     // requests per catch-up workgroup
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
    // FM second-order epilogue: the request's gradient row is  g + fm_scale[e] * (fm_sum[e, :] - fm_emb[e, f, :]);  nullptr: none
    const float* fm_scale; const float* fm_sum; const float* fm_emb;
};

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

// The plan's SLOT space.  Workgroup ("tile") w of count / place owns slots [256 w, 256 w + 256).  An id matrix
// [n_ex, F] is laid out FIELD-MAJOR: slot f * e256 + e  (e256 = n_ex rounded up to 256), so that a tile is 256 consecutive
// examples of ONE field.  Ragged sources stay row-major (e * F + f).  false: no request at this slot.
__device__ __forceinline__ bool slot_ef(const SrcDev& S, unsigned li, unsigned* e, unsigned* f) {
    if (S.offsets) {
        *e = li / S.F;
        *f = li - *e * S.F;
    } else {
        *f = li / S.e256;
        *e = li - *f * S.e256;
    }
    return *e < S.n_ex;
}
// arena row of request (e, f); -1: OOV id / beyond the sequence's length
__device__ __forceinline__ long long request_row(const SrcDev& S, unsigned e, unsigned f) {
    long long id;
    if (S.offsets) {
        const long long beg = S.offsets[e], len = S.offsets[e + 1] - beg;
        if ((long long)f >= len) return -1;
        id = S.ids[beg + f];
    } else {
        id = S.ids[(size_t)e * S.F + f];
    }
    if (id < 0) return -1;
    return id + S.base + (S.row_base ? S.row_base[f] : 0);
}
__device__ __forceinline__ long long slot_row(const SrcDev& S, unsigned li, unsigned* e, unsigned* f) {
    if (!slot_ef(S, li, e, f)) return -1;
    return request_row(S, *e, *f);
}
// the source a plan slot belongs to (unused sources have first = 0xffffffff)
__device__ __forceinline__ unsigned source_of(const SrcDev* lsrc, unsigned slot) {
    unsigned si = 0;
#pragma unroll
    for (int k = 1; k < kMaxSources; ++k) si += slot >= lsrc[k].first;
    return si;
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

using recalgo_deferred::adam1;
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
// g + a * (s - x): the FM second-order epilogue (the difference is rounded first, like torch.addcmul(g, a, s - x))
__device__ __forceinline__ float4 fm_epilogue(float4 g, float a, float4 s, float4 x) {
    return make_float4(fmaf(a, s.x - x.x, g.x), fmaf(a, s.y - x.y, g.y), fmaf(a, s.z - x.z, g.z), fmaf(a, s.w - x.w, g.w));
}
__device__ __forceinline__ float fm_epilogue(float g, float a, float s, float x) { return fmaf(a, s - x, g); }

struct Deferred {                  // deferred-Adam state of one arena
    float* w; float* m; float* v;
    int* last_step;                // [rows]: 0 = never touched (m = v = 0), s > 0 = (w, m, v) valid for step s, < 0 = claimed
    const float* lr_ring;          // [kLrRing]: lr_t(j) at j & (kLrRing - 1)
    float b1, b2, eps;
};

// replay the g = 0 updates of steps s+1 .. target on one lane's piece of a row
template <int VEC>
__device__ __forceinline__ void replay(typename Vec<VEC>::T& w, typename Vec<VEC>::T& m, typename Vec<VEC>::T& v, int s,
                                       int target, const float* lr_ring, float b1, float b2, float eps) {
    recalgo_deferred::replay(w, m, v, s, target, lr_ring, b1, b2, eps);
}

// gradient piece q of request (e, f) of source S
template <int VEC>
__device__ __forceinline__ typename Vec<VEC>::T load_req_g(const SrcDev& S, unsigned e, unsigned f, unsigned q, unsigned KV) {
    using V = typename Vec<VEC>::T;
    V v = *reinterpret_cast<const V*>(S.g + (size_t)e * S.g_stride + S.g_col + (size_t)f * S.g_fmul + q * VEC);
    if (S.fm_scale) {
        const float a = S.fm_scale[e];
        const V s = reinterpret_cast<const V*>(S.fm_sum)[(size_t)e * KV + q];
        const V x = reinterpret_cast<const V*>(S.fm_emb)[((size_t)e * S.F + f) * KV + q];
        v = fm_epilogue(v, a, s, x);
    }
    return v;
}
// the companion's scalar gradient of request (e, f) (0 for a source without one)
__device__ __forceinline__ float load_req_g1(const SrcDev& S, unsigned e, unsigned f) {
    if (!S.g1) return 0.f;
    return S.g1[(size_t)e * S.g1_stride + S.g1_col + (size_t)f * S.g1_fmul];
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
// 1. prepare: catch-up of the lookup's rows | the companion's | bucket counts | the sweep of both arenas
// ---------------------------------------------------------------------------------------------
// sources one `prepare` launch serves: a step's lookups into one arena that the model issues together (sparse.batch_lookups:
// DIN's profile fields, target item and history are three lookups into one arena — three launches of 6 .. 17 us until round 6)
constexpr int kPrepSources = 4;
struct PrepareArgs {
    SrcDev S;                      // source 0 (the only one of a single-lookup launch)
    SrcDev Sx[kPrepSources - 1];   // sources 1 .. n_src - 1
    unsigned n_src;
    unsigned nreq_x[kPrepSources - 1];        // n_ex * F of sources 1 ..
    unsigned catch_end[kPrepSources];         // catch-up workgroups [catch_end[i-1], catch_end[i]) walk source i's requests
    unsigned count_end[kPrepSources];         // the same for the count workgroups (relative to b_comp)
    unsigned* total;               // [nb << cs] entries per bucket (count workgroups), one counter every 1 << cs words
    unsigned nb_log2, cs;
    Deferred D;                    // D.last_step == nullptr: no deferred state (no catch-up, no sweep)
    Deferred D1;                   // the companion arena (one float per row, its own last_step); last_step == nullptr: none
    const long long* step;         // catch-up / sweep target = step[0] + step_off
    int step_off;
    unsigned K;
    unsigned n_req;                // n_ex * F: the catch-up walks the requests in REQUEST order (every workgroup the same field mix)
    unsigned b_catch, b_comp, b_count, b_sweep;       // first block of: companion catch-up, count, sweep, companion sweep
    // sweep: the arena is cut into blocks of 256 rows; step `target` takes the blocks c, c + P, c + 2 P, ... (c = target % P):
    // an even sample of every table each step (contiguous 1 / P ranges made the step time swing with the range's share of
    // live rows: 2.5 .. 25 us), coalesced `last_step` reads
    long long rows, rows1;
    int period;
    unsigned passes, passes1;      // a workgroup walks `passes` of the step's row blocks
    unsigned bshift, bshift1;      // sweep blocks are 256 << bshift rows (sweep_block_shift)
    int sweeping;                  // this launch carries the sweep: the catch-up leaves the rows of the step's blocks to it
};

// request r (request-major: e = r / F, f = r % F) -> arena row, -1 if none
__device__ __forceinline__ long long request_row_linear(const SrcDev& S, unsigned r) {
    const unsigned e = r / S.F, f = r - e * S.F;
    return request_row(S, e, f);
}

// claim a lagging row: exactly one caller gets true (and the step its state is valid for); hot rows are current, so only
// stale rows cost the atomic
__device__ __forceinline__ bool claim_row(int* last_step, long long row, int target, int* s_out) {
    const int s = __hip_atomic_load(&last_step[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s > 0 && s < target && atomicCAS(&last_step[row], s, -s) == s) {
        *s_out = s;
        return true;
    }
    return false;
}

// A workgroup's list of rows to bring up to date, and their replay.  The replay is a chain of ~9 dependent instructions per
// missed step and lane, 1 .. P + 1 steps long: the list is first ordered by lag (counting sort in LDS, longest first), so
// that the 64 tasks of a wave — (row, float) pairs, one float per lane — run loops of the same length; the state of a
// thread's next task is loaded before the current one is replayed.
struct Claims {
    int* row; int* s;              // [kThreads] as collected
    int* row2; int* s2;            // [kThreads] ordered by lag
    int* won;                      // [kThreads] (ordered) the claim was granted: its replay is written back
    unsigned* hist;                // [64]
    unsigned* n;
};
__device__ __forceinline__ Claims claims_carve(unsigned* lds) {
    Claims C;
    C.row = reinterpret_cast<int*>(lds);
    C.s = C.row + kThreads;
    C.row2 = C.s + kThreads;
    C.s2 = C.row2 + kThreads;
    C.won = C.s2 + kThreads;
    C.hist = reinterpret_cast<unsigned*>(C.won + kThreads);
    C.n = C.hist + 64;
    return C;
}
constexpr size_t kClaimsLdsBytes = (5 * kThreads + 64 + 4) * sizeof(unsigned);

// all threads call, after a barrier that made the collected list complete; ends with the list free for re-use.
// The thread that listed entry `my_slot` (`listed`) passes the outcome of its claim as (old, expect): granted iff equal.
// `old` may still be in flight (the compare-and-swap of catchup_requests): its round trip overlaps the sort and the first
// state loads, it is only waited for where the winners are published, right before the first replay.  A claim that was not
// granted is replayed for nothing and dropped.
__device__ __forceinline__ void replay_claims(const Deferred& D, unsigned K, int target, const Claims& C,
                                              const recalgo_deferred::LrWindow& W, bool listed, unsigned my_slot, int old, int expect,
                                              bool all_won = false) {
    const unsigned n = *C.n;
    if (n == 0) return;                                       // (uniform)
    if (threadIdx.x < 64) C.hist[threadIdx.x] = 0;
    __syncthreads();
    unsigned bkt = 0, pos = 0;
    if (threadIdx.x < n) {
        const int lag = target - C.s[threadIdx.x];
        bkt = 63u - (unsigned)(lag > 63 ? 63 : lag);           // longest first
        pos = atomicAdd(&C.hist[bkt], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {                                   // exclusive prefix of the 64 lag buckets (one wave)
        const unsigned v = C.hist[threadIdx.x];
        unsigned inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(inc, o, 64);
            if ((threadIdx.x & 63) >= (unsigned)o) inc += t;
        }
        C.hist[threadIdx.x] = inc - v;
    }
    __syncthreads();
    unsigned jsorted = 0;
    if (threadIdx.x < n) {
        jsorted = C.hist[bkt] + pos;
        C.row2[jsorted] = C.row[threadIdx.x];
        C.s2[jsorted] = C.s[threadIdx.x];
        C.row[threadIdx.x] = (int)jsorted;                    // (entry -> its place in the order, for the claimant)
    }
    __syncthreads();
    // whole waves walk the task list (the replay loop is wave-uniform); task = claim * K + float
    const unsigned ntask = n * K, lane = threadIdx.x & 63;
    unsigned t0 = threadIdx.x & ~63u;
    bool mine = t0 + lane < ntask;
    unsigned k = mine ? (t0 + lane) / K : 0, j = (t0 + lane) - k * K;
    size_t o = (size_t)(mine ? C.row2[k] : 0) * K + j;
    int sc = mine ? C.s2[k] : target;
    float w = 0.f, m = 0.f, v = 0.f;
    if (mine) { w = D.w[o]; m = D.m[o]; v = D.v[o]; }
    if (all_won) {                                            // (the sweep: its rows are its own, nothing was claimed)
        if (threadIdx.x < n) C.won[threadIdx.x] = 1;
    } else if (listed) {
        C.won[C.row[my_slot]] = old == expect ? 1 : 0;        // (waits for the claim's outcome; the state loads are in flight)
    }
    __syncthreads();
    while (t0 < ntask) {
        const unsigned t1 = t0 + kThreads;
        const bool mine1 = t1 + lane < ntask;
        const unsigned k1 = mine1 ? (t1 + lane) / K : 0, j1 = (t1 + lane) - k1 * K;
        const size_t o1 = (size_t)(mine1 ? C.row2[k1] : 0) * K + j1;
        const int sc1 = mine1 ? C.s2[k1] : target;
        float w1 = 0.f, m1 = 0.f, v1 = 0.f;
        if (mine1) { w1 = D.w[o1]; m1 = D.m[o1]; v1 = D.v[o1]; }      // (the next task's state: in flight during the replay)
        recalgo_deferred::replay_wave(w, m, v, sc, target, W, D.lr_ring, D.b1, D.b2, D.eps);
        if (mine && C.won[k]) {
            D.w[o] = w; D.m[o] = m; D.v[o] = v;
            if (j == 0) D.last_step[C.row2[k]] = target;
        }
        t0 = t1; mine = mine1; k = k1; j = j1; o = o1; sc = sc1; w = w1; m = m1; v = v1;
    }
    __syncthreads();
}

// one pass of the sweep: kSweepRows consecutive rows (one per thread of the first wave), the lagging ones listed (no claim:
// while a sweep runs, its rows are its own).  64 rows, not 256: a row block of a small, fully live table is 256 x K tasks —
// sixteen replay rounds on one workgroup, the tail of the whole launch
constexpr unsigned kSweepRows = 64;
// A workgroup's `passes` (<= 32) units of 64 rows, unit p starting at row unit_row(p).  The steps of ALL its rows are read up
// front — every thread's loads in flight together, four units per round over the 256 threads — and a round whose rows are all
// current or untouched (a 100 M-row table a few steps into training: nearly every round) then costs two barriers and no trip
// to memory.  (The first version walked unit after unit with 64 threads, a dependent load per unit: the sweep share of a
// 100 M-row table, 3.1 M rows a step, took 0.2 ms of a 0.27 ms step.)
constexpr unsigned kSweepMaxPasses = 32;
// one unit (a small share: sweep_passes == 1): its 64 rows by the first wave, no grouping to do — measurably leaner than the
// general form below (DCN's `prepare` 19.8 vs 21.3 us, DeepFM's two arenas 22.9 vs 27.0 us, same box)
__device__ __forceinline__ void sweep_one_unit(const Deferred& D, unsigned K, long long row, long long end, int target, const Claims& C,
                                               const recalgo_deferred::LrWindow& W) {
    if (threadIdx.x == 0) *C.n = 0;
    __syncthreads();
    bool listed = false;
    unsigned k = 0;
    if (threadIdx.x < kSweepRows && row < end) {
        const int s = D.last_step[row];
        if (s > 0 && s < target) {
            listed = true;
            k = atomicAdd(C.n, 1u);
            C.row[k] = (int)row;
            C.s[k] = s;
        }
    }
    __syncthreads();
    replay_claims(D, K, target, C, W, listed, k, 0, 0);
}
template <typename UnitRow>
__device__ __forceinline__ void sweep_units(const Deferred& D, unsigned K, unsigned passes, UnitRow unit_row, long long end, int target,
                                            const Claims& C, const recalgo_deferred::LrWindow& W) {
    if (passes == 1) {                                                    // (uniform over the launch)
        sweep_one_unit(D, K, unit_row(0) + threadIdx.x, end, target, C, W);
        return;
    }
    constexpr unsigned kPer = kThreads / kSweepRows;                      // units inspected per round
    constexpr unsigned kRounds = kSweepMaxPasses / kPer;
    const unsigned up = threadIdx.x / kSweepRows, ur = threadIdx.x % kSweepRows;
    int sv[kRounds];
    unsigned rw[kRounds];                                                 // (arena rows are < 2^31; registers are what bounds
#pragma unroll                                                            //  the occupancy of the whole `prepare` launch)
    for (unsigned j = 0; j < kRounds; ++j) {
        const unsigned pu = up + kPer * j;
        const long long r = pu < passes ? unit_row(pu) + ur : end;
        rw[j] = (unsigned)(r < end ? r : end);
        sv[j] = r < end ? D.last_step[r] : 0;
    }
    // how many rows of each round lag (block-wide counts: every thread knows them all) ...
    unsigned cntp[kRounds / 2];                                           // (two 16-bit counts per register)
#pragma unroll
    for (unsigned j = 0; j < kRounds / 2; ++j) cntp[j] = 0;
#pragma unroll
    for (unsigned j = 0; j < kRounds; ++j)
        if (kPer * j < passes) cntp[j / 2] |= (unsigned)__syncthreads_count(sv[j] > 0 && sv[j] < target) << (16 * (j & 1));
    auto cnt = [&](unsigned j) -> unsigned {
        unsigned v = 0;
#pragma unroll
        for (unsigned q = 0; q < kRounds / 2; ++q) v = (j / 2 == q) ? cntp[q] : v;
        return (v >> (16 * (j & 1))) & 0xffffu;
    };
    // ... so that the rounds are replayed in GROUPS of up to kThreads claims: a workgroup whose 2048 rows hold a handful of
    // lagging ones (a large table: the rows the recent batches touched, spread thin) replays them in ONE pass instead of one
    // pass — state loads, the replay loop, stores: ~5 us — per round that has any
    unsigned j0 = 0;
#pragma unroll 1
    while (j0 < kRounds && kPer * j0 < passes) {
        unsigned acc = cnt(j0), j1 = j0 + 1;
        while (j1 < kRounds && acc + cnt(j1) <= (unsigned)kThreads) acc += cnt(j1++);
        if (acc > 0) {                                                    // (uniform)
            if (threadIdx.x == 0) *C.n = 0;
            __syncthreads();
#pragma unroll
            for (unsigned j = 0; j < kRounds; ++j)
                if (j >= j0 && j < j1 && sv[j] > 0 && sv[j] < target) {
                    const unsigned k = atomicAdd(C.n, 1u);
                    C.row[k] = (int)rw[j];
                    C.s[k] = sv[j];
                }
            __syncthreads();
            replay_claims(D, K, target, C, W, false, 0, 0, 0, true);
        }
        j0 = j1;
    }
}
// first row of unit u (64 rows) of the share of step index c: the row blocks c, c + P, c + 2 P, ... of 256 << g rows each
// (4 << g units).  g = 0 for the tables of the reference's data sets (their hot rows sit at the low ids of every table: small
// blocks spread them over the steps and the workgroups); large tables take larger blocks (sweep_block_shift) so that a step's
// share is read in long contiguous runs — 12 k scattered 1 KB pieces of a 400 MB step array per step were 0.13 ms of TLB misses
__device__ __forceinline__ long long sweep_unit_row(int c, int period, unsigned g, long long u) {
    return (((long long)c + (long long)period * (u >> (2 + g))) << (8 + g)) + (long long)(u & ((4u << g) - 1u)) * kSweepRows;
}
inline unsigned sweep_block_shift(long long rows, int period) {
    unsigned g = 0;
    while (g < 8 && ((rows >> (8 + g)) / (period < 1 ? 1 : period)) > 512) ++g;
    return g;
}
inline unsigned sweep_passes(long long rows_in_launch) {
    // (small shares: ONE unit per workgroup — the launch lasts as long as its longest workgroup, and a workgroup with 64 rows
    // of a densely live table already has a full pass of replay work)
    const long long r = rows_in_launch / 65536;
    return (unsigned)(r < 1 ? 1 : (r > (long long)kSweepMaxPasses ? (long long)kSweepMaxPasses : r));
}

// the catch-up of up to kThreads requests: lagging rows claimed (one winner per row), listed, replayed
__device__ __forceinline__ void catchup_requests(const PrepareArgs& A, const SrcDev& S, unsigned n_req, const Deferred& D, unsigned K,
                                                 unsigned r, bool active, long long n_rows, unsigned bshift, int cidx, int target,
                                                 const Claims& C) {
    if (threadIdx.x == 0) *C.n = 0;
    __syncthreads();
    // a lagging row is LISTED at once and claimed by a compare-and-swap whose outcome is not waited for here (one winner per
    // row over all workgroups; hot rows are current: only stale rows cost the atomic)
    int s = 0, old = 0;
    bool lag = false;
    long long row = -1;
    if (active && r < n_req) {
        row = request_row_linear(S, r);
        // (rows of the blocks the same launch sweeps are the sweep's)
        if (row >= 0 && row < n_rows && !(A.sweeping && (int)((row >> (8 + bshift)) % A.period) == cidx)) {
            s = __hip_atomic_load(&D.last_step[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lag = s > 0 && s < target;
        }
    }
    if (lag) old = atomicCAS(&D.last_step[row], s, -s);        // (in flight until replay_claims publishes the winners)
    unsigned slot = 0;
    if (lag) {
        slot = atomicAdd(C.n, 1u);
        C.row[slot] = (int)row;
        C.s[slot] = s;
    }
    __syncthreads();
    replay_claims(D, K, target, C, recalgo_deferred::lr_window(D.lr_ring, target), lag, slot, old, s);
}

__global__ __launch_bounds__(kThreads, 8) void sparse_prepare_kernel(PrepareArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds_u[];
    const int target = A.D.last_step ? (int)(A.step[0] + A.step_off) : 0;
    const int cidx = A.period > 0 ? target % A.period : 0;
    if (blockIdx.x < A.b_catch) {
        // ---- catch-up: kCatchReq requests per workgroup, in request order ------------------------------------------------
        if (A.n_src > 1) {                                    // (uniform over the launch; the source: uniform over the workgroup)
            unsigned si = 0, b0 = 0;
#pragma unroll
            for (int i = 1; i < kPrepSources; ++i)
                if ((unsigned)i < A.n_src && blockIdx.x >= A.catch_end[i - 1]) { si = i; b0 = A.catch_end[i - 1]; }
            SrcDev S = A.S;
            unsigned n_req = A.n_req;
#pragma unroll
            for (int i = 1; i < kPrepSources; ++i)
                if (si == (unsigned)i) { S = A.Sx[i - 1]; n_req = A.nreq_x[i - 1]; }
            catchup_requests(A, S, n_req, A.D, A.K, (blockIdx.x - b0) * kCatchReq + threadIdx.x, threadIdx.x < kCatchReq, A.rows, A.bshift,
                             cidx, target, claims_carve(lds_u));
            return;
        }
        catchup_requests(A, A.S, A.n_req, A.D, A.K, blockIdx.x * kCatchReq + threadIdx.x, threadIdx.x < kCatchReq, A.rows, A.bshift, cidx,
                         target, claims_carve(lds_u));
        return;
    }
    if (blockIdx.x < A.b_comp) {
        // ---- the companion arena's rows of the same requests (one float per row; single-source launches only) -----------
        catchup_requests(A, A.S, A.n_req, A.D1, 1, (blockIdx.x - A.b_catch) * kThreads + threadIdx.x, true, A.rows1, A.bshift1, cidx,
                         target, claims_carve(lds_u));
        return;
    }
    if (blockIdx.x < A.b_count) {
        // ---- count: one entry per DISTINCT row of the tile (its first request, the leader) -----------------------------
        unsigned* hkey = lds_u;                               // [kSlots]     the tile's distinct rows ...
        unsigned* mask = hkey + kSlots;                       // [kSlots][8]  ... and which threads request them
        unsigned cb = blockIdx.x - A.b_comp;
        SrcDev S = A.S;
        if (A.n_src > 1) {
            unsigned si = 0, b0 = 0;
#pragma unroll
            for (int i = 1; i < kPrepSources; ++i)
                if ((unsigned)i < A.n_src && cb >= A.count_end[i - 1]) { si = i; b0 = A.count_end[i - 1]; }
#pragma unroll
            for (int i = 1; i < kPrepSources; ++i)
                if (si == (unsigned)i) S = A.Sx[i - 1];
            cb -= b0;
        }
        const unsigned li = cb * kThreads + threadIdx.x;
        unsigned e, f;
        const long long row = li < S.n ? slot_row(S, li, &e, &f) : -1;
        unsigned slot;
        const EqInfo eq = tile_equal(row >= 0 ? (unsigned)row : 0xffffffffu, hkey, mask, &slot);
        if (row >= 0 && eq.before == 0) atomicAdd(&A.total[(size_t)bucket_of((unsigned)row, A.nb_log2) << A.cs], 1u);
        return;
    }
    // ---- sweep (deferred Adam): rows [c * chunk, (c + 1) * chunk), c = target % period, brought to `target` -----------
    if (target <= 0) return;
    const Claims C = claims_carve(lds_u);
    if (blockIdx.x < A.b_sweep) {
        const recalgo_deferred::LrWindow W = recalgo_deferred::lr_window(A.D.lr_ring, target);
        // this workgroup's units (64 rows each) of the step's share: STRIDED over the launch's sweep workgroups — the live rows of
        // a model sit in a few dense regions of the arena (its small tables); consecutive units to one workgroup made ~40 of the
        // 1500 sweep workgroups of a 100 M-row arena replay nearly all of the step's lagging rows, 15 passes each
        const long long wg = blockIdx.x - A.b_count, nwg = A.b_sweep - A.b_count;
        sweep_units(A.D, A.K, A.passes, [&](unsigned pu) { return sweep_unit_row(cidx, A.period, A.bshift, wg + (long long)pu * nwg); },
                    A.rows, target, C, W);
        return;
    }
    {
        const recalgo_deferred::LrWindow W = recalgo_deferred::lr_window(A.D1.lr_ring, target);
        const long long wg = blockIdx.x - A.b_sweep, nwg = (long long)gridDim.x - A.b_sweep;
        sweep_units(A.D1, 1, A.passes1, [&](unsigned pu) { return sweep_unit_row(cidx, A.period, A.bshift1, wg + (long long)pu * nwg); },
                    A.rows1, target, C, W);
    }
}

// ---------------------------------------------------------------------------------------------
// 2. place: a tile's entries into their buckets, tile duplicates pre-combined
// ---------------------------------------------------------------------------------------------
struct PlaceArgs {
    SrcDev src[kMaxSources];
    int n_src;
    unsigned n_total;
    const unsigned* total; unsigned* cursor;   // [nb << cs], [nb << cs]
    unsigned cs;
    uint4* sched;                              // [nb]: (bucket, first entry, entries, -) in the order `apply` takes the buckets (the heavy ones first)
    // prescanned: offs_g / sched already hold the prefix of the totals (recalgo_scatter_plan_scan's record run by the optimizer
    // launch in front, plan_scan.h) — the tiles then skip their own scan of the counters and its nb words of LDS
    const unsigned* offs_g; int prescanned;
    // workgroup 0 also evaluates this step's lr_t ONCE for `apply` (header word 0) and records it in the rings
    float* hdr;
    float* lr_ring; float* lr_ring1;           // nullptr: none
    const long long* step; int step_off;       // t = step[0] + step_off (nullptr: no optimizer step, GRAD)
    float lr, b1, b2;
    unsigned long long* keys;                  // [n_total]
    float* partials;                           // [n_total][K]: the summed gradient rows of a tile's duplicated rows
    float* partials1;                          // [n_total]: the same for the companion's scalar gradients (nullptr: no companion)
    unsigned nb_log2;
    unsigned KV, L;
    int stage_ok;                              // the tile's duplicated gradient rows fit the LDS staging area ([256][K] floats)
};

constexpr unsigned kTileLong = 24;             // duplicates of a row in a tile above which the whole workgroup sums them

template <int VEC>
__global__ __launch_bounds__(kThreads) void sparse_place_kernel(PlaceArgs A) {
    using V = typename Vec<VEC>::T;
    extern __shared__ __attribute__((aligned(16))) unsigned lds_u[];
    const unsigned nb = 1u << A.nb_log2, bpt = nb / kThreads;   // nb is a multiple of kThreads
    unsigned* rows = lds_u;                                   // [kThreads] the tile's rows
    unsigned* samec = rows + kThreads;                        // [kThreads] requests of the thread's row in the tile
    unsigned* mbase = samec + kThreads;                       // [kThreads] (leaders of duplicated rows) first entry in mlist
    unsigned* mlist = mbase + kThreads;                       // [kThreads] member threads of the duplicated rows, in order
    unsigned* req_e = mlist + kThreads;                       // [kThreads] the thread's request: example ...
    unsigned* req_f = req_e + kThreads;                       // [kThreads] ... field ...
    unsigned* req_s = req_f + kThreads;                       // [kThreads] ... source
    unsigned* jobs = req_s + kThreads;                        // [kThreads] leaders of the duplicated rows
    float* red = reinterpret_cast<float*>(jobs + kThreads);   // [kThreads * 4]
    unsigned* offs = reinterpret_cast<unsigned*>(red + kThreads * 4);   // [nb] (not when the plan is prescanned: offs_g)
    unsigned* sh = offs + (A.prescanned ? 0u : nb);           // [8]; sh[6] = number of jobs, sh[7] = number of long jobs
    unsigned* ljobs = sh + 8;                                 // [16] leaders of the rows with > kTileLong duplicates (<= 10)
    unsigned* hkey = ljobs + 16;                              // [kSlots]     equal-key bookkeeping (tile_equal)
    unsigned* mask = hkey + kSlots;                           // [kSlots][8]
    unsigned* bbase = mask + kSlots * 8;                      // [kSlots] where the tile's entries of a bucket start in the bucket
    // the source descriptors go to LDS: a data-dependent index into the kernel-argument array would go through scratch
    SrcDev* lsrc = reinterpret_cast<SrcDev*>(bbase + kSlots); // [kMaxSources]
    copy_kernarg_words(reinterpret_cast<unsigned*>(lsrc), offsetof(PlaceArgs, src), sizeof(SrcDev) * kMaxSources);
    if (threadIdx.x == 0) { sh[6] = 0; sh[7] = 0; }
    __syncthreads();
    // this thread's slot: its row first (ids / row bases from memory), the scan of the bucket totals runs in its shadow
    const unsigned i = blockIdx.x * kThreads + threadIdx.x;
    long long row = -1;
    unsigned e = 0, f = 0, si = 0;
    if (i < A.n_total) {
        si = source_of(lsrc, i);
        const SrcDev& S = lsrc[si];
        if (i - S.first < S.n) row = slot_row(S, i - S.first, &e, &f);      // (padding between two sources: no request)
    }
    if (A.prescanned) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && A.step != nullptr) {
            const long long t = A.step[0] + A.step_off;
            const float lr_t = lr_t_of(A.lr, A.b1, A.b2, t);
            A.hdr[0] = lr_t;
            if (A.lr_ring) A.lr_ring[(unsigned)t & (kLrRing - 1)] = lr_t;
            if (A.lr_ring1) A.lr_ring1[(unsigned)t & (kLrRing - 1)] = lr_t;
        }
    } else {
        // (ONE pass over the counters — they are spread over cache lines, a load fetches a line per counter: the totals wait
        // in LDS for their prefix)
        unsigned sum = 0;
        for (unsigned k = 0; k < bpt; ++k) {
            const unsigned b = threadIdx.x * bpt + k, tb = A.total[(size_t)b << A.cs];
            offs[b] = tb;
            sum += tb;
        }
        unsigned total;
        unsigned run = block_excl_scan(sum, sh, total);
        for (unsigned k = 0; k < bpt; ++k) {
            const unsigned b = threadIdx.x * bpt + k, tb = offs[b];
            offs[b] = run;
            run += tb;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0 && A.step != nullptr) {
            const long long t = A.step[0] + A.step_off;
            const float lr_t = lr_t_of(A.lr, A.b1, A.b2, t);
            A.hdr[0] = lr_t;
            if (A.lr_ring) A.lr_ring[(unsigned)t & (kLrRing - 1)] = lr_t;
            if (A.lr_ring1) A.lr_ring1[(unsigned)t & (kLrRing - 1)] = lr_t;
        }
        if (blockIdx.x == 0) {
            // `apply` runs one workgroup per bucket, more of them than fit the chip at once for large plans: the buckets that
            // hold a hot row (many entries: a long tail of the launch when they start late) are dispatched first
            const unsigned heavy_min = 2u * (total >> A.nb_log2) + 64u;
            unsigned nh = 0;
            for (unsigned k = 0; k < bpt; ++k) nh += A.total[(size_t)(threadIdx.x * bpt + k) << A.cs] >= heavy_min;
            unsigned n_heavy;
            unsigned hrun = block_excl_scan(nh, sh, n_heavy);
            for (unsigned k = 0; k < bpt; ++k) {
                const unsigned b = threadIdx.x * bpt + k;
                const unsigned tb = A.total[(size_t)b << A.cs];
                const uint4 rec = make_uint4(b, offs[b], tb, 0u);
                if (tb >= heavy_min) A.sched[hrun++] = rec;
                else A.sched[n_heavy + b - hrun] = rec;     // (b - hrun = the light buckets before b)
            }
        }
    }
    rows[threadIdx.x] = row >= 0 ? (unsigned)row : 0xffffffffu;
    req_e[threadIdx.x] = e;
    req_f[threadIdx.x] = f;
    req_s[threadIdx.x] = si;
    unsigned slot;
    const EqInfo d = tile_equal(row >= 0 ? (unsigned)row : 0xffffffffu, hkey, mask, &slot);
    const bool leader = row >= 0 && d.before == 0, dupl = leader && d.same > 1;
    const unsigned b = leader ? bucket_of((unsigned)row, A.nb_log2) : 0xffffffffu;
    samec[threadIdx.x] = d.same;
    {
        unsigned total;
        mbase[threadIdx.x] = block_excl_scan(dupl ? d.same : 0u, sh, total);
    }
    if (dupl) jobs[atomicAdd(&sh[6], 1u)] = threadIdx.x;
    __syncthreads();
    if (row >= 0 && d.same > 1) mlist[mbase[d.leader] + d.before] = threadIdx.x;
    // the gradient rows of the tile's duplicated requests are needed further down (summed per row in LDS): their loads are
    // issued HERE, so that they are in flight during the bucket phase (a returning atomic + two barriers)
    const unsigned L = A.L, q = threadIdx.x & (L - 1), grp = threadIdx.x / L, ngrp = kThreads / L;
    V gq[4];
    float g1s = 0.f;
    if (A.stage_ok) {
#pragma unroll
        for (unsigned u = 0; u < 4; ++u) {
            const unsigned tm = grp + u * ngrp;
            gq[u] = vz<VEC>();
            if (u < L && q < A.KV && samec[tm] > 1 && rows[tm] != 0xffffffffu)
                gq[u] = load_req_g<VEC>(lsrc[req_s[tm]], req_e[tm], req_f[tm], q, A.KV);
        }
        if (A.partials1 && d.same > 1 && row >= 0) g1s = load_req_g1(lsrc[si], e, f);
    }
    // the tile's entries of one bucket take consecutive positions: the first of them draws the range from the bucket's cursor
    unsigned slot_b;
    const EqInfo db = tile_equal(b, hkey, mask, &slot_b);
    if (leader && db.before == 0) bbase[slot_b] = atomicAdd(&A.cursor[(size_t)b << A.cs], db.same);
    const unsigned first_b = !leader ? 0u : (A.prescanned ? A.offs_g[b] : offs[b]);
    __syncthreads();
    if (leader) {
        // a duplicated row's entry refers to the tile's partial sum (written below), a single request to its own row.  The key
        // (row, slot of the leader) is unique: `apply` orders a bucket by it, so the order the tiles arrive in does not matter
        A.keys[first_b + bbase[slot_b] + db.before] = ((unsigned long long)row << 32) | ((unsigned long long)i << 1) | (dupl ? 1u : 0u);
    }
    // ---- the duplicated rows of the tile: gradient rows added in request order ---------------------------------------
    const unsigned njobs = sh[6];
    if (njobs == 0) return;                                   // (uniform)
    if (A.stage_ok) {
        // ONE round trip to memory: every duplicated request's gradient row goes to LDS (a group of L lanes per row, all
        // loads of a thread in flight together), then each duplicated row is summed from LDS, in request order
        V* stage = reinterpret_cast<V*>(lsrc + kMaxSources);  // [kThreads][KV]
        float* stage1 = reinterpret_cast<float*>(stage + (size_t)kThreads * A.KV);      // [kThreads] (companion only)
        if (A.partials1) stage1[threadIdx.x] = g1s;
        for (unsigned r0 = 0; r0 < L; r0 += 4) {              // a group owns the member threads grp, grp + ngrp, ...
            if (r0 > 0) {                                     // (the first four were loaded above)
#pragma unroll
                for (unsigned u = 0; u < 4; ++u) {
                    const unsigned tm = grp + (r0 + u) * ngrp;
                    gq[u] = vz<VEC>();
                    if (r0 + u < L && q < A.KV && samec[tm] > 1 && rows[tm] != 0xffffffffu)
                        gq[u] = load_req_g<VEC>(lsrc[req_s[tm]], req_e[tm], req_f[tm], q, A.KV);
                }
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
    auto gload = [&](unsigned tm) { return load_req_g<VEC>(lsrc[req_s[tm]], req_e[tm], req_f[tm], q, A.KV); };
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
                const V g0 = gload(mlist[bs + m]), g1 = gload(mlist[bs + m + 1]);
                const V g2 = gload(mlist[bs + m + 2]), g3 = gload(mlist[bs + m + 3]);
                vadd(acc, g0); vadd(acc, g1); vadd(acc, g2); vadd(acc, g3);
            }
            for (; m < c; ++m) vadd(acc, gload(mlist[bs + m]));
            reinterpret_cast<V*>(A.partials)[((size_t)blockIdx.x * kThreads + ld) * A.KV + q] = acc;
        }
    }
    __syncthreads();
    const unsigned nlong = sh[7];
    for (unsigned k = 0; k < nlong; ++k) {                    // hot rows: all groups sum strided slices, fixed-order combination
        const unsigned ld = ljobs[k], c = samec[ld], bs = mbase[ld];
        V acc = vz<VEC>();
        if (q < A.KV) {
            for (unsigned m = grp; m < c; m += ngrp) vadd(acc, gload(mlist[bs + m]));
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
// 3. apply
// ---------------------------------------------------------------------------------------------
struct ApplyArgs {
    SrcDev src[kMaxSources];
    int n_src;
    const uint4* sched;            // [nb]: workgroup i takes bucket sched[i].x = entries [sched[i].y, + sched[i].z) of `keys`
    const float* hdr;              // hdr[0] = lr_t of this step (written by `place`)
    unsigned* total; unsigned* cursor;                        // cleared per bucket for the next step
    unsigned cs;
    const unsigned long long* keys; unsigned long long* keys_alt;   // keys_alt: scratch of the same size (large buckets)
    const float* partials;         // [slots][K]: gradient rows of the entries whose key has the partial bit set
    const float* partials1;        // [slots]: the companion's
    int mode;                      // RECALGO_SCATTER_GRAD / _ADAM / _LAZY_ADAM
    float* w; float* m; float* v; float* grad;                // grad: GRAD target; ADAM modes: rows zeroed when non-null
    int* last_step;                // ADAM (deferred-exact) only
    float* lr_ring;
    // the companion arena (one float per row; w1 / grad1 == nullptr: none)
    float* w1; float* m1; float* v1; float* grad1;
    int* last_step1;
    float* lr_ring1;
    long long rows1;
    int has1;
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


// where the gradient of the entry `key` is: (source, e, f) of the request, or the tile's partial sum
struct EntryRef { const SrcDev* S; unsigned e, f, slot; bool partial; };
__device__ __forceinline__ EntryRef entry_ref(const SrcDev* lsrc, unsigned long long key) {
    EntryRef r;
    const unsigned ref = (unsigned)key;
    r.slot = ref >> 1;
    r.partial = ref & 1u;
    r.S = lsrc;
    r.e = r.f = 0;
    if (!r.partial) {
        r.S = lsrc + source_of(lsrc, r.slot);
        slot_ef(*r.S, r.slot - r.S->first, &r.e, &r.f);
    }
    return r;
}
// gradient piece q of the request a key refers to
template <int VEC>
__device__ __forceinline__ typename Vec<VEC>::T load_g(const ApplyArgs& A, const SrcDev* lsrc, unsigned long long key, unsigned q) {
    using V = typename Vec<VEC>::T;
    const EntryRef r = entry_ref(lsrc, key);
    if (r.partial) return reinterpret_cast<const V*>(A.partials)[(size_t)r.slot * A.KV + q];
    return load_req_g<VEC>(*r.S, r.e, r.f, q, A.KV);
}
__device__ __forceinline__ float load_g1(const ApplyArgs& A, const SrcDev* lsrc, unsigned long long key) {
    const EntryRef r = entry_ref(lsrc, key);
    if (r.partial) return A.partials1[r.slot];
    return load_req_g1(*r.S, r.e, r.f);
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
        if (A.mode == RECALGO_SCATTER_ADAM && st.s > 0 && st.s < t - 1)       // (normally done by the lookup's catch-up; kept
            replay<VEC>(st.w, st.m, st.v, st.s, t - 1, A.lr_ring, A.b1, A.b2, A.eps);      //  for lookups registered without one)
        vadam(st.w, acc, st.m, st.v, lr_t, A.b1, A.b2, A.eps);
        reinterpret_cast<V*>(A.w)[o] = st.w;
        reinterpret_cast<V*>(A.m)[o] = st.m;
        reinterpret_cast<V*>(A.v)[o] = st.v;
        if (A.grad) reinterpret_cast<V*>(A.grad)[o] = vz<VEC>();
    }
    if (q == 0 && A.mode == RECALGO_SCATTER_ADAM) A.last_step[row] = t;
}

// the companion arena's row (one float), by ONE lane: the same three endings
__device__ __forceinline__ void finish_row1(const ApplyArgs& A, unsigned row, float acc1, int t, float lr_t) {
    if ((long long)row >= A.rows1) return;
    if (A.mode == RECALGO_SCATTER_GRAD) {
        A.grad1[row] += acc1;
        return;
    }
    float w = A.w1[row], m = A.m1[row], v = A.v1[row];
    if (A.mode == RECALGO_SCATTER_ADAM) {
        const int s = A.last_step1[row];
        if (s > 0 && s < t - 1) recalgo_deferred::replay1(w, m, v, s, t - 1, A.lr_ring1, A.b1, A.b2, A.eps);
    }
    adam1(w, acc1, m, v, lr_t, A.b1, A.b2, A.eps);
    A.w1[row] = w; A.m1[row] = m; A.v1[row] = v;
    if (A.grad1) A.grad1[row] = 0.f;
    if (A.mode == RECALGO_SCATTER_ADAM) A.last_step1[row] = t;
}

// one row (segment [lo, hi) of the ordered keys) by one group of L lanes: gradient rows added in key order
template <int VEC>
__device__ __forceinline__ void short_row(const ApplyArgs& A, const SrcDev* lsrc, const unsigned long long* keys, unsigned lo,
                                          unsigned hi, unsigned q, int t, float lr_t) {
    using V = typename Vec<VEC>::T;
    const unsigned row = key_row(keys[lo]);
    const RowState<VEC> st = load_state<VEC>(A, row, q);
    // up to eight gradient rows in flight per round trip (a hot row of a field arrives as B / 256 = 16 tile partials: a
    // load -> wait -> add loop over them was most of this launch), added in key order
    constexpr unsigned kU = RECALGO_APPLY_KU;
    V acc = vz<VEC>();
    if (q < A.KV) {
        for (unsigned j = lo; j < hi; j += kU) {
            V gq[kU];
#pragma unroll
            for (unsigned u = 0; u < kU; ++u)
                if (j + u < hi) gq[u] = load_g<VEC>(A, lsrc, keys[j + u], q);
#pragma unroll
            for (unsigned u = 0; u < kU; ++u)
                if (j + u < hi) vadd(acc, gq[u]);
        }
    }
    float acc1 = 0.f;
    if (A.has1 && q == 0) {                                   // the companion's scalars of the same entries, same order
        for (unsigned j = lo; j < hi; j += kU) {
            float a[kU];
#pragma unroll
            for (unsigned u = 0; u < kU; ++u)
                if (j + u < hi) a[u] = load_g1(A, lsrc, keys[j + u]);
#pragma unroll
            for (unsigned u = 0; u < kU; ++u)
                if (j + u < hi) acc1 += a[u];
        }
    }
    finish_row<VEC>(A, row, st, acc, q, t, lr_t);
    if (A.has1 && q == 0) finish_row1(A, row, acc1, t, lr_t);
}

// rows with many entries: all groups of the workgroup sum strided slices, fixed-order combination through LDS
template <int VEC>
__device__ __forceinline__ void long_rows(const ApplyArgs& A, const SrcDev* lsrc, const unsigned long long* keys,
                                          const unsigned* long_list, unsigned nl, float* red, float* red1, int t, float lr_t) {
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
        if (A.has1 && q == 0) {
            float a1 = 0.f;
            for (unsigned j = lo + grp; j < hi; j += ngrp) a1 += load_g1(A, lsrc, keys[j]);
            red1[grp] = a1;
        }
        __syncthreads();
        if (grp == 0) {
            V tot = vz<VEC>();
            if (q < A.KV)
                for (unsigned g2 = 0; g2 < ngrp; ++g2) vadd(tot, reinterpret_cast<const V*>(red)[g2 * A.KV + q]);
            finish_row<VEC>(A, row, st, tot, q, t, lr_t);
            if (A.has1 && q == 0) {
                float t1 = 0.f;
                for (unsigned g2 = 0; g2 < ngrp; ++g2) t1 += red1[g2];
                finish_row1(A, row, t1, t, lr_t);
            }
        }
        __syncthreads();
    }
}

// a row of more than kLongSeg entries joins the long list (true), unless the list is full
__device__ __forceinline__ bool defer_long(unsigned lo, unsigned hi, unsigned q, unsigned L, unsigned* long_list, unsigned* n_long) {
    if (hi - lo <= kLongSeg) return false;
    unsigned slot = kMaxLong;
    if (q == 0) slot = atomicAdd(n_long, 1u);
    slot = __shfl(slot, (int)((threadIdx.x & 63) & ~(L - 1)), 64);
    if (slot >= kMaxLong) return false;
    if (q == 0) { long_list[2 * slot] = lo; long_list[2 * slot + 1] = hi; }
    return true;
}

// segments listed in LDS (seg_lo / seg_n): groups take them round robin; long rows are deferred to long_rows()
template <int VEC>
__device__ __forceinline__ void process_segments(const ApplyArgs& A, const SrcDev* lsrc, const unsigned long long* keys,
                                                 const unsigned* seg_lo, const unsigned* seg_n, unsigned nseg,
                                                 unsigned* long_list, unsigned* n_long, float* red, float* red1, int t, float lr_t) {
    const unsigned L = A.L, q = threadIdx.x & (L - 1), grp = threadIdx.x / L, ngrp = kThreads / L;
    for (unsigned h = grp; h < nseg; h += ngrp) {
        const unsigned lo = seg_lo[h], len = seg_n[h];
        if (defer_long(lo, lo + len, q, L, long_list, n_long)) continue;
        short_row<VEC>(A, lsrc, keys, lo, lo + len, q, t, lr_t);
    }
    __syncthreads();
    long_rows<VEC>(A, lsrc, keys, long_list, min(*n_long, kMaxLong), red, red1, t, lr_t);
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


// SORTED keys whose rows are not listed: every group scans a contiguous block of entries for row heads and owns the rows
// that START in its block
template <int VEC>
__device__ __forceinline__ void process_sorted_scan(const ApplyArgs& A, const SrcDev* lsrc, const unsigned long long* keys, unsigned n,
                                                    unsigned* long_list, unsigned* n_long, float* red, float* red1, int t, float lr_t) {
    const unsigned L = A.L, q = threadIdx.x & (L - 1), grp = threadIdx.x / L, ngrp = kThreads / L;
    const unsigned per = (n + ngrp - 1) / ngrp;
    unsigned i = grp * per;
    const unsigned stop = min(n, i + per);
    while (i < stop) {
        const unsigned row = key_row(keys[i]);
        if (i > 0 && key_row(keys[i - 1]) == row) { ++i; continue; }          // not a head (only at the block start)
        const unsigned end = seg_end(keys, i, n, row);
        if (!defer_long(i, end, q, L, long_list, n_long)) short_row<VEC>(A, lsrc, keys, i, end, q, t, lr_t);
        i = end;
    }
    __syncthreads();
    long_rows<VEC>(A, lsrc, keys, long_list, min(*n_long, kMaxLong), red, red1, t, lr_t);
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
__global__ __launch_bounds__(kThreads, RECALGO_APPLY_WGS) void sparse_apply_kernel(ApplyArgs A) {   // 256 CUs x this many resident
    __shared__ unsigned long long lds_keys[kLdsKeys];         // the bucket's keys in order (buckets up to kLdsKeys entries)
    __shared__ unsigned seg_lo[kMaxSeg], seg_n[kMaxSeg];
    __shared__ unsigned long_list[2 * kMaxLong];
    __shared__ unsigned n_long, n_seg;
    __shared__ __attribute__((aligned(16))) float red[kThreads * 4];
    __shared__ float red1[kThreads];
    __shared__ SrcDev lsrc[kMaxSources];
    // (a small bucket's keys as they arrived: only needed until they are ranked, before `red` is)
    unsigned long long* ck = reinterpret_cast<unsigned long long*>(red);
    const uint4 sc = A.sched[blockIdx.x];                     // (one 16-byte record: bucket, first entry, entries)
    const unsigned b = sc.x, beg = sc.y, n = sc.z;
    const int t = (int)(A.step[0] + A.step_off);
    const float lr_t = A.mode != RECALGO_SCATTER_GRAD ? A.hdr[0] : 0.f;
    if (threadIdx.x == 0) {
        n_long = 0; n_seg = 0;
        A.total[(size_t)b << A.cs] = 0;                       // the plan is consumed: clean for the next step's counts
        A.cursor[(size_t)b << A.cs] = 0;
    }
    if (n == 0) return;                                       // (uniform)
    copy_kernarg_words(reinterpret_cast<unsigned*>(lsrc), offsetof(ApplyArgs, src), sizeof(SrcDev) * kMaxSources);
    const unsigned long long* in = A.keys + beg;              // the bucket's keys, in the order the tiles placed them
    __syncthreads();
    if (n <= kThreads) {
        // ---- small bucket: rank by comparison (one entry per thread); keys are unique ---------------------------------
        const unsigned long long key = threadIdx.x < n ? in[threadIdx.x] : kPadKey;
        const unsigned row = key_row(key);                    // (padding: 0xffffffff, greater than every row)
        ck[threadIdx.x] = key;
        __syncthreads();
        if (threadIdx.x < n) {
            unsigned less = 0, same_before = 0, same = 0;
            const ulonglong2* k2 = reinterpret_cast<const ulonglong2*>(ck);
#pragma unroll 8
            for (unsigned j2 = 0; j2 < (n + 1) / 2; ++j2) {
                const ulonglong2 kk = k2[j2];
                const unsigned r0 = key_row(kk.x), r1 = key_row(kk.y);
                less += (r0 < row) + (r1 < row);
                same += (r0 == row) + (r1 == row);
                same_before += (r0 == row && kk.x < key) + (r1 == row && kk.y < key);
            }
            lds_keys[less + same_before] = key;
            if (same_before == 0) {                           // the row's first entry lists the row
                const unsigned k = atomicAdd(&n_seg, 1u);
                seg_lo[k] = less;
                seg_n[k] = same;
            }
        }
        __syncthreads();
        process_segments<VEC>(A, lsrc, lds_keys, seg_lo, seg_n, n_seg, long_list, &n_long, red, red1, t, lr_t);
        return;
    }
    // ---- large bucket (a hot row's entries of many tiles, or a plan at the top of the bucket range): full sort ----------
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
    process_sorted_scan<VEC>(A, lsrc, sorted, n, long_list, &n_long, red, red1, t, lr_t);
}

// ---------------------------------------------------------------------------------------------
// standalone sweep: rows [row0, row1) brought to step[0] + step_off (flush before EVAL / PREDICT / checkpoint)
// ---------------------------------------------------------------------------------------------
struct SweepArgs {
    Deferred D;
    const long long* step;
    int step_off;
    unsigned K, passes;
    long long row0, row1;
};
__global__ __launch_bounds__(kThreads) void sparse_sweep_kernel(SweepArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds_u[];
    const int target = (int)(A.step[0] + A.step_off);
    if (target <= 0) return;
    const Claims C = claims_carve(lds_u);
    const recalgo_deferred::LrWindow W = recalgo_deferred::lr_window(A.D.lr_ring, target);
    sweep_units(A.D, A.K, A.passes, [&](unsigned pu) { return A.row0 + ((long long)blockIdx.x + (long long)pu * gridDim.x) * kSweepRows; },
                A.row1, target, C, W);
}

// ---- host helpers -----------------------------------------------------------------------------
struct Geometry { int vec; unsigned KV, L, G; };
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool geometry(int K, const recalgo_scatter_source_t* src, int n_src, Geometry* G) {
    if (K < 1 || K > 256) return false;
    bool v4 = (K & 3) == 0 && K / 4 <= 64;
    for (int i = 0; v4 && src && i < n_src; ++i) {
        const recalgo_scatter_source_t& s = src[i];
        if (s.g && (!aligned16(s.g) || (s.g_stride & 3) || (s.g_col & 3) || (s.g_fmul & 3))) v4 = false;
        if (s.fm_scale && (!aligned16(s.fm_sum) || !aligned16(s.fm_emb))) v4 = false;
    }
    G->vec = v4 ? 4 : 1;
    G->KV = v4 ? (unsigned)K / 4 : (unsigned)K;
    if (G->KV > 64) return false;
    unsigned L = 1;
    while (L < G->KV) L <<= 1;
    G->L = L;
    unsigned g = 1;
    while (g < (unsigned)K && g < 64) g <<= 1;
    G->G = g;                                               // (rows wider than a wave take several passes of the group)
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
        out[i] = SrcDev{nullptr, nullptr, nullptr, 0, 0, 1, 0xffffffffu, 0, kThreads, 0, nullptr, 0, 0, 0, nullptr, 0, 0, 0,
                        nullptr, nullptr, nullptr};
    for (int i = 0; i < n_src; ++i) {
        const recalgo_scatter_source_t& s = src[i];
        if (!s.ids || s.n_ex < 0 || s.F < 1 || (need_g && !s.g)) return false;
        if (s.fm_scale && (!s.fm_sum || !s.fm_emb)) return false;
        const int64_t n = recalgo_scatter_source_slots(s.n_ex, s.F, s.offsets != nullptr);
        if ((int64_t)first + n >= (1ll << 31)) return false;
        const unsigned e256 = ((unsigned)s.n_ex + kThreads - 1) / kThreads * kThreads;
        out[i] = SrcDev{s.ids, s.offsets, s.row_base, (long long)s.base, (unsigned)s.n_ex, (unsigned)s.F, first, (unsigned)n,
                        e256 ? e256 : kThreads, 0, s.g, (long long)s.g_stride, (unsigned)s.g_col, (unsigned)s.g_fmul,
                        comp ? comp[i].g : nullptr, comp ? (long long)comp[i].g_stride : 0, comp ? (unsigned)comp[i].g_col : 0u,
                        comp ? (unsigned)comp[i].g_fmul : 0u, s.fm_scale, s.fm_sum, s.fm_emb};
        // every source is a whole number of tiles
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
// The bucket counters take integer atomics from every tile of the plan (one per tile-distinct row in `prepare`, one
// returning add per (tile, bucket) in `place`).  Atomics on one cache line are served one after the other, so the
// counters are spread: one every 16 words = one per 64-byte line (1024 counters packed into 32 lines cost + 10 us in
// `prepare` and in `place`; measured in round 4).
inline unsigned counter_shift() { return 4u; }

struct Ws { float* hdr; unsigned* total; unsigned* cursor; uint4* sched; unsigned* offs; unsigned long long* keys;
            unsigned long long* keys_alt; float* partials; float* partials1; };
inline Ws carve(void* ws, int64_t cap, int nb_log2) {
    const int64_t nb = 1ll << nb_log2;
    char* p = static_cast<char*>(ws);
    Ws w;
    // header (zero-filled by the caller before the first use and whenever counted-but-unapplied sources are dropped):
    const int64_t nc = nb << counter_shift();
    w.hdr = reinterpret_cast<float*>(p);                    // 32 words (a whole cache line): [0] = lr_t of the step
    w.total = reinterpret_cast<unsigned*>(p) + 32;          // [nb << cs]
    w.cursor = w.total + nc;                                // [nb << cs]
    w.sched = reinterpret_cast<uint4*>(w.cursor + nc);      // [nb]  (16-byte aligned: 32 + 2 nc words in front)
    w.offs = reinterpret_cast<unsigned*>(w.sched + nb);     // [nb] first entry of every bucket (written by the plan scan)
    w.keys = reinterpret_cast<unsigned long long*>(w.offs + nb);
    w.keys_alt = w.keys + cap;
    w.partials1 = reinterpret_cast<float*>(w.keys_alt + cap);  // [cap]
    w.partials = w.partials1 + cap;                             // [cap][K]
    return w;
}

}  // namespace

RECALGO_EXPORT int recalgo_scatter_plan_buckets_log2(int64_t n_requests) {
    // ~64-128 entries per bucket (one workgroup each in `apply`, up to 256 ranked by comparison), 1024 .. 8192 buckets
    int l = 10;
    while (l < 13 && (n_requests >> l) > 128) ++l;          // (DIN's 311 k requests: 4096 buckets of ~76 — `apply` 74 -> 65 us)
    return l;
}

RECALGO_EXPORT int64_t recalgo_scatter_plan_header_bytes(int nb_log2) {
    if (!nb_ok(nb_log2)) return 0;
    return (32 + 2 * ((1ll << nb_log2) << counter_shift())) * (int64_t)sizeof(unsigned);
}

RECALGO_EXPORT int recalgo_scatter_plan_scan(void* plan_workspace, int64_t plan_requests, int nb_log2, recalgo_plan_scan_t* out) {
    RECALGO_REQUIRE(plan_workspace != nullptr && out != nullptr && nb_ok(nb_log2) && plan_requests >= 0 &&
                    plan_requests % kThreads == 0);
    const Ws ws = carve(plan_workspace, plan_requests, nb_log2);
    out->total = ws.total; out->offs = ws.offs; out->sched = ws.sched;
    out->counter_shift = counter_shift(); out->nb_log2 = (unsigned)nb_log2;
    return 0;
}

RECALGO_EXPORT int64_t recalgo_scatter_plan_workspace_bytes(int64_t n_slots, int nb_log2, int K) {
    if (n_slots < 0 || n_slots % kThreads != 0 || !nb_ok(nb_log2) || K < 1) return 0;
    const int64_t nb = 1ll << nb_log2;
    const int64_t cap = n_slots > 0 ? n_slots : kThreads;
    return (5 * nb + 2 * (nb << counter_shift()) + 32) * (int64_t)sizeof(unsigned) + 2 * cap * (int64_t)sizeof(unsigned long long) +
           cap * (int64_t)(K + 1) * (int64_t)sizeof(float) + 64;
}

static int scatter_prepare_impl(const recalgo_scatter_source_t* source, int n_sources, const int64_t* first_requests, int K,
                                void* plan_workspace, int64_t plan_requests, int nb_log2, int flags,
                                const recalgo_deferred_adam_t* deferred, const recalgo_deferred_adam_t* companion_deferred,
                                int64_t rows, int64_t companion_rows, int sweep_period, const int64_t* step_dev, int step_offset,
                                recalgo_stream_t stream);

RECALGO_EXPORT int recalgo_scatter_prepare(const recalgo_scatter_source_t* source, int K, void* plan_workspace,
                                           int64_t plan_requests, int nb_log2, int64_t first_request, int flags,
                                           const recalgo_deferred_adam_t* deferred,
                                           const recalgo_deferred_adam_t* companion_deferred, int64_t rows, int64_t companion_rows,
                                           int sweep_period, const int64_t* step_dev, int step_offset, recalgo_stream_t stream) {
    return scatter_prepare_impl(source, (source != nullptr && source->n_ex > 0) ? 1 : 0, &first_request, K, plan_workspace, plan_requests,
                                nb_log2, flags, deferred, companion_deferred, rows, companion_rows, sweep_period, step_dev, step_offset,
                                stream);
}

RECALGO_EXPORT int recalgo_scatter_prepare_multi(const recalgo_scatter_source_t* sources, int n_sources, const int64_t* first_requests,
                                                 int K, void* plan_workspace, int64_t plan_requests, int nb_log2, int flags,
                                                 const recalgo_deferred_adam_t* deferred, int64_t rows, int sweep_period,
                                                 const int64_t* step_dev, int step_offset, recalgo_stream_t stream) {
    RECALGO_REQUIRE(sources != nullptr && first_requests != nullptr && n_sources >= 1 && n_sources <= kPrepSources);
    for (int i = 0; i < n_sources; ++i) RECALGO_REQUIRE(sources[i].n_ex > 0);
    return scatter_prepare_impl(sources, n_sources, first_requests, K, plan_workspace, plan_requests, nb_log2, flags, deferred, nullptr,
                                rows, 0, sweep_period, step_dev, step_offset, stream);
}

static int scatter_prepare_impl(const recalgo_scatter_source_t* source, int n_sources, const int64_t* first_requests, int K,
                                void* plan_workspace, int64_t plan_requests, int nb_log2, int flags,
                                const recalgo_deferred_adam_t* deferred, const recalgo_deferred_adam_t* companion_deferred,
                                int64_t rows, int64_t companion_rows, int sweep_period, const int64_t* step_dev, int step_offset,
                                recalgo_stream_t stream) {
    RECALGO_REQUIRE(nb_ok(nb_log2) && plan_workspace != nullptr);
    RECALGO_REQUIRE(plan_requests >= 0 && plan_requests % kThreads == 0);
    RECALGO_REQUIRE(rows >= 0 && rows < (1ll << 31) && companion_rows >= 0 && companion_rows < (1ll << 31));
    const bool count = (flags & RECALGO_PREPARE_COUNT) != 0, sweep = (flags & RECALGO_PREPARE_SWEEP) != 0;
    const bool catchup = (flags & RECALGO_PREPARE_CATCHUP) != 0;
    const int64_t first_request = first_requests[0];
    RECALGO_REQUIRE(first_request >= 0 && first_request % kThreads == 0 && first_request < (1ll << 31));
    SrcDev S[kMaxSources];
    unsigned n = 0;
    if (n_sources >= 1) {
        RECALGO_REQUIRE(to_dev(source, 1, S, &n, false));
        RECALGO_REQUIRE(first_request + (int64_t)n <= plan_requests);
    } else {
        recalgo_scatter_source_t none{};
        to_dev(&none, 0, S, &n, false);
    }
    Geometry G;
    RECALGO_REQUIRE(geometry(K, nullptr, 0, &G));
    PrepareArgs A;
    A.S = S[0];
    A.S.first = (unsigned)first_request;
    A.n_src = n_sources > 1 ? (unsigned)n_sources : 1u;
    unsigned n_x[kPrepSources - 1] = {0, 0, 0};
    for (int i = 1; i < kPrepSources; ++i) {
        A.Sx[i - 1] = S[1];                                    // (an unused slot: first = 0xffffffff, n = 0)
        A.nreq_x[i - 1] = 0;
        if (i < n_sources) {
            SrcDev T[kMaxSources];
            unsigned ni = 0;
            RECALGO_REQUIRE(first_requests[i] >= 0 && first_requests[i] % kThreads == 0 && first_requests[i] < (1ll << 31));
            RECALGO_REQUIRE(to_dev(source + i, 1, T, &ni, false) && first_requests[i] + (int64_t)ni <= plan_requests);
            RECALGO_REQUIRE(companion_deferred == nullptr);   // (a companion arena rides on single-lookup launches only)
            A.Sx[i - 1] = T[0];
            A.Sx[i - 1].first = (unsigned)first_requests[i];
            A.nreq_x[i - 1] = (unsigned)((int64_t)source[i].n_ex * source[i].F);
            n_x[i - 1] = ni;
        }
    }
    A.total = carve(plan_workspace, plan_requests, nb_log2).total;
    A.nb_log2 = (unsigned)nb_log2;
    A.cs = counter_shift();
    A.D = deferred_of(deferred);
    A.D1 = deferred_of(companion_deferred);
    RECALGO_REQUIRE(A.D.last_step == nullptr || (step_dev != nullptr && A.D.lr_ring != nullptr));
    RECALGO_REQUIRE(A.D1.last_step == nullptr || (A.D.last_step != nullptr && A.D1.lr_ring != nullptr));
    RECALGO_REQUIRE(!sweep || (A.D.last_step != nullptr && sweep_period >= 1 && sweep_period <= (int)kLrRing - 8));
    A.step = reinterpret_cast<const long long*>(step_dev);
    A.step_off = step_offset;
    A.K = (unsigned)K;
    A.n_req = n ? (unsigned)((int64_t)source->n_ex * source->F) : 0u;
    RECALGO_REQUIRE(!catchup || A.D.last_step != nullptr);
    unsigned catch_blocks = (catchup && A.n_req) ? (unsigned)cdiv(A.n_req, kCatchReq) : 0u;
    const unsigned comp_blocks = (catchup && A.D1.last_step && A.n_req) ? (unsigned)cdiv(A.n_req, kThreads) : 0u;
    unsigned count_blocks = (count && n) ? (unsigned)cdiv(n, kThreads) : 0u;
    A.catch_end[0] = catch_blocks;
    A.count_end[0] = count_blocks;
    for (int i = 1; i < kPrepSources; ++i) {
        if (i < n_sources) {
            if (catchup) catch_blocks += (unsigned)cdiv(A.nreq_x[i - 1], kCatchReq);
            if (count) count_blocks += (unsigned)cdiv(n_x[i - 1], kThreads);
        }
        A.catch_end[i] = catch_blocks;
        A.count_end[i] = count_blocks;
    }
    A.period = sweep_period < 1 ? 1 : sweep_period;
    A.rows = rows; A.rows1 = companion_rows;
    // row blocks (of 256 << g rows) per step: every P-th block of the arena
    A.bshift = sweep_block_shift(rows, A.period);
    A.bshift1 = sweep_block_shift(companion_rows, A.period);
    const long long brows = (long long)kThreads << A.bshift, brows1 = (long long)kThreads << A.bshift1;
    const long long share = cdiv(cdiv(rows, brows), A.period), share1 = cdiv(cdiv(companion_rows, brows1), A.period);
    A.passes = sweep_passes(share * brows);
    A.passes1 = sweep_passes(share1 * brows1);
    A.sweeping = sweep ? 1 : 0;
    const unsigned sweep_blocks = sweep ? (unsigned)cdiv(share * (brows / kSweepRows), (long long)A.passes) : 0u;
    const unsigned sweep1_blocks = (sweep && A.D1.last_step) ? (unsigned)cdiv(share1 * (brows1 / kSweepRows), (long long)A.passes1) : 0u;
    A.b_catch = catch_blocks;
    A.b_comp = A.b_catch + comp_blocks;
    A.b_count = A.b_comp + count_blocks;
    A.b_sweep = A.b_count + sweep_blocks;
    const unsigned blocks = A.b_sweep + sweep1_blocks;
    if (blocks == 0) return 0;
    const size_t smem = kSlots * 9 * sizeof(unsigned);        // (count tiles; the claim lists of the other workgroups are smaller)
    static_assert(kClaimsLdsBytes <= kSlots * 9 * sizeof(unsigned), "claim lists fit the count tiles' LDS");
    hipLaunchKernelGGL(sparse_prepare_kernel, dim3(blocks), dim3(kThreads), smem, as_stream(stream), A);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_scatter_apply(const recalgo_scatter_source_t* sources, int n_sources,
                                         const recalgo_scatter_companion_t* companion, int K, void* plan_workspace,
                                         int64_t plan_requests, int nb_log2, int mode, float* w, float* m, float* v,
                                         float* grad, const recalgo_deferred_adam_t* deferred, int64_t rows,
                                         const recalgo_live_t* live, const int64_t* step_dev,
                                         int step_offset, float lr, float beta1, float beta2, float eps,
                                         recalgo_stream_t stream) {
    RECALGO_REQUIRE(sources != nullptr && n_sources >= 1 && n_sources <= kMaxSources && plan_workspace != nullptr);
    RECALGO_REQUIRE(nb_ok(nb_log2) && rows >= 0 && rows < (1ll << 31));
    const int prescanned = (mode & RECALGO_SCATTER_PRESCANNED) != 0;
    mode &= ~RECALGO_SCATTER_PRESCANNED;
    RECALGO_REQUIRE(mode == RECALGO_SCATTER_GRAD || mode == RECALGO_SCATTER_ADAM || mode == RECALGO_SCATTER_LAZY_ADAM);
    RECALGO_REQUIRE(mode != RECALGO_SCATTER_GRAD ? (w && m && v && step_dev) : grad != nullptr);
    RECALGO_REQUIRE(mode != RECALGO_SCATTER_ADAM || (deferred && deferred->last_step && deferred->lr_ring));
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
    const unsigned W = (unsigned)cdiv(n_total, kThreads);     // tiles of the plan
    P.n_src = n_sources;
    P.n_total = n_total;
    P.total = ws.total; P.cursor = ws.cursor; P.sched = ws.sched; P.keys = ws.keys; P.partials = ws.partials;
    P.hdr = ws.hdr;
    P.offs_g = ws.offs; P.prescanned = prescanned;
    P.lr_ring = mode == RECALGO_SCATTER_ADAM ? deferred->lr_ring : nullptr;
    P.lr_ring1 = (mode == RECALGO_SCATTER_ADAM && companion) ? companion->deferred->lr_ring : nullptr;
    P.step = mode != RECALGO_SCATTER_GRAD ? reinterpret_cast<const long long*>(step_dev) : nullptr;
    P.step_off = step_offset;
    P.lr = lr; P.b1 = beta1; P.b2 = beta2;
    P.partials1 = companion ? ws.partials1 : nullptr;
    P.nb_log2 = (unsigned)nb_log2;
    P.cs = counter_shift();
    P.KV = G.KV; P.L = G.L;
    P.stage_ok = (size_t)K * kThreads * sizeof(float) <= 32 * 1024;      // (K <= 32: every model of the reference)
    const size_t smem = ((prescanned ? 0 : (size_t)nb) + 8 * kThreads + 8 + 16 + kSlots * 10) * sizeof(unsigned) + kThreads * 4 * sizeof(float) +
                        kMaxSources * sizeof(SrcDev) + (P.stage_ok ? (size_t)(K + 1) * kThreads * sizeof(float) : 0);
    if (smem > 64 * 1024) {
        hipError_t e = G.vec == 4 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&sparse_place_kernel<4>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)
                                  : hipFuncSetAttribute(reinterpret_cast<const void*>(&sparse_place_kernel<1>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    const unsigned place_blocks = W ? W : 1;                  // (the scan of all-zero totals still publishes offs[])
    if (G.vec == 4)
        hipLaunchKernelGGL(sparse_place_kernel<4>, dim3(place_blocks), dim3(kThreads), smem, st, P);
    else
        hipLaunchKernelGGL(sparse_place_kernel<1>, dim3(place_blocks), dim3(kThreads), smem, st, P);
    ApplyArgs A;
    for (int i = 0; i < kMaxSources; ++i) A.src[i] = P.src[i];
    A.n_src = n_sources;
    A.sched = ws.sched; A.hdr = ws.hdr; A.total = ws.total; A.cursor = ws.cursor; A.cs = counter_shift();
    A.keys = ws.keys; A.keys_alt = ws.keys_alt; A.partials = ws.partials; A.partials1 = ws.partials1;
    A.mode = mode;
    A.w = w; A.m = m; A.v = v; A.grad = grad;
    A.last_step = mode == RECALGO_SCATTER_ADAM ? deferred->last_step : nullptr;
    A.lr_ring = mode == RECALGO_SCATTER_ADAM ? deferred->lr_ring : nullptr;
    A.w1 = A.m1 = A.v1 = A.grad1 = nullptr;
    A.last_step1 = nullptr; A.lr_ring1 = nullptr; A.rows1 = 0; A.has1 = 0;
    if (companion) {
        // the second arena: the same entries, walked by the lane that owns the row's first piece; only the gradient (a scalar
        // per request, the tile partial sums `place` wrote beside the main ones) and the arena differ
        A.has1 = 1;
        A.w1 = companion->w; A.m1 = companion->m; A.v1 = companion->v; A.grad1 = companion->grad;
        A.last_step1 = mode == RECALGO_SCATTER_ADAM ? companion->deferred->last_step : nullptr;
        A.lr_ring1 = mode == RECALGO_SCATTER_ADAM ? companion->deferred->lr_ring : nullptr;
        A.rows1 = companion->rows;
    }
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
        A.step = reinterpret_cast<const long long*>(ws.hdr);
        A.step_off = 0;
    }
    if (G.vec == 4)
        hipLaunchKernelGGL(sparse_apply_kernel<4>, dim3(nb), dim3(kThreads), 0, st, A);
    else
        hipLaunchKernelGGL(sparse_apply_kernel<1>, dim3(nb), dim3(kThreads), 0, st, A);
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
    A.K = (unsigned)K;
    A.row0 = row_begin; A.row1 = row_end;
    A.passes = sweep_passes(row_end - row_begin);
    const dim3 grid((unsigned)cdiv(row_end - row_begin, (int64_t)A.passes * kSweepRows));
    hipLaunchKernelGGL(sparse_sweep_kernel, grid, dim3(kThreads), kClaimsLdsBytes, as_stream(stream), A);
    RECALGO_RETURN_LAST();
}
