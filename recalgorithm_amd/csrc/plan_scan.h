// The prefix scan of a scatter plan's bucket totals (sparse.hip: `prepare` counts, `place` needs every bucket's first entry,
// `apply` the order the buckets are taken in), as ONE 256-thread workgroup's job — hosted by a launch that runs between the
// last count and `place` anyway (the dense optimizer step, tail.hip), so that the tiles of `place` read a compact offs[nb]
// instead of each scanning the cache-line-spread counters themselves (nb lines per tile: 160 MB of L2 reads for DIN's 1216
// tiles x 2048 buckets).
#pragma once
#include "common.h"

namespace recalgo_plan {

struct Scan {
    const unsigned* total;   // [nb << cs]: bucket b's count at word b << cs
    unsigned* offs;          // [nb] out: first entry of bucket b
    uint4* sched;            // [nb] out: (bucket, first entry, entries, 0) in the order `apply` takes them (heavy buckets first)
    unsigned cs, nb_log2;
};

// 256-thread exclusive scan of one value per thread; sh: 8 words of LDS
__device__ __forceinline__ unsigned excl_scan256(unsigned v, unsigned* sh, unsigned& total) {
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
    for (unsigned w = 0; w < 4; ++w) {
        const unsigned t = sh[w];
        if (w < wave) before += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return before + inc - v;
}

// all 256 threads of one workgroup; nb is a multiple of 256 (1024 .. 8192: 4 .. 32 buckets per thread).  Up to 16 buckets per
// thread the counters are read ONCE, all loads in flight together, and kept in registers; larger plans read them again in
// every pass (they stay in L2) rather than parking nb words in LDS: the host kernel keeps its LDS footprint.
constexpr unsigned kRegs = 16;
template <bool REGS>
__device__ __forceinline__ void scan_block_impl(const Scan& S, unsigned* sh) {
    const unsigned bpt = (1u << S.nb_log2) / 256u, b0 = threadIdx.x * bpt;
    unsigned cnt[kRegs];
    if (REGS) {
#pragma unroll
        for (unsigned k = 0; k < kRegs; ++k) cnt[k] = k < bpt ? S.total[(size_t)(b0 + k) << S.cs] : 0u;
    }
    auto count = [&](unsigned k) -> unsigned { return REGS ? cnt[k] : S.total[(size_t)(b0 + k) << S.cs]; };
    unsigned sum = 0;
    if (REGS) {
#pragma unroll
        for (unsigned k = 0; k < kRegs; ++k) sum += cnt[k];
    } else {
        for (unsigned k = 0; k < bpt; ++k) sum += count(k);
    }
    unsigned total;
    const unsigned first = excl_scan256(sum, sh, total);
    // the buckets that hold a hot row (many entries: a long tail of `apply` when they start late) are dispatched first
    const unsigned heavy_min = 2u * (total >> S.nb_log2) + 64u;
    unsigned nh = 0;
    if (REGS) {
#pragma unroll
        for (unsigned k = 0; k < kRegs; ++k) nh += (k < bpt && cnt[k] >= heavy_min) ? 1u : 0u;
    } else {
        for (unsigned k = 0; k < bpt; ++k) nh += count(k) >= heavy_min;
    }
    unsigned n_heavy;
    unsigned hrun = excl_scan256(nh, sh, n_heavy);
    unsigned run = first;
    auto emit = [&](unsigned k, unsigned tb) {
        const unsigned b = b0 + k;
        S.offs[b] = run;
        const uint4 rec = make_uint4(b, run, tb, 0u);
        run += tb;
        if (tb >= heavy_min) S.sched[hrun++] = rec;
        else S.sched[n_heavy + b - hrun] = rec;               // (b - hrun = the light buckets before b)
    };
    if (REGS) {
#pragma unroll
        for (unsigned k = 0; k < kRegs; ++k)
            if (k < bpt) emit(k, cnt[k]);
    } else {
        for (unsigned k = 0; k < bpt; ++k) emit(k, count(k));
    }
}
__device__ __forceinline__ void scan_block(const Scan& S, unsigned* sh) {
    if (S.nb_log2 <= 12) scan_block_impl<true>(S, sh);        // (workgroup-uniform)
    else scan_block_impl<false>(S, sh);
}

}  // namespace recalgo_plan
