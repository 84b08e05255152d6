// Register-blocked fp32-MFMA tile engine (round 5) for tf.layers.dense forward / input gradient / weight gradient
// (the context MLP of every model_fn, e.g. /root/reference algorithm/DCN/dcn.py:162-170, DeepFM/deepfm.py:204-212).
//
// Same arithmetic as the round-2 engine of dense.hip — v_mfma_f32_32x32x2_f32, exact fp32 fmaf chains — but a wave now owns
// WM x WN sub-tiles of 32 x 32 and an LDS operand read feeds several MFMAs:
//   * reduction-contiguous operand ("RC", [idx][red] in memory): LDS [T idx][32 red] with row stride 36 floats; lane
//     (l32, hi) reads ONE ds_read_b128 at [idx][8 g + 4 hi] = its operand values of FOUR consecutive MFMA steps (step 4 g + e
//     contracts red 8 g + e and 8 g + 4 + e; any assignment of reduction indices to steps is a valid fp32 chain as long as both
//     operands use the same one).  Stride 36: the 16 lanes the hardware services together (MI355X_MICROARCH.md, LDS table)
//     cover 16 distinct residues mod 16 of idx -> 16 distinct groups of 4 banks: conflict-free.
//   * reduction-major operand ("RM", [red][idx] in memory): LDS [32 red][T idx] exactly as in memory (one ds_write_b128 per
//     staged float4); the wave's W sub-tiles interleave along idx (sub-tile j holds idx = base + W * l32 + j), so ONE
//     ds_read_b32 / _b64 of W consecutive floats feeds the step's MFMAs of all W sub-tiles.
// LDS instructions per MFMA: 0.3 - 1.25 (round-2 engine: 2 ds_read_b32 + their share of the scalar tile stores).
// Pipeline: 3-slot LDS ring, ONE barrier per 32-deep reduction chunk; fragments double-buffered per group of 4 steps; global
// loads (raw buffer loads, out-of-range -> hardware zeros) run three chunks ahead through two register sets; every non-MFMA
// instruction of an iteration is pinned behind a fixed MFMA of the same iteration (sched_barrier), so that one wave per SIMD
// keeps the matrix pipe busy by itself.
#pragma once
#include "common.h"

namespace tv2 {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kThreads = 256;
constexpr int BK = 32;
constexpr int kLdRC = 36;
constexpr int kStages = 3;
constexpr int kOOB = (int)0x80000000u;

struct Operand {
    const float* p;        // RC: [n_idx][ld] (red contiguous)   RM: [n_red][ld] (idx contiguous)
    const float* mask;     // optional, same layout: element := 0 where mask <= 0   (ReLU backward)
    int ld;
    int bytes;             // extent in bytes (buffer descriptor num_records)
};

template <bool RC, int W>
struct Geom {
    static constexpr int T = 64 * W;                              // tile extent along idx (2 waves x W sub-tiles x 32)
    static constexpr int kFloats = RC ? T * kLdRC : BK * T;       // LDS floats per ring slot
    static constexpr int kLoads = 2 * W;                          // float4 per thread per chunk
    static constexpr int U = T / 4;                               // RM: float4 units per reduction row
};

template <bool MASK, int N>
struct StageRegs {
    float4 v[N];
    float4 m[MASK ? N : 1];
};

__device__ __forceinline__ float4 relu_mask4(float4 v, float4 m) {
    return make_float4(m.x > 0.f ? v.x : 0.f, m.y > 0.f ? v.y : 0.f, m.z > 0.f ? v.z : 0.f, m.w > 0.f ? v.w : 0.f);
}

template <bool RC, int W, bool MASK, bool EXACT>
struct Stager {
    using G = Geom<RC, W>;
    __amdgpu_buffer_rsrc_t rs, rm;
    int off[G::kLoads];        // byte offset of this thread's float4 j at chunk 0 (kOOB: idx out of range)
    int red[G::kLoads];        // its (first) reduction index at chunk 0
    int step_b, red_end;

    __device__ __forceinline__ void init(const Operand& op, int idx0, int red0, int n_idx, int red_end_) {
        const int t = threadIdx.x;
        red_end = red_end_;
        step_b = (int)((RC ? (size_t)BK : (size_t)BK * op.ld) * sizeof(float));
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.p), (short)0, op.bytes, 0x00020000);
        rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MASK ? op.mask : op.p), (short)0, op.bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < G::kLoads; ++j) {
            const int u = t + kThreads * j;
            int idx, r;
            if constexpr (RC) { idx = idx0 + (u >> 3); r = red0 + (u & 7) * 4; }
            else { r = red0 + u / G::U; idx = idx0 + (u % G::U) * 4; }
            red[j] = r;
            const size_t o = RC ? (size_t)idx * op.ld + r : (size_t)r * op.ld + idx;
            off[j] = idx < n_idx ? (int)(o * sizeof(float)) : kOOB;
        }
    }
    __device__ __forceinline__ void issue(int c, int j, StageRegs<MASK, G::kLoads>& st) const {
        if constexpr (EXACT) {
            const int so = c * step_b;                         // wave-uniform: an SGPR offset
            st.v[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off[j], so, 0));
            if constexpr (MASK) st.m[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rm, off[j], so, 0));
        } else {
            const int o = red[j] + c * BK < red_end ? off[j] + c * step_b : kOOB;
            st.v[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0));
            if constexpr (MASK) st.m[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rm, o, 0, 0));
        }
    }
    static __device__ __forceinline__ float4 value(const StageRegs<MASK, G::kLoads>& st, int j) {
        if constexpr (MASK) return relu_mask4(st.v[j], st.m[j]);
        else return st.v[j];
    }
    // registers -> LDS slot S, float4 j (one ds_write_b128)
    static __device__ __forceinline__ void store(float* __restrict__ S, const StageRegs<MASK, G::kLoads>& st, int j) {
        const int u = threadIdx.x + kThreads * j;
        float* d;
        if constexpr (RC) d = S + (u >> 3) * kLdRC + (u & 7) * 4;
        else d = S + (u / G::U) * G::T + (u % G::U) * 4;
        *reinterpret_cast<float4*>(d) = value(st, j);
    }
};

// operand values of one group of 4 MFMA steps: x[e][sub-tile]
template <int W>
struct Frag {
    float x[4][W];
};

// wb: first idx of this wave's W sub-tiles inside the tile (wave_select * 32 * W)
template <bool RC, int W>
__device__ __forceinline__ void read_item(const float* __restrict__ S, int wb, int l32, int hi, int g, int item, Frag<W>& f) {
    if constexpr (RC) {                                   // item = sub-tile
        const float4 q = *reinterpret_cast<const float4*>(S + (wb + 32 * item + l32) * kLdRC + 8 * g + 4 * hi);
        f.x[0][item] = q.x; f.x[1][item] = q.y; f.x[2][item] = q.z; f.x[3][item] = q.w;
    } else {                                              // item = step e
        const float* p = S + (8 * g + 4 * hi + item) * Geom<false, W>::T + wb + W * l32;
        if constexpr (W == 1) f.x[item][0] = p[0];
        else if constexpr (W == 2) { const float2 q = *reinterpret_cast<const float2*>(p); f.x[item][0] = q.x; f.x[item][1] = q.y; }
        else { const float4 q = *reinterpret_cast<const float4*>(p); f.x[item][0] = q.x; f.x[item][1] = q.y; f.x[item][2] = q.z; f.x[item][3] = q.w; }
    }
}
template <bool RC, int W> constexpr int frag_items() { return RC ? W : 4; }

// position of (sub-tile, lane index) along idx inside the wave's range (the MFMA's row / column index of that lane)
template <bool RC, int W>
__device__ __forceinline__ int idx_of(int sub, int lane_idx) { return RC ? 32 * sub + lane_idx : W * lane_idx + sub; }

// acc[i][j] (32 x 32 each) += sum over the reduction range [red_begin, red_end) of A B for the workgroup tile at (m0, n0).
// TWO: the groups of four steps alternate between acc and accb (two fmaf chains of half the length per element, summed by the
// caller: the rounding error of an 800-term batch-split chain otherwise exceeds the fp32 reference's blocked summation —
// tests/util.py's strict guard).  As / Bs: kStages * Geom::kFloats floats each.  COLSUM: also accumulate, per thread, the column sums of the B float4s this
// thread stages (wgrad: dbias); B must be RM then (a thread's float4s share their 4 columns).
template <bool A_RC, bool B_RC, int WM, int WN, bool MASK_A, bool MASK_B, bool COLSUM, bool EXACT, bool TWO = false>
__device__ __forceinline__ void mainloop(const Operand& A, const Operand& B, int m0, int n0, int M, int N, int red_begin,
                                         int red_end, float* __restrict__ As, float* __restrict__ Bs,
                                         f32x16 (&acc)[WM][WN], float4& colsum, f32x16 (*accb)[WN] = nullptr) {
    using GA = Geom<A_RC, WM>;
    using GB = Geom<B_RC, WN>;
    using SA = Stager<A_RC, WM, MASK_A, EXACT>;
    using SB = Stager<B_RC, WN, MASK_B, EXACT>;
    constexpr int nLA = GA::kLoads, nLB = GB::kLoads, nL = nLA + nLB;
    constexpr int nFA = frag_items<A_RC, WM>(), nFB = frag_items<B_RC, WN>(), nF = nFA + nFB;
    constexpr int G = 4 * WM * WN;                       // MFMAs (= side-work slots) per group of 4 steps
    static_assert(nL <= G, "one load / store item per slot");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, l32 = lane & 31;
    const int wbA = (wave >> 1) * 32 * WM, wbB = (wave & 1) * 32 * WN;
    const int nchunks = (red_end - red_begin + BK - 1) / BK;
    if (nchunks <= 0) return;
    SA ga;
    SB gb;
    ga.init(A, m0, red_begin, M, red_end);
    gb.init(B, n0, red_begin, N, red_end);
    StageRegs<MASK_A, nLA> sa[2];
    StageRegs<MASK_B, nLB> sb[2];
    // prologue: chunks 0, 1 -> slots 0, 1; chunk 2 in flight in set 0
#pragma unroll
    for (int j = 0; j < nLA; ++j) ga.issue(0, j, sa[0]);
#pragma unroll
    for (int j = 0; j < nLB; ++j) gb.issue(0, j, sb[0]);
#pragma unroll
    for (int j = 0; j < nLA; ++j) ga.issue(1, j, sa[1]);
#pragma unroll
    for (int j = 0; j < nLB; ++j) gb.issue(1, j, sb[1]);
    __syncthreads();                                      // previous users of the LDS ring are done
#pragma unroll
    for (int j = 0; j < nLA; ++j) SA::store(As, sa[0], j);
#pragma unroll
    for (int j = 0; j < nLB; ++j) SB::store(Bs, sb[0], j);
    if constexpr (COLSUM) {
#pragma unroll
        for (int j = 0; j < nLB; ++j) colsum = f4_add(colsum, SB::value(sb[0], j));
    }
#pragma unroll
    for (int j = 0; j < nLA; ++j) ga.issue(2, j, sa[0]);
#pragma unroll
    for (int j = 0; j < nLB; ++j) gb.issue(2, j, sb[0]);
#pragma unroll
    for (int j = 0; j < nLA; ++j) SA::store(As + GA::kFloats, sa[1], j);
#pragma unroll
    for (int j = 0; j < nLB; ++j) SB::store(Bs + GB::kFloats, sb[1], j);
    if constexpr (COLSUM) {
        if (nchunks > 1) {
#pragma unroll
            for (int j = 0; j < nLB; ++j) colsum = f4_add(colsum, SB::value(sb[1], j));
        }
    }
    __syncthreads();
    Frag<WM> fa[2];
    Frag<WN> fb[2];
#pragma unroll
    for (int it = 0; it < nFA; ++it) read_item<A_RC, WM>(As, wbA, l32, hi, 0, it, fa[0]);
#pragma unroll
    for (int it = 0; it < nFB; ++it) read_item<B_RC, WN>(Bs, wbB, l32, hi, 0, it, fb[0]);
    int s0 = 0, s1 = 1, s2 = 2;                           // ring slots of chunks c, c+1, c+2
    // cur: staged registers of chunk c+2 (loaded in the previous iteration); nxt: receives chunk c+3
    auto step = [&](int c, StageRegs<MASK_A, nLA>& curA, StageRegs<MASK_B, nLB>& curB, StageRegs<MASK_A, nLA>& nxtA,
                    StageRegs<MASK_B, nLB>& nxtB) {
        const float* rA0 = As + s0 * GA::kFloats;
        const float* rB0 = Bs + s0 * GB::kFloats;
        const float* rA1 = As + s1 * GA::kFloats;
        const float* rB1 = Bs + s1 * GB::kFloats;
        float* wA = As + s2 * GA::kFloats;
        float* wB = Bs + s2 * GB::kFloats;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float* rA = g == 3 ? rA1 : rA0;         // group 3 prefetches group 0 of the next chunk
            const float* rB = g == 3 ? rB1 : rB0;
            const int gn = (g + 1) & 3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int i = 0; i < WM; ++i) {
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        if (TWO && (g & 1)) accb[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1].x[e][i], fb[g & 1].x[e][j], accb[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1].x[e][i], fb[g & 1].x[e][j], acc[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        const int ls = (e * WM + i) * WN + j;             // slot inside the group, 0 .. G-1
                        // (1) fragments of the next group: item k in slot k (slots wrap when there are more items than slots)
#pragma unroll
                        for (int k = 0; k < nF; ++k) {
                            if (k % G == ls) {
                                if (k < nFA) read_item<A_RC, WM>(rA, wbA, l32, hi, gn, k, fa[(g + 1) & 1]);
                                else read_item<B_RC, WN>(rB, wbB, l32, hi, gn, k - nFA, fb[(g + 1) & 1]);
                            }
                        }
                        // (2) group 0: global loads of chunk c+3
                        if (g == 0 && ls < nL) {
                            if (ls < nLA) ga.issue(c + 3, ls, nxtA);
                            else gb.issue(c + 3, ls - nLA, nxtB);
                        }
                        // (3) group 2: registers (chunk c+2) -> LDS
                        if (g == 2 && ls < nL) {
                            if (ls < nLA) SA::store(wA, curA, ls);
                            else SB::store(wB, curB, ls - nLA);
                        }
                        if constexpr (COLSUM) {
                            if (g == 3 && ls < nLB) {
                                if (c + 2 < nchunks) colsum = f4_add(colsum, SB::value(curB, ls));
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        const int t = s0;
        s0 = s1; s1 = s2; s2 = t;
        __syncthreads();
    };
    for (int c = 0; c < nchunks; c += 2) {
        step(c, sa[0], sb[0], sa[1], sb[1]);
        if (c + 1 < nchunks) step(c + 1, sa[1], sb[1], sa[0], sb[0]);
    }
}

// XCD-aware linear block -> tile: block b runs on XCD b % 8; give every XCD a contiguous range of the tile order
__device__ __forceinline__ int xcd_swizzle(int b, int total) {
    if (total % 8) return b;
    return (b % 8) * (total / 8) + b / 8;
}

// row of accumulator register r of lane (.., hi) inside a 32 x 32 MFMA result (the column is l32)
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// Row-wise epilogue of one wave's 32 x 32 tile.  An accumulator register holds ONE column per lane (two rows per instruction,
// 128 bytes each): 16 dword stores per lane, and the 512 workgroups of a layer all issue them at once — 3 us of a 21 us launch
// (scripts/mfma_lab.hip timeline).  Instead the tile goes through `T` (32 x kLdT floats of LDS private to the wave: the operand
// ring is free after the main loop's last barrier) and every lane gets float4 (row, 4 columns) pieces: 8 lanes per 128-byte
// row segment, 8 rows per instruction, 4 dwordx4 stores per lane.  f(it, row, col, v): piece it < 4 (compile time) of this
// lane, row = 8 it + lane / 8 and col = 4 (lane % 8) inside the tile.
constexpr int kLdT = 36;
constexpr int kTileScratch = 32 * kLdT;
template <class F>
__device__ __forceinline__ void tile_rows(float* __restrict__ T, const f32x16& acc, F&& f) {
    const int lane = threadIdx.x & 63, hi = lane >> 5, l32 = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) T[acc_row(r, hi) * kLdT + l32] = acc[r];
    __builtin_amdgcn_wave_barrier();              // same wave writes and reads: its LDS operations complete in issue order
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), col = (lane & 7) * 4;
        f(it, row, col, *reinterpret_cast<const float4*>(T + row * kLdT + col));
    }
    __builtin_amdgcn_wave_barrier();
}

}  // namespace tv2
