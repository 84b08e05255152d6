// tf.layers.dense (the context MLP of every model_fn, e.g. algorithm/DCN/dcn.py:163-166, and the D-way
// contraction of the PNN product layer, algorithm/PNN/pnn.py:139,146-181) on the gfx950 fp32 matrix
// cores: v_mfma_f32_32x32x2_f32, exact fp32 (a k-ordered fmaf chain), 64 FLOP/clk/SIMD = the chip's fp32
// peak (157 TFLOP/s).  Three GEMM forms, one tile engine:
//
//   forward   Y[M,N]  = act( X[M,K] W[K,N] (+ X2[M,K2] W2[K2,N]) + bias[N] )            A: rows x red-contiguous, B: red-major
//   dgrad     dX[M,K] = (G (.) [Ymask > 0])[M,N] W[K,N]^T  (+ beta * C[M,K])           A: red-contiguous,      B: red-contiguous
//   wgrad     dW[K,N] = X[M,K]^T (G (.) [Ymask > 0])[M,N],  dbias[N] = colsum(G (.) mask)   A: red-major, B: red-major; split over M
//
// Fused epilogues / prologues (what the library GEMMs needed extra launches for): bias + ReLU in the
// forward store; the ReLU mask of the backward applied while the gradient tile is staged (G2 = G * [Y > 0]
// is never materialised); the bias gradient accumulated from the staged gradient tiles of the wgrad; the
// `beta * C` term of the first DIN layer's mini-batch-aware regulariser in the dgrad store.
//
// Tile engine: workgroup = 4 waves = 64 x 64 output tile, one 32 x 32 sub-tile per wave, reduction chunks of 32 through a
// 3-slot LDS ring with one barrier per chunk.  Float4-addressable operands (every layer of the benchmark models) run on the
// round-5 main loop of tile_v2.h: reduction-contiguous operands are fetched from LDS as ONE ds_read_b128 per four MFMA steps
// (row stride 36), tiles are staged with ds_write_b128, fragments are double-buffered per group of four steps — 0.3-1.25 LDS
// instructions per MFMA instead of 2.4, main loop at 0.93 of the matrix pipe's rate (round 2: 0.76; scripts/mfma_lab.hip,
// profiles/r05_mfma_lab.md).  Everything below about ds_read_b32 layouts describes the round-2 main loop, which remains for
// operands that are not float4-addressable (the reference's default DCN input width 82, odd leading dimensions): one 32 x 32
// sub-tile per wave held as TWO interleaved accumulator chains, every piece of non-MFMA work of an iteration issued in the
// shadow of the iteration's own 16 MFMAs at fixed slots (see tile_mainloop_impl), LDS layouts chosen per operand form so
// that every MFMA operand read is a conflict-free ds_read_b32:
//   red-contiguous operand ([idx][red] in memory): LDS [64 idx][32 red], row stride 33 (odd): lane l reads
//       [idx0 + (l & 31)][kk + (l >> 5)] -> 32 distinct banks per half wave; staged by 4 scalar stores;
//   red-major operand ([red][idx] in memory): LDS [32 red][64 idx], row stride 64: lanes read 32 consecutive
//       floats; staged by one ds_write_b128.
// A [4096 x 512 x 416] layer is 512 workgroups (2 per CU).  blockIdx -> tile is XCD-aware: the 8 XCDs (block b runs
// on XCD b % 8) each own a contiguous range of row tiles, so an XCD's L2 holds its own 1/8 of X plus the (small) W.
// The weight gradient is split over the batch into deterministic slabs (summed in fixed order by
// dense_sum_slabs_kernel — per layer, or once per training step for all layers: recalgo_dense_bwd_weights_reduce);
// recalgo_dense_bwd runs a layer's input- and weight-gradient tiles in ONE grid.
#include <cstdlib>

#include "common.h"
#include "cross_bwd.h"
#include "act.h"
#include "tile_v2.h"
#include "dropout.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kThreads = 256;
constexpr int BM = 64, BN = 64, BK = 32;
constexpr int kLdRC = BK + 1;                     // red-contiguous operand: [64][33]
constexpr int kLdRM = 64;                         // red-major operand:      [32][64]
constexpr int kBufFloats = 64 * tv2::kLdRC;       // 2304: a ring slot of either main loop (round 2: 64 * 33, 32 * 64)

struct Operand {
    const float* p;        // RC: [n_idx][ld] (red contiguous)   RM: [n_red][ld] (idx contiguous)
    const float* mask;     // optional, same layout: element := 0 where mask <= 0   (ReLU backward)
    int ld;
    int vec;               // base 16-byte aligned and ld % 4 == 0: float4 loads (else four scalar loads, e.g. the
                           // reference's default DCN input d = 82, or the 351 Gram features of IPNN at F = 26)
    int bytes;             // extent of the operand in bytes (buffer descriptor num_records); 0: not buffer-addressable
};

__device__ __forceinline__ float4 load4_guard(const float* p, int remaining, int vec) {
    if (remaining >= 4) {
        if (vec) return *reinterpret_cast<const float4*>(p);
        return make_float4(p[0], p[1], p[2], p[3]);
    }
    float4 v = f4_zero();
    if (remaining > 0) v.x = p[0];
    if (remaining > 1) v.y = p[1];
    if (remaining > 2) v.z = p[2];
    return v;
}
__device__ __forceinline__ float4 relu_mask(float4 v, float4 m) {
    return make_float4(m.x > 0.f ? v.x : 0.f, m.y > 0.f ? v.y : 0.f, m.z > 0.f ? v.z : 0.f, m.w > 0.f ? v.w : 0.f);
}

// ---------------------------------------------------------------------------------------------
// Operand staging: global -> registers -> LDS.
// FAST (operand float4-addressable: base 16-byte aligned, ld % 4 == 0, extent of the contiguous dimension
// % 4 == 0, < 2 GiB): raw buffer loads through a buffer descriptor whose num_records is the operand's extent —
// an out-of-range coordinate gets the offset 0x80000000 and the hardware returns zeros, so a chunk's loads are
// branch-free, issue back to back, and may run past the end of the reduction (the software pipeline below never
// tests "is there a next chunk").
// !FAST: element-wise guarded loads (any alignment / extent; e.g. the reference's default DCN width 82).
// ---------------------------------------------------------------------------------------------
using u32x4 = __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int;
constexpr int kOOB = (int)0x80000000u;

template <bool MASK>
struct StageRegs {
    float4 v[2];
    float4 m[2];           // raw mask values (MASK only); applied when the tile is written to LDS
};

// EXACT (FAST only): the reduction extent is a whole number of chunks — the chunk advance is then a scalar offset of the
// buffer load and no per-load bounds arithmetic is left (loads past the end fetch in-bounds garbage or OOB zeros into
// ring slots that are never consumed).
template <bool RC, bool FAST, bool MASK, bool EXACT>
struct Stager {
    __amdgpu_buffer_rsrc_t rs, rm;     // FAST
    int off[2];                        // FAST: byte offsets of this thread's two float4 at chunk 0 (kOOB: never valid)
    int step_b;                        // FAST: bytes per chunk
    const float* p[2];                 // !FAST
    const float* m[2];
    size_t step;
    int red[2], idx[2];
    bool idx_ok[2];
    int n_idx, red_end, vec;

    __device__ __forceinline__ void init(const Operand& op, int idx0, int red0, int n_idx_, int red_end_) {
        const int t = threadIdx.x;
        n_idx = n_idx_; red_end = red_end_; vec = op.vec;
        step = RC ? (size_t)BK : (size_t)BK * op.ld;
        step_b = (int)(step * sizeof(float));
        if constexpr (FAST) {
            rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.p), (short)0, op.bytes, 0x00020000);
            rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MASK ? op.mask : op.p), (short)0, op.bytes, 0x00020000);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (RC) { idx[j] = idx0 + (t >> 3) + 32 * j; red[j] = red0 + (t & 7) * 4; }
            else { red[j] = red0 + (t >> 4) + 16 * j; idx[j] = idx0 + (t & 15) * 4; }
            idx_ok[j] = idx[j] < n_idx;
            const size_t o = RC ? (size_t)idx[j] * op.ld + red[j] : (size_t)red[j] * op.ld + idx[j];
            if constexpr (FAST) {
                off[j] = idx_ok[j] ? (int)(o * sizeof(float)) : kOOB;
            } else {
                p[j] = op.p + o;
                m[j] = MASK ? op.mask + o : nullptr;
            }
        }
    }
    // issue the load of this thread's float4 `j` of chunk `c` (relative to init's red0)
    __device__ __forceinline__ void issue(int c, int j, StageRegs<MASK>& st) const {
        const int r = red[j] + c * BK;
        if constexpr (FAST) {
            // beyond red_end (the split's end, or the matrix's): zeros.  idx beyond n_idx: off[j] is already kOOB
            // (kOOB + c * step_b stays >= 2^31 as an unsigned offset for every c the loop can reach)
            if constexpr (EXACT) {
                const int so = c * step_b;                 // wave-uniform: an SGPR offset
                st.v[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off[j], so, 0));
                if constexpr (MASK) st.m[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rm, off[j], so, 0));
            } else {
                const int o = r < red_end ? off[j] + c * step_b : kOOB;
                st.v[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0));
                if constexpr (MASK) st.m[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rm, o, 0, 0));
            }
        } else {
            float4 v = f4_zero(), mk = make_float4(1.f, 1.f, 1.f, 1.f);
            const size_t o = (size_t)c * step;
            if (RC ? (idx_ok[j] && r < red_end) : r < red_end) {
                const int remaining = RC ? red_end - r : n_idx - idx[j];
                v = load4_guard(p[j] + o, remaining, vec);
                if (MASK) mk = load4_guard(m[j] + o, remaining, vec);
            }
            st.v[j] = v;
            if constexpr (MASK) st.m[j] = mk;
        }
    }
};

template <bool MASK>
__device__ __forceinline__ float4 staged(const StageRegs<MASK>& st, int j) {
    if constexpr (MASK) return relu_mask(st.v[j], st.m[j]);
    else return st.v[j];
}

// registers -> LDS, in pieces so that the stores can be spread between the MFMAs of a chain:
// RC: piece q in 0..3 = (row j = q >> 1, half = q & 1) -> one ds_write2_b32;   RM: piece q in 0..1 -> one ds_write_b128
template <bool RC, bool MASK>
__device__ __forceinline__ void stage_store_piece(float* __restrict__ S, const StageRegs<MASK>& st, int q) {
    const int t = threadIdx.x;
    if constexpr (RC) {
        const int j = q >> 1, h = q & 1;
        const float4 v = staged<MASK>(st, j);
        float* d = S + ((t >> 3) + 32 * j) * kLdRC + (t & 7) * 4 + 2 * h;
        d[0] = h ? v.z : v.x;
        d[1] = h ? v.w : v.y;
    } else {
        *reinterpret_cast<float4*>(S + ((t >> 4) + 16 * q) * kLdRM + (t & 15) * 4) = staged<MASK>(st, q);
    }
}
template <bool RC, bool MASK>
__device__ __forceinline__ void stage_store(float* __restrict__ S, const StageRegs<MASK>& st) {
#pragma unroll
    for (int q = 0; q < (RC ? 4 : 2); ++q) stage_store_piece<RC, MASK>(S, st, q);
}

// MFMA operand values k, k+1 (k even) of this lane for one staged chunk
template <bool RC>
__device__ __forceinline__ void read_frag_pair(const float* __restrict__ S, int idx_local, int hi, int k, float (&f)[BK / 2]) {
    if constexpr (RC) {
        f[k] = S[idx_local * kLdRC + 2 * k + hi];
        f[k + 1] = S[idx_local * kLdRC + 2 * k + 2 + hi];
    } else {
        f[k] = S[(2 * k + hi) * kLdRM + idx_local];
        f[k + 1] = S[(2 * k + 2 + hi) * kLdRM + idx_local];
    }
}

struct Segment {           // one (A, B) pair contracted over n_red; a launch accumulates up to 2 of them
    Operand a, b;
    int n_red;
};

constexpr int kStages = 3;                        // LDS ring

// acc(32x32 of this wave) += sum over the segment's reduction range [red_begin, red_end) of A B.
// COLSUM: also accumulate, per thread, the column sums of the B tiles this thread stages (wgrad: dbias).
//
// Software pipeline.  One wave per SIMD must keep the matrix pipe busy by itself (co-resident workgroups run in
// lockstep, so their non-MFMA phases coincide instead of complementing each other), hence ALL other work of an
// iteration is issued in the shadow of its own 16 MFMAs (an MFMA occupies the pipe for 64 cycles but only one issue
// slot), in this fixed order (`sched_barrier` pins it):
//   iteration c:  MFMA k of chunk c            k = 0..15, alternating between two accumulator chains
//                 + global loads of chunk c+3  (k = 0..3; consumed at the END of iteration c+1: ~2 iterations of latency)
//                 + LDS -> operand registers of chunk c+1, one read pair per MFMA
//                 + registers (loaded in iteration c-1) -> LDS slot of chunk c+2   (k = 8..13)
//                 one barrier
// LDS ring of 3: while chunk c+1 is read out of slot (c+1) % 3, chunk c+2 lands in slot (c+2) % 3 = (c-1) % 3, whose
// last readers passed two barriers ago.  Loads and slots past the end of the reduction are zeros / unused.
template <bool A_RC, bool B_RC, bool FAST, bool MASK_A, bool MASK_B, bool COLSUM, bool EXACT>
__device__ __forceinline__ void tile_mainloop_impl(const Segment& sg, int m0, int n0, int M, int N, int red_begin, int red_end,
                                              float* __restrict__ As, float* __restrict__ Bs, f32x16& acc, f32x16& acc1, float4& colsum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, l32 = lane & 31;
    const int am = (wave >> 1) * 32 + l32, bn = (wave & 1) * 32 + l32;
    const int nchunks = (red_end - red_begin + BK - 1) / BK;
    if (nchunks <= 0) return;
    Stager<A_RC, FAST, MASK_A, EXACT> ga;
    Stager<B_RC, FAST, MASK_B, EXACT> gb;
    ga.init(sg.a, m0, red_begin, M, red_end);
    gb.init(sg.b, n0, red_begin, N, red_end);
    StageRegs<MASK_A> sa0, sa1;
    StageRegs<MASK_B> sb0, sb1;
    // prologue: chunks 0, 1 -> slots 0, 1; chunk 2 in flight in set 0
#pragma unroll
    for (int j = 0; j < 2; ++j) { ga.issue(0, j, sa0); gb.issue(0, j, sb0); ga.issue(1, j, sa1); gb.issue(1, j, sb1); }
    __syncthreads();                                      // previous users of the LDS ring are done
    stage_store<A_RC, MASK_A>(As, sa0);
    stage_store<B_RC, MASK_B>(Bs, sb0);
    stage_store<A_RC, MASK_A>(As + kBufFloats, sa1);
    stage_store<B_RC, MASK_B>(Bs + kBufFloats, sb1);
    if constexpr (COLSUM) {
        colsum = f4_add(colsum, f4_add(staged<MASK_B>(sb0, 0), staged<MASK_B>(sb0, 1)));
        if (nchunks > 1) colsum = f4_add(colsum, f4_add(staged<MASK_B>(sb1, 0), staged<MASK_B>(sb1, 1)));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) { ga.issue(2, j, sa0); gb.issue(2, j, sb0); }
    __syncthreads();
    float fa0[BK / 2], fb0[BK / 2], fa1[BK / 2], fb1[BK / 2];
#pragma unroll
    for (int k = 0; k < BK / 2; k += 2) {
        read_frag_pair<A_RC>(As, am, hi, k, fa0);
        read_frag_pair<B_RC>(Bs, bn, hi, k, fb0);
    }
    int s1 = 1, s2 = 2;                                   // ring slots of chunks c+1, c+2
    // cur: staged registers of chunk c+2 (loaded in the previous iteration); nxt: receives chunk c+3
    auto step = [&](int c, float (&fa)[BK / 2], float (&fb)[BK / 2], float (&na)[BK / 2], float (&nb)[BK / 2],
                    StageRegs<MASK_A>& curA, StageRegs<MASK_B>& curB, StageRegs<MASK_A>& nxtA, StageRegs<MASK_B>& nxtB) {
        const float* rA = As + s1 * kBufFloats;
        const float* rB = Bs + s1 * kBufFloats;
        float* wA = As + s2 * kBufFloats;
        float* wB = Bs + s2 * kBufFloats;
#pragma unroll
        for (int k = 0; k < BK / 2; ++k) {
            if (k & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k], fb[k], acc1, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[k], fb[k], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // side work of the iteration, spread so that no MFMA shadow is overfull and the last two are free (the
            // LDS traffic has drained by the time the barrier is reached)
            if (k < 2) ga.issue(c + 3, k, nxtA);
            else if (k < 4) gb.issue(c + 3, k - 2, nxtB);
            if (k < 8) {
                read_frag_pair<A_RC>(rA, am, hi, 2 * k, na);
                read_frag_pair<B_RC>(rB, bn, hi, 2 * k, nb);
            }
            constexpr int nA = A_RC ? 4 : 2, nB = B_RC ? 4 : 2, w0 = 14 - nA - nB;       // stores end at k = 13
            if (k >= w0 && k < w0 + nA) stage_store_piece<A_RC, MASK_A>(wA, curA, k - w0);
            if (k >= w0 + nA && k < w0 + nA + nB) stage_store_piece<B_RC, MASK_B>(wB, curB, k - w0 - nA);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (COLSUM) {
            // chunk c+2; staged tiles past the end of the reduction hold garbage (EXACT) and must not be counted
            if (c + 2 < nchunks) colsum = f4_add(colsum, f4_add(staged<MASK_B>(curB, 0), staged<MASK_B>(curB, 1)));
        }
        s1 = s2;
        s2 = s2 + 1 == kStages ? 0 : s2 + 1;
        __syncthreads();
    };
    for (int c = 0; c < nchunks; c += 2) {
        step(c, fa0, fb0, fa1, fb1, sa0, sb0, sa1, sb1);
        if (c + 1 < nchunks) step(c + 1, fa1, fb1, fa0, fb0, sa1, sb1, sa0, sb0);
    }
}

template <bool A_RC, bool B_RC, bool FAST, bool MASK_A, bool MASK_B, bool COLSUM>
__device__ __forceinline__ void tile_mainloop(const Segment& sg, int m0, int n0, int M, int N, int red_begin, int red_end,
                                              float* __restrict__ As, float* __restrict__ Bs, f32x16& acc, f32x16& acc1, float4& colsum) {
    // whole chunks only (and, for a batch split, split boundaries on chunk boundaries): wave-uniform
    const bool exact = FAST && ((red_end - red_begin) % BK == 0) && (red_begin % BK == 0);
    if constexpr (FAST) {
        // round-5 main loop (tile_v2.h), one sub-tile per wave: same workgroup tile, same wave -> (row, column) map, same
        // staging thread -> column map (COLSUM) as the round-2 loop below, so every epilogue of this file serves both
        static_assert(tv2::Geom<true, 1>::kFloats <= kBufFloats && tv2::Geom<false, 1>::kFloats <= kBufFloats, "ring slot size");
        const tv2::Operand a{sg.a.p, sg.a.mask, sg.a.ld, sg.a.bytes}, b{sg.b.p, sg.b.mask, sg.b.ld, sg.b.bytes};
        // (two chains per element, as in the round-2 loop: acc takes the even groups of four steps, acc1 the odd ones)
        tv2::f32x16 (&acc2)[1][1] = *reinterpret_cast<tv2::f32x16 (*)[1][1]>(&acc);
        tv2::f32x16 (*accb)[1] = reinterpret_cast<tv2::f32x16 (*)[1]>(&acc1);
        if (exact) tv2::mainloop<A_RC, B_RC, 1, 1, MASK_A, MASK_B, COLSUM, true, true>(a, b, m0, n0, M, N, red_begin, red_end, As, Bs, acc2, colsum, accb);
        else tv2::mainloop<A_RC, B_RC, 1, 1, MASK_A, MASK_B, COLSUM, false, true>(a, b, m0, n0, M, N, red_begin, red_end, As, Bs, acc2, colsum, accb);
    } else {
        tile_mainloop_impl<A_RC, B_RC, FAST, MASK_A, MASK_B, COLSUM, false>(sg, m0, n0, M, N, red_begin, red_end, As, Bs, acc, acc1, colsum);
    }
}

// XCD-aware linear block -> (tile_m, tile_n): block b runs on XCD b % 8; give every XCD a contiguous
// range of the m-major tile order (all n-tiles of an m-tile land on one XCD).
__device__ __forceinline__ int xcd_swizzle(int b, int total) {
    if (total % 8) return b;
    return (b % 8) * (total / 8) + b / 8;
}

struct FwdArgs {
    Segment seg[2];
    int nseg;
    const float* bias;     // [N] or null
    int relu;
    float* y;
    int ldy;
    int M, N;
    // optional: the BatchNorm layer that consumes y (tf.layers.dense -> tf.layers.batch_normalization, deepfm.py:207-211) gets
    // the batch moments of this workgroup's 64-row tile from the epilogue — [tile][0:N] = mean, [tile][N:2N] = sum of squared
    // deviations from it, the partial-row layout of recalgo_batchnorm_moments — instead of a pass of its own over y
    float* bn_partials;
    // optional, with bn_partials: the per-channel activation between the two (tf.layers.dense -> dice | prelu ->
    // tf.layers.batch_normalization, din.py:262-266): z = x W + b goes to `z` (the activation's backward needs it), y and the
    // moments are those of act(z)
    const float* act_alpha;    // [N]
    int act_kind;              // 0: none, 1 + RECALGO_ACT_PRELU, 1 + RECALGO_ACT_DICE
    float* z;                  // [M][ldy]
    int vec_store;             // y is float4-addressable (base, ldy, N): the plain epilogue writes whole row segments
    // optional (ldy == N): the tf.layers.dropout that follows the layer (tf.layers.dense(relu) -> tf.layers.dropout [-> batch_normalization],
    // deepfm.py:207-211): y := y * keep / (1 - rate) before the store and the moments (csrc/dropout.h; element index row * N + col)
    recalgo_drop::Spec drop;
};

template <bool FAST>
__global__ __launch_bounds__(kThreads) void dense_fwd_kernel(FwdArgs P) {
    __shared__ __attribute__((aligned(16))) float As[kStages * kBufFloats];
    __shared__ __attribute__((aligned(16))) float Bs[kStages * kBufFloats];
    const int tn_count = (P.N + BN - 1) / BN;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int m0 = (tile / tn_count) * BM, n0 = (tile % tn_count) * BN;
    f32x16 acc, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = 0.f;
    float4 unused = f4_zero();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, l32 = lane & 31;
    // (row-wise epilogue: the lane's four bias values are requested before the main loop, not behind it)
    const bool rowwise = FAST && P.vec_store && P.bn_partials == nullptr;
    const int c4 = n0 + (wave & 1) * 32 + (lane & 7) * 4;
    float4 bias4 = f4_zero();
    if (rowwise && P.bias && c4 < P.N) bias4 = *reinterpret_cast<const float4*>(P.bias + c4);
    for (int s = 0; s < P.nseg; ++s)
        tile_mainloop<true, false, FAST, false, false, false>(P.seg[s], m0, n0, P.M, P.N, 0, P.seg[s].n_red, As, Bs, acc, acc1, unused);
    acc += acc1;
    const int col = n0 + (wave & 1) * 32 + l32;
    const bool dropping = recalgo_drop::enabled(P.drop);
    const recalgo_drop::Key dkey = dropping ? recalgo_drop::make_key(P.drop) : recalgo_drop::Key{0u, 0u};
    if (rowwise) {
        const int r0 = m0 + (wave >> 1) * 32;
        tv2::tile_rows(As + wave * tv2::kTileScratch, acc, [&](int, int row, int, float4 v) {
            v = f4_add(v, bias4);
            if (P.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            if (r0 + row < P.M && c4 < P.N) {
                if (dropping) {
                    const float4 f = recalgo_drop::factor4(P.drop, dkey, (uint32_t)(r0 + row) * (uint32_t)P.N + (uint32_t)c4);
                    v = make_float4(v.x * f.x, v.y * f.y, v.z * f.z, v.w * f.w);
                }
                *reinterpret_cast<float4*>(P.y + (size_t)(r0 + row) * P.ldy + c4) = v;
            }
        });
        return;
    }
    if (P.bn_partials == nullptr) {
        if (col >= P.N) return;
        const float bv = P.bias ? P.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (row < P.M) {
                float v = acc[r] + bv;
                if (P.relu) v = fmaxf(v, 0.f);
                if (dropping) v *= recalgo_drop::factor(P.drop, dkey, (uint32_t)row * (uint32_t)P.N + (uint32_t)col);
                P.y[(size_t)row * P.ldy + col] = v;
            }
        }
        return;
    }
    // ---- the same store + the tile's column moments (two-pass inside the tile: mean first, then deviations) ----------
    const bool cok = col < P.N;
    const bool rows_y = FAST && P.vec_store;
    const float bv = (cok && P.bias) ? P.bias[col] : 0.f;
    const float av = (cok && P.act_kind) ? P.act_alpha[col] : 0.f;
    float vals[16];
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        float v = acc[r] + bv;
        if (P.relu) v = fmaxf(v, 0.f);
        const bool ok = cok && row < P.M;
        if (P.act_kind) {
            if (ok) P.z[(size_t)row * P.ldy + col] = v;
            v = P.act_kind == 1 + RECALGO_ACT_DICE ? recalgo_act::dice(v, av) : recalgo_act::prelu(v, av);
        }
        if (dropping && ok) v *= recalgo_drop::factor(P.drop, dkey, (uint32_t)row * (uint32_t)P.N + (uint32_t)col);
        vals[r] = ok ? v : 0.f;
        if (ok && !rows_y) P.y[(size_t)row * P.ldy + col] = v;
        s += vals[r];
    }
    if (rows_y) {
        // y leaves as whole row segments (the wave's scratch lies in Bs: the moment sums below go through As)
        f32x16 yv;
#pragma unroll
        for (int r = 0; r < 16; ++r) yv[r] = vals[r];
        const int r0 = m0 + (wave >> 1) * 32;
        tv2::tile_rows(Bs + wave * tv2::kTileScratch, yv, [&](int, int row, int, float4 v) {
            if (r0 + row < P.M && c4 < P.N) *reinterpret_cast<float4*>(P.y + (size_t)(r0 + row) * P.ldy + c4) = v;
        });
    }
    s += __shfl_xor(s, 32, 64);                               // the wave's 32 rows of this column
    float* red = As;                                          // [2 row halves][64 columns] (the operand ring is free)
    __syncthreads();
    if (hi == 0) red[(wave >> 1) * 64 + (wave & 1) * 32 + l32] = s;
    __syncthreads();
    const int cl = (wave & 1) * 32 + l32;
    const int nrows = min(P.M - m0, BM);
    const float mean = (red[cl] + red[64 + cl]) * (1.0f / (float)nrows);
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float dlt = vals[r] - mean;
        if (cok && row < P.M) q = fmaf(dlt, dlt, q);
    }
    q += __shfl_xor(q, 32, 64);
    __syncthreads();
    if (hi == 0) red[(wave >> 1) * 64 + cl] = q;
    __syncthreads();
    if ((wave >> 1) == 0 && hi == 0 && cok) {
        float* prow = P.bn_partials + (size_t)(m0 / BM) * 2 * P.N;
        prow[col] = mean;
        prow[P.N + col] = red[cl] + red[64 + cl];
    }
}

struct DgradArgs {
    Segment seg;           // a = G (+mask) [M][N] red-contiguous, b = W [K][N] red-contiguous, n_red = N
    const float* c_in;     // [M][ldc] or null: dx += beta * c_in
    float beta;
    int ldc;
    float* dx;
    int lddx;
    int M, K;              // output is [M][K]
    int accumulate;        // dx += result (the second operand pair of the PNN layer)
    int tiles_per_block;   // consecutive output tiles one workgroup computes (>= 1; see bwd_balance)
    int vec_store;         // dx (and c_in) float4-addressable: the plain epilogue writes whole row segments
    // optional: dx := dx_mask > 0 ? dx : 0, dx_mask [M][ld_mask] — the ReLU output this layer's input IS (tf.layers.dense(...,
    // relu) -> tf.layers.dense, dcn.py:163-166): the producing layer's backward then gets its gradient already masked and
    // stages it without mask loads (3 us of a 24 us half-launch, profiles/r05_mfma_lab.md "mask cost")
    const float* dx_mask;
    int ld_mask;
    // optional: this layer's input is the output of a training-mode BatchNorm over bn_x [M][K] (contiguous) with the batch
    // statistics bn_mean / bn_rstd [K] — the epilogue leaves the sums that BatchNorm's backward starts with, per 64-row tile:
    // bn_partials[tile][0:K] = colsum(dx), [K:2K] = colsum(dx * xhat) (the partial rows of recalgo_batchnorm_bwd_sums)
    const float* bn_x;
    const float* bn_mean;
    const float* bn_rstd;
    float* bn_partials;
};

template <bool FAST, bool MASK>
__device__ __forceinline__ void dgrad_tile(const DgradArgs& P, int block, int nblocks, float* As, float* Bs) {
    const int tn_count = (P.K + BN - 1) / BN;
    const int total = ((P.M + BM - 1) / BM) * tn_count;
    const int first = xcd_swizzle(block, nblocks) * P.tiles_per_block;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, l32 = lane & 31;
    for (int t = 0; t < P.tiles_per_block; ++t) {
        const int tile = first + t;                       // consecutive tiles: same gradient rows, adjacent columns
        if (tile >= total) break;                         // (workgroup-uniform)
        const int m0 = (tile / tn_count) * BM, n0 = (tile % tn_count) * BN;
        f32x16 acc, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = 0.f;
        float4 unused = f4_zero();
        const int col = n0 + (wave & 1) * 32 + l32;
        // (BatchNorm sums: the tile of bn_x this thread will need in the epilogue is requested BEFORE the main loop — the loads
        // complete in its shadow instead of extending the workgroup's critical path by a round trip to memory)
        float xb[16];
        float mu = 0.f, rs = 0.f;
        if (P.bn_partials != nullptr && col < P.K) {
            mu = P.bn_mean[col];
            rs = P.bn_rstd[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                xb[r] = row < P.M ? P.bn_x[(size_t)row * P.K + col] : 0.f;
            }
        }
        // (row-wise epilogue with an output mask: the lane's four float4 of the mask are requested before the main loop)
        const bool rowwise = FAST && P.vec_store && P.bn_partials == nullptr;
        float4 mk[4];
        if (rowwise && P.dx_mask != nullptr) {
            const int r0 = m0 + (wave >> 1) * 32 + (lane >> 3), c4 = n0 + (wave & 1) * 32 + (lane & 7) * 4;
#pragma unroll
            for (int it = 0; it < 4; ++it)
                mk[it] = (r0 + 8 * it < P.M && c4 < P.K) ? *reinterpret_cast<const float4*>(P.dx_mask + (size_t)(r0 + 8 * it) * P.ld_mask + c4)
                                                         : f4_zero();
        }
        tile_mainloop<true, true, FAST, MASK, false, false>(P.seg, m0, n0, P.M, P.K, 0, P.seg.n_red, As, Bs, acc, acc1, unused);
        acc += acc1;
        if (P.bn_partials == nullptr) {
            if (rowwise) {
                const int r0 = m0 + (wave >> 1) * 32, c4 = n0 + (wave & 1) * 32 + (lane & 7) * 4;
                tv2::tile_rows(As + wave * tv2::kTileScratch, acc, [&](int it, int row, int, float4 v) {
                    if (r0 + row < P.M && c4 < P.K) {
                        if (P.dx_mask) v = relu_mask(v, mk[it]);
                        if (P.c_in) v = f4_fma(*reinterpret_cast<const float4*>(P.c_in + (size_t)(r0 + row) * P.ldc + c4), P.beta, v);
                        float4* o = reinterpret_cast<float4*>(P.dx + (size_t)(r0 + row) * P.lddx + c4);
                        *o = P.accumulate ? f4_add(*o, v) : v;
                    }
                });
                continue;                                         // (the next tile's main loop starts with a barrier)
            }
            if (col < P.K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (row < P.M) {
                        float v = acc[r];
                        if (P.dx_mask && !(P.dx_mask[(size_t)row * P.ld_mask + col] > 0.f)) v = 0.f;
                        if (P.c_in) v = fmaf(P.beta, P.c_in[(size_t)row * P.ldc + col], v);
                        float* o = P.dx + (size_t)row * P.lddx + col;
                        *o = P.accumulate ? *o + v : v;
                    }
                }
            }
            continue;
        }
        // ---- the same store + the tile's column sums of dx and dx * xhat (fixed order: a lane's 16 rows, the wave's two row
        //      halves, the workgroup's two wave rows) ----
        const bool cok = col < P.K;
        const bool rows_dx = FAST && P.vec_store;
        f32x16 dxv;
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            float v = 0.f;
            if (cok && row < P.M) {
                v = acc[r];
                if (P.dx_mask && !(P.dx_mask[(size_t)row * P.ld_mask + col] > 0.f)) v = 0.f;
                if (P.c_in) v = fmaf(P.beta, P.c_in[(size_t)row * P.ldc + col], v);
                if (!rows_dx) P.dx[(size_t)row * P.lddx + col] = v;
                const float xh = (xb[r] - mu) * rs;
                sg += v;
                sgx = fmaf(v, xh, sgx);
            }
            dxv[r] = v;
        }
        if (rows_dx) {
            const int r0 = m0 + (wave >> 1) * 32, c4 = n0 + (wave & 1) * 32 + (lane & 7) * 4;
            tv2::tile_rows(Bs + wave * tv2::kTileScratch, dxv, [&](int, int row, int, float4 v) {
                if (r0 + row < P.M && c4 < P.K) *reinterpret_cast<float4*>(P.dx + (size_t)(r0 + row) * P.lddx + c4) = v;
            });
        }
        sg += __shfl_xor(sg, 32, 64);
        sgx += __shfl_xor(sgx, 32, 64);
        float* red = As;                                      // [2 sums][2 wave rows][64 columns] (the operand ring is free)
        __syncthreads();
        const int cl = (wave & 1) * 32 + l32;
        if (hi == 0) {
            red[(wave >> 1) * 64 + cl] = sg;
            red[128 + (wave >> 1) * 64 + cl] = sgx;
        }
        __syncthreads();
        if ((wave >> 1) == 0 && hi == 0 && cok) {
            float* prow = P.bn_partials + (size_t)(m0 / BM) * 2 * P.K;
            prow[col] = red[cl] + red[64 + cl];
            prow[P.K + col] = red[128 + cl] + red[192 + cl];
        }
        __syncthreads();                                      // (the next tile's staging reuses the ring)
    }
}

template <bool FAST, bool MASK>
__global__ __launch_bounds__(kThreads) void dense_dgrad_kernel(DgradArgs P) {
    __shared__ __attribute__((aligned(16))) float As[kStages * kBufFloats];
    __shared__ __attribute__((aligned(16))) float Bs[kStages * kBufFloats];
    dgrad_tile<FAST, MASK>(P, blockIdx.x, gridDim.x, As, Bs);
}

struct WgradArgs {
    Segment seg;           // a = X [M][K] red-major (red = m), b = G (+mask) [M][N] red-major, n_red = M
    int K, N;              // output [K][N]
    int splits, rows_per_split;
    float* out;            // splits == 1: dW itself (ld N); else partials [splits][K*N + Npad]
    float* dbias_out;      // splits == 1: dbias or null
    int want_dbias;
    size_t slab;           // floats per split slab (K*N + N rounded)
    int vec_store;         // the output (slab or dW) is float4-addressable
};

template <bool FAST, bool MASK>
__device__ __forceinline__ void wgrad_tile(const WgradArgs& P, int block, float* As, float* Bs) {
    const int tn_count = (P.N + BN - 1) / BN, tm_count = (P.K + BM - 1) / BM;
    const int ntiles = tn_count * tm_count;
    // XCD placement: all tiles of one split read the same rows of X and G -> keep a split on one XCD
    int split, tile;
    if (P.splits % 8 == 0) {
        const int b = block, x = b % 8, j = b / 8;
        split = x + 8 * (j / ntiles);
        tile = j % ntiles;
    } else {
        // any other split count: every XCD takes a contiguous range of the split-major (split, tile) list, so the rows of a
        // split are read by one or two XCDs instead of all eight
        const int l = xcd_swizzle(block, ntiles * P.splits);
        split = l / ntiles;
        tile = l % ntiles;
    }
    const int m0 = (tile / tn_count) * BM, n0 = (tile % tn_count) * BN;
    const int r_begin = split * P.rows_per_split;
    const int r_end = min(P.seg.n_red, r_begin + P.rows_per_split);
    f32x16 acc, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = 0.f;
    float4 colsum = f4_zero();
    const bool do_bias = P.want_dbias && m0 == 0;
    if (do_bias) tile_mainloop<false, false, FAST, false, MASK, true>(P.seg, m0, n0, P.K, P.N, r_begin, r_end, As, Bs, acc, acc1, colsum);
    else tile_mainloop<false, false, FAST, false, MASK, false>(P.seg, m0, n0, P.K, P.N, r_begin, r_end, As, Bs, acc, acc1, colsum);
    acc += acc1;
    float* base = P.out + (size_t)split * P.slab;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, l32 = lane & 31;
    const int col = n0 + (wave & 1) * 32 + l32;
    if (FAST && P.vec_store) {
        const int r0 = m0 + (wave >> 1) * 32, c4 = n0 + (wave & 1) * 32 + (lane & 7) * 4;
        // (the wave's scratch lies in Bs: the bias sums below go through As)
        tv2::tile_rows(Bs + wave * tv2::kTileScratch, acc, [&](int, int row, int, float4 v) {
            if (r0 + row < P.K && c4 < P.N) *reinterpret_cast<float4*>(base + (size_t)(r0 + row) * P.N + c4) = v;
        });
    } else if (col < P.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (row < P.K) base[(size_t)row * P.N + col] = acc[r];
        }
    }
    if (do_bias) {
        // thread t staged rows (t >> 4) + 16 j, columns (t & 15) * 4 .. + 3: reduce the 16 row lanes in fixed order
        float4* sh = reinterpret_cast<float4*>(As);
        __syncthreads();
        sh[threadIdx.x] = colsum;
        __syncthreads();
        if (threadIdx.x < 16) {
            float4 t = sh[threadIdx.x];
#pragma unroll
            for (int k = 1; k < 16; ++k) t = f4_add(t, sh[k * 16 + threadIdx.x]);
            float* db = (P.splits == 1 ? P.dbias_out : base + (size_t)P.K * P.N);
            const int c = n0 + threadIdx.x * 4;
            if (c + 0 < P.N) db[c + 0] = t.x;
            if (c + 1 < P.N) db[c + 1] = t.y;
            if (c + 2 < P.N) db[c + 2] = t.z;
            if (c + 3 < P.N) db[c + 3] = t.w;
        }
    }
}

template <bool FAST, bool MASK>
__global__ __launch_bounds__(kThreads) void dense_wgrad_kernel(WgradArgs P) {
    __shared__ __attribute__((aligned(16))) float As[kStages * kBufFloats];
    __shared__ __attribute__((aligned(16))) float Bs[kStages * kBufFloats];
    wgrad_tile<FAST, MASK>(P, blockIdx.x, As, Bs);
}

// Both gradients of a layer in ONE launch: workgroups [0, dgrad_blocks) compute input-gradient tiles (the next
// layer's backward waits for them), the rest weight-gradient tiles.  The two GEMMs share nothing but their inputs;
// what the merge buys is one launch ramp / drain / boundary instead of two (each ~6 us per layer at these sizes,
// scripts/bench_dense_k.py) and a fuller chip for the small layers (256 + 256 workgroups for [4096 x 256 x 128]).
template <bool FAST, bool MASK>
__global__ __launch_bounds__(kThreads) void dense_bwd_kernel(DgradArgs D, WgradArgs W, int dgrad_blocks) {
    __shared__ __attribute__((aligned(16))) float As[kStages * kBufFloats];
    __shared__ __attribute__((aligned(16))) float Bs[kStages * kBufFloats];
    if ((int)blockIdx.x < dgrad_blocks) dgrad_tile<FAST, MASK>(D, blockIdx.x, dgrad_blocks, As, Bs);
    else wgrad_tile<FAST, MASK>(W, blockIdx.x - dgrad_blocks, As, Bs);
}

// The same launch carrying RIDERS: work of OTHER layers over the same batch whose operands are ready and which nothing in this
// launch depends on —
//   * the weight-gradient tiles of the layer above, when its input gradient came out of a fused kernel (csrc/tailfuse.hip):
//     0.27 GFLOP that would otherwise be a 7 us launch of their own.  Rider tiles never stage a mask (their gradient arrives masked);
//   * the CrossNet backward (csrc/cross_bwd.h; dcn.py:157-160: the cross branch's gradient is ready as soon as the head's is):
//     an HBM / latency-bound kernel of 11 us beside this MFMA-bound one, 4 waves per workgroup here instead of 8.
// Riders are dispatched first: they share the CUs with the first round of the layer's own tiles instead of forming a round of
// their own behind them.  (A second HIP stream inside the captured graph costs more than it hides: scripts/lab_fork_join.py,
// 39.8 us forked against 32.0 in series for exactly this pair.)
struct CrossRider {
    const float* x0; const float4* w; const float4* b; const float* g;
    float* dx0; float* partials;
    unsigned x_stride, g_stride, B, d4, blocks;
};

template <bool MASK, int NV, int L>
__global__ __launch_bounds__(kThreads) void dense_bwd_rider_kernel(DgradArgs D, WgradArgs W, WgradArgs R, CrossRider C, int dgrad_blocks,
                                                                   int rider_blocks) {
    __shared__ __attribute__((aligned(16))) float As[kStages * kBufFloats];
    __shared__ __attribute__((aligned(16))) float Bs[kStages * kBufFloats];
    int b = (int)blockIdx.x;
    if (NV > 0) {
        if (b < (int)C.blocks) {
            static_assert(kThreads / 64 * 512 + (kThreads / 64 + 1) * 6 <= kStages * kBufFloats, "the cross rider's scratch lies in As");
            recalgo_cross::cross_stack_bwd_block<(NV > 0 ? NV : 1), (L > 0 ? L : 1), kThreads / 64>(
                C.x0, C.x_stride, C.w, C.b, C.g, C.g_stride, nullptr, C.B, C.d4, C.dx0, C.partials, (unsigned)b, C.blocks, As);
            return;
        }
        b -= (int)C.blocks;
    }
    b -= rider_blocks;
    if (b < 0) wgrad_tile<true, false>(R, b + rider_blocks, As, Bs);
    else if (b < dgrad_blocks) dgrad_tile<true, MASK>(D, b, dgrad_blocks, As, Bs);
    else wgrad_tile<true, MASK>(W, b - dgrad_blocks, As, Bs);
}

// fixed-order sum of split slabs, batched over up to kMaxSplitJobs weight gradients (one launch for all the layers
// of a backward pass): out[i] = sum_s partials[s][i]; elements [0, n0) -> out0 (dW), [n0, n) -> out1 (dbias).
// (Tried instead: letting the last-arriving workgroup of each tile do the sum inside the wgrad kernel, with the
// agent-scope release / ticket / acquire hand-off — correct, but the per-workgroup L2 write-back made the
// [4096 x 256 x 128] wgrad 56 us instead of 18 + 7; the partials of one launch stay cheap only across a kernel boundary.)
constexpr int kMaxSplitJobs = 32;
struct SplitJob {
    const float* partials;
    float* out0;
    float* out1;
    unsigned long long slab, n0, n;
    int S, vec;                // vec: 1 float4 per thread, 0 one float per thread, 2 "tall" (S >> n): 16 columns x 16 row
                               //      groups per workgroup (a thread walking hundreds of partial rows alone is latency bound)
    unsigned first_block;      // blocks [first_block, next job's first_block) belong to this job
};
struct SplitJobs {
    SplitJob job[kMaxSplitJobs];
    int n_jobs;
    long long* step;           // optional: the optimizer's step counter, advanced by this launch (it runs once per
                               // step, right before recalgo_adam_tf1_step, which then only READS the counter)
};

__global__ __launch_bounds__(256) void dense_sum_slabs_kernel(SplitJobs J) {
    if (J.step != nullptr && blockIdx.x == 0 && threadIdx.x == 0) J.step[0] += 1;
    if (J.n_jobs == 0) return;
    int j = 0;
#pragma unroll
    for (int k = 1; k < kMaxSplitJobs; ++k)
        if (k < J.n_jobs && blockIdx.x >= J.job[k].first_block) j = k;
    const SplitJob& jb = J.job[j];
    const size_t t = (size_t)(blockIdx.x - jb.first_block) * 256 + threadIdx.x;
    if (jb.vec == 2) {
        __shared__ float sh[16][17];
        const unsigned cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
        const size_t c = (size_t)(blockIdx.x - jb.first_block) * 16 + cl;
        float acc = 0.f;
        if (c < jb.n) {
            // the partial rows of a column are independent loads: eight in flight per thread, added in row order (a loop of
            // load -> wait -> add over the 512 partial rows of the loss tail was most of this launch: ~13 us in the step)
            const float* col = jb.partials + c;
            int r = rg;
            for (; r + 16 * 7 < jb.S; r += 16 * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = col[(size_t)(r + 16 * u) * jb.slab];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
            for (; r < jb.S; r += 16) acc += col[(size_t)r * jb.slab];
        }
        sh[rg][cl] = acc;
        __syncthreads();
        if (rg == 0 && c < jb.n) {
            float tt = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) tt += sh[g][cl];
            jb.out0[c] = tt;
        }
        return;
    }
    if (jb.vec) {
        const size_t i = t * 4;
        if (i >= jb.n) return;
        // the slab loads are independent: four in flight per thread, added in slab order (the sum's order is fixed)
        const float* base = jb.partials + i;
        float4 acc = *reinterpret_cast<const float4*>(base);
        int s = 1;
        for (; s + 4 <= jb.S; s += 4) {
            const float4 a0 = *reinterpret_cast<const float4*>(base + (size_t)s * jb.slab);
            const float4 a1 = *reinterpret_cast<const float4*>(base + (size_t)(s + 1) * jb.slab);
            const float4 a2 = *reinterpret_cast<const float4*>(base + (size_t)(s + 2) * jb.slab);
            const float4 a3 = *reinterpret_cast<const float4*>(base + (size_t)(s + 3) * jb.slab);
            acc = f4_add(f4_add(f4_add(f4_add(acc, a0), a1), a2), a3);
        }
        for (; s < jb.S; ++s) acc = f4_add(acc, *reinterpret_cast<const float4*>(base + (size_t)s * jb.slab));
        *reinterpret_cast<float4*>(i < jb.n0 ? jb.out0 + i : jb.out1 + (i - jb.n0)) = acc;
    } else {
        if (t >= jb.n) return;
        float acc = jb.partials[t];
        for (int s = 1; s < jb.S; ++s) acc += jb.partials[(size_t)s * jb.slab + t];
        *(t < jb.n0 ? jb.out0 + t : jb.out1 + (t - jb.n0)) = acc;
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int vec_ok(const float* p, int ld) { return (aligned16(p) && ld % 4 == 0) ? 1 : 0; }
// rows x cols: the extent of the matrix in memory (row-major, leading dimension ld)
inline Operand operand(const float* p, const float* mask, int ld, int64_t rows, int64_t cols) {
    const int64_t bytes = rows > 0 ? ((rows - 1) * ld + cols) * (int64_t)sizeof(float) : 0;
    return Operand{p, mask, ld, (vec_ok(p, ld) && (mask == nullptr || aligned16(mask))) ? 1 : 0,
                   bytes < (1ll << 31) - (1ll << 24) ? (int)bytes : 0};
}

// FAST-path predicates: the contiguous extent of the operand must be float4-addressable
inline bool fast_rc(const Operand& o, int n_red) { return o.p == nullptr || (o.vec && o.bytes > 0 && n_red % 4 == 0); }
inline bool fast_rm(const Operand& o, int n_idx) { return o.p == nullptr || (o.vec && o.bytes > 0 && n_idx % 4 == 0); }

// Balance of a layer's backward launch (recalgo_dense_bwd: input- and weight-gradient tiles in one grid).  The tile engine
// needs 122-162 VGPRs + 32 AGPRs: two workgroups (8 waves) are resident per CU, 512 on the chip.  A grid of 896 equal
// workgroups (the first version: 448 dgrad tiles + 56 wgrad tiles x 8 batch splits for 416 -> 512) therefore runs as two
// rounds, the second 3/4 full, and pays the launch ramp + pipeline fill + epilogue (~6 us) twice: 42 us for 22 us of
// matrix work.  Instead the grid is sized to ONE resident round with equal work per workgroup: u = total chunk-tiles /
// resident; a dgrad workgroup takes round(u / chunks per tile) consecutive output tiles, the weight gradient is split over
// the batch into slabs of ~u chunks.
constexpr int kResidentBlocks = 512;

struct BwdBalance {
    int tiles_per_block;   // dgrad
    int splits;            // wgrad
};
inline BwdBalance bwd_balance(int M, int K, int N) {
    const int gd = cdiv(M, BM) * cdiv(K, BN), cd = cdiv(N, BK);           // dgrad: tiles, chunks per tile
    const int tw = cdiv(K, BM) * cdiv(N, BN), cw = cdiv(M, BK);           // wgrad: tiles, chunks per tile over the whole batch
    const int max_s = cdiv(M, 4 * BK) < 1 ? 1 : cdiv(M, 4 * BK);         // at least four chunks per split
    BwdBalance b{1, 1};
    // chunk-tiles per workgroup for one resident round; a layer too large for one round (AFM's attention net: 1.3 M rows)
    // runs several rounds of workgroups of at most 64 chunks
    double u = ((double)gd * cd + (double)tw * cw) / kResidentBlocks;
    if (u > 64.0) u = 64.0;
    if (u < 1.0) u = 1.0;
    const int tpb = (int)(u / cd + 0.5);
    b.tiles_per_block = tpb < 1 ? 1 : (tpb > 8 ? 8 : tpb);
    int s = (int)(cw / u + 0.5);                        // every batch split of a weight-gradient tile: ~u chunks
    if (s > max_s) s = max_s;
    if (s >= 16) s = s / 8 * 8;                         // many splits: whole XCD groups (a split's rows stay in one XCD's L2)
    b.splits = s < 1 ? 1 : s;
    return b;
}
inline int wgrad_splits(int M, int K, int N) { return bwd_balance(M, K, N).splits; }
inline size_t wgrad_slab(int K, int N) { return ((size_t)K * N + N + 3) / 4 * 4; }

}  // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
RECALGO_EXPORT int recalgo_dense_fwd(const float* x, int ldx, const float* w, int K, const float* x2, int ldx2,
                                     const float* w2, int K2, const float* bias, int M, int N, int relu, float* y,
                                     int ldy, recalgo_stream_t stream) {
    return recalgo_dense_fwd_bn(x, ldx, w, K, x2, ldx2, w2, K2, bias, M, N, relu, y, ldy, nullptr, stream);
}

RECALGO_EXPORT int recalgo_dense_fwd_bn(const float* x, int ldx, const float* w, int K, const float* x2, int ldx2,
                                        const float* w2, int K2, const float* bias, int M, int N, int relu, float* y,
                                        int ldy, float* bn_partials, recalgo_stream_t stream) {
    return recalgo_dense_fwd_act_bn(x, ldx, w, K, x2, ldx2, w2, K2, bias, M, N, relu, RECALGO_ACT_NONE, nullptr, nullptr, y, ldy,
                                    bn_partials, stream);
}

static int dense_fwd_impl(const float* x, int ldx, const float* w, int K, const float* x2, int ldx2, const float* w2, int K2,
                          const float* bias, int M, int N, int relu, int act_kind, const float* act_alpha, float* z, float* y, int ldy,
                          float* bn_partials, const recalgo_dropout_t* drop, recalgo_stream_t stream);

RECALGO_EXPORT int recalgo_dense_fwd_act_bn(const float* x, int ldx, const float* w, int K, const float* x2, int ldx2,
                                            const float* w2, int K2, const float* bias, int M, int N, int relu, int act_kind,
                                            const float* act_alpha, float* z, float* y, int ldy, float* bn_partials,
                                            recalgo_stream_t stream) {
    return dense_fwd_impl(x, ldx, w, K, x2, ldx2, w2, K2, bias, M, N, relu, act_kind, act_alpha, z, y, ldy, bn_partials, nullptr, stream);
}

RECALGO_EXPORT int recalgo_dense_fwd_drop(const float* x, int ldx, const float* w, int K, const float* x2, int ldx2,
                                          const float* w2, int K2, const float* bias, int M, int N, int relu, float* y, int ldy,
                                          float* bn_partials, const recalgo_dropout_t* drop, recalgo_stream_t stream) {
    RECALGO_REQUIRE(drop == nullptr || (ldy == N && (int64_t)M * N < ((int64_t)1 << 32) && recalgo_drop::abi_ok(drop)));
    return dense_fwd_impl(x, ldx, w, K, x2, ldx2, w2, K2, bias, M, N, relu, RECALGO_ACT_NONE, nullptr, nullptr, y, ldy, bn_partials, drop, stream);
}

static int dense_fwd_impl(const float* x, int ldx, const float* w, int K, const float* x2, int ldx2, const float* w2, int K2,
                          const float* bias, int M, int N, int relu, int act_kind, const float* act_alpha, float* z, float* y, int ldy,
                          float* bn_partials, const recalgo_dropout_t* drop, recalgo_stream_t stream) {
    RECALGO_REQUIRE(M >= 0 && N > 0 && K > 0 && y != nullptr && ldy >= N);
    RECALGO_REQUIRE(act_kind == RECALGO_ACT_NONE || ((act_kind == RECALGO_ACT_PRELU || act_kind == RECALGO_ACT_DICE) &&
                                                     act_alpha != nullptr && z != nullptr && bn_partials != nullptr && !relu));
    RECALGO_REQUIRE(x != nullptr && ldx >= K && w != nullptr);
    RECALGO_REQUIRE(x2 == nullptr || (ldx2 >= K2 && K2 > 0 && w2 != nullptr));
    if (M == 0) return 0;
    FwdArgs P;
    P.seg[0] = Segment{operand(x, nullptr, ldx, M, K), operand(w, nullptr, N, K, N), K};
    P.seg[1] = Segment{operand(x2, nullptr, ldx2, M, K2), operand(w2, nullptr, N, K2, N), K2};
    P.nseg = x2 ? 2 : 1;
    P.bias = bias; P.relu = relu; P.y = y; P.ldy = ldy; P.M = M; P.N = N;
    P.bn_partials = bn_partials;
    P.act_alpha = act_alpha; P.act_kind = act_kind == RECALGO_ACT_NONE ? 0 : 1 + act_kind; P.z = z;
    P.vec_store = (aligned16(y) && ldy % 4 == 0 && N % 4 == 0 && (bias == nullptr || aligned16(bias))) ? 1 : 0;
    P.drop = recalgo_drop::from_abi(drop);
    const int grid = cdiv(M, BM) * cdiv(N, BN);
    const bool fast = fast_rc(P.seg[0].a, K) && fast_rm(P.seg[0].b, N) && fast_rc(P.seg[1].a, K2) && fast_rm(P.seg[1].b, N);
    if (fast) hipLaunchKernelGGL(dense_fwd_kernel<true>, dim3(grid), dim3(kThreads), 0, as_stream(stream), P);
    else hipLaunchKernelGGL(dense_fwd_kernel<false>, dim3(grid), dim3(kThreads), 0, as_stream(stream), P);
    RECALGO_RETURN_LAST();
}

static bool build_dgrad(DgradArgs& P, const float* g, int ldg, const float* y_mask, const float* w, int M, int N, int K,
                        const float* c_in, int ldc, float beta, float* dx, int lddx, int accumulate) {
    if (!(M >= 0 && N > 0 && K > 0 && dx != nullptr && lddx >= K)) return false;
    if (!(g != nullptr && ldg >= N && w != nullptr)) return false;
    if (!(c_in == nullptr || ldc >= K)) return false;
    P.seg = Segment{operand(g, y_mask, ldg, M, N), operand(w, nullptr, N, K, N), N};
    P.c_in = c_in; P.beta = beta; P.ldc = ldc; P.dx = dx; P.lddx = lddx; P.M = M; P.K = K; P.accumulate = accumulate;
    P.tiles_per_block = 1;
    P.bn_x = P.bn_mean = P.bn_rstd = nullptr; P.bn_partials = nullptr;
    P.vec_store = (aligned16(dx) && lddx % 4 == 0 && K % 4 == 0 && (c_in == nullptr || (aligned16(c_in) && ldc % 4 == 0))) ? 1 : 0;
    P.dx_mask = nullptr; P.ld_mask = 0;
    return true;
}
static bool dgrad_fast(const DgradArgs& P) { return fast_rc(P.seg.a, P.seg.n_red) && fast_rc(P.seg.b, P.seg.n_red); }

RECALGO_EXPORT int recalgo_dense_bwd_input(const float* g, int ldg, const float* y_mask, const float* w, int M, int N,
                                           int K, const float* c_in, int ldc, float beta, float* dx, int lddx,
                                           int accumulate, recalgo_stream_t stream) {
    DgradArgs P;
    RECALGO_REQUIRE(build_dgrad(P, g, ldg, y_mask, w, M, N, K, c_in, ldc, beta, dx, lddx, accumulate));
    if (M == 0) return 0;
    const int grid = cdiv(M, BM) * cdiv(K, BN);
    const bool fast = dgrad_fast(P);
    hipStream_t st = as_stream(stream);
    if (fast && y_mask) hipLaunchKernelGGL((dense_dgrad_kernel<true, true>), dim3(grid), dim3(kThreads), 0, st, P);
    else if (fast) hipLaunchKernelGGL((dense_dgrad_kernel<true, false>), dim3(grid), dim3(kThreads), 0, st, P);
    else if (y_mask) hipLaunchKernelGGL((dense_dgrad_kernel<false, true>), dim3(grid), dim3(kThreads), 0, st, P);
    else hipLaunchKernelGGL((dense_dgrad_kernel<false, false>), dim3(grid), dim3(kThreads), 0, st, P);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int64_t recalgo_dense_bwd_weights_workspace_bytes(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    const int S = wgrad_splits(M, K, N);
    return S <= 1 ? 0 : (int64_t)S * (int64_t)wgrad_slab(K, N) * (int64_t)sizeof(float);
}

static int launch_split_jobs(const SplitJobs& J, unsigned blocks, hipStream_t st) {
    if (J.n_jobs == 0 && J.step == nullptr) return 0;
    hipLaunchKernelGGL(dense_sum_slabs_kernel, dim3(blocks ? blocks : 1u), dim3(256), 0, st, J);
    return (int)hipGetLastError();
}

static SplitJob make_split_job(const float* ws, int S, int K, int N, float* dw, float* dbias, unsigned first_block,
                               unsigned* blocks) {
    SplitJob jb;
    jb.partials = ws; jb.out0 = dw; jb.out1 = dbias; jb.S = S;
    jb.slab = wgrad_slab(K, N);
    jb.n0 = (unsigned long long)K * N;
    jb.n = jb.n0 + (dbias ? (unsigned long long)N : 0ull);
    jb.vec = (jb.n0 % 4 == 0 && N % 4 == 0 && aligned16(dw) && (dbias == nullptr || aligned16(dbias)) && aligned16(ws)) ? 1 : 0;
    jb.first_block = first_block;
    *blocks = (unsigned)cdiv((int64_t)(jb.vec ? (jb.n + 3) / 4 : jb.n), 256);
    return jb;
}

// -> number of splits (>= 1), or 0 on bad arguments
static int build_wgrad(WgradArgs& P, const float* x, int ldx, const float* g, int ldg, const float* y_mask, int M, int K,
                       int N, float* dw, float* dbias, void* workspace) {
    if (!(M > 0 && N > 0 && K > 0 && dw != nullptr)) return 0;
    if (!(x != nullptr && ldx >= K && g != nullptr && ldg >= N)) return 0;
    const int S = wgrad_splits(M, K, N);
    if (!(S == 1 || (workspace != nullptr && aligned16(workspace)))) return 0;
    P.seg = Segment{operand(x, nullptr, ldx, M, K), operand(g, y_mask, ldg, M, N), M};
    P.K = K; P.N = N; P.splits = S;
    P.rows_per_split = cdiv(cdiv(M, S), BK) * BK;
    P.want_dbias = dbias != nullptr;
    P.slab = S == 1 ? 0 : wgrad_slab(K, N);
    P.out = S == 1 ? dw : static_cast<float*>(workspace);
    P.dbias_out = dbias;
    P.vec_store = (aligned16(P.out) && N % 4 == 0) ? 1 : 0;
    return S;
}
static bool wgrad_fast(const WgradArgs& P) { return fast_rm(P.seg.a, P.K) && fast_rm(P.seg.b, P.N); }

static int finish_wgrad(int S, int defer_reduce, int K, int N, float* dw, float* dbias, void* workspace, hipStream_t st) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    if (S > 1 && !defer_reduce) {
        SplitJobs J;
        unsigned blocks = 0;
        J.job[0] = make_split_job(static_cast<const float*>(workspace), S, K, N, dw, dbias, 0, &blocks);
        J.n_jobs = 1;
        J.step = nullptr;
        return launch_split_jobs(J, blocks, st);
    }
    return 0;
}

RECALGO_EXPORT int recalgo_dense_bwd_weights(const float* x, int ldx, const float* g, int ldg, const float* y_mask,
                                             int M, int K, int N, float* dw, float* dbias, void* workspace,
                                             int defer_reduce, recalgo_stream_t stream) {
    WgradArgs P;
    const int S = build_wgrad(P, x, ldx, g, ldg, y_mask, M, K, N, dw, dbias, workspace);
    RECALGO_REQUIRE(S >= 1);
    hipStream_t st = as_stream(stream);
    const int grid = cdiv(K, BM) * cdiv(N, BN) * S;
    const bool fast = wgrad_fast(P);
    if (fast && y_mask) hipLaunchKernelGGL((dense_wgrad_kernel<true, true>), dim3(grid), dim3(kThreads), 0, st, P);
    else if (fast) hipLaunchKernelGGL((dense_wgrad_kernel<true, false>), dim3(grid), dim3(kThreads), 0, st, P);
    else if (y_mask) hipLaunchKernelGGL((dense_wgrad_kernel<false, true>), dim3(grid), dim3(kThreads), 0, st, P);
    else hipLaunchKernelGGL((dense_wgrad_kernel<false, false>), dim3(grid), dim3(kThreads), 0, st, P);
    return finish_wgrad(S, defer_reduce, K, N, dw, dbias, workspace, st);
}

RECALGO_EXPORT int recalgo_dense_bwd(const float* x, int ldx, const float* g, int ldg, const float* y_mask, const float* w,
                                     int M, int K, int N, const float* c_in, int ldc, float beta, float* dx, int lddx,
                                     float* dw, float* dbias, void* workspace, int defer_reduce, recalgo_stream_t stream) {
    return recalgo_dense_bwd_bn(x, ldx, g, ldg, y_mask, w, M, K, N, c_in, ldc, beta, dx, lddx, dw, dbias, workspace, defer_reduce,
                                nullptr, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

RECALGO_EXPORT int recalgo_dense_bwd_bn(const float* x, int ldx, const float* g, int ldg, const float* y_mask, const float* w,
                                        int M, int K, int N, const float* c_in, int ldc, float beta, float* dx, int lddx,
                                        float* dw, float* dbias, void* workspace, int defer_reduce, const float* bn_x,
                                        const float* bn_mean, const float* bn_rstd, float* bn_partials,
                                        const float* dx_relu_mask, int ld_mask, recalgo_stream_t stream) {
    DgradArgs D;
    WgradArgs W;
    RECALGO_REQUIRE(M > 0 && build_dgrad(D, g, ldg, y_mask, w, M, N, K, c_in, ldc, beta, dx, lddx, 0));
    RECALGO_REQUIRE(bn_partials == nullptr || (bn_x != nullptr && bn_mean != nullptr && bn_rstd != nullptr));
    RECALGO_REQUIRE(dx_relu_mask == nullptr || ld_mask >= K);
    D.dx_mask = dx_relu_mask; D.ld_mask = ld_mask;
    if (dx_relu_mask != nullptr && !(aligned16(dx_relu_mask) && ld_mask % 4 == 0)) D.vec_store = 0;
    D.bn_x = bn_x; D.bn_mean = bn_mean; D.bn_rstd = bn_rstd; D.bn_partials = bn_partials;
    const int S = build_wgrad(W, x, ldx, g, ldg, y_mask, M, K, N, dw, dbias, workspace);
    RECALGO_REQUIRE(S >= 1);
    hipStream_t st = as_stream(stream);
    D.tiles_per_block = bwd_balance(M, K, N).tiles_per_block;
    const int gd = cdiv(cdiv(M, BM) * cdiv(K, BN), D.tiles_per_block), gw = cdiv(K, BM) * cdiv(N, BN) * S;
    const bool fast = dgrad_fast(D) && wgrad_fast(W);
    if (fast && y_mask) hipLaunchKernelGGL((dense_bwd_kernel<true, true>), dim3(gd + gw), dim3(kThreads), 0, st, D, W, gd);
    else if (fast) hipLaunchKernelGGL((dense_bwd_kernel<true, false>), dim3(gd + gw), dim3(kThreads), 0, st, D, W, gd);
    else if (y_mask) hipLaunchKernelGGL((dense_bwd_kernel<false, true>), dim3(gd + gw), dim3(kThreads), 0, st, D, W, gd);
    else hipLaunchKernelGGL((dense_bwd_kernel<false, false>), dim3(gd + gw), dim3(kThreads), 0, st, D, W, gd);
    return finish_wgrad(S, defer_reduce, K, N, dw, dbias, workspace, st);
}

// partial rows (= workgroups) of a CrossNet backward riding in recalgo_dense_bwd_rider: recalgo_cross_bwd_partial_rows(B)
static int cross_rider_blocks(int B) { const int need = cdiv(B, 8); return need < 1 ? 1 : (need > 256 ? 256 : need); }

RECALGO_EXPORT int recalgo_dense_bwd_cross_rider_supported(int d, int L) {
    return (d > 0 && d % 4 == 0 && d <= 512 && L >= 1 && L <= 4) ? 1 : 0;
}

RECALGO_EXPORT int recalgo_dense_bwd_rider(const float* x, int ldx, const float* g, int ldg, const float* y_mask, const float* w,
                                           int M, int K, int N, const float* c_in, int ldc, float beta, float* dx, int lddx,
                                           float* dw, float* dbias, void* workspace, int defer_reduce, const float* bn_x,
                                           const float* bn_mean, const float* bn_rstd, float* bn_partials,
                                           const float* dx_relu_mask, int ld_mask, const float* r_x, int r_ldx, const float* r_g,
                                           int r_ldg, int r_K, int r_N, float* r_dw, float* r_dbias, void* r_workspace,
                                           const float* c_x0, int c_x_stride, const float* c_w, const float* c_b, const float* c_g,
                                           int c_g_stride, int c_d, int c_L, float* c_dx0, void* c_workspace,
                                           recalgo_stream_t stream) {
    DgradArgs D;
    WgradArgs W, R;
    RECALGO_REQUIRE(M > 0 && build_dgrad(D, g, ldg, y_mask, w, M, N, K, c_in, ldc, beta, dx, lddx, 0));
    RECALGO_REQUIRE(bn_partials == nullptr || (bn_x != nullptr && bn_mean != nullptr && bn_rstd != nullptr));
    RECALGO_REQUIRE(dx_relu_mask == nullptr || ld_mask >= K);
    RECALGO_REQUIRE(r_x != nullptr || c_x0 != nullptr);
    D.dx_mask = dx_relu_mask; D.ld_mask = ld_mask;
    if (dx_relu_mask != nullptr && !(aligned16(dx_relu_mask) && ld_mask % 4 == 0)) D.vec_store = 0;
    D.bn_x = bn_x; D.bn_mean = bn_mean; D.bn_rstd = bn_rstd; D.bn_partials = bn_partials;
    const int S = build_wgrad(W, x, ldx, g, ldg, y_mask, M, K, N, dw, dbias, workspace);
    RECALGO_REQUIRE(S >= 1);
    // the riders' partials are always left to recalgo_dense_bwd_weights_reduce (a weight-gradient rider with a single split writes dw itself)
    int gr = 0;
    if (r_x != nullptr) {
        const int SR = build_wgrad(R, r_x, r_ldx, r_g, r_ldg, nullptr, M, r_K, r_N, r_dw, r_dbias, r_workspace);
        RECALGO_REQUIRE(SR >= 1 && wgrad_fast(R));
        gr = cdiv(r_K, BM) * cdiv(r_N, BN) * SR;
    } else {
        R = W;
    }
    CrossRider C{};
    if (c_x0 != nullptr) {
        RECALGO_REQUIRE(recalgo_dense_bwd_cross_rider_supported(c_d, c_L) && c_w && c_b && c_g && c_dx0 && c_workspace);
        RECALGO_REQUIRE(c_x_stride % 4 == 0 && c_g_stride % 4 == 0 && c_x_stride >= c_d && c_g_stride >= c_d);
        RECALGO_REQUIRE(aligned16(c_x0) && aligned16(c_w) && aligned16(c_b) && aligned16(c_g) && aligned16(c_dx0));
        C = CrossRider{c_x0, reinterpret_cast<const float4*>(c_w), reinterpret_cast<const float4*>(c_b), c_g, c_dx0,
                       static_cast<float*>(c_workspace), (unsigned)c_x_stride, (unsigned)c_g_stride, (unsigned)M, (unsigned)(c_d / 4),
                       (unsigned)cross_rider_blocks(M)};
    }
    hipStream_t st = as_stream(stream);
    D.tiles_per_block = bwd_balance(M, K, N).tiles_per_block;
    const int gd = cdiv(cdiv(M, BM) * cdiv(K, BN), D.tiles_per_block), gw = cdiv(K, BM) * cdiv(N, BN) * S;
    RECALGO_REQUIRE(dgrad_fast(D) && wgrad_fast(W));
    const dim3 grid(gd + gw + gr + (int)C.blocks), block(kThreads);
    const int nv = c_x0 ? cdiv(c_d / 4, 64) : 0;
#define RIDER_LAUNCH(MASK, NV, L) hipLaunchKernelGGL((dense_bwd_rider_kernel<MASK, NV, L>), grid, block, 0, st, D, W, R, C, gd, gr)
#define RIDER_L(MASK, NV)                                             \
    switch (c_L) {                                                    \
        case 1: RIDER_LAUNCH(MASK, NV, 1); break;                     \
        case 2: RIDER_LAUNCH(MASK, NV, 2); break;                     \
        case 3: RIDER_LAUNCH(MASK, NV, 3); break;                     \
        default: RIDER_LAUNCH(MASK, NV, 4); break;                    \
    }
    if (nv == 0) {
        if (y_mask) RIDER_LAUNCH(true, 0, 0);
        else RIDER_LAUNCH(false, 0, 0);
    } else {
        RECALGO_REQUIRE(y_mask == nullptr);          // (a cross rider only beside a layer whose gradient arrives masked: dcn.py's stack)
        if (nv == 1) { RIDER_L(false, 1) } else { RIDER_L(false, 2) }
    }
#undef RIDER_L
#undef RIDER_LAUNCH
    return finish_wgrad(S, defer_reduce, K, N, dw, dbias, workspace, st);
}

// 1: recalgo_dense_bwd_rider serves these shapes (the vectorised tile paths of all three GEMMs; r_x == NULL: no weight-gradient
// rider), 0: launch the riders on their own
RECALGO_EXPORT int recalgo_dense_bwd_rider_supported(const float* x, int ldx, const float* g, int ldg, const float* y_mask,
                                                     const float* w, int M, int K, int N, float* dx, int lddx, const float* r_x,
                                                     int r_ldx, const float* r_g, int r_ldg, int r_K, int r_N) {
    if (!(M > 0 && K > 0 && N > 0 && x && g && w && dx)) return 0;
    DgradArgs D;
    WgradArgs W, R;
    static float dummy[4];
    alignas(16) static char ws[16];
    if (!build_dgrad(D, g, ldg, y_mask, w, M, N, K, nullptr, 0, 0.f, dx, lddx, 0)) return 0;
    if (build_wgrad(W, x, ldx, g, ldg, y_mask, M, K, N, dummy, nullptr, ws) < 1) return 0;
    if (!(dgrad_fast(D) && wgrad_fast(W))) return 0;
    if (r_x != nullptr) {
        if (!(r_K > 0 && r_N > 0 && r_g)) return 0;
        if (build_wgrad(R, r_x, r_ldx, r_g, r_ldg, nullptr, M, r_K, r_N, dummy, nullptr, ws) < 1) return 0;
        if (!wgrad_fast(R)) return 0;
    }
    return 1;
}

RECALGO_EXPORT int recalgo_dense_bwd_weights_reduce(const recalgo_dense_split_t* jobs, int n_jobs,
                                                    const recalgo_colsum_t* sums, int n_sums, int64_t* step_dev,
                                                    recalgo_stream_t stream) {
    RECALGO_REQUIRE(n_jobs >= 0 && (n_jobs == 0 || jobs != nullptr) && n_sums >= 0 && (n_sums == 0 || sums != nullptr));
    hipStream_t st = as_stream(stream);
    SplitJobs J;
    J.n_jobs = 0;
    J.step = reinterpret_cast<long long*>(step_dev);
    unsigned blocks = 0;
    for (int i = 0; i < n_sums; ++i) {                       // plain column sums of partial rows (the loss tail's)
        const recalgo_colsum_t& c = sums[i];
        RECALGO_REQUIRE(c.partials != nullptr && c.out != nullptr && c.rows >= 1 && c.n >= 1 && c.row_stride >= c.n);
        SplitJob jb;
        jb.partials = c.partials; jb.out0 = c.out; jb.out1 = nullptr; jb.S = c.rows;
        jb.slab = (unsigned long long)c.row_stride; jb.n0 = jb.n = (unsigned long long)c.n;
        jb.vec = 2;
        jb.first_block = blocks;
        blocks += (unsigned)cdiv((int64_t)jb.n, 16);
        J.job[J.n_jobs++] = jb;
        if (J.n_jobs == kMaxSplitJobs) {
            int rc = launch_split_jobs(J, blocks, st);
            if (rc) return rc;
            J.n_jobs = 0;
            J.step = nullptr;
            blocks = 0;
        }
    }
    for (int i = 0; i < n_jobs; ++i) {
        const recalgo_dense_split_t& d = jobs[i];
        RECALGO_REQUIRE(d.M > 0 && d.K > 0 && d.N > 0 && d.dw != nullptr);
        const int S = wgrad_splits(d.M, d.K, d.N);
        if (S <= 1) continue;                               // nothing was deferred for this shape
        RECALGO_REQUIRE(d.workspace != nullptr);
        unsigned nb = 0;
        J.job[J.n_jobs++] = make_split_job(static_cast<const float*>(d.workspace), S, d.K, d.N, d.dw, d.dbias, blocks, &nb);
        blocks += nb;
        if (J.n_jobs == kMaxSplitJobs) {
            int rc = launch_split_jobs(J, blocks, st);
            if (rc) return rc;
            J.n_jobs = 0;
            J.step = nullptr;
            blocks = 0;
        }
    }
    return launch_split_jobs(J, blocks, st);
}
