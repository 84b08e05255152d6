// Loss tail (a14), TF1 Adam (a15), DIN activations (a12) — small streaming kernels, gfx950.
#include "deferred.h"
#include "act.h"
#include "plan_scan.h"

namespace {

// ---------------------------------------------------------------------------------------
// sigmoid + mean sigmoid-CE + dlogit.  B is a few thousand: a single 1024-thread workgroup
// keeps the mean a deterministic tree reduction and needs no zero-initialised accumulator.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sigmoid_ce_kernel(
    const float* __restrict__ logits, const float* __restrict__ labels, unsigned B,
    float grad_scale, float* __restrict__ prob, float* __restrict__ loss,
    float* __restrict__ dlogit) {
    __shared__ float red[16];
    float acc = 0.f;
    const float invB = 1.0f / (float)B;
    for (unsigned i = threadIdx.x; i < B; i += 1024) {
        float x = logits[i], z = labels[i];
        float ax = fabsf(x);
        float e = expf(-ax);
        // tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log1p(exp(-|x|))
        acc += fmaxf(x, 0.f) - x * z + log1pf(e);
        float r = e / (1.0f + e);
        float p = x >= 0.f ? 1.0f / (1.0f + e) : r;
        prob[i] = p;
        // d/dx in the form TF's autodiff of the three terms produces:
        //   [x>=0] - z -/+ e/(1+e)     (keeps 1e-13-size gradients at |x| ~ 30)
        if (dlogit) {
            float d = ((x >= 0.f ? 1.0f : 0.f) - z) + (x >= 0.f ? -r : r);
            dlogit[i] = d * grad_scale * invB;
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = threadIdx.x < 16 ? red[threadIdx.x] : 0.f;
        v = wave_sum(v);
        if (threadIdx.x == 0) loss[0] = v * invB;
    }
}

// ---------------------------------------------------------------------------------------
// TF1 Adam, dense.  Pure stream: 4 reads + 3 writes (+ sparse zeroing of g) per element.
// ---------------------------------------------------------------------------------------
// (the update itself: deferred.h adam1 — shared with the sparse optimizer so that the deferred form stays bit-identical)
using recalgo_deferred::adam1;

__global__ __launch_bounds__(256) void adam_tf1_kernel(float* __restrict__ p, float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       int64_t n4, int64_t n, float lr_t_val,
                                                       const float* __restrict__ lr_t_dev, float b1,
                                                       float b2, float eps, int zero_grad) {
    const float lr_t = lr_t_dev ? lr_t_dev[0] : lr_t_val;
    const int64_t stride = (int64_t)gridDim.x * 256;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* g4 = reinterpret_cast<float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 gg = g4[i];
        float4 mm = m4[i], vv = v4[i];
        // Exact shortcut of the dense update: where g, m and v are all zero (embedding rows no batch
        // has touched yet), m' = v' = 0 and p' = p - lr_t*0/(0+eps) = p — nothing to read or write.
        // TF1's dense semantics are kept bit for bit; only the traffic of inert words is skipped
        // (12 instead of 28 bytes per parameter).
        const bool inert = gg.x == 0.f && gg.y == 0.f && gg.z == 0.f && gg.w == 0.f && mm.x == 0.f && mm.y == 0.f &&
                           mm.z == 0.f && mm.w == 0.f && vv.x == 0.f && vv.y == 0.f && vv.z == 0.f && vv.w == 0.f;
        if (inert) continue;
        float4 pp = p4[i];
        adam1(pp.x, gg.x, mm.x, vv.x, lr_t, b1, b2, eps);
        adam1(pp.y, gg.y, mm.y, vv.y, lr_t, b1, b2, eps);
        adam1(pp.z, gg.z, mm.z, vv.z, lr_t, b1, b2, eps);
        adam1(pp.w, gg.w, mm.w, vv.w, lr_t, b1, b2, eps);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (zero_grad && (gg.x != 0.f || gg.y != 0.f || gg.z != 0.f || gg.w != 0.f)) g4[i] = f4_zero();
    }
    // tail (n % 4)
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - n4 * 4)) {
        int64_t i = n4 * 4 + threadIdx.x;
        float pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        adam1(pp, gg, mm, vv, lr_t, b1, b2, eps);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (zero_grad) g[i] = 0.f;
    }
}

// TF1 dense Adam over an embedding arena [rows, K] with a per-row liveness byte.  A row is inert
// until some batch first touches it (g != 0): until then g = m = v = 0 and the dense update is the
// identity, so nothing but g (to detect the first touch) and the byte is read.  Bit-identical to
// adam_tf1_kernel over the same buffers; invariant: live[r] == 0  =>  m[r,:] == v[r,:] == 0.
// K4 = K/4 lanes own one row (K4 divides 64, so a row never straddles a wave).
template <int K4>
__global__ __launch_bounds__(256) void adam_tf1_rows_kernel(float* __restrict__ p, float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v,
                                                            unsigned char* __restrict__ live, int64_t total4,
                                                            float lr_t_val, const float* __restrict__ lr_t_dev,
                                                            float b1, float b2, float eps, int zero_grad) {
    const float lr_t = lr_t_dev ? lr_t_dev[0] : lr_t_val;
    const int64_t stride = (int64_t)gridDim.x * 256;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* g4 = reinterpret_cast<float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
        const int64_t row = i / K4;
        float4 gg = g4[i];
        int nz = (gg.x != 0.f) | (gg.y != 0.f) | (gg.z != 0.f) | (gg.w != 0.f);
        int any = nz;
#pragma unroll
        for (int o = 1; o < K4; o <<= 1) any |= __shfl_xor(any, o, 64);
        const unsigned char was = live[row];
        if (!was && !any) continue;                       // inert row: identity update
        if (!was && (i % K4) == 0) live[row] = 1;
        float4 mm = m4[i], vv = v4[i], pp = p4[i];
        adam1(pp.x, gg.x, mm.x, vv.x, lr_t, b1, b2, eps);
        adam1(pp.y, gg.y, mm.y, vv.y, lr_t, b1, b2, eps);
        adam1(pp.z, gg.z, mm.z, vv.z, lr_t, b1, b2, eps);
        adam1(pp.w, gg.w, mm.w, vv.w, lr_t, b1, b2, eps);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (zero_grad && nz) g4[i] = f4_zero();
    }
}

// ---------------------------------------------------------------------------------------
// Live-row bookkeeping for embedding arenas (optimizer cost proportional to the
// rows a model has touched, with TF1's dense semantics kept exactly).
//   live[r] (one byte per row) == 1 and r is in list[0 .. count)  <=>  some gradient has reached row r.
// mark: every valid (b, f) id of a batch is checked; the first toucher of a row (atomicOr on the
// aligned 32-bit word holding the byte) appends it to the list.  adam_list: the dense TF1 update
// applied to the listed rows only — all other rows have g = m = v = 0, for which it is the identity.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mark_live_rows_kernel(const int64_t* __restrict__ ids,
                                                             const int64_t* __restrict__ row_base, int64_t n,
                                                             unsigned F, unsigned* __restrict__ live_words,
                                                             int* __restrict__ list, int* __restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t id = ids[i];
    if (id < 0) return;
    const int64_t row = id + (row_base ? row_base[i % F] : 0);
    const unsigned bit = 1u << (8 * (unsigned)(row & 3));
    unsigned* w = live_words + (row >> 2);
    if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) return;      // common case: already live
    const unsigned old = atomicOr(w, bit);
    if (!(old & bit)) list[atomicAdd(count, 1)] = (int)row;
}

// Address-ordered rebuild of the live-row list from the liveness bytes (housekeeping between steps:
// mark_live_rows appends in first-touch order, i.e. at random, and the list Adam then walks HBM at
// random — measured 40 us vs 32 us in address order for 345 k rows x 64 B x 7 streams).  Ordered
// compaction in three small launches: per-chunk popcounts, one-block exclusive scan of the chunk
// counts, ordered write.  A chunk = 1024 liveness words = 4096 rows, one word per thread and round.
constexpr unsigned kChunkWords = 1024;

__device__ __forceinline__ unsigned live_bytes_in(unsigned w) {       // number of non-zero bytes in a word
    return ((w & 0xffu) != 0) + ((w & 0xff00u) != 0) + ((w & 0xff0000u) != 0) + ((w & 0xff000000u) != 0);
}

__global__ __launch_bounds__(256) void live_chunk_count_kernel(const unsigned* __restrict__ live_words,
                                                               int64_t n_words, int* __restrict__ chunk_count) {
    __shared__ int part[4];
    const int64_t base = (int64_t)blockIdx.x * kChunkWords;
    int c = 0;
#pragma unroll
    for (unsigned r = 0; r < kChunkWords / 256; ++r) {
        const int64_t w = base + r * 256 + threadIdx.x;
        if (w < n_words) c += (int)live_bytes_in(live_words[w]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) chunk_count[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of chunk_count in place, total -> live_count[0]; one block, any number of chunks
__global__ __launch_bounds__(1024) void live_chunk_scan_kernel(int* __restrict__ chunk_count, int n_chunks,
                                                               int* __restrict__ live_count) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n_chunks; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const int x = i < n_chunks ? chunk_count[i] : 0;
        int incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(incl, o, 64);
            if ((int)lane >= o) incl += y;
        }
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        int before = carry_s;
        for (unsigned k = 0; k < wv; ++k) before += wave_tot[k];
        if (i < n_chunks) chunk_count[i] = before + incl - x;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) live_count[0] = carry_s;
}

__global__ __launch_bounds__(256) void live_chunk_write_kernel(const unsigned* __restrict__ live_words,
                                                               int64_t n_words, int64_t rows,
                                                               const int* __restrict__ chunk_base,
                                                               int* __restrict__ list) {
    __shared__ int wave_tot[4];
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * kChunkWords;
    int out = chunk_base[blockIdx.x];
    for (unsigned r = 0; r < kChunkWords / 256; ++r) {
        const int64_t w = base + r * 256 + threadIdx.x;
        const unsigned word = w < n_words ? live_words[w] : 0u;
        const int x = (int)live_bytes_in(word);
        int incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(incl, o, 64);
            if ((int)lane >= o) incl += y;
        }
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        int pos = out + incl - x;
        for (unsigned k = 0; k < wv; ++k) pos += wave_tot[k];
#pragma unroll
        for (unsigned b = 0; b < 4; ++b)
            if ((word >> (8 * b)) & 0xffu) {
                const int64_t row = w * 4 + b;
                if (row < rows) list[pos++] = (int)row;
            }
        out += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        __syncthreads();
    }
}

template <int K4>
__global__ __launch_bounds__(256) void adam_tf1_list_kernel(float* __restrict__ p, float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v,
                                                            const int* __restrict__ list,
                                                            const int* __restrict__ count, float lr_t_val,
                                                            const float* __restrict__ lr_t_dev, float b1,
                                                            float b2, float eps, int zero_grad) {
    const float lr_t = lr_t_dev ? lr_t_dev[0] : lr_t_val;
    const int64_t total4 = (int64_t)count[0] * K4;
    const int64_t stride = (int64_t)gridDim.x * 256;
    float4* p4 = reinterpret_cast<float4*>(p);
    float4* g4 = reinterpret_cast<float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total4; t += stride) {
        const int64_t i = (int64_t)list[t / K4] * K4 + (t % K4);
        float4 gg = g4[i], mm = m4[i], vv = v4[i], pp = p4[i];
        const bool nz = gg.x != 0.f || gg.y != 0.f || gg.z != 0.f || gg.w != 0.f;
        adam1(pp.x, gg.x, mm.x, vv.x, lr_t, b1, b2, eps);
        adam1(pp.y, gg.y, mm.y, vv.y, lr_t, b1, b2, eps);
        adam1(pp.z, gg.z, mm.z, vv.z, lr_t, b1, b2, eps);
        adam1(pp.w, gg.w, mm.w, vv.w, lr_t, b1, b2, eps);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (zero_grad && nz) g4[i] = f4_zero();
    }
}

// ---------------------------------------------------------------------------------------
// PReLU / Dice (algorithm/DIN/activations.py:4-37), x: [rows, C], alpha: [C].
// ---------------------------------------------------------------------------------------
using recalgo_act::kDiceInvStd;                      // 1/sqrt(1 + 1e-3): BN inference, stats (0,1)

template <bool DICE>
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x,
                                                      const float* __restrict__ alpha, int64_t n,
                                                      unsigned C, float* __restrict__ y) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float xv = x[i], a = alpha[i % C];
    if (DICE) {
        float px = 1.0f / (1.0f + expf(-xv * kDiceInvStd));
        y[i] = xv * px + a * xv * (1.0f - px);
    } else {
        y[i] = fmaxf(0.f, xv) + a * fminf(0.f, xv);
    }
}

// dx and per-workgroup partial dalpha ([gridDim.x][C], rows of one workgroup are contiguous).
template <bool DICE>
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ x,
                                                      const float* __restrict__ alpha,
                                                      const float* __restrict__ gy, unsigned rows,
                                                      unsigned C, unsigned rows_per_blk,
                                                      float* __restrict__ dx,
                                                      float* __restrict__ partial) {
    const unsigned r0 = blockIdx.x * rows_per_blk;
    const unsigned r1 = min(rows, r0 + rows_per_blk);
    for (unsigned c = threadIdx.x; c < C; c += 256) {
        float a = alpha[c], da = 0.f;
        for (unsigned r = r0; r < r1; ++r) {
            size_t i = (size_t)r * C + c;
            float xv = x[i], g = gy[i];
            if (DICE) {
                float px = 1.0f / (1.0f + expf(-xv * kDiceInvStd));
                float dpx = px * (1.0f - px) * kDiceInvStd;
                // y = x*px + a*x*(1-px)
                dx[i] = g * (px + a * (1.0f - px) + xv * dpx * (1.0f - a));
                da += g * xv * (1.0f - px);
            } else {
                dx[i] = g * (xv > 0.f ? 1.0f : (xv < 0.f ? a : 0.f));
                da += g * fminf(0.f, xv);
            }
        }
        partial[(size_t)blockIdx.x * C + c] = da;
    }
}

constexpr unsigned kActRowsPerBlk = 16;

// C % 4 == 0: the same pass on 64-row x 64-column tiles (16 float4 column groups x 16 row lanes, 4
// rows per thread): 8x the workgroups of the row-block kernel for a [4096, 128] layer and 4x fewer
// partial rows for the fixed-order dalpha sum.
constexpr unsigned kActTileRows = 64;
template <bool DICE>
__global__ __launch_bounds__(256) void act_bwd_tile_kernel(const float4* __restrict__ x,
                                                           const float4* __restrict__ alpha,
                                                           const float4* __restrict__ gy, unsigned rows,
                                                           unsigned C4, float4* __restrict__ dx,
                                                           float4* __restrict__ partial) {
    __shared__ float4 sh[256];
    const unsigned cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const unsigned c4 = blockIdx.x * 16 + cl;
    const unsigned r0 = blockIdx.y * kActTileRows;
    float4 da = f4_zero();
    if (c4 < C4) {
        const float4 a4 = alpha[c4];
        const float a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (unsigned k = 0; k < kActTileRows / 16; ++k) {
            const unsigned r = r0 + rl + 16 * k;
            if (r < rows) {
                const size_t i = (size_t)r * C4 + c4;
                const float4 x4 = x[i], g4 = gy[i];
                const float xv[4] = {x4.x, x4.y, x4.z, x4.w}, g[4] = {g4.x, g4.y, g4.z, g4.w};
                float o[4], d[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (DICE) {
                        const float px = 1.0f / (1.0f + expf(-xv[j] * kDiceInvStd));
                        const float dpx = px * (1.0f - px) * kDiceInvStd;
                        o[j] = g[j] * (px + a[j] * (1.0f - px) + xv[j] * dpx * (1.0f - a[j]));
                        d[j] = g[j] * xv[j] * (1.0f - px);
                    } else {
                        o[j] = g[j] * (xv[j] > 0.f ? 1.0f : (xv[j] < 0.f ? a[j] : 0.f));
                        d[j] = g[j] * fminf(0.f, xv[j]);
                    }
                }
                dx[i] = make_float4(o[0], o[1], o[2], o[3]);
                da = f4_add(da, make_float4(d[0], d[1], d[2], d[3]));
            }
        }
    }
    sh[threadIdx.x] = da;
    __syncthreads();
    if (rl == 0 && c4 < C4) {
        float4 t = sh[cl];
#pragma unroll
        for (unsigned k = 1; k < 16; ++k) t = f4_add(t, sh[k * 16 + cl]);
        partial[(size_t)blockIdx.y * C4 + c4] = t;
    }
}

// ---------------------------------------------------------------------------------------
// One launch per optimizer step: the step counter, TF1 Adam over the flat dense-variable buffer and
// over the live rows of up to kAdamMaxArenas embedding arenas (previously adam_advance + adam_tf1 +
// one adam_tf1_list per arena: 3-4 launches of 5-30 us, the first two at the launch floor).
// Every workgroup reads step[0] when it starts and derives lr_t itself (two pow and a sqrt in double);
// the LAST workgroup to finish (arrival ticket) publishes step[0] = t and re-arms the ticket — by then
// every workgroup has read the old value, so there is no race, and the next launch (a kernel boundary
// later) sees the new one.  hipGraph replayable: nothing depends on host-side state.
// ---------------------------------------------------------------------------------------
constexpr int kAdamMaxArenas = 4;
struct AdamArena {
    float* p; float* g; float* m; float* v;
    const int* list;
    const int* count;
    int K;
    int k4_shift;             // log2(K / 4) when K / 4 is a power of two, else -1
    int lazy;                 // LazyAdam: rows whose gradient is all zero this step are left untouched
    unsigned first_block, n_blocks;
};
struct AdamStepArgs {
    float* p; float* g; float* m; float* v;        // flat dense buffer (n may be 0)
    long long n;
    unsigned dense_blocks;
    AdamArena ar[kAdamMaxArenas];
    int n_arenas;
    long long* step;
    int* ticket;
    int advance;
    float lr, b1, b2, eps;
    int zero_grad;
    // the scatter plans of the arenas on the owner-computes path (sparse.hip): one extra workgroup each turns the plan's bucket
    // totals into the prefix `place` needs — this launch runs between the step's last count and `place` anyway
    recalgo_plan::Scan scan[kAdamMaxArenas];
    unsigned scan_first;      // blocks [scan_first, scan_first + n_scans) are those workgroups
};

__global__ __launch_bounds__(256) void adam_tf1_step_kernel(AdamStepArgs A) {
    __shared__ float s_lr_t;
    if (blockIdx.x >= A.scan_first) {                          // (workgroup-uniform)
        __shared__ unsigned scan_sh[8];
        recalgo_plan::scan_block(A.scan[blockIdx.x - A.scan_first], scan_sh);
        return;
    }
    const long long t = A.step[0] + (A.advance ? 1 : 0);
    if (threadIdx.x == 0) {
        const double td = (double)t;
        s_lr_t = (float)((double)A.lr * sqrt(1.0 - pow((double)A.b2, td)) / (1.0 - pow((double)A.b1, td)));
    }
    __syncthreads();
    const float lr_t = s_lr_t, b1 = A.b1, b2 = A.b2, eps = A.eps;
    if (blockIdx.x < A.dense_blocks) {
        const long long n4 = A.n / 4;
        const long long stride = (long long)A.dense_blocks * 256;
        float4* p4 = reinterpret_cast<float4*>(A.p);
        float4* g4 = reinterpret_cast<float4*>(A.g);
        float4* m4 = reinterpret_cast<float4*>(A.m);
        float4* v4 = reinterpret_cast<float4*>(A.v);
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            float4 gg = g4[i], mm = m4[i], vv = v4[i];
            const bool inert = gg.x == 0.f && gg.y == 0.f && gg.z == 0.f && gg.w == 0.f && mm.x == 0.f && mm.y == 0.f &&
                               mm.z == 0.f && mm.w == 0.f && vv.x == 0.f && vv.y == 0.f && vv.z == 0.f && vv.w == 0.f;
            if (inert) continue;                       // exact: the update of an all-zero word is the identity
            float4 pp = p4[i];
            adam1(pp.x, gg.x, mm.x, vv.x, lr_t, b1, b2, eps);
            adam1(pp.y, gg.y, mm.y, vv.y, lr_t, b1, b2, eps);
            adam1(pp.z, gg.z, mm.z, vv.z, lr_t, b1, b2, eps);
            adam1(pp.w, gg.w, mm.w, vv.w, lr_t, b1, b2, eps);
            p4[i] = pp; m4[i] = mm; v4[i] = vv;
            if (A.zero_grad && (gg.x != 0.f || gg.y != 0.f || gg.z != 0.f || gg.w != 0.f)) g4[i] = f4_zero();
        }
        if (blockIdx.x == 0 && threadIdx.x < (unsigned)(A.n - n4 * 4)) {
            const long long i = n4 * 4 + threadIdx.x;
            float pp = A.p[i], gg = A.g[i], mm = A.m[i], vv = A.v[i];
            adam1(pp, gg, mm, vv, lr_t, b1, b2, eps);
            A.p[i] = pp; A.m[i] = mm; A.v[i] = vv;
            if (A.zero_grad) A.g[i] = 0.f;
        }
    } else {
        int a = 0;
#pragma unroll
        for (int k = 1; k < kAdamMaxArenas; ++k)
            if (k < A.n_arenas && blockIdx.x >= A.ar[k].first_block) a = k;
        AdamArena R = A.ar[0];                       // select by value (a dynamic index into the kernel arguments
#pragma unroll                                       // would go through scratch)
        for (int k = 1; k < kAdamMaxArenas; ++k)
            if (a == k) R = A.ar[k];
        const unsigned blk = blockIdx.x - R.first_block;
        const long long stride = (long long)R.n_blocks * 256;
        const int K = R.K;
        if ((K & 3) == 0) {
            const int K4 = K >> 2;
            const long long total4 = (long long)R.count[0] * K4;
            float4* p4 = reinterpret_cast<float4*>(R.p);
            float4* g4 = reinterpret_cast<float4*>(R.g);
            float4* m4 = reinterpret_cast<float4*>(R.m);
            float4* v4 = reinterpret_cast<float4*>(R.v);
            const int sh = R.k4_shift;
            for (long long tt = (long long)blk * 256 + threadIdx.x; tt < total4; tt += stride) {
                const long long r = sh >= 0 ? tt >> sh : tt / K4;
                const long long i = (long long)R.list[r] * K4 + (tt - r * K4);
                float4 gg = g4[i];
                const bool nz = gg.x != 0.f || gg.y != 0.f || gg.z != 0.f || gg.w != 0.f;
                if (R.lazy) {
                    // tf.contrib.opt.LazyAdamOptimizer (the reference's DIEN, dien.py:328): only the rows of this step's
                    // IndexedSlices move — here: rows with a non-zero gradient (the K4 lanes of a row vote)
                    int any = nz;
                    if (sh > 0)
                        for (int o = 1; o < K4; o <<= 1) any |= __shfl_xor(any, o, 64);
                    if (!any) continue;
                }
                float4 mm = m4[i], vv = v4[i], pp = p4[i];
                adam1(pp.x, gg.x, mm.x, vv.x, lr_t, b1, b2, eps);
                adam1(pp.y, gg.y, mm.y, vv.y, lr_t, b1, b2, eps);
                adam1(pp.z, gg.z, mm.z, vv.z, lr_t, b1, b2, eps);
                adam1(pp.w, gg.w, mm.w, vv.w, lr_t, b1, b2, eps);
                p4[i] = pp; m4[i] = mm; v4[i] = vv;
                if (A.zero_grad && nz) g4[i] = f4_zero();
            }
        } else {
            const long long total = (long long)R.count[0] * K;
            for (long long tt = (long long)blk * 256 + threadIdx.x; tt < total; tt += stride) {
                const long long r = tt / K;
                const long long i = (long long)R.list[r] * K + (tt - r * K);
                float gg = R.g[i];
                const bool nz = gg != 0.f;
                if (R.lazy && !nz) continue;                 // (element-wise vote for widths that are not 4 * 2^n)
                float mm = R.m[i], vv = R.v[i], pp = R.p[i];
                adam1(pp, gg, mm, vv, lr_t, b1, b2, eps);
                R.p[i] = pp; R.m[i] = mm; R.v[i] = vv;
                if (A.zero_grad && nz) R.g[i] = 0.f;
            }
        }
    }
    if (!A.advance) return;
    // the last workgroup to arrive publishes the new step count
    __syncthreads();
    if (threadIdx.x == 0) {
        const int arrived = __hip_atomic_fetch_add(A.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == (int)A.scan_first - 1) {                 // (the plan-scan workgroups behind scan_first take no ticket)
            __hip_atomic_store(A.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            A.step[0] = t;
        }
    }
}

__global__ void adam_advance_kernel(int64_t* step, float lr, float b1, float b2, float* lr_t) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int64_t t = step[0] + 1;
        step[0] = t;
        double td = (double)t;
        lr_t[0] = (float)((double)lr * sqrt(1.0 - pow((double)b2, td)) / (1.0 - pow((double)b1, td)));
    }
}
// ---------------------------------------------------------------------------------------
// tf.concat of up to 4 row-major [B, w_p] parts + the value of DIN's mini-batch-aware regulariser over the result
// (din.py:249-257: l2_lambda / 2 / B * sum(ev^2), ev = concat[category, target, attention output]) in ONE launch:
// a workgroup copies a block of rows and accumulates their squares; per-workgroup partial sums are added IN FIXED ORDER by
// the workgroup that arrives last (ticket), so the scalar is bit-reproducible.  Replaces a concat copy, a dot product
// (two library launches) and a scaling launch.
// ---------------------------------------------------------------------------------------
constexpr int kCatMaxParts = 4;
struct CatArgs {
    const float* x[kCatMaxParts];
    int width[kCatMaxParts], off[kCatMaxParts];
    int n, B, C;              // C = total width (the row stride of `out`)
    float* out;
    float scale;
    float* sum_out;           // scale * sum(out^2)
    float* partials;          // [gridDim.x]
    unsigned* ticket;         // zero before the first launch; the last workgroup leaves it zero again
    int rows_per_block;
};
__global__ __launch_bounds__(256) void concat_sumsq_kernel(CatArgs A) {
    __shared__ float red[4];
    __shared__ unsigned s_last;
    const int r0 = blockIdx.x * A.rows_per_block, r1 = min(A.B, r0 + A.rows_per_block);
    float acc = 0.f;
    for (int p = 0; p < A.n; ++p) {
        const int w = A.width[p], n = (r1 - r0) * w;
        const float* __restrict__ src = A.x[p] + (size_t)r0 * w;
        float* __restrict__ dst = A.out + (size_t)r0 * A.C + A.off[p];
        if ((w & 3) == 0 && (A.C & 3) == 0 && (A.off[p] & 3) == 0 && (reinterpret_cast<uintptr_t>(A.x[p]) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(A.out) & 15) == 0) {
            const int w4 = w >> 2;
            for (int i = threadIdx.x; i < (n >> 2); i += 256) {
                const int r = i / w4, c = i - r * w4;
                const float4 v = reinterpret_cast<const float4*>(src)[i];
                *reinterpret_cast<float4*>(dst + (size_t)r * A.C + 4 * c) = v;
                acc = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, acc))));
            }
        } else {
            for (int i = threadIdx.x; i < n; i += 256) {
                const int r = i / w, c = i - r * w;
                const float v = src[i];
                dst[(size_t)r * A.C + c] = v;
                acc = fmaf(v, v, acc);
            }
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        // publish the partial with a write-through (sc1) store, drain it, then draw a ticket; the workgroup that draws the last
        // ticket reads every partial with sc1 loads (guides/cdna_hip_programming.md Guideline 16, the drained-sc1 form: no
        // release fence — a fence would first write back the 27 KB of `out` this workgroup has just dirtied, ~6 us)
        __hip_atomic_store(&A.partials[blockIdx.x], (red[0] + red[1]) + (red[2] + red[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    // fixed order: thread t adds partials t, t + 256, ...; then the 256 thread sums in thread order
    float tot = 0.f;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += 256)
        tot += __hip_atomic_load(&A.partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ float all[256];
    all[threadIdx.x] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t2 = 0.f;
        for (int i = 0; i < 256; ++i) t2 += all[i];
        A.sum_out[0] = A.scale * t2;
        __hip_atomic_store(A.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

RECALGO_EXPORT int64_t recalgo_concat_sumsq_workspace_bytes(int B) {
    return B > 0 ? (int64_t)(cdiv(B, 16) + 16) * 4 : 64;
}
RECALGO_EXPORT int recalgo_concat_sumsq(const float* const* parts, const int* widths, int n_parts, int B, float* out,
                                        float scale, float* sum_out, void* workspace, recalgo_stream_t stream) {
    RECALGO_REQUIRE(parts && widths && n_parts >= 1 && n_parts <= kCatMaxParts && B >= 0 && out && sum_out && workspace);
    CatArgs A;
    int C = 0;
    for (int p = 0; p < kCatMaxParts; ++p) {
        A.x[p] = p < n_parts ? parts[p] : nullptr;
        A.width[p] = p < n_parts ? widths[p] : 0;
        A.off[p] = C;
        if (p < n_parts) {
            RECALGO_REQUIRE(parts[p] != nullptr && widths[p] >= 1);
            C += widths[p];
        }
    }
    A.n = n_parts; A.B = B; A.C = C; A.out = out; A.scale = scale; A.sum_out = sum_out;
    A.rows_per_block = 16;
    const int blocks = B > 0 ? cdiv(B, A.rows_per_block) : 1;
    A.ticket = static_cast<unsigned*>(workspace);             // word 0 (zero-initialised by the caller, once)
    A.partials = reinterpret_cast<float*>(static_cast<unsigned*>(workspace) + 16);
    hipLaunchKernelGGL(concat_sumsq_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), A);
    RECALGO_RETURN_LAST();
}

namespace {
}  // namespace

namespace {
// 16 bytes per thread and pass (8 in flight per thread for large spans); a head / tail of single bytes for any alignment
__global__ __launch_bounds__(256) void copy_bytes_kernel(unsigned char* __restrict__ dst, const unsigned char* __restrict__ src,
                                                         size_t head, size_t n16, size_t nbytes) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, T = (size_t)gridDim.x * 256;
    const uint4* s16 = reinterpret_cast<const uint4*>(src + head);
    uint4* d16 = reinterpret_cast<uint4*>(dst + head);
    size_t i = t;
    for (; i + 3 * T < n16; i += 4 * T) {
        const uint4 a = s16[i], b = s16[i + T], c = s16[i + 2 * T], d = s16[i + 3 * T];
        d16[i] = a; d16[i + T] = b; d16[i + 2 * T] = c; d16[i + 3 * T] = d;
    }
    for (; i < n16; i += T) d16[i] = s16[i];
    if (t < head) dst[t] = src[t];
    const size_t tail0 = head + n16 * 16;
    if (tail0 + t < nbytes && t < 16) dst[tail0 + t] = src[tail0 + t];
}
}  // namespace

RECALGO_EXPORT int recalgo_copy_bytes(void* dst, const void* src, int64_t nbytes, recalgo_stream_t stream) {
    RECALGO_REQUIRE(nbytes >= 0 && (nbytes == 0 || (dst != nullptr && src != nullptr)));
    if (nbytes == 0) return 0;
    const uintptr_t d = reinterpret_cast<uintptr_t>(dst), s = reinterpret_cast<uintptr_t>(src);
    size_t head = 0, n16 = 0;
    if ((d & 15) == (s & 15)) {                       // same phase: vector body between a byte head and a byte tail
        head = (16 - (d & 15)) & 15;
        if (head > (size_t)nbytes) head = (size_t)nbytes;
        n16 = ((size_t)nbytes - head) / 16;
    } else {
        // different phase: whole span by the byte lanes (rare: every batch allocation of the library is 256-byte aligned);
        // handled as repeated launches of the byte paths below would be slow, so fall back to the runtime's copy
        return (int)hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyDeviceToDevice, as_stream(stream));
    }
    const size_t want = (n16 + 255) / 256;
    const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    hipLaunchKernelGGL(copy_bytes_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), static_cast<unsigned char*>(dst),
                       static_cast<const unsigned char*>(src), head, n16, (size_t)nbytes);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_abi_version(void) { return RECALGO_ABI_VERSION; }
RECALGO_EXPORT const char* recalgo_target_arch(void) { return "gfx950"; }

RECALGO_EXPORT int recalgo_sigmoid_ce_fwd_bwd(const float* logits, const float* labels, int B,
                                              float grad_scale, float* prob, float* loss,
                                              float* dlogit, recalgo_stream_t stream) {
    RECALGO_REQUIRE(B > 0);
    hipLaunchKernelGGL(sigmoid_ce_kernel, dim3(1), dim3(1024), 0, as_stream(stream), logits, labels,
                       (unsigned)B, grad_scale, prob, loss, dlogit);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_adam_tf1_advance(int64_t* step_dev, float lr, float beta1, float beta2,
                                            float* lr_t_dev, recalgo_stream_t stream) {
    RECALGO_REQUIRE(step_dev != nullptr && lr_t_dev != nullptr);
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, as_stream(stream), step_dev, lr, beta1,
                       beta2, lr_t_dev);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_adam_tf1_dense(float* p, float* g, float* m, float* v, int64_t n,
                                          float lr_t, const float* lr_t_dev, float beta1, float beta2,
                                          float eps, int zero_grad, recalgo_stream_t stream) {
    RECALGO_REQUIRE(n >= 0);
    if (n == 0) return 0;
    int64_t n4 = n / 4;
    int64_t want = (n4 + 255) / 256;
    int blocks = (int)(want < 1 ? 1 : (want > 256 * 16 ? 256 * 16 : want));
    hipLaunchKernelGGL(adam_tf1_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), p, g, m, v, n4,
                       n, lr_t, lr_t_dev, beta1, beta2, eps, zero_grad);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_adam_tf1_rows(float* p, float* g, float* m, float* v, unsigned char* row_live,
                                         int64_t rows, int K, float lr_t, const float* lr_t_dev, float beta1,
                                         float beta2, float eps, int zero_grad, recalgo_stream_t stream) {
    RECALGO_REQUIRE(rows >= 0 && row_live != nullptr);
    RECALGO_REQUIRE(K == 4 || K == 8 || K == 16 || K == 32 || K == 64);
    if (rows == 0) return 0;
    const int64_t total4 = rows * (K / 4);
    int64_t want = (total4 + 255) / 256;
    int blocks = (int)(want < 1 ? 1 : (want > 256 * 16 ? 256 * 16 : want));
    hipStream_t st = as_stream(stream);
#define LAUNCH(KK4)                                                                                       \
    hipLaunchKernelGGL(adam_tf1_rows_kernel<KK4>, dim3(blocks), dim3(256), 0, st, p, g, m, v, row_live, total4, \
                       lr_t, lr_t_dev, beta1, beta2, eps, zero_grad)
    switch (K / 4) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 4: LAUNCH(4); break;
        case 8: LAUNCH(8); break;
        default: LAUNCH(16); break;
    }
#undef LAUNCH
    RECALGO_RETURN_LAST();
}

// any row width (e.g. the K = 1 first-order weights of DeepFM): one float per thread
__global__ __launch_bounds__(256) void adam_tf1_list_scalar_kernel(float* __restrict__ p, float* __restrict__ g,
                                                                   float* __restrict__ m, float* __restrict__ v,
                                                                   const int* __restrict__ list,
                                                                   const int* __restrict__ count, unsigned K,
                                                                   float lr_t_val, const float* __restrict__ lr_t_dev,
                                                                   float b1, float b2, float eps, int zero_grad) {
    const float lr_t = lr_t_dev ? lr_t_dev[0] : lr_t_val;
    const int64_t total = (int64_t)count[0] * K;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride) {
        const int64_t i = (int64_t)list[t / K] * K + (t % K);
        float gg = g[i], mm = m[i], vv = v[i], pp = p[i];
        const bool nz = gg != 0.f;
        adam1(pp, gg, mm, vv, lr_t, b1, b2, eps);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (zero_grad && nz) g[i] = 0.f;
    }
}

RECALGO_EXPORT int recalgo_mark_live_rows(const int64_t* ids, const int64_t* row_base, int64_t n, int F,
                                          unsigned char* row_live, int* live_list, int* live_count,
                                          recalgo_stream_t stream) {
    RECALGO_REQUIRE(n >= 0 && F >= 1 && row_live != nullptr && live_list != nullptr && live_count != nullptr);
    RECALGO_REQUIRE((reinterpret_cast<uintptr_t>(row_live) & 3) == 0);
    if (n == 0) return 0;
    hipLaunchKernelGGL(mark_live_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), ids, row_base, n,
                       (unsigned)F, reinterpret_cast<unsigned*>(row_live), live_list, live_count);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int64_t recalgo_order_live_list_workspace_bytes(int64_t rows) {
    const int64_t n_words = (rows + 3) / 4;
    return ((n_words + kChunkWords - 1) / kChunkWords) * (int64_t)sizeof(int);
}

RECALGO_EXPORT int recalgo_order_live_list(const unsigned char* row_live, int64_t rows, int* live_list,
                                           int* live_count, void* workspace, recalgo_stream_t stream) {
    RECALGO_REQUIRE(rows >= 0 && row_live != nullptr && live_list != nullptr && live_count != nullptr);
    RECALGO_REQUIRE((reinterpret_cast<uintptr_t>(row_live) & 3) == 0);
    if (rows == 0) return 0;
    RECALGO_REQUIRE(workspace != nullptr);
    hipStream_t st = as_stream(stream);
    const int64_t n_words = (rows + 3) / 4;
    const int n_chunks = cdiv(n_words, kChunkWords);
    const unsigned* words = reinterpret_cast<const unsigned*>(row_live);
    int* chunk = static_cast<int*>(workspace);
    hipLaunchKernelGGL(live_chunk_count_kernel, dim3(n_chunks), dim3(256), 0, st, words, n_words, chunk);
    hipLaunchKernelGGL(live_chunk_scan_kernel, dim3(1), dim3(1024), 0, st, chunk, n_chunks, live_count);
    hipLaunchKernelGGL(live_chunk_write_kernel, dim3(n_chunks), dim3(256), 0, st, words, n_words, rows, chunk, live_list);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_adam_tf1_list(float* p, float* g, float* m, float* v, const int* live_list,
                                         const int* live_count, int64_t max_rows, int K, float lr_t,
                                         const float* lr_t_dev, float beta1, float beta2, float eps,
                                         int zero_grad, recalgo_stream_t stream) {
    RECALGO_REQUIRE(max_rows >= 0 && K >= 1 && live_list != nullptr && live_count != nullptr);
    if (max_rows == 0) return 0;
    hipStream_t st = as_stream(stream);
    if (!(K == 4 || K == 8 || K == 16 || K == 32 || K == 64)) {
        int64_t want1 = (max_rows * K + 255) / 256;
        int blocks1 = (int)(want1 < 1 ? 1 : (want1 > 256 * 8 ? 256 * 8 : want1));
        hipLaunchKernelGGL(adam_tf1_list_scalar_kernel, dim3(blocks1), dim3(256), 0, st, p, g, m, v, live_list,
                           live_count, (unsigned)K, lr_t, lr_t_dev, beta1, beta2, eps, zero_grad);
        RECALGO_RETURN_LAST();
    }
    // the launch is sized for the largest possible list (graph replayable); surplus workgroups exit at once
    int64_t want = (max_rows * (K / 4) + 255) / 256;
    int blocks = (int)(want < 1 ? 1 : (want > 256 * 8 ? 256 * 8 : want));
#define LAUNCH(KK4)                                                                                         \
    hipLaunchKernelGGL(adam_tf1_list_kernel<KK4>, dim3(blocks), dim3(256), 0, st, p, g, m, v, live_list, live_count, \
                       lr_t, lr_t_dev, beta1, beta2, eps, zero_grad)
    switch (K / 4) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 4: LAUNCH(4); break;
        case 8: LAUNCH(8); break;
        default: LAUNCH(16); break;
    }
#undef LAUNCH
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_activation_fwd(const float* x, const float* alpha, int rows, int C,
                                          int kind, float* y, recalgo_stream_t stream) {
    RECALGO_REQUIRE(rows >= 0 && C > 0 && (kind == RECALGO_ACT_PRELU || kind == RECALGO_ACT_DICE));
    int64_t n = (int64_t)rows * C;
    if (n == 0) return 0;
    if (kind == RECALGO_ACT_DICE)
        hipLaunchKernelGGL(act_fwd_kernel<true>, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), x,
                           alpha, n, (unsigned)C, y);
    else
        hipLaunchKernelGGL(act_fwd_kernel<false>, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), x,
                           alpha, n, (unsigned)C, y);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int64_t recalgo_activation_bwd_workspace_bytes(int rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    return (int64_t)cdiv(rows, kActRowsPerBlk) * C * (int64_t)sizeof(float);
}

RECALGO_EXPORT int recalgo_activation_bwd_partial_rows(int rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    return C % 4 == 0 ? cdiv(rows, kActTileRows) : cdiv(rows, kActRowsPerBlk);
}
RECALGO_EXPORT int recalgo_activation_bwd(const float* x, const float* alpha, const float* gy,
                                          int rows, int C, int kind, float* dx, float* dalpha,
                                          void* workspace, recalgo_stream_t stream) {
    RECALGO_REQUIRE(rows > 0 && C > 0 && workspace != nullptr);
    RECALGO_REQUIRE(kind == RECALGO_ACT_PRELU || kind == RECALGO_ACT_DICE);
    float* partial = static_cast<float*>(workspace);
    hipStream_t st = as_stream(stream);
    if (C % 4 == 0) {
        const unsigned nt = (unsigned)cdiv(rows, kActTileRows), C4 = (unsigned)C / 4;
        const dim3 grid(cdiv(C4, 16), nt);
        if (kind == RECALGO_ACT_DICE)
            hipLaunchKernelGGL(act_bwd_tile_kernel<true>, grid, dim3(256), 0, st, reinterpret_cast<const float4*>(x),
                               reinterpret_cast<const float4*>(alpha), reinterpret_cast<const float4*>(gy),
                               (unsigned)rows, C4, reinterpret_cast<float4*>(dx), reinterpret_cast<float4*>(partial));
        else
            hipLaunchKernelGGL(act_bwd_tile_kernel<false>, grid, dim3(256), 0, st, reinterpret_cast<const float4*>(x),
                               reinterpret_cast<const float4*>(alpha), reinterpret_cast<const float4*>(gy),
                               (unsigned)rows, C4, reinterpret_cast<float4*>(dx), reinterpret_cast<float4*>(partial));
        if (dalpha) launch_colsum16(partial, nt, (unsigned)C, dalpha, (unsigned)C, static_cast<float*>(nullptr), st);
        RECALGO_RETURN_LAST();
    }
    const unsigned nblk = (unsigned)cdiv(rows, kActRowsPerBlk);
    if (kind == RECALGO_ACT_DICE)
        hipLaunchKernelGGL(act_bwd_kernel<true>, dim3(nblk), dim3(256), 0, st, x, alpha, gy, (unsigned)rows,
                           (unsigned)C, kActRowsPerBlk, dx, partial);
    else
        hipLaunchKernelGGL(act_bwd_kernel<false>, dim3(nblk), dim3(256), 0, st, x, alpha, gy, (unsigned)rows,
                           (unsigned)C, kActRowsPerBlk, dx, partial);
    if (dalpha) launch_colsum16(partial, nblk, (unsigned)C, dalpha, (unsigned)C, static_cast<float*>(nullptr), st);
    RECALGO_RETURN_LAST();
}

RECALGO_EXPORT int recalgo_adam_tf1_step(float* p, float* g, float* m, float* v, int64_t n,
                                         const recalgo_adam_arena_t* arenas, int n_arenas, int64_t* step_dev,
                                         int* ticket_dev, int advance, float lr, float beta1, float beta2, float eps,
                                         int zero_grad, recalgo_stream_t stream) {
    return recalgo_adam_tf1_step_plans(p, g, m, v, n, arenas, n_arenas, step_dev, ticket_dev, advance, lr, beta1, beta2, eps,
                                       zero_grad, nullptr, 0, stream);
}

RECALGO_EXPORT int recalgo_adam_tf1_step_plans(float* p, float* g, float* m, float* v, int64_t n,
                                               const recalgo_adam_arena_t* arenas, int n_arenas, int64_t* step_dev,
                                               int* ticket_dev, int advance, float lr, float beta1, float beta2, float eps,
                                               int zero_grad, const recalgo_plan_scan_t* scans, int n_scans,
                                               recalgo_stream_t stream) {
    RECALGO_REQUIRE(n >= 0 && n_arenas >= 0 && n_arenas <= kAdamMaxArenas && step_dev != nullptr);
    RECALGO_REQUIRE(n_scans >= 0 && n_scans <= kAdamMaxArenas && (n_scans == 0 || scans != nullptr));
    RECALGO_REQUIRE(!advance || ticket_dev != nullptr);
    RECALGO_REQUIRE(n == 0 || (p && g && m && v));
    RECALGO_REQUIRE(n_arenas == 0 || arenas != nullptr);
    AdamStepArgs A;
    A.p = p; A.g = g; A.m = m; A.v = v; A.n = n;
    const int64_t want = (n / 4 + 255) / 256;
    A.dense_blocks = n == 0 ? 0u : (unsigned)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
    unsigned blocks = A.dense_blocks;
    A.n_arenas = 0;
    for (int i = 0; i < n_arenas; ++i) {
        const recalgo_adam_arena_t& a = arenas[i];
        if (a.max_rows <= 0) continue;
        RECALGO_REQUIRE(a.p && a.g && a.m && a.v && a.live_list && a.live_count && a.K >= 1);
        AdamArena& R = A.ar[A.n_arenas++];
        R.p = a.p; R.g = a.g; R.m = a.m; R.v = a.v; R.list = a.live_list; R.count = a.live_count; R.K = a.K;
        R.k4_shift = -1;
        R.lazy = a.lazy;
        if ((a.K & 3) == 0)
            for (int sft = 0; sft < 16; ++sft)
                if ((a.K >> 2) == (1 << sft)) R.k4_shift = sft;
        // the lazy vote of this (live-list) path is per row only when the K/4 lanes of a row are an aligned power-of-two
        // group of one wave; other widths would vote per float4 / per element, which is not LazyAdam (the owner-computes
        // path, recalgo_scatter_apply RECALGO_SCATTER_LAZY_ADAM, has no such limit and is what the host uses)
        RECALGO_REQUIRE(!a.lazy || (R.k4_shift >= 0 && (a.K >> 2) <= 64));
        // sized for the largest possible list (graph replayable); surplus workgroups find nothing to do
        const int64_t per = (a.K & 3) == 0 ? a.max_rows * (a.K / 4) : a.max_rows * a.K;
        const int64_t w = (per + 255) / 256;
        R.n_blocks = (unsigned)(w < 1 ? 1 : (w > 2048 ? 2048 : w));
        R.first_block = blocks;
        blocks += R.n_blocks;
    }
    for (int i = A.n_arenas; i < kAdamMaxArenas; ++i) A.ar[i] = AdamArena{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 4, 0, 0, 0xffffffffu, 0};
    A.step = reinterpret_cast<long long*>(step_dev); A.ticket = ticket_dev; A.advance = advance;
    A.lr = lr; A.b1 = beta1; A.b2 = beta2; A.eps = eps; A.zero_grad = zero_grad;
    if (blocks == 0) blocks = 1;                      // still advances the step counter
    A.scan_first = blocks;
    for (int i = 0; i < n_scans; ++i) {
        const recalgo_plan_scan_t& c = scans[i];
        RECALGO_REQUIRE(c.total && c.offs && c.sched && c.nb_log2 >= 8 && c.nb_log2 <= 13 && c.counter_shift <= 5);
        A.scan[i] = recalgo_plan::Scan{c.total, c.offs, static_cast<uint4*>(c.sched), c.counter_shift, c.nb_log2};
    }
    for (int i = n_scans; i < kAdamMaxArenas; ++i) A.scan[i] = recalgo_plan::Scan{nullptr, nullptr, nullptr, 0, 8};
    blocks += (unsigned)n_scans;
    hipLaunchKernelGGL(adam_tf1_step_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), A);
    RECALGO_RETURN_LAST();
}
