// Kernel lab for the fp32-MFMA tile engine (csrc/tile_v2.h): a standalone binary (no torch: starts in a second on a GPU box)
// that checks the register-blocked main loop against a naive fp32 GEMM and times it per layer shape and tile configuration
// beside the round-2 engine of librecalgo_hip.so.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude scripts/mfma_lab.hip -Lrecalgorithm_amd -lrecalgo_hip \
//         -Wl,-rpath,'$ORIGIN/../recalgorithm_amd' -o scripts/mfma_lab.bin
//   scripts/mfma_lab.bin [M]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../recalgorithm_amd/csrc/tile_v2.h"

using namespace tv2;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------------
struct FwdP {
    Operand a, b;      // a = X [M][K] RC, b = W [K][N] RM
    const float* bias;
    float* y;
    int M, N, K, relu;
    unsigned long long* trace;     // optional: [grid][8] timestamps (wall clock, 100 MHz) and shader cycles
};

__device__ __forceinline__ void stamp(unsigned long long* trace, int slot) {
    if (trace != nullptr && threadIdx.x == 0) {
        trace[(size_t)blockIdx.x * 8 + slot] = wall_clock64();
        trace[(size_t)blockIdx.x * 8 + 4 + slot] = clock64();
    }
}

// EPI: 0 = bias loaded after the main loop, one scalar (WN = 2: float2) store per accumulator register;
//      1 = the same with the bias requested before the main loop;
//      2 = bias before the main loop, the wave's tile transposed through LDS and written as whole 128 / 256-byte row
//          segments (one dwordx4 per lane and instruction)
template <int WM, int WN, int EPI>
__global__ __launch_bounds__(kThreads) void fwd_kernel(FwdP P) {
    stamp(P.trace, 0);
    using GA = Geom<true, WM>;
    using GB = Geom<false, WN>;
    __shared__ __attribute__((aligned(16))) float As[kStages * GA::kFloats];
    __shared__ __attribute__((aligned(16))) float Bs[kStages * GB::kFloats];
    const int tn = (P.N + GB::T - 1) / GB::T;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int m0 = (tile / tn) * GA::T, n0 = (tile % tn) * GB::T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, l32 = lane & 31;
    const int rb = m0 + (wave >> 1) * 32 * WM, cb = n0 + (wave & 1) * 32 * WN;
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // EPI 2: lane -> (row group, float4 column) of the transposed tile: CW = 32 * WN columns = 8 * WN float4 per row
    constexpr int CW = 32 * WN, Q = CW / 4, RPI = 64 / Q;        // rows per store instruction
    const int tq = lane % Q, tr = lane / Q;
    float4 bias4 = f4_zero();
    float bv[WN];
    if constexpr (EPI == 2) {
        if (P.bias && cb + 4 * tq + 3 < P.N) bias4 = *reinterpret_cast<const float4*>(P.bias + cb + 4 * tq);
    } else if constexpr (EPI == 1) {
#pragma unroll
        for (int j = 0; j < WN; ++j) bv[j] = (P.bias && cb + WN * l32 + j < P.N) ? P.bias[cb + WN * l32 + j] : 0.f;
    }
    float4 unused = f4_zero();
    if (P.K % BK == 0) mainloop<true, false, WM, WN, false, false, false, true>(P.a, P.b, m0, n0, P.M, P.N, 0, P.K, As, Bs, acc, unused);
    else mainloop<true, false, WM, WN, false, false, false, false>(P.a, P.b, m0, n0, P.M, P.N, 0, P.K, As, Bs, acc, unused);
    stamp(P.trace, 1);
    if constexpr (EPI == 2) {
        // the ring is free after the main loop's last barrier; every wave uses its own 32 x (CW + 4) floats, WM times
        constexpr int LD = CW + 4;
        constexpr bool useB = GB::kFloats > GA::kFloats;
        float* T = (useB ? Bs : As) + wave * (32 * LD);
        static_assert(4 * 32 * LD <= kStages * (useB ? GB::kFloats : GA::kFloats), "transpose buffer fits an operand ring");
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = acc_row(r, hi);
#pragma unroll
                for (int j = 0; j < WN; ++j) T[row * LD + WN * l32 + j] = acc[i][j][r];
            }
            // (same wave wrote and reads: no barrier, the LDS operations of a wave complete in order)
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int row = it * RPI + tr;
                float4 v = *reinterpret_cast<const float4*>(T + row * LD + 4 * tq);
                v = f4_add(v, bias4);
                if (P.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                const int grow = rb + 32 * i + row;
                if (grow < P.M && cb + 4 * tq + 3 < P.N) *reinterpret_cast<float4*>(P.y + (size_t)grow * P.N + cb + 4 * tq) = v;
            }
        }
        stamp(P.trace, 2);
        return;
    }
    // columns of this lane: cb + WN * l32 + j (j < WN): WN consecutive floats
    const int c0 = cb + WN * l32;
    if (c0 >= P.N) return;
    if constexpr (EPI == 0) {
#pragma unroll
        for (int j = 0; j < WN; ++j) bv[j] = P.bias ? P.bias[c0 + j] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb + idx_of<true, WM>(i, acc_row(r, hi));
            if (row < P.M) {
                float v[WN];
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    v[j] = acc[i][j][r] + bv[j];
                    if (P.relu) v[j] = fmaxf(v[j], 0.f);
                }
                float* o = P.y + (size_t)row * P.N + c0;
                if constexpr (WN == 1) o[0] = v[0];
                else if constexpr (WN == 2) *reinterpret_cast<float2*>(o) = make_float2(v[0], v[1]);
                else *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
    stamp(P.trace, 2);
}

struct DgradP {
    Operand a, b;      // a = G (+mask Y) [M][N] RC, b = W [K][N] RC
    float* dx;
    int M, N, K;
};

template <int WM, int WN, bool MASK>
__global__ __launch_bounds__(kThreads) void dgrad_kernel(DgradP P) {
    using GA = Geom<true, WM>;
    using GB = Geom<true, WN>;
    __shared__ __attribute__((aligned(16))) float As[kStages * GA::kFloats];
    __shared__ __attribute__((aligned(16))) float Bs[kStages * GB::kFloats];
    const int tn = (P.K + GB::T - 1) / GB::T;
    const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
    const int m0 = (tile / tn) * GA::T, n0 = (tile % tn) * GB::T;
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 unused = f4_zero();
    if (P.N % BK == 0) mainloop<true, true, WM, WN, MASK, false, false, true>(P.a, P.b, m0, n0, P.M, P.K, 0, P.N, As, Bs, acc, unused);
    else mainloop<true, true, WM, WN, MASK, false, false, false>(P.a, P.b, m0, n0, P.M, P.K, 0, P.N, As, Bs, acc, unused);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, l32 = lane & 31;
    const int rb = m0 + (wave >> 1) * 32 * WM, cb = n0 + (wave & 1) * 32 * WN;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int col = cb + idx_of<true, WN>(j, l32);
        if (col >= P.K) continue;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb + idx_of<true, WM>(i, acc_row(r, hi));
                if (row < P.M) P.dx[(size_t)row * P.K + col] = acc[i][j][r];
            }
        }
    }
}

struct WgradP {
    Operand a, b;      // a = X [M][K] RM, b = G (+mask) [M][N] RM
    float* out;        // [splits][K*N + N]
    int M, N, K, splits, rows_per_split;
    size_t slab;
};

template <int WM, int WN, bool MASK>
__global__ __launch_bounds__(kThreads) void wgrad_kernel(WgradP P) {
    using GA = Geom<false, WM>;
    using GB = Geom<false, WN>;
    __shared__ __attribute__((aligned(16))) float As[kStages * GA::kFloats];
    __shared__ __attribute__((aligned(16))) float Bs[kStages * GB::kFloats];
    const int tn = (P.N + GB::T - 1) / GB::T, tm = (P.K + GA::T - 1) / GA::T;
    const int ntiles = tn * tm;
    const int l = xcd_swizzle(blockIdx.x, gridDim.x);
    const int split = l / ntiles, tile = l % ntiles;
    const int m0 = (tile / tn) * GA::T, n0 = (tile % tn) * GB::T;
    const int r_begin = split * P.rows_per_split;
    const int r_end = min(P.M, r_begin + P.rows_per_split);
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 colsum = f4_zero();
    const bool do_bias = m0 == 0;
    const bool exact = (r_end - r_begin) % BK == 0;
    if (do_bias) {
        if (exact) mainloop<false, false, WM, WN, false, MASK, true, true>(P.a, P.b, m0, n0, P.K, P.N, r_begin, r_end, As, Bs, acc, colsum);
        else mainloop<false, false, WM, WN, false, MASK, true, false>(P.a, P.b, m0, n0, P.K, P.N, r_begin, r_end, As, Bs, acc, colsum);
    } else {
        if (exact) mainloop<false, false, WM, WN, false, MASK, false, true>(P.a, P.b, m0, n0, P.K, P.N, r_begin, r_end, As, Bs, acc, colsum);
        else mainloop<false, false, WM, WN, false, MASK, false, false>(P.a, P.b, m0, n0, P.K, P.N, r_begin, r_end, As, Bs, acc, colsum);
    }
    float* base = P.out + (size_t)split * P.slab;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hi = lane >> 5, l32 = lane & 31;
    const int rb = m0 + (wave >> 1) * 32 * WM, cb = n0 + (wave & 1) * 32 * WN;
    const int c0 = cb + WN * l32;
    if (c0 < P.N) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb + idx_of<false, WM>(i, acc_row(r, hi));
                if (row < P.K) {
                    float* o = base + (size_t)row * P.N + c0;
                    if constexpr (WN == 1) o[0] = acc[i][0][r];
                    else *reinterpret_cast<float2*>(o) = make_float2(acc[i][0][r], acc[i][1][r]);
                }
            }
        }
    }
    if (do_bias) {
        // thread t staged columns (t % U) * 4 .. + 3 of rows t / U + (256 / U) j: reduce over the threads of a column group
        constexpr int U = GB::U, R = kThreads / U;
        float4* sh = reinterpret_cast<float4*>(As);
        __syncthreads();
        sh[threadIdx.x] = colsum;
        __syncthreads();
        if ((int)threadIdx.x < U) {
            float4 t = sh[threadIdx.x];
#pragma unroll
            for (int k = 1; k < R; ++k) t = f4_add(t, sh[k * U + threadIdx.x]);
            float* db = base + (size_t)P.K * P.N;
            const int c = n0 + threadIdx.x * 4;
            if (c + 3 < P.N) *reinterpret_cast<float4*>(db + c) = t;
        }
    }
}

// naive references (fp32 fmaf chains in k order)
__global__ void ref_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int relu) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float a = 0.f;
    for (int k = 0; k < K; ++k) a = fmaf(x[(size_t)m * K + k], w[(size_t)k * N + n], a);
    a += bias[n];
    y[(size_t)m * N + n] = relu ? fmaxf(a, 0.f) : a;
}
__global__ void ref_dgrad(const float* g, const float* ymask, const float* w, float* dx, int M, int N, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (k >= K) return;
    float a = 0.f;
    for (int n = 0; n < N; ++n) {
        const float gv = ymask[(size_t)m * N + n] > 0.f ? g[(size_t)m * N + n] : 0.f;
        a = fmaf(gv, w[(size_t)k * N + n], a);
    }
    dx[(size_t)m * K + k] = a;
}
__global__ void ref_wgrad(const float* x, const float* g, const float* ymask, float* dw, float* db, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;     // k == K: the bias row
    if (n >= N) return;
    double a = 0.0;
    for (int m = 0; m < M; ++m) {
        const float gv = ymask[(size_t)m * N + n] > 0.f ? g[(size_t)m * N + n] : 0.f;
        a += (double)(k < K ? x[(size_t)m * K + k] : 1.0f) * (double)gv;
    }
    if (k < K) dw[(size_t)k * N + n] = (float)a;
    else db[n] = (float)a;
}
__global__ void sum_slabs(const float* partials, float* out, size_t slab, size_t n, int S) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += partials[(size_t)s * slab + i];
    out[i] = a;
}

// the round-2 engine (librecalgo_hip.so): declared by include/recalgo.h (via common.h)

// ---------------------------------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------------------------------
static float* dev_rand(size_t n, unsigned seed, float scale, bool integer = false) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        const float u = (float)((s >> 8) & 0xffff) / 65536.0f - 0.5f;
        h[i] = integer ? (float)((int)(u * 8.0f)) : u * 2.0f * scale;
    }
    float* d;
    CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return d;
}
static float* dev_zero(size_t n) {
    float* d;
    CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemset(d, 0, n * sizeof(float)));
    return d;
}
static Operand operand(const float* p, const float* mask, int ld, size_t rows, size_t cols) {
    return Operand{p, mask, ld, (int)(((rows - 1) * ld + cols) * sizeof(float))};
}
// worst |a - b| / (|b| + rms(b))
static double compare(const float* a_dev, const float* b_dev, size_t n) {
    std::vector<float> a(n), b(n);
    CK(hipMemcpy(a.data(), a_dev, n * sizeof(float), hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), b_dev, n * sizeof(float), hipMemcpyDeviceToHost));
    double ss = 0;
    for (size_t i = 0; i < n; ++i) ss += (double)b[i] * b[i];
    const double rms = std::sqrt(ss / (double)n);
    double worst = 0;
    for (size_t i = 0; i < n; ++i) {
        const double e = std::fabs((double)a[i] - b[i]) / (std::fabs((double)b[i]) + rms + 1e-30);
        if (!(e <= worst)) worst = e;       // (NaN-propagating)
    }
    return worst;
}
template <class F>
static double time_us(F&& f, int iters = 40) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f();
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) f();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms * 1e3 / iters);
    }
    CK(hipGetLastError());
    return best;
}

struct Row { std::string what; double us, tflops, err; };
static std::vector<Row> rows;
static void report(const std::string& what, double us, double flop, double err) {
    rows.push_back({what, us, flop / us * 1e-6, err});
    printf("| %-58s | %8.2f | %7.1f | %5.3f | %9.2e |\n", what.c_str(), us, flop / us * 1e-6, flop / us * 1e-6 / 157.3, err);
    fflush(stdout);
}

template <int WM, int WN, int EPI = 0>
static void run_fwd(const char* tag, int M, int K, int N, const float* x, const float* w, const float* b, float* y, const float* yref) {
    FwdP P{operand(x, nullptr, K, M, K), operand(w, nullptr, N, K, N), b, y, M, N, K, 1, nullptr};
    const int grid = ((M + 64 * WM - 1) / (64 * WM)) * ((N + 64 * WN - 1) / (64 * WN));
    CK(hipMemset(y, 0, (size_t)M * N * sizeof(float)));
    hipLaunchKernelGGL((fwd_kernel<WM, WN, EPI>), dim3(grid), dim3(kThreads), 0, 0, P);
    CK(hipDeviceSynchronize());
    const double err = yref ? compare(y, yref, (size_t)M * N) : -1.0;
    const double us = time_us([&] { hipLaunchKernelGGL((fwd_kernel<WM, WN, EPI>), dim3(grid), dim3(kThreads), 0, 0, P); });
    char buf[160];
    snprintf(buf, sizeof buf, "%s fwd v2<%d,%d> epi %d %dx%dx%d grid %d", tag, WM, WN, EPI, M, K, N, grid);
    report(buf, us, 2.0 * M * K * N, err);
}
template <int WM, int WN, bool MASK = true>
static void run_dgrad(const char* tag, int M, int K, int N, const float* g, const float* ymask, const float* w, float* dx, const float* ref) {
    DgradP P{operand(g, ymask, N, M, N), operand(w, nullptr, N, K, N), dx, M, N, K};
    const int grid = ((M + 64 * WM - 1) / (64 * WM)) * ((K + 64 * WN - 1) / (64 * WN));
    CK(hipMemset(dx, 0, (size_t)M * K * sizeof(float)));
    hipLaunchKernelGGL((dgrad_kernel<WM, WN, MASK>), dim3(grid), dim3(kThreads), 0, 0, P);
    CK(hipDeviceSynchronize());
    const double err = ref ? compare(dx, ref, (size_t)M * K) : -1.0;
    const double us = time_us([&] { hipLaunchKernelGGL((dgrad_kernel<WM, WN, MASK>), dim3(grid), dim3(kThreads), 0, 0, P); });
    char buf[160];
    snprintf(buf, sizeof buf, "%s dgrad v2<%d,%d> mask %d %dx%dx%d grid %d", tag, WM, WN, (int)MASK, M, K, N, grid);
    report(buf, us, 2.0 * M * K * N, err);
}
template <int WM, int WN, bool MASK = true>
static void run_wgrad(const char* tag, int M, int K, int N, int splits, const float* x, const float* g, const float* ymask, float* ws,
                      float* dw, const float* ref_dw, const float* ref_db) {
    WgradP P;
    P.a = operand(x, nullptr, K, M, K);
    P.b = operand(g, ymask, N, M, N);
    P.out = ws; P.M = M; P.N = N; P.K = K; P.splits = splits;
    P.rows_per_split = ((M + splits - 1) / splits + BK - 1) / BK * BK;
    P.slab = ((size_t)K * N + N + 3) / 4 * 4;
    const int grid = ((K + 64 * WM - 1) / (64 * WM)) * ((N + 64 * WN - 1) / (64 * WN)) * splits;
    CK(hipMemset(ws, 0, P.slab * splits * sizeof(float)));
    hipLaunchKernelGGL((wgrad_kernel<WM, WN, MASK>), dim3(grid), dim3(kThreads), 0, 0, P);
    const size_t n = (size_t)K * N + N;
    hipLaunchKernelGGL(sum_slabs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ws, dw, P.slab, n, splits);
    CK(hipDeviceSynchronize());
    double err = -1.0;
    if (ref_dw) err = std::max(compare(dw, ref_dw, (size_t)K * N), compare(dw + (size_t)K * N, ref_db, (size_t)N));
    const double us = time_us([&] { hipLaunchKernelGGL((wgrad_kernel<WM, WN, MASK>), dim3(grid), dim3(kThreads), 0, 0, P); });
    char buf[160];
    snprintf(buf, sizeof buf, "%s wgrad v2<%d,%d> mask %d %dx%dx%d splits %d grid %d", tag, WM, WN, (int)MASK, M, K, N, splits, grid);
    report(buf, us, 2.0 * M * K * N, err);
}

// timeline of three back-to-back launches: when do workgroups start, leave the main loop, finish?
template <int WM, int WN, int EPI = 0>
static void trace_fwd(int M, int K, int N) {
    float* x = dev_rand((size_t)M * K, 1, 1.0f);
    float* w = dev_rand((size_t)K * N, 2, 1.0f / std::sqrt((float)K));
    float* b = dev_rand(N, 3, 0.5f);
    float* y = dev_zero((size_t)M * N);
    const int grid = ((M + 64 * WM - 1) / (64 * WM)) * ((N + 64 * WN - 1) / (64 * WN));
    const int L = 4;
    unsigned long long* tr;
    CK(hipMalloc(&tr, (size_t)L * grid * 8 * sizeof(unsigned long long)));
    CK(hipMemset(tr, 0, (size_t)L * grid * 8 * sizeof(unsigned long long)));
    FwdP P{operand(x, nullptr, K, M, K), operand(w, nullptr, N, K, N), b, y, M, N, K, 1, nullptr};
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fwd_kernel<WM, WN, EPI>), dim3(grid), dim3(kThreads), 0, 0, P);
    CK(hipDeviceSynchronize());
    for (int l = 0; l < L; ++l) {
        P.trace = tr + (size_t)l * grid * 8;
        hipLaunchKernelGGL((fwd_kernel<WM, WN, EPI>), dim3(grid), dim3(kThreads), 0, 0, P);
    }
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)L * grid * 8);
    CK(hipMemcpy(h.data(), tr, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    printf("\n### timeline fwd v2<%d,%d> epi %d %dx%dx%d grid %d (wall clock, us relative to the first start of launch 0; cycles = shader clock)\n", WM, WN, EPI, M, K, N, grid);
    printf("| launch | first start | last start | first loop-end | median loop-end | last loop-end | first end | last end | median cycles start->loop-end | median cycles loop-end->end |\n|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n");
    unsigned long long t00 = ~0ull;
    for (int g = 0; g < grid; ++g) t00 = std::min(t00, h[(size_t)g * 8]);
    for (int l = 0; l < L; ++l) {
        std::vector<double> s0, s1, s2, c01, c12;
        for (int g = 0; g < grid; ++g) {
            const unsigned long long* r = &h[((size_t)l * grid + g) * 8];
            s0.push_back((double)(r[0] - t00) * 0.01);
            s1.push_back((double)(r[1] - t00) * 0.01);
            s2.push_back((double)(r[2] - t00) * 0.01);
            c01.push_back((double)(r[5] - r[4]));
            c12.push_back((double)(r[6] - r[5]));
        }
        auto srt = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); };
        srt(s0); srt(s1); srt(s2); srt(c01); srt(c12);
        printf("| %d | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.0f | %.0f |\n", l, s0.front(), s0.back(), s1.front(), s1[grid / 2], s1.back(),
               s2.front(), s2.back(), c01[grid / 2], c12[grid / 2]);
    }
    for (float* p : {x, w, b, y}) CK(hipFree(p));
    CK(hipFree(tr));
}

static void layer(int M, int K, int N, bool check) {
    printf("\n### layer %d x %d -> %d\n| kernel | us | TFLOP/s | of 157.3 | max rel err |\n|---|---:|---:|---:|---:|\n", M, K, N);
    float* x = dev_rand((size_t)M * K, 1, 1.0f);
    float* w = dev_rand((size_t)K * N, 2, 1.0f / std::sqrt((float)K));
    float* b = dev_rand(N, 3, 0.5f);
    float* g = dev_rand((size_t)M * N, 4, 1.0f);
    float* y = dev_zero((size_t)M * N);
    float* yref = dev_zero((size_t)M * N);
    float* dx = dev_zero((size_t)M * K);
    float* dxref = dev_zero((size_t)M * K);
    float* dw = dev_zero((size_t)K * N + N + 16);
    float* dwref = dev_zero((size_t)K * N + N);
    const size_t slab = ((size_t)K * N + N + 3) / 4 * 4;
    float* ws = dev_zero(slab * 64);
    const long long wsb = recalgo_dense_bwd_weights_workspace_bytes(M, K, N);
    float* ws1 = dev_zero((size_t)wsb / 4 + 16);
    if (check) {
        hipLaunchKernelGGL(ref_fwd, dim3((N + 63) / 64, M), dim3(64), 0, 0, x, w, b, yref, M, N, K, 1);
        hipLaunchKernelGGL(ref_dgrad, dim3((K + 63) / 64, M), dim3(64), 0, 0, g, yref, w, dxref, M, N, K);
        hipLaunchKernelGGL(ref_wgrad, dim3((N + 63) / 64, K + 1), dim3(64), 0, 0, x, g, yref, dwref, dwref + (size_t)K * N, M, N, K);
        CK(hipDeviceSynchronize());
    } else {
        recalgo_dense_fwd(x, K, w, K, nullptr, 0, nullptr, 0, b, M, N, 1, yref, N, nullptr);
        CK(hipDeviceSynchronize());
    }
    const float* yr = check ? yref : nullptr;
    const float* dxr = check ? dxref : nullptr;
    const float* dwr = check ? dwref : nullptr;
    const float* dbr = check ? dwref + (size_t)K * N : nullptr;
    const double fl = 2.0 * M * K * N;
    // ---- round-2 engine ----
    {
        recalgo_dense_fwd(x, K, w, K, nullptr, 0, nullptr, 0, b, M, N, 1, y, N, nullptr);
        CK(hipDeviceSynchronize());
        const double e = yr ? compare(y, yr, (size_t)M * N) : -1.0;
        report("v1 fwd", time_us([&] { recalgo_dense_fwd(x, K, w, K, nullptr, 0, nullptr, 0, b, M, N, 1, y, N, nullptr); }), fl, e);
        recalgo_dense_bwd_input(g, N, yref, w, M, N, K, nullptr, 0, 0.f, dx, K, 0, nullptr);
        CK(hipDeviceSynchronize());
        const double e2 = dxr ? compare(dx, dxr, (size_t)M * K) : -1.0;
        report("v1 dgrad", time_us([&] { recalgo_dense_bwd_input(g, N, yref, w, M, N, K, nullptr, 0, 0.f, dx, K, 0, nullptr); }), fl, e2);
        report("v1 wgrad (slabs, no reduce)", time_us([&] { recalgo_dense_bwd_weights(x, K, g, N, yref, M, K, N, dw, dw + (size_t)K * N, ws1, 1, nullptr); }), fl, -1.0);
        report("v1 bwd merged (dgrad + wgrad, no reduce)",
               time_us([&] { recalgo_dense_bwd(x, K, g, N, yref, w, M, K, N, nullptr, 0, 0.f, dx, K, dw, dw + (size_t)K * N, ws1, 1, nullptr); }), 2 * fl, -1.0);
    }
    // ---- v2 ----
    run_fwd<1, 1>("", M, K, N, x, w, b, y, yr);
    run_fwd<1, 1, 1>("", M, K, N, x, w, b, y, yr);
    run_fwd<1, 1, 2>("", M, K, N, x, w, b, y, yr);
    run_fwd<1, 2>("", M, K, N, x, w, b, y, yr);
    run_fwd<1, 2, 2>("", M, K, N, x, w, b, y, yr);
    run_fwd<2, 1>("", M, K, N, x, w, b, y, yr);
    run_fwd<2, 2>("", M, K, N, x, w, b, y, yr);
    run_dgrad<1, 1>("", M, K, N, g, yref, w, dx, dxr);
    run_dgrad<1, 2>("", M, K, N, g, yref, w, dx, dxr);
    run_dgrad<2, 1>("", M, K, N, g, yref, w, dx, dxr);
    run_dgrad<2, 2>("", M, K, N, g, yref, w, dx, dxr);
    for (int splits : {1, 4, 8, 16, 32}) {
        if (splits == 1 ? M > 1024 : (M / splits) < 4 * BK) continue;
        run_wgrad<1, 1>("", M, K, N, splits, x, g, yref, ws, dw, dwr, dbr);
        run_wgrad<1, 2>("", M, K, N, splits, x, g, yref, ws, dw, dwr, dbr);
        run_wgrad<2, 2>("", M, K, N, splits, x, g, yref, ws, dw, dwr, dbr);
    }
    for (float* p : {x, w, b, g, y, yref, dx, dxref, dw, dwref, ws, ws1}) CK(hipFree(p));
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096;
    const bool trace_only = argc > 2 && std::string(argv[2]) == "trace";
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("# mfma_lab on %s (%d CUs, %d MHz)\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    // exactness on small integers (any summation order gives the same fp32 result) at an odd shape, then the MLP shapes
    if (trace_only) trace_fwd<1, 1, 0>(M, 416, 512);
    if (trace_only) trace_fwd<1, 1, 1>(M, 416, 512);
    if (trace_only) trace_fwd<1, 1, 2>(M, 416, 512);
    if (trace_only) trace_fwd<1, 2, 0>(M, 416, 512);
    if (trace_only) trace_fwd<1, 2, 2>(M, 416, 512);
    if (trace_only) trace_fwd<1, 1, 2>(M, 512, 256);
    if (trace_only) trace_fwd<1, 1, 0>(M, 256, 128);
    if (trace_only) trace_fwd<1, 1, 2>(M, 256, 128);
    if (trace_only) return 0;
    if (argc > 2 && std::string(argv[2]) == "mask") {
        for (int K : {416, 1664}) {
            const int N = 512;
            printf("\n### %d x %d -> %d: mask cost\n| kernel | us | TFLOP/s | of 157.3 | max rel err |\n|---|---:|---:|---:|---:|\n", M, K, N);
            float* x = dev_rand((size_t)M * K, 1, 1.0f);
            float* w = dev_rand((size_t)K * N, 2, 1.0f / std::sqrt((float)K));
            float* g = dev_rand((size_t)M * N, 4, 1.0f);
            float* y = dev_rand((size_t)M * N, 5, 1.0f);
            float* dx = dev_zero((size_t)M * K);
            float* dw = dev_zero((size_t)K * N + N + 16);
            float* ws = dev_zero((((size_t)K * N + N + 3) / 4 * 4) * 16);
            run_dgrad<1, 1, true>("", M, K, N, g, y, w, dx, nullptr);
            run_dgrad<1, 1, false>("", M, K, N, g, y, w, dx, nullptr);
            run_dgrad<1, 2, true>("", M, K, N, g, y, w, dx, nullptr);
            run_dgrad<1, 2, false>("", M, K, N, g, y, w, dx, nullptr);
            run_wgrad<1, 1, true>("", M, K, N, 8, x, g, y, ws, dw, nullptr, nullptr);
            run_wgrad<1, 1, false>("", M, K, N, 8, x, g, y, ws, dw, nullptr, nullptr);
            run_wgrad<1, 2, true>("", M, K, N, 16, x, g, y, ws, dw, nullptr, nullptr);
            run_wgrad<1, 2, false>("", M, K, N, 16, x, g, y, ws, dw, nullptr, nullptr);
            for (float* p : {x, w, g, y, dx, dw, ws}) CK(hipFree(p));
        }
        return 0;
    }
    layer(200, 96, 72, true);
    layer(M, 416, 512, true);
    layer(M, 512, 256, true);
    layer(M, 256, 128, true);
    // per-chunk cost: the same tiles with a 4 x longer reduction
    layer(M, 1664, 512, false);
    return 0;
}
