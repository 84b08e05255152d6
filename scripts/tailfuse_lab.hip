// Timeline of tail_dense_head_kernel (csrc/tailfuse.hip built with -DTAILFUSE_TIMELINE): shader-clock totals per phase of
// wave 0 of workgroup 0 at the DCN default shapes (B = 4096, K2 = 256, Cs = 416), and the plain kernel time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Iinclude [-DTAIL_LAB_PLAIN] scripts/tailfuse_lab.hip -o /tmp/tail_lab
#ifndef TAIL_LAB_PLAIN
#define TAILFUSE_TIMELINE 1
#endif
#include "../recalgorithm_amd/csrc/tailfuse.hip"

#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, K2 = 256, N3 = 128, Cs = argc > 2 ? atoi(argv[2]) : 416;
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX - 0.5f; };
    std::vector<float> h2((size_t)B * K2), w3(K2 * N3), b3(N3), side((size_t)B * Cs), wh(Cs + N3), bh(1), y(B);
    for (auto* v : {&h2, &w3, &b3, &side, &wh, &bh}) for (auto& x : *v) x = rnd() * 0.3f;
    for (auto& x : h2) x = x > 0 ? x : 0;
    for (auto& x : y) x = rand() % 3 == 0;
    auto up = [](const std::vector<float>& h) { float* d; hipMalloc(&d, h.size() * 4 + 16); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); return d; };
    float *dh2 = up(h2), *dw3 = up(w3), *db3 = up(b3), *dside = up(side), *dwh = up(wh), *dbh = up(bh), *dy = up(y);
    auto out = [](size_t n) { float* d; hipMalloc(&d, n * 4 + 16); return d; };
    float *logit = out(B), *prob = out(B), *dlogit = out(B), *d_side = out((size_t)B * Cs), *dz3 = out((size_t)B * N3), *g_h2 = out((size_t)B * K2);
    float* partials = out((size_t)recalgo_tail_partial_rows(B) * (Cs + N3 + 2));
    float* flush; hipMalloc(&flush, 512u << 20);
    auto run = [&] {
        return recalgo_tail_dense_head_fwd_bwd(dh2, K2, dw3, db3, N3, dside, Cs, 1, dwh, dwh + Cs, dbh, dy, nullptr, B, 1.0f, logit, prob, dlogit,
                                               d_side, dz3, g_h2, partials, nullptr);
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
#ifdef TAIL_LAB_PLAIN
    for (int i = 0; i < 3; ++i) run();
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) run();
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("tail fused B %d Cs %d: %.2f us back to back (inputs cache-warm)\n", B, Cs, ms * 1e3 / 50);
    float cold = 0;
    for (int i = 0; i < 10; ++i) {
        hipMemsetAsync(flush, i, 512u << 20, nullptr);
        hipEventRecord(e0);
        run();
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        cold += ms;
    }
    printf("tail fused B %d Cs %d: %.2f us after a 512 MiB flush\n", B, Cs, cold * 1e3 / 10);
#else
    for (int rep = 0; rep < 3; ++rep) {
        unsigned long long zero[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(tailfuse_tl), zero, sizeof(zero));
        if (rep == 2) hipMemsetAsync(flush, 1, 512u << 20, nullptr);
        hipEventRecord(e0);
        int rc = run();
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long tl[16];
        hipMemcpyFromSymbol(tl, HIP_SYMBOL(tailfuse_tl), sizeof(tl));
        printf("rc %d  %.1f us (instrumented%s)\n", rc, ms * 1e3, rep == 2 ? ", after flush" : "");
        const char* names[] = {"loads -> LDS (h2, side, weights)", "side dot products (LDS)", "forward GEMM", "relu + row sums (+ bwd B loads)",
                               "(sync wait)", "logit / loss per row", "dz3 + dw_h3", "d_side + dw_side", "(sync wait)", "backward GEMM", "dh2 stores"};
        unsigned long long tot = 0;
        for (int i = 0; i < 11; ++i) tot += tl[i];
        for (int i = 0; i < 11; ++i) printf("  %-30s %8llu ticks  %5.1f %%\n", names[i], tl[i], 100.0 * tl[i] / tot);
        printf("  total %llu ticks (clock64)\n", tot);
    }
#endif
    return 0;
}
