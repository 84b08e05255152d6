// Timeline of din16::bwd_kernel (csrc/din.hip built with -DDIN16_TIMELINE): shader-clock totals per phase of wave 0 of
// workgroup 0, B = 4096, T = 50.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Iinclude scripts/din_lab.hip -o /tmp/din_lab
#ifndef DIN_LAB_PLAIN
#define DIN16_TIMELINE 1
#endif
#include "../recalgorithm_amd/csrc/din.hip"

#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
    const int B = 4096, T = argc > 1 ? atoi(argv[1]) : 50, H = 16;
    std::vector<float> hq(B * H), hk((size_t)B * T * H), hg(B * H), w1(64 * 64), b1(64), w2(64 * 32), b2(32), w3(32), b3(1);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX - 0.5f; };
    for (auto* v : {&hq, &hk, &hg, &w1, &b1, &w2, &b2, &w3, &b3}) for (auto& x : *v) x = rnd() * 0.5f;
    std::vector<int> hl(B, T);
    if (argc > 2 && atoi(argv[2]) != 0)                    // ragged: lengths ~ U{0..T}
        for (auto& l : hl) l = rand() % (T + 1);
    float *q, *k, *g, *W1, *B1, *W2, *B2, *W3, *B3, *dq, *dk, *ws, *o;
    int* len;
    auto up = [](float** d, const std::vector<float>& h) { hipMalloc(d, h.size() * 4); hipMemcpy(*d, h.data(), h.size() * 4, hipMemcpyHostToDevice); };
    up(&q, hq); up(&k, hk); up(&g, hg); up(&W1, w1); up(&B1, b1); up(&W2, w2); up(&B2, b2); up(&W3, w3); up(&B3, b3);
    hipMalloc(&len, B * 4); hipMemcpy(len, hl.data(), B * 4, hipMemcpyHostToDevice);
    hipMalloc(&dq, B * H * 4); hipMalloc(&dk, (size_t)B * T * H * 4); hipMalloc(&o, B * H * 4);
    hipMalloc(&ws, recalgo_din_attention_bwd_workspace_bytes(B, T, H));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
#ifdef DIN_LAB_PLAIN
    for (int sm = 0; sm < 2; ++sm) {
        for (int i = 0; i < 3; ++i) recalgo_din_attention_bwd(q, k, len, W1, B1, W2, B2, W3, B3, g, B, T, H, sm, dq, dk, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws, nullptr);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) recalgo_din_attention_bwd(q, k, len, W1, B1, W2, B2, W3, B3, g, B, T, H, sm, dq, dk, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms0; hipEventElapsedTime(&ms0, e0, e1);
        printf("bwd (softmax %d, T %d) %.1f us\n", sm, T, ms0 * 1e3 / 20);
    }
#else
    for (int rep = 0; rep < 3; ++rep) {
        unsigned long long zero[32] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(din16::din16_tl), zero, sizeof(zero));
        hipEventRecord(e0);
        int rc = recalgo_din_attention_bwd(q, k, len, W1, B1, W2, B2, W3, B3, g, B, T, H, 0, dq, dk, nullptr, nullptr, nullptr, nullptr, nullptr,
                                           nullptr, ws, nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long tl[32];
        hipMemcpyFromSymbol(tl, HIP_SYMBOL(din16::din16_tl), sizeof(tl));
        printf("bwd rc %d  %.1f us (instrumented)\n", rc, ms * 1e3);
        const char* names[] = {"begin_example (q / g -> scratch, cq)", "tile prologue (next key row requested)", "fwd_tile (layers 1-3)",
                               "d2 + P / Q stores", "dH1 chain", "dW2 GEMM", "mask + P / R stores", "dX chain", "dWx GEMM + row sums",
                               "dcq / dWq / dk / dq + stores", "dq finalize", "final reduction + partial row", "weights staging + first row"};
        unsigned long long tot = 0;
        for (int i = 0; i < 13; ++i) tot += tl[i];
        for (int i = 0; i < 13; ++i) printf("  %-34s %9llu cycles  %5.1f %%\n", names[i], tl[i], 100.0 * tl[i] / tot);
        printf("  total %llu cycles\n", tot);
    }
#endif
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) recalgo_din_attention_fwd(q, k, len, W1, B1, W2, B2, W3, B3, B, T, H, 0, o, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("fwd %.1f us\n", ms * 1e3 / 20);
    return 0;
}
