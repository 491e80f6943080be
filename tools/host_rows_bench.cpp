// PCIe-inclusive rate of the reference's own container through the C ABI:
// N separately allocated f64 rows (Vec<Array1<f64>>, src/vq/vq.rs:142) -> vqhip_assign_rows_f64, and the
// same samples as one contiguous f32 matrix -> vqhip_assign.  Never bench.py's `value`.
//   g++ -O2 -std=c++17 -Iinclude tools/host_rows_bench.cpp -o tools/host_rows_bench -Lvq_amd -lvqhip
//   tools/host_rows_bench [rows]        (c2 shape: D = 128, K = 256)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "vqhip.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const uint64_t N = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4000000ull;
    const uint32_t D = 128, K = 256;
    vqhip_ctx *ctx = nullptr;
    if (vqhip_create(0, &ctx) != 0) {
        fprintf(stderr, "vqhip_create: %s\n", vqhip_last_error(nullptr));
        return 1;
    }
    std::vector<float> W((size_t)K * D), X((size_t)N * D);
    vqhip_synth_fill_host(W.data(), W.size(), 0, 42, 1, 1.0f, nullptr, 0);
    vqhip_synth_fill_host(X.data(), X.size(), 0, 42, 2, 0.25f, W.data(), W.size());    // sample i around prototype i mod K
    if (vqhip_set_prototypes(ctx, W.data(), K, D) != 0) return 2;
    std::vector<std::unique_ptr<double[]>> rows(N);
    std::vector<const double *> ptr(N);
    for (uint64_t i = 0; i < N; ++i) {
        rows[i].reset(new double[D]);
        for (uint32_t d = 0; d < D; ++d) rows[i][d] = X[i * D + d];
        ptr[i] = rows[i].get();
    }
    std::vector<uint32_t> idx(N), idx2(N);
    std::vector<float> dist(N);
    const uint64_t warm = N < 100000 ? N : 100000;
    vqhip_assign(ctx, X.data(), warm, idx.data(), dist.data());                            // staging allocation, warm-up
    double t = now();
    if (vqhip_assign(ctx, X.data(), N, idx.data(), dist.data()) != 0) return 3;
    double dt = now() - t;
    printf("vqhip_assign          %llu contiguous f32 rows : %.4g samples/s, %.1f GB/s over the host link\n", (unsigned long long)N,
           N / dt, N * (D * 4.0 + 8.0) / dt / 1e9);
    t = now();
    if (vqhip_assign_rows_f64(ctx, ptr.data(), N, D, idx2.data(), dist.data()) != 0) return 4;
    dt = now() - t;
    printf("vqhip_assign_rows_f64 %llu separate f64 rows   : %.4g samples/s, %.1f GB/s of f64 rows gathered and narrowed\n",
           (unsigned long long)N, N / dt, N * D * 8.0 / dt / 1e9);
    uint64_t bad = 0;
    for (uint64_t i = 0; i < N; ++i) bad += idx[i] != i % K || idx2[i] != idx[i];
    printf("wrong assignments: %llu\n", (unsigned long long)bad);
    vqhip_destroy(ctx);
    return bad != 0;
}
