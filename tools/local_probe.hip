// What does the tile loop of k_local_pj (kernels_local.hip) cost with nothing around it?  One workgroup of 4 waves per CU
// (one wave per SIMD), a Geom3<128> tile in LDS, S sample tiles as pieces in registers, ITER x tile_mma_pj + the squared
// sums -- no DMA, no barrier, no global memory inside the loop.  Prints shader cycles per MFMA.
// build: hipcc -O3 --offload-arch=gfx950 -Iinclude -Ivq_amd/csrc tools/local_probe.hip -o tools/local_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "mfma_device.h"

using namespace vq;

template <int S, bool SUMS>
__global__ void __launch_bounds__(256, 1) k_probe(unsigned long long *out, int iters, float seed)
{
    using G3 = Geom3<128>;
    __shared__ __attribute__((aligned(16))) char tile[G3::TILE_BYTES];
    for (int i = threadIdx.x; i < G3::TILE_BYTES / 4; i += 256) ((uint32_t *)tile)[i] = 0x3c003c00u + (uint32_t)i;   // small bf16 pairs
    __syncthreads();
    const int lane = threadIdx.x & 63, s_col = lane & 31, h = lane >> 5;
    u32x4 xp[S][8][3];
    for (int s = 0; s < S; ++s)
        for (int q = 0; q < 8; ++q)
            for (int c = 0; c < 3; ++c) {
                const uint32_t v = 0x3a003a00u + (uint32_t)(lane + 7 * s + 3 * q + c) + __builtin_bit_cast(uint32_t, seed);
                xp[s][q][c] = u32x4{v, v + 1, v + 2, v + 3};
            }
    float dsum[S];
    for (int s = 0; s < S; ++s) dsum[s] = 0.0f;
    const unsigned long long t0 = stamp();
    for (int it = 0; it < iters; ++it) {
        f32x16 acc[S];
        tile_mma_pj<128, S>(tile, xp, acc, s_col, h);
        if (SUMS) {
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) dsum[s] = fmaf(acc[s][r], acc[s][r], dsum[s]);
        } else {
#pragma unroll
            for (int s = 0; s < S; ++s) dsum[s] += acc[s][0];
        }
    }
    const unsigned long long t1 = stamp();
    float tot = 0;
    for (int s = 0; s < S; ++s) tot += dsum[s];
    if (tot == 12345.678f) out[8191] = 1;
    if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int S, bool SUMS>
static void run(unsigned long long *d, int iters)
{
    hipLaunchKernelGGL((k_probe<S, SUMS>), dim3(256), dim3(256), 0, 0, d, iters, 0.0f);
    hipLaunchKernelGGL((k_probe<S, SUMS>), dim3(256), dim3(256), 0, 0, d, iters, 0.0f);
    std::vector<unsigned long long> h(1024);
    if (hipMemcpy(h.data(), d, 1024 * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("failed\n"); return; }
    double s = 0;
    for (int i = 0; i < 1024; ++i) s += (double)h[i];
    const double per_tile = s / 1024.0 / iters;
    printf("S=%d sums=%d: %8.1f cycles per tile, %5.1f per MFMA (%d MFMAs)\n", S, (int)SUMS, per_tile, per_tile / (48.0 * S), 48 * S);
}

int main()
{
    unsigned long long *d;
    if (hipMalloc(&d, 8192 * 8) != hipSuccess) return 1;
    run<3, true>(d, 2000);
    run<3, false>(d, 2000);
    run<2, true>(d, 2000);
    run<1, true>(d, 2000);
    hipFree(d);
    return 0;
}
