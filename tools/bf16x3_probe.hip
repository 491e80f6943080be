// How exact is a dot product on the bf16 matrix cores when both operands are split into two bf16 pieces (hi + lo,
// round to nearest) and three of the four piece products are accumulated in f32 (hi.hi + hi.lo + lo.hi)?
// Prints, over TRIALS random 32 x 32 x D blocks, the largest |score - exact| / (||x|| ||w||) and the same against
// (||x||^2 + ||w||^2) / 2, next to the f32 MFMA's (v_mfma_f32_32x32x2_f32) on the same data: the numbers the screening
// margin of the bf16 search is built on.  Also checks the operand layout the kernels assume.
// build: hipcc -O3 --offload-arch=gfx950 tools/bf16x3_probe.hip -o tools/bf16x3_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2(float w0, float w1, uint32_t &hp, uint32_t &lp)
{
    hp = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){w0, w1}, bf16x2));
    const float h0 = __builtin_bit_cast(float, hp << 16), h1 = __builtin_bit_cast(float, hp & 0xFFFF0000u);
    lp = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){w0 - h0, w1 - h1}, bf16x2));
}
__device__ __forceinline__ void split8(const f32x4 &a0, const f32x4 &a1, u32x4 &hi, u32x4 &lo)
{
    uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
    split2(a0[0], a0[1], h0, l0);
    split2(a0[2], a0[3], h1, l1);
    split2(a1[0], a1[1], h2, l2);
    split2(a1[2], a1[3], h3, l3);
    hi = u32x4{h0, h1, h2, h3};
    lo = u32x4{l0, l1, l2, l3};
}

// one wave: W 32 x D, X 32 x D (row major) -> S[m][n] = sum_d W[m][d] X[n][d], both ways
template <int D>
__global__ void k_probe(const float *W, const float *X, float *Sbf, float *Sf32)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc = {0}, ref = {0};
    for (int q = 0; q < D / 16; ++q) {
        // lane half h owns the 16-byte chunks c with (c & 1) == h: chunks 4q + h and 4q + 2 + h of this step
        const f32x4 a0 = *(const f32x4 *)(W + r * D + 16 * q + 4 * h), a1 = *(const f32x4 *)(W + r * D + 16 * q + 8 + 4 * h);
        const f32x4 b0 = *(const f32x4 *)(X + r * D + 16 * q + 4 * h), b1 = *(const f32x4 *)(X + r * D + 16 * q + 8 + 4 * h);
        u32x4 ah, al, bh, bl;
        split8(a0, a1, ah, al);
        split8(b0, b1, bh, bl);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
        for (int e = 0; e < 4; ++e) ref = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], ref, 0, 0, 0);
        for (int e = 0; e < 4; ++e) ref = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], ref, 0, 0, 0);
    }
    for (int g = 0; g < 4; ++g)
        for (int e = 0; e < 4; ++e) {
            const int m = 8 * g + 4 * h + e;           // C layout: column = lane & 31 (sample), rows 8g + 4h + e
            Sbf[m * 32 + r] = acc[4 * g + e];
            Sf32[m * 32 + r] = ref[4 * g + e];
        }
}

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static double urand()
{
    rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
    return (double)(rng >> 11) / 9007199254740992.0;
}

template <int D>
static int run(int trials, int mode)
{
    float *dW, *dX, *dA, *dB;
    hipMalloc(&dW, 32 * D * 4); hipMalloc(&dX, 32 * D * 4); hipMalloc(&dA, 4096); hipMalloc(&dB, 4096);
    std::vector<float> W(32 * D), X(32 * D), A(1024), B(1024);
    double worst_bf = 0, worst_f32 = 0, worst_bf2 = 0, worst_f322 = 0;
    for (int t = 0; t < trials; ++t) {
        for (int i = 0; i < 32 * D; ++i) {
            double w = 2 * urand() - 1, x = 2 * urand() - 1;
            if (mode == 1) { w = 100 + w; x = 100 + x; }                 // all one sign: no cancellation in the sum
            if (mode == 2) { w = std::exp(8 * w); x = std::exp(8 * x) * (urand() < 0.5 ? -1 : 1); }   // wide dynamic range
            W[i] = (float)w; X[i] = (float)x;
        }
        hipMemcpy(dW, W.data(), 32 * D * 4, hipMemcpyHostToDevice);
        hipMemcpy(dX, X.data(), 32 * D * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_probe<D>, dim3(1), dim3(64), 0, 0, dW, dX, dA, dB);
        if (hipMemcpy(A.data(), dA, 4096, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        hipMemcpy(B.data(), dB, 4096, hipMemcpyDeviceToHost);
        for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n) {
                double s = 0, ww = 0, xx = 0;
                for (int d = 0; d < D; ++d) {
                    s += (double)W[m * D + d] * X[n * D + d];
                    ww += (double)W[m * D + d] * W[m * D + d];
                    xx += (double)X[n * D + d] * X[n * D + d];
                }
                const double sc = std::sqrt(ww * xx), sc2 = 0.5 * (ww + xx);
                worst_bf = std::fmax(worst_bf, std::fabs(A[m * 32 + n] - s) / sc);
                worst_f32 = std::fmax(worst_f32, std::fabs(B[m * 32 + n] - s) / sc);
                worst_bf2 = std::fmax(worst_bf2, std::fabs(A[m * 32 + n] - s) / sc2);
                worst_f322 = std::fmax(worst_f322, std::fabs(B[m * 32 + n] - s) / sc2);
            }
    }
    printf("D=%4d mode=%d trials=%d  bf16x3: %.3e of |x||w| (%.3e of (xx+ww)/2)   f32 mfma: %.3e (%.3e)   [bound 3*2^-16 = %.3e]\n", D, mode, trials,
           worst_bf, worst_bf2, worst_f32, worst_f322, 3.0 / 65536.0);
    hipFree(dW); hipFree(dX); hipFree(dA); hipFree(dB);
    return 0;
}

int main(int argc, char **argv)
{
    const int trials = argc > 1 ? atoi(argv[1]) : 200;
    for (int mode = 0; mode < 3; ++mode) {
        if (run<16>(trials, mode) || run<128>(trials, mode) || run<256>(trials, mode)) return 1;
    }
    return 0;
}
