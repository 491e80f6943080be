// How exact is a dot product on the bf16 matrix cores when both operands are split into THREE bf16 pieces
// (v = p1 + p2 + p3 exactly for a normal f32: 3 x 8 significant bits) and the six piece products down to 2^-16 of the
// leading one are accumulated in f32 (p1.p1 | p1.p2 + p2.p1 | p1.p3 + p2.p2 + p3.p1)?  This is the projection y = Omega x
// of the Omega models moved from v_mfma_f32_32x32x2_f32 (64 cycles per 2 reduction values) to v_mfma_f32_32x32x16_bf16
// (6 x 32 cycles per 16 values: 2.7 times the rate) -- worth it only if the result is as good as the f32 chain's.
// Prints, over TRIALS random 32 x 32 x D blocks, the largest and the rms |score - exact| / (||x|| ||w||) for
//   f32   the f32 MFMA chain (what the kernels do today)
//   x6a   six products into one accumulator, smallest class first inside every step
//   x6b   one accumulator per class (big / mid / small), added small-to-big at the end
//   x9    all nine products, three accumulators
// build: hipcc -O3 --offload-arch=gfx950 tools/bf16x6_probe.hip -o tools/bf16x6_probe
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

// two values -> three packed bf16 pairs (round to nearest even each time; the remainders are exact in f32)
__device__ __forceinline__ void split3(float w0, float w1, uint32_t &p1, uint32_t &p2, uint32_t &p3)
{
    p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){w0, w1}, bf16x2));
    const float r0 = w0 - __builtin_bit_cast(float, p1 << 16), r1 = w1 - __builtin_bit_cast(float, p1 & 0xFFFF0000u);
    p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){r0, r1}, bf16x2));
    const float s0 = r0 - __builtin_bit_cast(float, p2 << 16), s1 = r1 - __builtin_bit_cast(float, p2 & 0xFFFF0000u);
    p3 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){s0, s1}, bf16x2));
}
__device__ __forceinline__ void split24(const f32x4 &a0, const f32x4 &a1, u32x4 &p1, u32x4 &p2, u32x4 &p3)
{
    uint32_t a[4], b[4], c[4];
    split3(a0[0], a0[1], a[0], b[0], c[0]);
    split3(a0[2], a0[3], a[1], b[1], c[1]);
    split3(a1[0], a1[1], a[2], b[2], c[2]);
    split3(a1[2], a1[3], a[3], b[3], c[3]);
    p1 = u32x4{a[0], a[1], a[2], a[3]};
    p2 = u32x4{b[0], b[1], b[2], b[3]};
    p3 = u32x4{c[0], c[1], c[2], c[3]};
}
__device__ __forceinline__ f32x16 mm(const u32x4 &a, const u32x4 &b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// one wave: W 32 x D, X 32 x D (row major) -> S[v][m][n] = sum_d W[m][d] X[n][d], four ways
template <int D>
__global__ void k_probe(const float *W, const float *X, float *S)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 ref = {0}, one = {0}, big = {0}, mid = {0}, sml = {0}, tiny = {0};
    for (int q = 0; q < D / 16; ++q) {
        const f32x4 a0 = *(const f32x4 *)(W + r * D + 16 * q + 4 * h), a1 = *(const f32x4 *)(W + r * D + 16 * q + 8 + 4 * h);
        const f32x4 b0 = *(const f32x4 *)(X + r * D + 16 * q + 4 * h), b1 = *(const f32x4 *)(X + r * D + 16 * q + 8 + 4 * h);
        u32x4 a[3], b[3];
        split24(a0, a1, a[0], a[1], a[2]);
        split24(b0, b1, b[0], b[1], b[2]);
        one = mm(a[2], b[0], one); one = mm(a[1], b[1], one); one = mm(a[0], b[2], one);
        one = mm(a[1], b[0], one); one = mm(a[0], b[1], one);
        one = mm(a[0], b[0], one);
        sml = mm(a[2], b[0], sml); sml = mm(a[1], b[1], sml); sml = mm(a[0], b[2], sml);
        mid = mm(a[1], b[0], mid); mid = mm(a[0], b[1], mid);
        big = mm(a[0], b[0], big);
        tiny = mm(a[2], b[1], tiny); tiny = mm(a[1], b[2], tiny); tiny = mm(a[2], b[2], tiny);
        for (int e = 0; e < 4; ++e) ref = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], ref, 0, 0, 0);
        for (int e = 0; e < 4; ++e) ref = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], ref, 0, 0, 0);
    }
    for (int g = 0; g < 4; ++g)
        for (int e = 0; e < 4; ++e) {
            const int m = 8 * g + 4 * h + e, i = 4 * g + e;   // C layout: column = lane & 31, rows 8g + 4h + e
            S[0 * 1024 + m * 32 + r] = ref[i];
            S[1 * 1024 + m * 32 + r] = one[i];
            S[2 * 1024 + m * 32 + r] = (sml[i] + mid[i]) + big[i];
            S[3 * 1024 + m * 32 + r] = ((tiny[i] + sml[i]) + mid[i]) + big[i];
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
    float *dW, *dX, *dS;
    if (hipMalloc(&dW, 32 * D * 4) != hipSuccess || hipMalloc(&dX, 32 * D * 4) != hipSuccess || hipMalloc(&dS, 4 * 4096) != hipSuccess) return 1;
    std::vector<float> W(32 * D), X(32 * D), S(4 * 1024);
    double worst[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
    for (int t = 0; t < trials; ++t) {
        for (int i = 0; i < 32 * D; ++i) {
            double w = 2 * urand() - 1, x = 2 * urand() - 1;
            if (mode == 1) { w = 100 + w; x = 100 + x; }                 // all one sign: no cancellation in the sum
            if (mode == 2) { w = std::exp(8 * w); x = std::exp(8 * x) * (urand() < 0.5 ? -1 : 1); }   // wide dynamic range
            W[i] = (float)w; X[i] = (float)x;
        }
        hipMemcpy(dW, W.data(), 32 * D * 4, hipMemcpyHostToDevice);
        hipMemcpy(dX, X.data(), 32 * D * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_probe<D>, dim3(1), dim3(64), 0, 0, dW, dX, dS);
        if (hipMemcpy(S.data(), dS, 4 * 4096, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n) {
                double s = 0, ww = 0, xx = 0;
                for (int d = 0; d < D; ++d) {
                    s += (double)W[m * D + d] * X[n * D + d];
                    ww += (double)W[m * D + d] * W[m * D + d];
                    xx += (double)X[n * D + d] * X[n * D + d];
                }
                const double sc = std::sqrt(ww * xx);
                for (int v = 0; v < 4; ++v) {
                    const double e = std::fabs(S[v * 1024 + m * 32 + n] - s) / sc;
                    worst[v] = std::fmax(worst[v], e);
                    sq[v] += e * e;
                }
            }
    }
    const double cnt = 1024.0 * trials;
    printf("D=%4d mode=%d trials=%d  |err|/(|x||w|) max (rms):  f32 %.3e (%.3e)  x6a %.3e (%.3e)  x6b %.3e (%.3e)  x9 %.3e (%.3e)\n", D, mode, trials,
           worst[0], std::sqrt(sq[0] / cnt), worst[1], std::sqrt(sq[1] / cnt), worst[2], std::sqrt(sq[2] / cnt), worst[3], std::sqrt(sq[3] / cnt));
    hipFree(dW); hipFree(dX); hipFree(dS);
    return 0;
}

int main(int argc, char **argv)
{
    const int trials = argc > 1 ? atoi(argv[1]) : 100;
    for (int mode = 0; mode < 3; ++mode) {
        if (run<16>(trials, mode) || run<128>(trials, mode) || run<256>(trials, mode)) return 1;
    }
    return 0;
}
