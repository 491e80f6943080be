// What do the vector instructions of the bf16 search cost beside v_mfma_f32_32x32x16_bf16, for ONE wave per SIMD and
// for two?  Every variant is a loop of ITER x [one MFMA + a block of vector instructions]; prints shader cycles per
// iteration (s_memtime, lane 0 of wave 0 of every workgroup, averaged).  No memory traffic inside the loops.
// build: hipcc -O3 --offload-arch=gfx950 tools/issue_probe.hip -o tools/issue_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int ITER = 2000;

#define MFMA "v_mfma_f32_32x32x16_bf16 %[acc], %[a], %[b], %[acc]\n\t"
#define CVT6 "v_cvt_pk_bf16_f32 %[t0], %[x0], %[x1]\n\tv_cvt_pk_bf16_f32 %[t1], %[x2], %[x3]\n\tv_cvt_pk_bf16_f32 %[t2], %[x0], %[x2]\n\tv_cvt_pk_bf16_f32 %[t3], %[x1], %[x3]\n\tv_cvt_pk_bf16_f32 %[t4], %[x0], %[x3]\n\tv_cvt_pk_bf16_f32 %[t5], %[x1], %[x2]\n\t"
#define SUB6 "v_sub_f32 %[t0], %[x0], %[x1]\n\tv_sub_f32 %[t1], %[x2], %[x3]\n\tv_sub_f32 %[t2], %[x0], %[x2]\n\tv_sub_f32 %[t3], %[x1], %[x3]\n\tv_sub_f32 %[t4], %[x0], %[x3]\n\tv_sub_f32 %[t5], %[x1], %[x2]\n\t"
#define MED6 "v_med3_f32 %[t0], %[x0], %[x1], %[x2]\n\tv_med3_f32 %[t1], %[x2], %[x3], %[x0]\n\tv_med3_f32 %[t2], %[x0], %[x2], %[x3]\n\tv_med3_f32 %[t3], %[x1], %[x3], %[x0]\n\tv_med3_f32 %[t4], %[x0], %[x3], %[x1]\n\tv_med3_f32 %[t5], %[x1], %[x2], %[x3]\n\t"
#define ANDOR6 "v_and_or_b32 %[t0], %[x0], %[m], 1\n\tv_and_or_b32 %[t1], %[x2], %[m], 2\n\tv_and_or_b32 %[t2], %[x0], %[m], 3\n\tv_and_or_b32 %[t3], %[x1], %[m], 4\n\tv_and_or_b32 %[t4], %[x0], %[m], 5\n\tv_and_or_b32 %[t5], %[x1], %[m], 6\n\t"
#define SHIFT6 "v_lshlrev_b32 %[t0], 16, %[x0]\n\tv_lshlrev_b32 %[t1], 16, %[x1]\n\tv_and_b32 %[t2], %[m], %[x2]\n\tv_and_b32 %[t3], %[m], %[x3]\n\tv_lshlrev_b32 %[t4], 16, %[x2]\n\tv_and_b32 %[t5], %[m], %[x0]\n\t"
// the screening block of kernels_pair.hip: two interleaved dependent chains
#define SCREEN8 "v_and_or_b32 %[t0], %[x0], %[m], 0\n\tv_and_or_b32 %[t1], %[x1], %[m], 0\n\tv_med3_f32 %[t2], %[x2], %[t2], %[t0]\n\tv_med3_f32 %[t3], %[x3], %[t3], %[t1]\n\tv_med3_f32 %[t4], %[t2], %[t4], %[t0]\n\tv_med3_f32 %[t5], %[t3], %[t5], %[t1]\n\tv_min_f32 %[t2], %[t2], %[t0]\n\tv_min_f32 %[t3], %[t3], %[t1]\n\t"
// split4a / split4b of kernels_pair.hip
#define SPLITA "v_cvt_pk_bf16_f32 %[t0], %[x0], %[x1]\n\tv_cvt_pk_bf16_f32 %[t1], %[x2], %[x3]\n\tv_lshlrev_b32 %[t2], 16, %[t0]\n\tv_lshlrev_b32 %[t3], 16, %[t1]\n\tv_and_b32 %[t4], %[m], %[t0]\n\tv_and_b32 %[t5], %[m], %[t1]\n\t"
#define SPLITB "v_sub_f32 %[t2], %[x0], %[t2]\n\tv_sub_f32 %[t3], %[x2], %[t3]\n\tv_sub_f32 %[t4], %[x1], %[t4]\n\tv_sub_f32 %[t5], %[x3], %[t5]\n\tv_cvt_pk_bf16_f32 %[t0], %[t2], %[t4]\n\tv_cvt_pk_bf16_f32 %[t1], %[t3], %[t5]\n\t"
#define ACCRD8 "v_accvgpr_read_b32 %[t0], %[ag]\n\tv_accvgpr_read_b32 %[t1], %[ag]\n\tv_accvgpr_read_b32 %[t2], %[ag]\n\tv_accvgpr_read_b32 %[t3], %[ag]\n\tv_accvgpr_read_b32 %[t4], %[ag]\n\tv_accvgpr_read_b32 %[t5], %[ag]\n\tv_accvgpr_read_b32 %[t0], %[ag]\n\tv_accvgpr_read_b32 %[t1], %[ag]\n\t"

#define BODY(NAME, TEXT)                                                                                                          \
    __global__ void __launch_bounds__(1024) NAME(unsigned long long *out, float seed)                                              \
    {                                                                                                                             \
        f32x16 acc = {0};                                                                                                         \
        u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;                                                    \
        float x0 = seed + threadIdx.x, x1 = x0 * 1.5f, x2 = x0 + 3.0f, x3 = x0 - 7.0f, t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, ag = 1.0f; \
        asm volatile("" : "+a"(ag));                                                                                              \
        unsigned long long c0, c1;                                                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");                                              \
        for (int i = 0; i < ITER; ++i)                                                                                            \
            asm volatile(TEXT : [acc] "+v"(acc), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3), [t4] "+v"(t4), [t5] "+v"(t5) \
                         : [a] "v"(a), [b] "v"(b), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [m] "s"(0xFFFF0000u), [ag] "a"(ag)); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");                                              \
        if (t0 + t1 + t2 + t3 + t4 + t5 + acc[0] == 12345.678f) out[4096] = 1;                                                    \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = c1 - c0;                                          \
    }

BODY(k_mfma, MFMA)
BODY(k_mfma2, MFMA MFMA)
BODY(k_cvt6, MFMA CVT6)
BODY(k_sub6, MFMA SUB6)
BODY(k_med6, MFMA MED6)
BODY(k_andor6, MFMA ANDOR6)
BODY(k_shift6, MFMA SHIFT6)
BODY(k_screen8, MFMA SCREEN8)
BODY(k_splita, MFMA SPLITA)
BODY(k_splitb, MFMA SPLITB)
BODY(k_accrd8, MFMA ACCRD8)
BODY(k_only_cvt6, CVT6)
BODY(k_only_sub6, SUB6)
BODY(k_only_med6, MED6)
BODY(k_only_screen8, SCREEN8)
BODY(k_only_splitab, SPLITA SPLITB)
BODY(k_step, MFMA SPLITA MFMA SPLITB MFMA SPLITA MFMA SPLITB MFMA SCREEN8 MFMA SCREEN8)

template <class K>
static void run(const char *name, K kern, unsigned long long *d, int threads)
{
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, 1.0f);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, 1.0f);
    std::vector<unsigned long long> h(256 * 16);
    if (hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("%s: failed\n", name); return; }
    double s = 0;
    const int waves = threads / 64;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < waves; ++w) s += (double)h[b * 16 + w];
    printf("%-16s %d waves/SIMD: %7.1f cycles per iteration\n", name, waves / 4, s / (256.0 * waves) / ITER);
}

int main()
{
    unsigned long long *d;
    if (hipMalloc(&d, 8192 * 8) != hipSuccess) return 1;
    hipMemset(d, 0, 8192 * 8);
    for (int threads : {256, 512, 768, 1024}) {
#define R(k) run(#k, k, d, threads)
        R(k_mfma); R(k_mfma2); R(k_cvt6); R(k_sub6); R(k_med6); R(k_andor6); R(k_shift6); R(k_screen8); R(k_splita); R(k_splitb); R(k_accrd8);
        R(k_only_cvt6); R(k_only_sub6); R(k_only_med6); R(k_only_screen8); R(k_only_splitab); R(k_step);
    }
    hipFree(d);
    return 0;
}
