// Batched GLVQ gradient (second half; vqhip_glvq_batch_gradient in include/vqhip.h).
//
// The reference's fit loop (src/glvq/glvq.rs:163-206) moves two prototypes after EVERY sample:
//     w+ <- w+ - lr * ( -4 d- / (d+ + d-)^2 ) (x - w+)          glvq.rs:193, 200
//     w- <- w- - lr * ( +4 d+ / (d+ + d-)^2 ) (x - w-)          glvq.rs:194, 201
// so sample i+1 is measured against prototypes sample i has moved: inherently sequential.  This file computes the BATCH
// form of the same two terms with the prototypes held fixed -- a different algorithm (mini-batch / full-batch GLVQ), labelled
// as such everywhere it is exposed:
//     G[k] = sum over samples with j+ = k of  -4 d- / (d+ + d-)^2 (x - w_k)
//          + sum over samples with j- = k of  +4 d+ / (d+ + d-)^2 (x - w_k)
// after the matched search has produced (j+, d+, j-, d-) for every sample.  sum c (x - w_k) = (sum c x) - (sum c) w_k, so
// the pass over the samples accumulates S[k] = sum c x (K x D) and s[k] = sum c, and one small kernel finishes G.
//
// The pass is a scatter-add of N rows into K bins and reads every row once: HBM-bound.  Models of up to ~36K prototype
// values accumulate in LDS (ds_add_f32) and leave one partial per workgroup, summed afterwards in a fixed order; larger
// ones add into global memory directly.
#include "mfma_host.h"
#include "vq_common.h"

namespace vq {

namespace {
constexpr int GRAD_THREADS = 512;
constexpr int GRAD_LDS_FLOATS = RES_BYTES / 4;             // S and s of one workgroup

struct GradArgs {
    const float *X;
    uint64_t N;
    uint32_t D, K;
    const uint32_t *jp, *jn;
    const float *dp, *dn;
    float *acc;            // LDS mode: nwg partials of K * D + K floats; global mode: one zeroed set, added to atomically
    double *cost;          // += sum of (d+ - d-) / (d+ + d-) over the samples that count, or null
    unsigned long long *used;   // += number of samples that count, or null
};

// One wave per sample, lanes over the row; coefficient pair per sample as in glvq.rs:193-194 with f' = 1 (glvq.rs:192).
// A sample counts when both searches found a prototype and d+ + d- > 0 (the reference would divide by zero there).
template <bool LDSACC>
__global__ void __launch_bounds__(GRAD_THREADS) k_glvq_accumulate(const GradArgs a)
{
    __shared__ __attribute__((aligned(16))) float lds[LDSACC ? GRAD_LDS_FLOATS : 4];
    const uint32_t total = a.K * a.D + a.K;
    float *acc = LDSACC ? lds : a.acc;
    if (LDSACC) {
        for (uint32_t e = threadIdx.x; e < total; e += GRAD_THREADS) lds[e] = 0.0f;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (GRAD_THREADS / 64) + (threadIdx.x >> 6), nwaves = (uint64_t)gridDim.x * (GRAD_THREADS / 64);
    double cost = 0.0;
    unsigned long long used = 0;
    auto add = [&](uint32_t e, float v) {
        if (LDSACC) atomicAdd(&lds[e], v);                  // ds_add_f32
        else unsafeAtomicAdd(&acc[e], v);                   // global_atomic_add_f32, no return
    };
    for (uint64_t i = wave; i < a.N; i += nwaves) {
        const uint32_t jp = a.jp[i], jn = a.jn[i];
        const float d1 = a.dp[i], d2 = a.dn[i], s = d1 + d2;
        if (!(s > 0.0f) || !(s < INFINITY) || jp >= a.K || jn >= a.K) continue;    // wave-uniform
        const float inv = 4.0f / (s * s);
        const float cp = -d2 * inv, cn = d1 * inv;          // glvq.rs:193 / :194
        const float *x = a.X + i * a.D;
        for (uint32_t d = lane; d < a.D; d += 64) {
            const float v = x[d];
            add(jp * a.D + d, cp * v);
            add(jn * a.D + d, cn * v);
        }
        if (lane == 0) {
            add(a.K * a.D + jp, cp);
            add(a.K * a.D + jn, cn);
            cost += (double)((d1 - d2) / s);
            ++used;
        }
    }
    if (LDSACC) {
        __syncthreads();
        float *out = a.acc + (size_t)blockIdx.x * total;
        for (uint32_t e = threadIdx.x; e < total; e += GRAD_THREADS) out[e] = lds[e];
    }
    if (lane == 0 && used != 0) {
        if (a.cost) unsafeAtomicAdd(a.cost, cost);
        if (a.used) atomicAdd(a.used, used);
    }
}

// G[k][d] (+)= sum over the partials of S[k][d] - (sum over the partials of s[k]) * W[k][d]; partials summed in index order
__global__ void k_glvq_finish(const float *acc, uint32_t nparts, const float *W, uint32_t K, uint32_t D, float *G, int accumulate)
{
    const uint32_t total = K * D + K;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= K * D) return;
    const uint32_t k = e / D;
    float S = 0.0f, s = 0.0f;
    for (uint32_t p = 0; p < nparts; ++p) {
        S += acc[(size_t)p * total + e];
        s += acc[(size_t)p * total + K * D + k];
    }
    const float g = S - s * W[e];
    G[e] = accumulate ? G[e] + g : g;
}
}  // namespace

size_t glvq_gradient_scratch_floats(uint32_t K, uint32_t D, uint32_t *nparts)
{
    const size_t total = (size_t)K * D + K;
    *nparts = total <= (size_t)GRAD_LDS_FLOATS ? 256u : 1u;           // one workgroup per CU, or one global set
    return total * *nparts;
}

hipError_t launch_glvq_gradient(const float *dX, uint64_t N, uint32_t D, uint32_t K, const float *dW, const uint32_t *jp, const float *dp,
                                const uint32_t *jn, const float *dn, float *scratch, float *dG, bool accumulate, double *dcost,
                                unsigned long long *dused, hipStream_t s)
{
    uint32_t nparts = 0;
    const size_t floats = glvq_gradient_scratch_floats(K, D, &nparts);
    GradArgs a{dX, N, D, K, jp, jn, dp, dn, scratch, dcost, dused};
    if (nparts == 1) {
        hipError_t e = hipMemsetAsync(scratch, 0, floats * sizeof(float), s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_glvq_accumulate<false>), dim3(2048), dim3(GRAD_THREADS), 0, s, a);
    } else {
        hipLaunchKernelGGL((k_glvq_accumulate<true>), dim3(nparts), dim3(GRAD_THREADS), 0, s, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_glvq_finish, dim3((K * D + 255) / 256), dim3(256), 0, s, scratch, nparts, dW, K, D, dG, accumulate ? 1 : 0);
    return hipGetLastError();
}

}  // namespace vq
