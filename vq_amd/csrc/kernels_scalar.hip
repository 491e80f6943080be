// Scalar kernels: one lane per sample, the reference's loops restated in f32.
//
//   k_assign_scalar        find_closest_prototype{,_matched}        src/helpers.rs:55-77, 91-132
//                          with euclidean_distance                   src/helpers.rs:11-24
//   k_project_scalar       Omega . x (the projection inside           src/helpers.rs:35-45,
//                          generalized_distance, hoisted per sample)  src/liramlvq/liramlvq.rs:429-446
//   k_assign_local_scalar  find_closest_local_prototype{,_matched}   src/lgmlvq/lgmlvq.rs:436-455, 468-506
//
// They serve every shape (any D, K, M) and are the fallback of the MFMA engine; they are also
// the on-device cross-check for it.  Sums run in index order with separate multiply and add
// (no fma contraction), like the reference's `sum += (a - b).powi(2)`.
#include <cmath>

#include "vq_common.h"

#pragma clang fp contract(off)

namespace vq {

namespace {

template <class T>
struct BestT {
    T d;
    uint32_t i;
    __device__ BestT() : d((T)INFINITY), i(0u) {}                 // helpers.rs:58-59
    __device__ void offer(T dist, uint32_t k)
    {
        if (dist < d) {                                           // helpers.rs:70 (strict; NaN never wins)
            d = dist;
            i = k;
        }
    }
};

using Best = BestT<float>;

template <bool MATCHED, class O, class B>
__device__ inline void store(const O &out, uint64_t i, const B &pos, const B &neg)
{
    if (out.idx) out.idx[i] = pos.i;
    if (out.dist) out.dist[i] = pos.d;
    if (MATCHED) {
        if (out.jneg) out.jneg[i] = neg.i;
        if (out.dneg) out.dneg[i] = neg.d;
    }
}

// T = float: the product path.  T = double: the reference's own number type (src/lib.rs:32), so
// index AND distance can be compared bit for bit with the f64 oracle.
template <bool MATCHED, class T, class O>
__global__ void __launch_bounds__(256)
k_assign_scalar(const T *__restrict__ X, uint64_t N, uint32_t D, const T *__restrict__ W, uint32_t K,
                const uint32_t *__restrict__ cls, const uint32_t *__restrict__ y, int rooted, O out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
        const T *x = X + i * D;
        const uint32_t label = MATCHED ? y[i] : 0u;
        BestT<T> pos, neg;
        for (uint32_t k = 0; k < K; ++k) {
            const T *w = W + (size_t)k * D;
            T sum = (T)0;
            for (uint32_t d = 0; d < D; ++d) {
                const T diff = w[d] - x[d];                       // helpers.rs:19 (prototype - sample)
                sum += diff * diff;
            }
            const T dist = rooted ? (T)sqrt(sum) : sum;           // helpers.rs:23 / :44 (sqrt: correctly rounded)
            if (!MATCHED || cls[k] == label) pos.offer(dist, k);  // helpers.rs:113
            else neg.offer(dist, k);                              // helpers.rs:122
        }
        store<MATCHED>(out, i, pos, neg);
    }
}

__global__ void __launch_bounds__(256)
k_project_scalar(const float *__restrict__ X, uint64_t N, uint32_t D, const float *__restrict__ Om, uint32_t M,
                 float *__restrict__ Y)
{
    const uint64_t total = N * M;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const uint64_t i = e / M;
        const uint32_t m = (uint32_t)(e % M);
        const float *x = X + i * D;
        const float *o = Om + (size_t)m * D;
        float t = 0.0f;
        for (uint32_t d = 0; d < D; ++d) t += o[d] * x[d];
        Y[e] = t;
    }
}

template <bool MATCHED>
__global__ void __launch_bounds__(256)
k_assign_local_scalar(const float *__restrict__ X, uint64_t N, uint32_t D, const float *__restrict__ W,
                      uint32_t K, const float *__restrict__ Oms, uint32_t M, const uint32_t *__restrict__ cls,
                      const uint32_t *__restrict__ y, Outputs out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
        const float *x = X + i * D;
        const uint32_t label = MATCHED ? y[i] : 0u;
        Best pos, neg;
        for (uint32_t j = 0; j < K; ++j) {
            const float *w = W + (size_t)j * D;
            const float *om = Oms + (size_t)j * M * D;            // lgmlvq.rs:445: omegas[index]
            float dist = 0.0f;
            for (uint32_t m = 0; m < M; ++m) {
                float t = 0.0f;
                for (uint32_t d = 0; d < D; ++d) t += om[(size_t)m * D + d] * (x[d] - w[d]);
                dist += t * t;                                    // ||Omega_j (x - w_j)||^2, squared
            }
            if (!MATCHED || cls[j] == label) pos.offer(dist, j);
            else neg.offer(dist, j);
        }
        store<MATCHED>(out, i, pos, neg);
    }
}

// LGMLVQ::project (src/lgmlvq/lgmlvq.rs:372-393): sample i is projected with the Omega of its nearest
// prototype j = idx[i].  T[i][m] = Omega_j[m] . x_i, then Y[i][d] = sum_m Omega_j[m][d] T[i][m] = (Lambda_j x_i)[d].
__global__ void __launch_bounds__(256)
k_project_local_fwd(const float *__restrict__ X, uint64_t N, uint32_t D, const float *__restrict__ Oms, uint32_t M,
                    const uint32_t *__restrict__ idx, float *__restrict__ T)
{
    const uint64_t total = N * M;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const uint64_t i = e / M;
        const uint32_t m = (uint32_t)(e % M);
        const float *x = X + i * D;
        const float *o = Oms + ((size_t)idx[i] * M + m) * D;
        float t = 0.0f;
        for (uint32_t d = 0; d < D; ++d) t += o[d] * x[d];
        T[e] = t;
    }
}

__global__ void __launch_bounds__(256)
k_project_local_bwd(const float *__restrict__ T, uint64_t N, uint32_t D, const float *__restrict__ Oms, uint32_t M,
                    const uint32_t *__restrict__ idx, float *__restrict__ Y)
{
    const uint64_t total = N * D;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const uint64_t i = e / D;
        const uint32_t d = (uint32_t)(e % D);
        const float *t = T + i * M;
        const float *o = Oms + (size_t)idx[i] * M * D + d;
        float y = 0.0f;
        for (uint32_t m = 0; m < M; ++m) y += o[(size_t)m * D] * t[m];
        Y[e] = y;
    }
}

__global__ void __launch_bounds__(256) k_square(float *d, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = d[i] * d[i];
}

inline unsigned grid_for(uint64_t work)
{
    uint64_t blocks = (work + 255) / 256;
    const uint64_t cap = 256ull * 32ull;
    return (unsigned)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

}  // namespace

hipError_t launch_assign_scalar(const float *dX, uint64_t N, uint32_t D, const float *dW, uint32_t K,
                                const uint32_t *dcls, const uint32_t *dy, bool rooted, Outputs out,
                                hipStream_t s)
{
    if (N == 0) return hipSuccess;
    if (dy)
        hipLaunchKernelGGL((k_assign_scalar<true, float, Outputs>), dim3(grid_for(N)), dim3(256), 0, s, dX, N, D, dW, K,
                           dcls, dy, (int)rooted, out);
    else
        hipLaunchKernelGGL((k_assign_scalar<false, float, Outputs>), dim3(grid_for(N)), dim3(256), 0, s, dX, N, D, dW, K,
                           dcls, dy, (int)rooted, out);
    return hipGetLastError();
}

hipError_t launch_assign_scalar_f64(const double *dX, uint64_t N, uint32_t D, const double *dW, uint32_t K,
                                    const uint32_t *dcls, const uint32_t *dy, Outputs64 out, hipStream_t s)
{
    if (N == 0) return hipSuccess;
    if (dy)
        hipLaunchKernelGGL((k_assign_scalar<true, double, Outputs64>), dim3(grid_for(N)), dim3(256), 0, s, dX, N, D, dW,
                           K, dcls, dy, 1, out);
    else
        hipLaunchKernelGGL((k_assign_scalar<false, double, Outputs64>), dim3(grid_for(N)), dim3(256), 0, s, dX, N, D, dW,
                           K, dcls, dy, 1, out);
    return hipGetLastError();
}

hipError_t launch_square(float *d, uint64_t n, hipStream_t s)
{
    if (n == 0 || !d) return hipSuccess;
    hipLaunchKernelGGL(k_square, dim3(grid_for(n)), dim3(256), 0, s, d, n);
    return hipGetLastError();
}

hipError_t launch_project_scalar(const float *dX, uint64_t N, uint32_t D, const float *dOm, uint32_t M,
                                 float *dY, hipStream_t s)
{
    if (N == 0) return hipSuccess;
    hipLaunchKernelGGL(k_project_scalar, dim3(grid_for(N * M)), dim3(256), 0, s, dX, N, D, dOm, M, dY);
    return hipGetLastError();
}

hipError_t launch_project_local(const float *dX, uint64_t N, uint32_t D, const float *dOms, uint32_t M,
                                const uint32_t *didx, float *dT, float *dY, hipStream_t s)
{
    if (N == 0) return hipSuccess;
    hipLaunchKernelGGL(k_project_local_fwd, dim3(grid_for(N * M)), dim3(256), 0, s, dX, N, D, dOms, M, didx, dT);
    hipLaunchKernelGGL(k_project_local_bwd, dim3(grid_for(N * D)), dim3(256), 0, s, dT, N, D, dOms, M, didx, dY);
    return hipGetLastError();
}

hipError_t launch_assign_local_scalar(const float *dX, uint64_t N, uint32_t D, const float *dW, uint32_t K,
                                      const float *dOms, uint32_t M, const uint32_t *dcls,
                                      const uint32_t *dy, Outputs out, hipStream_t s)
{
    if (N == 0) return hipSuccess;
    if (dy)
        hipLaunchKernelGGL(k_assign_local_scalar<true>, dim3(grid_for(N)), dim3(256), 0, s, dX, N, D, dW, K, dOms,
                           M, dcls, dy, out);
    else
        hipLaunchKernelGGL(k_assign_local_scalar<false>, dim3(grid_for(N)), dim3(256), 0, s, dX, N, D, dW, K,
                           dOms, M, dcls, dy, out);
    return hipGetLastError();
}

}  // namespace vq
