// Internal declarations shared by the translation units of libvqhip (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/vqhip.h"

namespace vq {

// ---------------------------------------------------------------- synthetic values
// value(seed, stream, i): SplitMix64 of a per-(seed, stream) key plus the counter; the top
// 24 bits become an f32 in [-1, 1).  Identical on host and device (integer ops + one exact
// int->float conversion + exact scaling).
__host__ __device__ inline uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ inline uint64_t synth_key(uint64_t seed, uint32_t stream)
{
    return splitmix64(seed ^ (0xA0761D6478BD642Full * (uint64_t)(stream + 1u)));
}
__host__ __device__ inline float synth_value(uint64_t key, uint64_t i)
{
    uint32_t top = (uint32_t)(splitmix64(key + i) >> 40);          // 24 bits
    return (float)(int32_t)top * (1.0f / 8388608.0f) - 1.0f;       // exact: [0,2^24) * 2^-23 - 1
}

// ---------------------------------------------------------------- model as kernels see it
enum OmegaKind : int { OMEGA_NONE = 0, OMEGA_GLOBAL = 1, OMEGA_LOCAL = 2 };

struct Outputs {           // any pointer may be null
    uint32_t *idx;  float *dist;       // plain search, or the matching search
    uint32_t *jneg; float *dneg;       // non-matching search (matched calls only)
};

struct Outputs64 {         // the f64 instantiation of the Euclidean scalar kernel
    uint32_t *idx;  double *dist;
    uint32_t *jneg; double *dneg;
};

// ---------------------------------------------------------------- scalar kernels (kernels_scalar.hip)
// X: N x D (ld = D).  Direct (x-w)^2 form, sums in index order, strict '<' from (0, +INF).
hipError_t launch_assign_scalar(const float *dX, uint64_t N, uint32_t D, const float *dW, uint32_t K,
                                const uint32_t *dcls, const uint32_t *dy, bool rooted, Outputs out,
                                hipStream_t s);
hipError_t launch_assign_scalar_f64(const double *dX, uint64_t N, uint32_t D, const double *dW, uint32_t K,
                                    const uint32_t *dcls, const uint32_t *dy, Outputs64 out, hipStream_t s);
// Y[i][m] = sum_d Om[m][d] * X[i][d]
hipError_t launch_project_scalar(const float *dX, uint64_t N, uint32_t D, const float *dOm, uint32_t M,
                                 float *dY, hipStream_t s);
// LGMLVQ::project: Y[i] = Omega_j^T (Omega_j x_i) with j = didx[i]; dT is N x M scratch
hipError_t launch_project_local(const float *dX, uint64_t N, uint32_t D, const float *dOms, uint32_t M,
                                const uint32_t *didx, float *dT, float *dY, hipStream_t s);
// dist_j = sum_m ( sum_d Om_j[m][d] * (x[d] - w_j[d]) )^2
hipError_t launch_assign_local_scalar(const float *dX, uint64_t N, uint32_t D, const float *dW, uint32_t K,
                                      const float *dOms, uint32_t M, const uint32_t *dcls,
                                      const uint32_t *dy, Outputs out, hipStream_t s);

// ---------------------------------------------------------------- MFMA kernels (engine_host.hip builds the model; kernels_engine / _resident / _omega.hip)
struct EngineModel;   // device tile images + shapes, built by the engine from host copies

// Where the matrix-core kernels may run their products (vqhip_set_option, include/vqhip.h; resolved against the
// environment's defaults in api.cpp): the candidates' ranking as three bf16 piece products, the projections as six.
struct BuildOptions {
    bool bf_rank = true;
    bool bf_proj = true;
};

// Returns nullptr (and sets why) when the shape is not eligible for the engine.
EngineModel *engine_build(const std::vector<float> &W, uint32_t K, uint32_t D, const std::vector<float> &Om,
                          int omega_kind, uint32_t M, const std::vector<uint32_t> &cls, const BuildOptions &opt,
                          std::string &why, hipError_t &herr);
void engine_free(EngineModel *m);
const char *engine_name(const EngineModel *m);
const char *engine_ranking(const EngineModel *m);     // "bf16x3" | "f32": where the candidates are ranked; "none": no ranking stage
const char *engine_projection(const EngineModel *m);  // "bf16x6" | "f32": where y = Omega x is computed; "none": no projection
// queue: batch counters to use instead of the model's own slots (QUEUE_BYTES of device memory; the launch zeroes
// them), or null
hipError_t engine_assign(const EngineModel *m, const float *dX, uint64_t N, const uint32_t *dy, Outputs out,
                         hipStream_t s, uint32_t *queue = nullptr);

// Y = X . A^T on the matrix cores for an A (rows x D) resident in LDS; nullptr when D is not 32/64/128/256.
struct ProjectModel;
ProjectModel *project_build(const std::vector<float> &A, uint32_t rows, uint32_t D, const BuildOptions &opt, hipError_t &herr);
void project_free(ProjectModel *m);
bool project_on_bf16(const ProjectModel *m);          // the products run as six bf16 piece products (Geom3 tiles)
hipError_t project_run(const ProjectModel *m, const float *dX, uint64_t N, float *dY, uint32_t ldy, hipStream_t s);

// project / project_omega for a global Omega of at most 128 rows over rows above 256 values (kernels_long.hip)
struct LongProjectModel;
LongProjectModel *long_project_build(const std::vector<float> &Om, uint32_t M, uint32_t D, hipError_t &herr);
void long_project_free(LongProjectModel *m);
hipError_t long_project_run(const LongProjectModel *m, const float *dX, uint64_t N, bool lambda, float *dY, hipStream_t s);
bool long_project_has_lambda(const LongProjectModel *m);     // project (Lambda): Omega of at most 128 rows
// d[i] = d[i]^2 (rooted Euclidean distances of a search in the projected space -> helpers.rs:44)
hipError_t launch_square(float *d, uint64_t n, hipStream_t s);

// LGMLVQ::project on the matrix cores (kernels_project.hip): samples grouped by winner, Lambda_j from L2
struct LocalProjectModel;
LocalProjectModel *local_project_build(const std::vector<float> &Oms, uint32_t K, uint32_t M, uint32_t D, hipError_t &herr);
void local_project_free(LocalProjectModel *m);
uint64_t local_project_max_rows(const LocalProjectModel *m);
size_t local_project_scratch_words(const LocalProjectModel *m, uint64_t n);
hipError_t local_project_run(const LocalProjectModel *m, const float *dX, uint64_t n, const uint32_t *dwin, uint32_t *scratch,
                             float *dY, hipStream_t s);

// ---------------------------------------------------------------- batched GLVQ gradient (kernels_glvq.hip)
// scratch floats the pass needs for a K x D model; *nparts: partial sets it leaves (256: one per workgroup, LDS; 1: global atomics)
size_t glvq_gradient_scratch_floats(uint32_t K, uint32_t D, uint32_t *nparts);
// G (K x D, device) = (accumulate ? G : 0) + the batch gradient of the samples dX (N x D) whose matched search gave
// (jp, dp, jn, dn); dcost / dused (device, nullable) are added to
hipError_t launch_glvq_gradient(const float *dX, uint64_t N, uint32_t D, uint32_t K, const float *dW, const uint32_t *jp, const float *dp,
                                const uint32_t *jn, const float *dn, float *scratch, float *dG, bool accumulate, double *dcost,
                                unsigned long long *dused, hipStream_t s);

// ---------------------------------------------------------------- synth (synth.hip)
hipError_t launch_synth_fill(float *dout, uint64_t count, uint64_t first, uint64_t seed, uint32_t stream_id,
                             float scale, const float *dbase, uint64_t base_len, hipStream_t s);

}  // namespace vq
