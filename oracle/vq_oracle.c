/*
 * vq_oracle.c -- CPU restatement (f64) of the nearest-prototype hot path of
 * GerritLuimstra/vq.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.  The product library (vq_amd/libvqhip.so)
 * never links, loads or calls anything in this directory.
 *
 * What is restated (reference file:line, all under src/ of the reference):
 *   vqo_euclidean_distance            helpers.rs:11-24
 *   vqo_generalized_distance          helpers.rs:35-45   (faithful: Lambda rebuilt per call)
 *   vqo_generalized_distance_lambda   helpers.rs:41-44   (Lambda hoisted by the caller)
 *   vqo_find_closest_prototype        helpers.rs:55-77
 *   vqo_find_closest_prototype_matched helpers.rs:91-132
 *   vqo_find_closest_local_prototype          lgmlvq/lgmlvq.rs:436-455
 *   vqo_find_closest_local_prototype_matched  lgmlvq/lgmlvq.rs:468-506
 *   vqo_normalize_omega               gmlvq/gmlvq.rs:112-122
 *   vqo_project_lambda / vqo_project_omega    gmlvq/gmlvq.rs:354-372, liramlvq/liramlvq.rs:429-446
 *
 * Pinning status ("parity unpinned" at the ndarray boundary):
 *   - The reference is Rust and cannot be built in this environment (no cargo/rustc, no
 *     crate sources), so this file was never run against the reference itself.
 *   - euclidean_distance and the argmin loops contain no third-party arithmetic; they
 *     are restated operation for operation (same loop order, x*x for powi(2), sqrt,
 *     strict '<' from (index 0, +INF)).
 *   - generalized_distance goes through ndarray 0.15's `.dot` (Cargo.toml:12, caret
 *     requirement, no lockfile).  ndarray's source is absent, its f64 summation order is
 *     not determinable here; this file uses plain index-order sums.  No reference test
 *     asserts any distance value, so at that boundary the oracle is PARITY UNPINNED;
 *     it is pinned at label level by the known-answer fixtures in tests/golden/
 *     derived from the reference's own tests (tests/vq.rs:164-241, tests/lvq.rs:283-334,
 *     tests/gmlvq.rs:283-325, ...).
 *
 * Label strings: the reference compares prototype names by value
 * (helpers.rs:113,122).  Here names are interned to uint32 class ids by the caller;
 * equality of ids == equality of strings.
 *
 * Extras that are NOT in the reference (for the parity harness only): every search
 * also reports the winning distance and the runner-up distance among the eligible
 * prototypes so near-ties can be flagged.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ distances */

/* helpers.rs:11-24.  The reference asserts equal lengths (:14); here one length is passed. */
double vqo_euclidean_distance(const double *vec1, const double *vec2, size_t n)
{
    double sum = 0.0;
    for (size_t index = 0; index < n; ++index) {
        double diff = vec1[index] - vec2[index];
        sum += diff * diff;            /* powi(2) */
    }
    return sqrt(sum);
}

/* Lambda = Omega^T Omega  (helpers.rs:38).  omega is m x n row-major, lambda n x n. */
void vqo_lambda(const double *omega, size_t m, size_t n, double *lambda)
{
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j) {
            double s = 0.0;
            for (size_t r = 0; r < m; ++r)
                s += omega[r * n + i] * omega[r * n + j];
            lambda[i * n + j] = s;
        }
}

/* helpers.rs:41-44 with Lambda given: diff = vec1 - vec2; diff^T . Lambda . diff (squared, no sqrt).
 * scratch: n doubles. */
double vqo_generalized_distance_lambda(const double *lambda, size_t n, const double *vec1,
                                       const double *vec2, double *scratch)
{
    double *diff = scratch;
    for (size_t i = 0; i < n; ++i)
        diff[i] = vec1[i] - vec2[i];
    /* (diff^T . Lambda)[j] = sum_i diff[i] * Lambda[i][j];  then . diff */
    double total = 0.0;
    for (size_t j = 0; j < n; ++j) {
        double t = 0.0;
        for (size_t i = 0; i < n; ++i)
            t += diff[i] * lambda[i * n + j];
        total += t * diff[j];
    }
    return total;
}

/* helpers.rs:35-45, faithful: Lambda rebuilt on every call (:38).
 * scratch: n*n + n doubles. */
double vqo_generalized_distance(const double *omega, size_t m, size_t n, const double *vec1,
                                const double *vec2, double *scratch)
{
    double *lambda = scratch;
    vqo_lambda(omega, m, n, lambda);
    return vqo_generalized_distance_lambda(lambda, n, vec1, vec2, scratch + n * n);
}

/* --------------------------------------------------------------------- argmin */

/* How the distance of prototype `index` to `sample` is obtained. */
typedef struct {
    const double *prototypes; /* K x D row-major                                          */
    size_t K, D;
    const double *omega;      /* NULL (euclidean), M x D (global), or K x M x D (local)    */
    size_t M;
    int local;                /* 1: prototype j uses omega + j*M*D  (lgmlvq.rs:445)        */
    const double *lambda;     /* optional hoisted Lambda(s): D x D, or K x D x D if local  */
    double *scratch;          /* D*D + D doubles                                           */
} vqo_model;

static double model_distance(const vqo_model *mdl, size_t index, const double *sample)
{
    const double *proto = mdl->prototypes + index * mdl->D;
    if (!mdl->omega)
        return vqo_euclidean_distance(proto, sample, mdl->D);       /* helpers.rs:65 (prototype, sample) */
    if (mdl->lambda) {
        const double *lam = mdl->lambda + (mdl->local ? index * mdl->D * mdl->D : 0);
        return vqo_generalized_distance_lambda(lam, mdl->D, sample, proto, mdl->scratch);
    }
    const double *om = mdl->omega + (mdl->local ? index * mdl->M * mdl->D : 0);
    return vqo_generalized_distance(om, mdl->M, mdl->D, sample, proto, mdl->scratch); /* helpers.rs:66 */
}

/*
 * helpers.rs:55-77 (labels == NULL) and helpers.rs:91-132 (labels given);
 * lgmlvq.rs:436-455 / 468-506 when mdl->local.
 *   find_closest_matching != 0: only prototypes with cls == label are eligible (:113)
 *   find_closest_matching == 0: only prototypes with cls != label are eligible (:122)
 * Returns the index; init (0, +INF), update on strict '<' -> lowest index wins ties,
 * NaN / +INF never win, nothing eligible -> 0.
 * best_out / second_out (may be NULL): harness extras, see header.
 */
static size_t closest_impl(const vqo_model *mdl, const double *sample, const uint32_t *proto_cls,
                           int use_label, uint32_t label, int find_closest_matching,
                           double *best_out, double *second_out)
{
    size_t closest_prototype_index = 0;
    double smallest_distance = INFINITY;
    double second = INFINITY;

    for (size_t index = 0; index < mdl->K; ++index) {
        double distance = model_distance(mdl, index, sample);
        int eligible = 1;
        if (use_label)
            eligible = find_closest_matching ? (proto_cls[index] == label) : (proto_cls[index] != label);
        if (distance < smallest_distance && eligible) {
            second = smallest_distance;
            closest_prototype_index = index;
            smallest_distance = distance;
        } else if (eligible && distance < second) {
            second = distance;
        }
    }
    if (best_out) *best_out = smallest_distance;
    if (second_out) *second_out = second;
    return closest_prototype_index;
}

size_t vqo_find_closest_prototype(const double *prototypes, size_t K, size_t D, const double *sample,
                                  const double *omega, size_t M, double *scratch)
{
    vqo_model mdl = {prototypes, K, D, omega, M, 0, NULL, scratch};
    return closest_impl(&mdl, sample, NULL, 0, 0, 0, NULL, NULL);
}

size_t vqo_find_closest_prototype_matched(const double *prototypes, const uint32_t *proto_cls, size_t K,
                                          size_t D, const double *sample, uint32_t label,
                                          int find_closest_matching, const double *omega, size_t M,
                                          double *scratch)
{
    vqo_model mdl = {prototypes, K, D, omega, M, 0, NULL, scratch};
    return closest_impl(&mdl, sample, proto_cls, 1, label, find_closest_matching, NULL, NULL);
}

size_t vqo_find_closest_local_prototype(const double *prototypes, size_t K, size_t D, const double *sample,
                                        const double *omegas, size_t M, double *scratch)
{
    vqo_model mdl = {prototypes, K, D, omegas, M, 1, NULL, scratch};
    return closest_impl(&mdl, sample, NULL, 0, 0, 0, NULL, NULL);
}

size_t vqo_find_closest_local_prototype_matched(const double *prototypes, const uint32_t *proto_cls,
                                                size_t K, size_t D, const double *sample, uint32_t label,
                                                int find_closest_matching, const double *omegas, size_t M,
                                                double *scratch)
{
    vqo_model mdl = {prototypes, K, D, omegas, M, 1, NULL, scratch};
    return closest_impl(&mdl, sample, proto_cls, 1, label, find_closest_matching, NULL, NULL);
}

/* ------------------------------------------------------------ batch drivers */

/*
 * The per-sample loop of the six `predict`s (vq.rs:149-157, lvq.rs:200-208, glvq.rs:224-232,
 * gmlvq.rs:277-287, lgmlvq.rs:285-295, liramlvq.rs:294-304) without the name clone:
 * for every sample, the winner index.
 *
 *   X            N x D samples, f32 (x_is_f32) widened to f64, or f64
 *   omega_kind   0 none, 1 global (M x D), 2 local (K x M x D)
 *   hoist        0 faithful (Lambda rebuilt per (sample, prototype) pair, helpers.rs:38)
 *                1 Lambda built once per model -- same numbers, fewer flops
 *   labels       NULL for the plain search; else N sample class ids, and `matching`
 *                selects helpers.rs:113 (1) or :122 (0)
 *   threads      worker threads over samples (the reference itself is single-threaded)
 * Outputs: idx[N]; best[N], second[N] may be NULL.
 * Returns 0, or -1 on allocation failure.
 */
int vqo_assign_batch(const void *X, int x_is_f32, size_t N, size_t D, const double *prototypes,
                     size_t K, const uint32_t *proto_cls, const double *omega, int omega_kind,
                     size_t M, int hoist, const uint32_t *labels, int matching, int threads,
                     uint32_t *idx, double *best, double *second)
{
    double *lambda = NULL;
    if (omega_kind != 0 && hoist) {
        size_t nl = (omega_kind == 2) ? K : 1;
        lambda = (double *)malloc(nl * D * D * sizeof(double));
        if (!lambda) return -1;
        for (size_t j = 0; j < nl; ++j)
            vqo_lambda(omega + j * M * D, M, D, lambda + j * D * D);
    }
    int failed = 0;
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
        double *scratch = (double *)malloc((D * D + 2 * D) * sizeof(double));
        if (!scratch) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            failed = 1;
        } else {
            double *row = scratch + D * D + D;
            vqo_model mdl = {prototypes, K, D, omega_kind ? omega : NULL, M, omega_kind == 2, lambda, scratch};
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
            for (long long i = 0; i < (long long)N; ++i) {
                const double *sample;
                if (x_is_f32) {
                    const float *xf = (const float *)X + (size_t)i * D;
                    for (size_t d = 0; d < D; ++d) row[d] = (double)xf[d];
                    sample = row;
                } else {
                    sample = (const double *)X + (size_t)i * D;
                }
                double b, s;
                size_t w = closest_impl(&mdl, sample, proto_cls, labels != NULL, labels ? labels[i] : 0,
                                        matching, &b, &s);
                idx[i] = (uint32_t)w;
                if (best) best[i] = b;
                if (second) second[i] = s;
            }
            free(scratch);
        }
    }
    free(lambda);
    return failed ? -1 : 0;
}

/* All K distances of one sample (harness helper: lets tests inspect the full row). */
int vqo_all_distances(const double *sample, size_t D, const double *prototypes, size_t K,
                      const double *omega, int omega_kind, size_t M, double *out)
{
    double *scratch = (double *)malloc((D * D + D) * sizeof(double));
    if (!scratch) return -1;
    vqo_model mdl = {prototypes, K, D, omega_kind ? omega : NULL, M, omega_kind == 2, NULL, scratch};
    for (size_t k = 0; k < K; ++k) out[k] = model_distance(&mdl, k, sample);
    free(scratch);
    return 0;
}

/* ------------------------------------------------------------ model helpers */

/* gmlvq.rs:112-122: omega / sqrt(trace(omega^T omega)).  In place. */
void vqo_normalize_omega(double *omega, size_t m, size_t n)
{
    /* diag(omega^T omega)[i] = sum_r omega[r][i]^2 ; diagonal_sum = sum_i of that */
    double diagonal_sum = 0.0;
    for (size_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (size_t r = 0; r < m; ++r) s += omega[r * n + i] * omega[r * n + i];
        diagonal_sum += s;
    }
    double root = sqrt(diagonal_sum);
    for (size_t e = 0; e < m * n; ++e) omega[e] /= root;
}

/* gmlvq.rs:354-372 / liramlvq.rs:401-419: projected = Lambda . sample, Lambda = omega^T omega. */
int vqo_project_lambda(const double *omega, size_t m, size_t n, const double *X, size_t N, double *out)
{
    double *lambda = (double *)malloc(n * n * sizeof(double));
    if (!lambda) return -1;
    vqo_lambda(omega, m, n, lambda);
    for (size_t s = 0; s < N; ++s)
        for (size_t i = 0; i < n; ++i) {
            double t = 0.0;
            for (size_t j = 0; j < n; ++j) t += lambda[i * n + j] * X[s * n + j];
            out[s * n + i] = t;
        }
    free(lambda);
    return 0;
}

/* liramlvq.rs:429-446: projected = omega . sample (M values per sample). */
void vqo_project_omega(const double *omega, size_t m, size_t n, const double *X, size_t N, double *out)
{
    for (size_t s = 0; s < N; ++s)
        for (size_t r = 0; r < m; ++r) {
            double t = 0.0;
            for (size_t j = 0; j < n; ++j) t += omega[r * n + j] * X[s * n + j];
            out[s * m + r] = t;
        }
}

int vqo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
