/*
 * vqhip.h -- C ABI of the MI355X (gfx950) nearest-prototype library, libvqhip.
 *
 * This is the drop-in boundary for ONE path of GerritLuimstra/vq: batched distance
 * computation + winner (argmin) assignment, i.e. the bodies of
 *     find_closest_prototype              src/helpers.rs:55-77
 *     find_closest_prototype_matched      src/helpers.rs:91-132
 *     LGMLVQ::find_closest_local_prototype{,_matched}   src/lgmlvq/lgmlvq.rs:436-455, 468-506
 * as they are driven by the six `predict` loops (src/vq/vq.rs:142-160, src/lvq/lvq.rs:193-211,
 * src/glvq/glvq.rs:217-235, src/gmlvq/gmlvq.rs:270-290, src/lgmlvq/lgmlvq.rs:278-298,
 * src/liramlvq/liramlvq.rs:287-307) and by the matched double call of the fit loops
 * (src/glvq/glvq.rs:171-176, src/lgmlvq/lgmlvq.rs:194-199).
 *
 * The reference has no FFI of its own (pure Rust).  These are the entry points a Rust
 * `extern "C"` block for this path would bind; INTEGRATION.md shows that binding.
 *
 * Conventions
 *   - plain pointers and sizes only; all matrices row-major and contiguous
 *   - every function returns VQHIP_OK (0) or a negative vqhip_status; nothing throws or
 *     aborts across the boundary; vqhip_last_error() gives the text
 *   - a context is bound to one GPU and serves one caller at a time (not re-entrant),
 *     like `predict(&mut self)`.  One process per GPU; use one context per process/GPU.
 *   - the context owns all device memory and pinned staging; the caller owns every pointer
 *     it passes in; outputs are caller-allocated
 *   - numbers: the reference is f64; the device computes in f32 (BASELINE.json north_star).
 *     Euclidean results are ROOTED distances (src/helpers.rs:23), Omega results are the
 *     SQUARED quadratic form (src/helpers.rs:44).
 *   - winner semantics (src/helpers.rs:58-59,70): start at (index 0, +INF), replace on
 *     strict '<' => lowest index wins ties, NaN/+INF never win, no eligible winner =>
 *     index 0 with distance +INF.
 */
#ifndef VQHIP_H
#define VQHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vqhip_ctx vqhip_ctx; /* opaque */

typedef enum {
    VQHIP_OK = 0,
    VQHIP_ERR_INVALID = -1,     /* bad argument (NULL, zero size, shape mismatch)             */
    VQHIP_ERR_STATE = -2,       /* call order: prototypes / labels / omega not set            */
    VQHIP_ERR_HIP = -3,         /* a HIP runtime call failed (text has hipGetErrorString)     */
    VQHIP_ERR_NOMEM = -4,       /* host or device allocation failed                           */
    VQHIP_ERR_UNSUPPORTED = -5  /* forced kernel policy cannot run this shape                 */
} vqhip_status;

/* Which kernel family serves assign calls.  AUTO picks the MFMA engine whenever the shape
 * qualifies and the scalar kernels otherwise; the others are for tests and A/B timing. */
typedef enum {
    VQHIP_KERNEL_AUTO = 0,
    VQHIP_KERNEL_SCALAR = 1,    /* one lane per sample, direct (x-w)^2 form, index-order sums  */
    VQHIP_KERNEL_MFMA = 2       /* fp32 MFMA tile engine; VQHIP_ERR_UNSUPPORTED if not eligible */
} vqhip_kernel_policy;

/* ---- lifetime ------------------------------------------------------------------------ */

/* Creates a context on HIP device `device`.  Replaces nothing in the reference (it has no
 * device); it is what a Rust `Drop`-owning wrapper would hold inside the model struct. */
int vqhip_create(int device, vqhip_ctx **out);   /* (several devices behind one handle: vqhip_multi_create below) */
int vqhip_destroy(vqhip_ctx *ctx);

/* Text of the last failure on this context ("" if none).  ctx == NULL: last vqhip_create failure. */
const char *vqhip_last_error(const vqhip_ctx *ctx);

/* ---- model state (replaces the fields of src/lib.rs:30-34,186,226,270) ------------------ */

/* Prototype vectors: `prototypes[j].vector` flattened, K x D (src/lib.rs:30-34).  Resets omega,
 * local omegas and labels, which depend on K / D. */
int vqhip_set_prototypes(vqhip_ctx *ctx, const float *W, uint32_t K, uint32_t D);

/* Prototype class ids: `prototypes[j].name` interned to dense ids (src/helpers.rs:113 compares
 * the strings by value; equal ids <=> equal names).  Needed only by the matched calls. */
int vqhip_set_labels(vqhip_ctx *ctx, const uint32_t *cls, uint32_t K);

/* Global Omega, M x D: GMLVQ `omega` (src/lib.rs:186, M == D) or LiRaMLVQ `omega`
 * (src/lib.rs:270, M = max_rank).  Om == NULL clears it (back to Euclidean). */
int vqhip_set_omega(vqhip_ctx *ctx, const float *Om, uint32_t M, uint32_t D);

/* Local Omegas, K x M x D: LGMLVQ `omegas[j]` (src/lib.rs:226).  Oms == NULL clears. */
int vqhip_set_local_omegas(vqhip_ctx *ctx, const float *Oms, uint32_t K, uint32_t M, uint32_t D);

int vqhip_set_kernel_policy(vqhip_ctx *ctx, int policy /* vqhip_kernel_policy */);

/* Which matrix cores the MFMA kernels use.  No counterpart in the reference (f64 scalar code); results do not depend on
 * VQHIP_OPT_RANKING at all (vqhip_last_ranking below) and on VQHIP_OPT_PROJECTION only to rounding, both settings inside
 * the 1e-5 bar (vqhip_last_projection).  Setting an option invalidates the device images like any vqhip_set_*; the
 * next assign / project / vqhip_prepare call rebuilds them.
 *   VQHIP_OPT_RANKING     where candidates are ranked: VQHIP_VALUE_BF16 = split-bf16 products ("bf16x3"), VQHIP_VALUE_F32
 *   VQHIP_OPT_PROJECTION  where y = Omega x is computed: VQHIP_VALUE_BF16 = six bf16 piece products ("bf16x6", needs the
 *                         bf16 ranking too), VQHIP_VALUE_F32
 *   VQHIP_VALUE_DEFAULT   bf16 for both, unless the process environment holds VQHIP_NO_BF (both f32) or VQHIP_NO_BF_PROJ
 *                         (projection f32): the environment only moves the default, an explicit value always wins. */
typedef enum { VQHIP_OPT_RANKING = 0, VQHIP_OPT_PROJECTION = 1 } vqhip_option;
typedef enum { VQHIP_VALUE_DEFAULT = 0, VQHIP_VALUE_F32 = 1, VQHIP_VALUE_BF16 = 2 } vqhip_option_value;
int vqhip_set_option(vqhip_ctx *ctx, int option /* vqhip_option */, int value /* vqhip_option_value */);

/* ---- the hot path ------------------------------------------------------------------------ */

/* find_closest_prototype for N samples (src/helpers.rs:55-77; local Omegas:
 * src/lgmlvq/lgmlvq.rs:436-455).  X is N x D.  idx_out[i] = winner index, dist_out[i] = its
 * distance (rooted for Euclidean, squared for Omega models); dist_out may be NULL.
 * Host-pointer form: samples go up in pinned, double-buffered chunks. */
int vqhip_assign(vqhip_ctx *ctx, const float *X, uint64_t N, uint32_t *idx_out, float *dist_out);

/* The same for the reference's own container: `data: &Vec<Array1<f64>>` (src/vq/vq.rs:142), N separately
 * allocated rows of D doubles.  rows[i] points at row i; every row must hold D values (the reference
 * checks only data[0], src/vq/vq.rs:69, and panics inside euclidean_distance on a later ragged row,
 * src/helpers.rs:14 -- the caller asserts the lengths).  The rows are gathered and narrowed to f32
 * straight into the pinned upload buffers by the host cores, overlapped with the previous chunk's
 * upload and kernel. */
int vqhip_assign_rows_f64(vqhip_ctx *ctx, const double *const *rows, uint64_t N, uint32_t D, uint32_t *idx_out,
                          float *dist_out);

/* Same with device-resident X / outputs, enqueued on `stream` (a hipStream_t; NULL = the
 * default stream) without host synchronisation.  Up to three kernels and one memset go on the stream: samples
 * whose ranking the first kernel cannot vouch for (a third candidate within the rounding margin of the best: dense or
 * duplicated prototypes, binary features, a model in the reference's initial state, src/lvq/lvq.rs:120) are marked, packed
 * into a list and finished exactly by a second launch; the context keeps about 22 B per row of its
 * largest call for that (four round-robin slots of marks and list), allocated in the first call of a size (which
 * synchronises the device once). */
int vqhip_assign_device(vqhip_ctx *ctx, const float *dX, uint64_t N, uint32_t *didx_out,
                        float *ddist_out, void *stream);

/* Builds the device images of the model now instead of inside the first assign call after a vqhip_set_*
 * (they are device allocations + uploads).  No counterpart in the reference: its model lives in host Vecs. */
int vqhip_prepare(vqhip_ctx *ctx);

/* vqhip_assign_device for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  It has NO counterpart in the reference; results are those of
 * vqhip_assign_device bit for bit.  It calls no hipMalloc / hipFree / allocating routine: the only device memory the
 * kernels need beyond inputs and outputs -- the batch counters they claim work from, 64 KiB -- lives in a
 * file-scope __device__ array of the library, one copy per GPU for the life of the process.
 *   - the model must be prepared (vqhip_prepare, or any assign call since the last vqhip_set_*): VQHIP_ERR_STATE otherwise
 *   - it serves the models one MFMA kernel serves (what vqhip_last_kernel reports as mfma_* except
 *     mfma_omega_two_pass); for the others VQHIP_ERR_UNSUPPORTED (their scratch grows with N)
 *   - N <= VQHIP_NOALLOC_MAX_ROWS, VQHIP_ERR_INVALID beyond (the counters are 32-bit)
 *   - every call on one GPU shares that GPU's counters: calls that overlap on one GPU (two streams, two contexts,
 *     two threads) race.  Order them on one stream or use vqhip_assign_device.
 *   - one launch, no finish pass (its list would have to be allocated): samples the ranking cannot settle are scanned
 *     exactly inside their wave, as in round 2 -- same results, slower on the inputs DESIGN.md 4.12 lists. */
#define VQHIP_NOALLOC_MAX_ROWS (1ull << 36)
int vqhip_assign_device_noalloc(vqhip_ctx *ctx, const float *dX, uint64_t N, uint32_t *didx_out,
                                float *ddist_out, void *stream);

/* find_closest_prototype_matched, both calls of the fit loop at once (src/helpers.rs:91-132 as
 * called at src/glvq/glvq.rs:171-176; local: src/lgmlvq/lgmlvq.rs:468-506 as called at :194-199).
 * y[i] = class id of sample i.  jpos/dpos: nearest prototype with cls == y[i];
 * jneg/dneg: nearest with cls != y[i].  Any output pointer may be NULL. */
int vqhip_assign_matched(vqhip_ctx *ctx, const float *X, const uint32_t *y, uint64_t N,
                         uint32_t *jpos, float *dpos, uint32_t *jneg, float *dneg);
int vqhip_assign_matched_device(vqhip_ctx *ctx, const float *dX, const uint32_t *dy, uint64_t N,
                                uint32_t *djpos, float *ddpos, uint32_t *djneg, float *ddneg,
                                void *stream);
/* The same for the reference's own container (N separately allocated rows of D doubles, as vqhip_assign_rows_f64):
 * what an evaluation of a whole data set at the start of an epoch passes in (src/glvq/glvq.rs:147-149, 171-184). */
int vqhip_assign_matched_rows_f64(vqhip_ctx *ctx, const double *const *rows, const uint32_t *y, uint64_t N, uint32_t D,
                                  uint32_t *jpos, float *dpos, uint32_t *jneg, float *dneg);

/* ---- batched GLVQ gradient (second half) -- NOT the reference's algorithm ------------------
 * The reference's GLVQ::fit (src/glvq/glvq.rs:163-206) moves the two nearest prototypes after every single sample, so
 * sample i + 1 is measured against what sample i left behind; that loop stays on the host.  This call is the BATCH form
 * of its two update terms with the prototypes held fixed (full-batch / mini-batch GLVQ: a different algorithm whose
 * iterates differ from the reference's):
 *     dW[k] = sum over samples whose nearest same-class prototype is k of      -4 d- / (d+ + d-)^2 (x - w_k)   (glvq.rs:193)
 *           + sum over samples whose nearest other-class prototype is k of     +4 d+ / (d+ + d-)^2 (x - w_k)   (glvq.rs:194)
 * i.e. a caller's step is W <- W - lr * dW (glvq.rs:200-201 summed).  One matched search (vqhip_assign_matched) plus one
 * HBM-bound scatter-add pass over the rows on the device.  Euclidean models with labels only (VQHIP_ERR_UNSUPPORTED with an
 * Omega).  Samples with no prototype of / outside their class, or with d+ + d- = 0, are left out (the reference would
 * divide by zero).  cost (nullable): sum of (d+ - d-) / (d+ + d-) over the samples that count; used (nullable): how many
 * did.  dW: K x D, overwritten.  Sums are f32 and their order is not fixed: results agree with an f64 evaluation to
 * rounding (tests: 1e-4 of the largest entry), not bit for bit. */
int vqhip_glvq_batch_gradient(vqhip_ctx *ctx, const float *X, const uint32_t *y, uint64_t N, float *dW, double *cost, uint64_t *used);
/* Device-resident form, enqueued on `stream`: ddW K x D floats on the device; dcost (nullable): 16 bytes on the device,
 * {double cost, uint64 used}, zeroed and filled by the call. */
int vqhip_glvq_batch_gradient_device(vqhip_ctx *ctx, const float *dX, const uint32_t *dy, uint64_t N, float *ddW, double *dcost,
                                     void *stream);

/* ---- several GPUs of one node behind one handle (SURVEY.md section 8 rows b / e) ----------------------------------
 * For a single-process caller (the Rust crate's `predict`, src/glvq/glvq.rs:217-235, whose samples are independent,
 * :224-232): one context per listed device, the model replicated on each, the samples of a call cut into contiguous
 * shards [g * ceil(N / n), (g + 1) * ceil(N / n)), one host thread per device driving that context's pinned upload
 * pipeline, results written into the caller's arrays.  No exchange between the GPUs is needed for host-resident
 * results.  (Device-resident shards, one process per GPU, with the RCCL all-gather of the results: vq_amd/dist.py,
 * bench.py.)  A device may be listed more than once (two contexts on it).  Same semantics, status codes and
 * threading contract (one caller at a time) as the single-device calls they forward to. */
typedef struct vqhip_multi vqhip_multi; /* opaque */
int vqhip_multi_create(const int *devices, int n_devices, vqhip_multi **out);
int vqhip_multi_destroy(vqhip_multi *m);
const char *vqhip_multi_last_error(const vqhip_multi *m);   /* m == NULL: last vqhip_multi_create failure */
int vqhip_multi_device_count(const vqhip_multi *m);
int vqhip_multi_set_prototypes(vqhip_multi *m, const float *W, uint32_t K, uint32_t D);
int vqhip_multi_set_labels(vqhip_multi *m, const uint32_t *cls, uint32_t K);
int vqhip_multi_set_omega(vqhip_multi *m, const float *Om, uint32_t M, uint32_t D);
int vqhip_multi_set_local_omegas(vqhip_multi *m, const float *Oms, uint32_t K, uint32_t M, uint32_t D);
int vqhip_multi_set_option(vqhip_multi *m, int option, int value);
int vqhip_multi_assign(vqhip_multi *m, const float *X, uint64_t N, uint32_t *idx_out, float *dist_out);
int vqhip_multi_assign_rows_f64(vqhip_multi *m, const double *const *rows, uint64_t N, uint32_t D, uint32_t *idx_out,
                                float *dist_out);
int vqhip_multi_assign_matched(vqhip_multi *m, const float *X, const uint32_t *y, uint64_t N, uint32_t *jpos,
                               float *dpos, uint32_t *jneg, float *dneg);
/* vqhip_glvq_batch_gradient over the handle's devices: every device the gradient of its shard, summed on the host */
int vqhip_multi_glvq_batch_gradient(vqhip_multi *m, const float *X, const uint32_t *y, uint64_t N, float *dW, double *cost,
                                    uint64_t *used);

/* The same search for shards that LIVE on the GPUs, with the RCCL exchange of the results (north star: "RCCL over xGMI is
 * used only to gather winner indices and distances"; the samples of `predict` are independent, src/glvq/glvq.rs:224-232).
 * dX[g]: the shard of device g (n_rows[g] x D, on that device; global rows [first_g, first_g + n_rows[g]) with
 * first_g = n_rows[0] + .. + n_rows[g-1]).  didx_all[g] / ddist_all[g]: arrays of sum(n_rows) elements ON device g.
 * Every device searches its shard on its own stream into its own slice, then the slices are exchanged in place by one
 * grouped ncclAllGather per array (equal shards; unequal ones: one ncclBroadcast per shard, grouped) over a
 * single-process communicator clique (ncclCommInitAll) held by the handle -- so every device ends up with the winners
 * of ALL samples.  ddist_all may be NULL.  Returns after all streams have drained.  A device listed twice in the handle:
 * VQHIP_ERR_UNSUPPORTED (one rank per GPU).  librccl.so.1 is bound at the first call (dlopen), so that a process which
 * already maps a copy -- PyTorch ships its own -- keeps exactly that one, and callers that never gather need no RCCL;
 * VQHIP_ERR_UNSUPPORTED when it cannot be found.  Measured on one GPU only so far (a clique of one rank); the 8-GPU exchange
 * is 8 bytes per sample against >= 520 bytes read. */
int vqhip_multi_assign_device(vqhip_multi *m, const float *const *dX, const uint64_t *n_rows, uint32_t *const *didx_all,
                              float *const *ddist_all);

/* ---- projections ---------------------------------------------------------
 * project / project_omega for N samples.  X is N x D, row-major.
 *   VQHIP_PROJECT_OMEGA   Y = Omega x,  N x M   LiRaMLVQ::project_omega   src/liramlvq/liramlvq.rs:429-446
 *   VQHIP_PROJECT_LAMBDA  Y = Lambda x = Omega^T (Omega x),  N x D
 *                                               GMLVQ::project     src/gmlvq/gmlvq.rs:354-372
 *                                               LiRaMLVQ::project  src/liramlvq/liramlvq.rs:401-419
 *                         with local Omegas: Y_i = Lambda_j x_i, j = nearest local prototype of x_i
 *                                               LGMLVQ::project    src/lgmlvq/lgmlvq.rs:372-393
 * A global Omega with D <= 256 runs on the matrix cores (k_project: Omega / Lambda resident in LDS; Lambda is
 * built once in f64 as the reference builds it once, gmlvq.rs:357-358); LGMLVQ with D <= 256 groups the samples
 * by winner on the device and runs every group against its Lambda_j (k_project_local); a global Omega of at most
 * 128 rows over D > 256 runs as Omega^T (Omega x) on the matrix cores (k_project_long); the rest on the scalar kernels. */
typedef enum { VQHIP_PROJECT_OMEGA = 0, VQHIP_PROJECT_LAMBDA = 1 } vqhip_projection;
int vqhip_project(vqhip_ctx *ctx, const float *X, uint64_t N, int which /* vqhip_projection */, float *Y);
int vqhip_project_device(vqhip_ctx *ctx, const float *dX, uint64_t N, int which, float *dY, void *stream);
/* The same for the reference's container on both sides (`data: &Vec<Array1<f64>>` in, `Vec<Array1<f64>>` out:
 * src/gmlvq/gmlvq.rs:354-372): rows[i] points at D doubles, out_rows[i] at room for the projected row (M doubles for
 * VQHIP_PROJECT_OMEGA, D for VQHIP_PROJECT_LAMBDA).  Rows are gathered and narrowed into the pinned upload buffers and the
 * results widened straight out of the pinned download buffers by the host cores, overlapped with the transfers. */
int vqhip_project_rows_f64(vqhip_ctx *ctx, const double *const *rows, uint64_t N, uint32_t D, int which, double *const *out_rows);

/* ---- the Euclidean path in the reference's own number type ------------
 * euclidean_distance + find_closest_prototype{,_matched} in f64 (src/helpers.rs:11-24, 55-77, 91-132):
 * the same operations in the same order as the reference (index-order sum, separate multiply and
 * add, IEEE sqrt), so index AND distance are bit-identical to it.  This is the on-device check of
 * the argmin / tie / NaN logic without f32 rounding in the way; it is not the fast path.
 * vqhip_set_prototypes_f64 also installs the narrowed f32 model (resetting labels and Omega like
 * vqhip_set_prototypes), so both paths describe the same prototypes; set labels afterwards. */
int vqhip_set_prototypes_f64(vqhip_ctx *ctx, const double *W, uint32_t K, uint32_t D);
int vqhip_assign_f64(vqhip_ctx *ctx, const double *X, uint64_t N, uint32_t *idx_out, double *dist_out);
int vqhip_assign_matched_f64(vqhip_ctx *ctx, const double *X, const uint32_t *y, uint64_t N, uint32_t *jpos,
                             double *dpos, uint32_t *jneg, double *dneg);

/* ---- introspection / measurement support ------------------------------------------------- */

/* Name of the kernel family the last assign call on this context used ("scalar", "mfma_euclid",
 * "mfma_omega", "mfma_local", ...).  For tests and bench reports. */
const char *vqhip_last_kernel(const vqhip_ctx *ctx);

/* Where the last assign call ranked its candidates: "f32" (v_mfma_f32_32x32x2_f32), "bf16x3" (every f32 operand split
 * into two bf16 pieces, three piece products on v_mfma_f32_32x32x16_bf16; vqhip_set_option(VQHIP_OPT_RANKING, VQHIP_VALUE_F32)
 * turns it off) or "none" (kernels without a ranking stage: direct, local, scalar).  Ranking only picks the candidates: the
 * winners and distances that come back are measured in f32 in the reference's direct form either way, and an in-kernel
 * guard sends every sample whose ranking could be wrong to an exact scan, so both settings return the same results. */
const char *vqhip_last_ranking(const vqhip_ctx *ctx);

/* Where the last assign call computed the projection y = Omega x of the Omega models (src/helpers.rs:41-43): "f32"
 * (v_mfma_f32_32x32x2_f32 / 16x16x4), "bf16x6" (Omega and x each split into three bf16 pieces that add up to the f32
 * value exactly, the six piece products down to 2^-16 of the leading one on v_mfma_f32_32x32x16_bf16, accumulated in
 * f32: measured at or below the f32 chain's error, tools/bf16x6_probe.hip; vqhip_set_option(VQHIP_OPT_PROJECTION, VQHIP_VALUE_F32) turns it off) or "none"
 * (Euclidean models).  Unlike the ranking this is part of the result: the distances are measured on this y, to the
 * same 1e-5 bar against the f64 oracle either way. */
const char *vqhip_last_projection(const vqhip_ctx *ctx);

/* Counter-based synthetic data: value(seed, stream, i) in [-1, 1) as f32.
 * The host and device fills produce identical bits, so the CPU checker can regenerate any index
 * range of a device-resident matrix.  out[i] = base_scale * (base ? base[(first+i) % base_len] : 0)
 *                                            + scale * value(seed, stream, first + i)           */
int vqhip_synth_fill_host(float *out, uint64_t count, uint64_t first, uint64_t seed, uint32_t stream_id,
                          float scale, const float *base, uint64_t base_len);
int vqhip_synth_fill_device(vqhip_ctx *ctx, float *dout, uint64_t count, uint64_t first, uint64_t seed,
                            uint32_t stream_id, float scale, const float *dbase, uint64_t base_len,
                            void *stream);

const char *vqhip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VQHIP_H */
