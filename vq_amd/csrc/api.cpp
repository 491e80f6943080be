// libvqhip: context, model upload, dispatch, pinned double-buffered host path.
// The ABI is declared (with the reference citations) in include/vqhip.h.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include "mfma_host.h"
#include "vq_common.h"

using namespace vq;

// ---- batch counters of vqhip_assign_device_noalloc (include/vqhip.h) ----------------------------------------------
// The kernels claim their batches from QUEUE_GROUPS counters in device memory (BlockQueue, mfma_device.h), which every
// launch zeroes first.  The other entry points take them from the model's own hipMalloc'ed slots; the entry point
// that may not allocate takes them from this array.  Every GPU the library runs on holds its own copy (64 KiB) at its
// own address until the process ends.  In host code the NAME below is only a handle for hipGetSymbolAddress: what a
// kernel or hipMemsetAsync gets is the per-device address looked up once in noalloc_scratch_setup().
static __device__ __align__(256) uint32_t scratch[QUEUE_BYTES / sizeof(uint32_t)];

namespace {
thread_local std::string g_create_error;

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
uint32_t *g_scratch_ptr[MAX_DEVICES] = {};      // device address of `scratch` per GPU, null until looked up

// With `device` current: the address of this GPU's copy of `scratch`, looked up once per device and process.
hipError_t noalloc_scratch_setup(int device, uint32_t **out)
{
    *out = nullptr;
    if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = (uint32_t *)p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

struct Staging {                 // one of the two halves of the host-path pipeline
    hipStream_t stream = nullptr;
    float *hX = nullptr;     float *dX = nullptr;          // samples
    uint32_t *hY = nullptr;  uint32_t *dY = nullptr;       // sample class ids (matched)
    uint32_t *hI[2] = {nullptr, nullptr};  uint32_t *dI[2] = {nullptr, nullptr};
    float *hD[2] = {nullptr, nullptr};     float *dD[2] = {nullptr, nullptr};
    float *hP = nullptr;     float *dP = nullptr;          // projected rows (vqhip_project), p_cap floats each
    size_t p_cap = 0;
};
}  // namespace

struct vqhip_ctx {
    int device = 0;
    std::string err;
    const char *last_kernel = "none";
    const char *last_ranking = "none";
    const char *last_projection = "none";
    int policy = VQHIP_KERNEL_AUTO;
    int opt_ranking = VQHIP_VALUE_DEFAULT, opt_projection = VQHIP_VALUE_DEFAULT;   // vqhip_set_option

    uint32_t K = 0, D = 0, M = 0;
    int omega_kind = OMEGA_NONE;
    std::vector<float> hW, hOm;
    std::vector<uint32_t> hcls;

    float *dW = nullptr;       // K x D
    double *dW64 = nullptr;    // K x D in the reference's own number type (f64 entry points only)
    uint32_t *dcls = nullptr;  // K
    float *dOm = nullptr;      // M x D  or  K x M x D
    float *dP = nullptr;       // K x M: W . Omega^T (global Omega only)
    float *dOmT = nullptr;     // D x M: Omega^T (global Omega only; second half of the Lambda projection)
    float *dProj = nullptr;    // projection scratch: chunk x M intermediate, chunk x u32 winners
    size_t dProj_cap = 0;
    float *dYproj = nullptr;   // scalar-path scratch: projected samples, chunk x M
    size_t dYproj_cap = 0;
    // The scratch buffers are shared by every stream the context is driven from (the host path alternates
    // two): the last kernel that uses one records this event, the next user waits for it on its stream.
    hipEvent_t scratch_done = nullptr;

    uint32_t *noalloc_queue = nullptr;   // this GPU's copy of `scratch` (device address, set in vqhip_create)

    // vqhip_glvq_batch_gradient: the matched search's four result arrays for one chunk of rows, the accumulation scratch
    // (partials), the gradient itself and the cost / count pair; kept between calls, grown when a call needs more
    uint32_t *gJ[2] = {nullptr, nullptr};
    float *gD[2] = {nullptr, nullptr};
    uint64_t g_rows = 0;
    float *gScratch = nullptr, *gOut = nullptr;
    size_t gScratch_cap = 0, gOut_cap = 0;
    double *gCost = nullptr;             // {cost (double), used (u64)} on the device

    EngineModel *engine = nullptr;
    // A global Omega no single-pass kernel holds (more than 128 rows over rows above 256 values; more than 64 rows
    // over 129..256 values; more than 256 rows): the samples are projected into scratch (k_project_long / k_project)
    // and searched there by a Euclidean engine over P = W Omega^T; the rooted distances are squared afterwards
    // (helpers.rs:44).
    EngineModel *engine2 = nullptr;
    std::vector<float> hP;     // K x M: W . Omega^T (global Omega only)
    bool engine_tried = false, engine2_tried = false;
    std::string engine_why;
    ProjectModel *proj_omega = nullptr, *proj_lambda = nullptr;   // matrix-core projections (global Omega)
    LocalProjectModel *proj_local = nullptr;                      // LGMLVQ::project (local Omegas)
    LongProjectModel *proj_long = nullptr;                        // global Omega over rows above 256 values
    // ... with more than 128 rows: project (Lambda) = Omega^T (Omega x) in two passes, the second with Omega^T
    // (D rows of M values) as the matrix, through whichever projection kernel holds rows of M values
    ProjectModel *proj_back = nullptr;
    LongProjectModel *proj_back_long = nullptr;
    bool proj_tried = false;

    Staging st[2];
    size_t st_chunk = 0;       // samples per staging buffer
    uint32_t st_D = 0;

    // device buffers of the f64 entry points (one chunk of rows and its results), kept between calls
    double *f64_dX = nullptr, *f64_dD[2] = {nullptr, nullptr};
    uint32_t *f64_dY = nullptr, *f64_dI[2] = {nullptr, nullptr};
    uint64_t f64_rows = 0;
    uint32_t f64_D = 0;

    int fail(int code, const std::string &msg)
    {
        err = msg;
        return code;
    }
    int fail_hip(hipError_t e, const char *what)
    {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? VQHIP_ERR_NOMEM : VQHIP_ERR_HIP;
    }
};

// Nothing may unwind across the C ABI (include/vqhip.h): every extern "C" body is a function-try-block that ends in
// VQHIP_CATCH.  bad_alloc -> VQHIP_ERR_NOMEM, anything else (std::system_error from a thread that would not start,
// length_error, ...) -> VQHIP_ERR_INVALID, the text in vqhip_last_error.
static int on_exception(vqhip_ctx *c) noexcept;
#define VQHIP_CATCH(ctx) \
    catch (...) { return on_exception(ctx); }

#define HIPCHK(ctx, call)                                   \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) return (ctx)->fail_hip(e_, #call); \
    } while (0)

static int on_exception(vqhip_ctx *c) noexcept
{
    int code = VQHIP_ERR_INVALID;
    char text[256] = "unknown C++ exception";
    try {
        throw;
    } catch (const std::bad_alloc &) {
        code = VQHIP_ERR_NOMEM;
        std::snprintf(text, sizeof text, "out of host memory");
    } catch (const std::exception &e) {
        std::snprintf(text, sizeof text, "%s", e.what());
    } catch (...) {
    }
    try {
        (c ? c->err : g_create_error) = std::string("internal error: ") + text;
    } catch (...) {                                       // not even the message fits: the code still tells
    }
    return code;
}

namespace {

void drop_engine(vqhip_ctx *c)
{
    engine_free(c->engine);
    engine_free(c->engine2);
    c->engine = c->engine2 = nullptr;
    c->engine_tried = c->engine2_tried = false;
    c->engine_why.clear();
    project_free(c->proj_omega);
    project_free(c->proj_lambda);
    local_project_free(c->proj_local);
    long_project_free(c->proj_long);
    long_project_free(c->proj_back_long);
    project_free(c->proj_back);
    c->proj_long = c->proj_back_long = nullptr;
    c->proj_back = nullptr;
    c->proj_omega = c->proj_lambda = nullptr;
    c->proj_local = nullptr;
    c->proj_tried = false;
}

void free_staging(vqhip_ctx *c)
{
    for (Staging &s : c->st) {
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        (void)hipHostFree(s.hX); (void)hipFree(s.dX);
        (void)hipHostFree(s.hY); (void)hipFree(s.dY);
        for (int j = 0; j < 2; ++j) {
            (void)hipHostFree(s.hI[j]); (void)hipFree(s.dI[j]);
            (void)hipHostFree(s.hD[j]); (void)hipFree(s.dD[j]);
        }
        (void)hipHostFree(s.hP); (void)hipFree(s.dP);
        hipStream_t keep = s.stream;
        s = Staging();
        s.stream = keep;
    }
    c->st_chunk = 0;
    c->st_D = 0;
}

void free_omega(vqhip_ctx *c)
{
    (void)hipFree(c->dOm); c->dOm = nullptr;
    (void)hipFree(c->dP);  c->dP = nullptr;
    c->hP.clear();
    (void)hipFree(c->dOmT); c->dOmT = nullptr;
    c->hOm.clear();
    c->M = 0;
    c->omega_kind = OMEGA_NONE;
}

template <class T>
int upload(vqhip_ctx *c, T **dst, const T *src, size_t count)
{
    (void)hipFree(*dst);
    *dst = nullptr;
    HIPCHK(c, hipMalloc((void **)dst, count * sizeof(T)));
    HIPCHK(c, hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return VQHIP_OK;
}

int prepare_project(vqhip_ctx *c);

// vqhip_set_option resolved: an explicit value wins; VQHIP_VALUE_DEFAULT is the bf16 matrix cores unless the environment
// says otherwise (VQHIP_NO_BF: nothing there; VQHIP_NO_BF_PROJ: the ranking only) -- the environment is a default, read
// when a model's images are built, never an override.
BuildOptions build_options(const vqhip_ctx *c)
{
    BuildOptions o;
    const bool env_no_bf = getenv("VQHIP_NO_BF") != nullptr, env_no_pj = env_no_bf || getenv("VQHIP_NO_BF_PROJ") != nullptr;
    o.bf_rank = c->opt_ranking == VQHIP_VALUE_DEFAULT ? !env_no_bf : c->opt_ranking == VQHIP_VALUE_BF16;
    o.bf_proj = c->opt_projection == VQHIP_VALUE_DEFAULT ? !env_no_pj : c->opt_projection == VQHIP_VALUE_BF16;
    return o;
}

// Makes sure the MFMA engine's tile images exist when the policy wants them.
int prepare_engine(vqhip_ctx *c)
{
    if (c->policy == VQHIP_KERNEL_SCALAR) return VQHIP_OK;
    if (!c->engine_tried) {
        c->engine_tried = true;
        hipError_t herr = hipSuccess;
        c->engine = engine_build(c->hW, c->K, c->D, c->hOm, c->omega_kind, c->M, c->hcls, build_options(c), c->engine_why, herr);
        if (herr != hipSuccess) return c->fail_hip(herr, "engine_build");
    }
    if (!c->engine && !c->engine2_tried && c->omega_kind == OMEGA_GLOBAL && !getenv("VQHIP_NO_TWO_PASS")) {
        c->engine2_tried = true;
        int rc = prepare_project(c);
        if (rc != VQHIP_OK) return rc;
        if (c->proj_long || c->proj_omega) {
            hipError_t herr = hipSuccess;
            std::string why2;
            c->engine2 = engine_build(c->hP, c->K, c->M, std::vector<float>(), OMEGA_NONE, 0, c->hcls, build_options(c), why2, herr);
            if (herr != hipSuccess) return c->fail_hip(herr, "engine_build (projected space)");
            if (!c->engine2) c->engine_why += "; projected space: " + why2;
        }
    }
    if (!c->engine && !c->engine2 && c->policy == VQHIP_KERNEL_MFMA)
        return c->fail(VQHIP_ERR_UNSUPPORTED, "MFMA engine not eligible for this model: " + c->engine_why);
    return VQHIP_OK;
}

int scratch_acquire(vqhip_ctx *c, hipStream_t s)
{
    if (!c->scratch_done) {
        HIPCHK(c, hipEventCreateWithFlags(&c->scratch_done, hipEventDisableTiming));
        return VQHIP_OK;
    }
    HIPCHK(c, hipStreamWaitEvent(s, c->scratch_done, 0));
    return VQHIP_OK;
}

int scratch_release(vqhip_ctx *c, hipStream_t s)
{
    HIPCHK(c, hipEventRecord(c->scratch_done, s));
    return VQHIP_OK;
}

// The scratch for projected samples holds at least `need` floats (grown after the stream has drained).
int ensure_yproj(vqhip_ctx *c, size_t need, hipStream_t s)
{
    if (c->dYproj_cap >= need) return VQHIP_OK;
    HIPCHK(c, hipStreamSynchronize(s));
    (void)hipFree(c->dYproj);
    c->dYproj = nullptr;
    c->dYproj_cap = 0;
    HIPCHK(c, hipMalloc((void **)&c->dYproj, need * sizeof(float)));
    c->dYproj_cap = need;
    return VQHIP_OK;
}

// All kernels for one device-resident batch, enqueued on `s`.
int run_device(vqhip_ctx *c, const float *dX, uint64_t N, const uint32_t *dy, Outputs out, hipStream_t s)
{
    if (c->K == 0) return c->fail(VQHIP_ERR_STATE, "no prototypes set (vqhip_set_prototypes)");
    if (dy && c->hcls.empty()) return c->fail(VQHIP_ERR_STATE, "matched search needs vqhip_set_labels");
    if (N == 0) return VQHIP_OK;
    // the kernels fetch rows through 4-byte aligned buffer descriptors and store 4-byte results
    if (((uintptr_t)dX | (uintptr_t)dy | (uintptr_t)out.idx | (uintptr_t)out.dist | (uintptr_t)out.jneg | (uintptr_t)out.dneg) & 3)
        return c->fail(VQHIP_ERR_INVALID, "device pointers must be 4-byte aligned");
    int rc = prepare_engine(c);
    if (rc != VQHIP_OK) return rc;

    if (c->engine && c->policy != VQHIP_KERNEL_SCALAR) {      // (an engine built before the policy was set to scalar stays unused)
        c->last_kernel = engine_name(c->engine);
        c->last_ranking = engine_ranking(c->engine);
        c->last_projection = engine_projection(c->engine);
        HIPCHK(c, engine_assign(c->engine, dX, N, dy, out, s));
        return VQHIP_OK;
    }

    if (c->engine2 && (c->proj_long || c->proj_omega) && c->policy != VQHIP_KERNEL_SCALAR && ((uintptr_t)dX & 3) == 0) {
        const uint64_t chunk = std::min<uint64_t>(N, std::max<uint64_t>(1, (1024ull << 20) / (c->M * 4ull)));
        if ((rc = ensure_yproj(c, chunk * c->M, s)) != VQHIP_OK) return rc;
        if ((rc = scratch_acquire(c, s)) != VQHIP_OK) return rc;
        for (uint64_t off = 0; off < N; off += chunk) {
            const uint64_t n = std::min(chunk, N - off);
            Outputs o = out;
            if (o.idx) o.idx += off;
            if (o.dist) o.dist += off;
            if (o.jneg) o.jneg += off;
            if (o.dneg) o.dneg += off;
            if (c->proj_long)
                HIPCHK(c, long_project_run(c->proj_long, dX + off * c->D, n, false, c->dYproj, s));
            else
                HIPCHK(c, project_run(c->proj_omega, dX + off * c->D, n, c->dYproj, c->M, s));
            HIPCHK(c, engine_assign(c->engine2, c->dYproj, n, dy ? dy + off : nullptr, o, s));
            HIPCHK(c, launch_square(o.dist, n, s));
            if (dy) HIPCHK(c, launch_square(o.dneg, n, s));
        }
        c->last_kernel = "mfma_omega_two_pass";
        c->last_ranking = engine_ranking(c->engine2);
        c->last_projection = !c->proj_long && project_on_bf16(c->proj_omega) ? "bf16x6" : "f32";
        return scratch_release(c, s);
    }

    c->last_kernel = "scalar";
    c->last_ranking = "none";
    c->last_projection = c->omega_kind == OMEGA_NONE ? "none" : "f32";
    if (c->omega_kind == OMEGA_NONE) {
        HIPCHK(c, launch_assign_scalar(dX, N, c->D, c->dW, c->K, c->dcls, dy, /*rooted=*/true, out, s));
    } else if (c->omega_kind == OMEGA_GLOBAL) {
        // Project a chunk of samples into the M-dim space, then search there against W.Omega^T.
        const uint64_t chunk = std::min<uint64_t>(N, std::max<uint64_t>(1, (64ull << 20) / (c->M * 4ull)));
        if ((rc = ensure_yproj(c, chunk * c->M, s)) != VQHIP_OK) return rc;
        if ((rc = scratch_acquire(c, s)) != VQHIP_OK) return rc;
        for (uint64_t off = 0; off < N; off += chunk) {
            const uint64_t n = std::min(chunk, N - off);
            Outputs o = out;
            if (o.idx) o.idx += off;
            if (o.dist) o.dist += off;
            if (o.jneg) o.jneg += off;
            if (o.dneg) o.dneg += off;
            HIPCHK(c, launch_project_scalar(dX + off * c->D, n, c->D, c->dOm, c->M, c->dYproj, s));
            HIPCHK(c, launch_assign_scalar(c->dYproj, n, c->M, c->dP, c->K, c->dcls, dy ? dy + off : nullptr,
                                           /*rooted=*/false, o, s));
        }
        if ((rc = scratch_release(c, s)) != VQHIP_OK) return rc;
    } else {
        HIPCHK(c, launch_assign_local_scalar(dX, N, c->D, c->dW, c->K, c->dOm, c->M, c->dcls, dy, out, s));
    }
    return VQHIP_OK;
}

int ensure_staging(vqhip_ctx *c, size_t chunk, bool matched)
{
    if (c->st_chunk >= chunk && c->st_D == c->D && (!matched || c->st[0].hY)) return VQHIP_OK;
    free_staging(c);
    for (Staging &s : c->st) {
        if (!s.stream) HIPCHK(c, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        HIPCHK(c, hipHostMalloc((void **)&s.hX, chunk * c->D * sizeof(float), hipHostMallocDefault));
        HIPCHK(c, hipMalloc((void **)&s.dX, chunk * c->D * sizeof(float)));
        HIPCHK(c, hipHostMalloc((void **)&s.hY, chunk * sizeof(uint32_t), hipHostMallocDefault));
        HIPCHK(c, hipMalloc((void **)&s.dY, chunk * sizeof(uint32_t)));
        for (int j = 0; j < 2; ++j) {
            HIPCHK(c, hipHostMalloc((void **)&s.hI[j], chunk * sizeof(uint32_t), hipHostMallocDefault));
            HIPCHK(c, hipMalloc((void **)&s.dI[j], chunk * sizeof(uint32_t)));
            HIPCHK(c, hipHostMalloc((void **)&s.hD[j], chunk * sizeof(float), hipHostMallocDefault));
            HIPCHK(c, hipMalloc((void **)&s.dD[j], chunk * sizeof(float)));
        }
    }
    c->st_chunk = chunk;
    c->st_D = c->D;
    return VQHIP_OK;
}

// Host-pointer path: pinned, double-buffered chunks.  While the GPU works on chunk c (stream
// c%2) the CPU packs chunk c+1 into the other pinned buffer and unpacks the results of c-1.
// Splits [0, n) over the host's cores (at most 16 workers: the GPU box gives one GPU 16 of them).
void parallel_rows(uint64_t n, const std::function<void(uint64_t, uint64_t)> &fn)
{
    unsigned workers = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (n < 4096 || workers == 1) {
        fn(0, n);
        return;
    }
    const uint64_t per = (n + workers - 1) / workers;
    std::vector<std::thread> pool;
    pool.reserve(workers);                                // (may throw before any thread exists)
    uint64_t inline_from = n;                             // slices a thread could not be started for run here
    for (unsigned w = 1; w < workers && w * per < n; ++w) {
        try {
            pool.emplace_back(fn, w * per, std::min<uint64_t>(n, (w + 1) * per));
        } catch (...) {                                   // std::system_error: out of threads
            inline_from = w * per;
            break;
        }
    }
    fn(0, std::min<uint64_t>(n, per));                    // the slices' bodies are memcpy / narrowing loops: they do not throw
    if (inline_from < n) fn(inline_from, n);
    for (std::thread &t : pool) t.join();
}

// pack(dst, first_row, rows): writes rows [first_row, first_row + rows) as contiguous f32 into pinned dst.
using Packer = std::function<void(float *, uint64_t, uint64_t)>;

int run_host(vqhip_ctx *c, const Packer &pack, const uint32_t *y, uint64_t N, uint32_t *o0, float *d0, uint32_t *o1,
             float *d1)
{
    if (c->K == 0) return c->fail(VQHIP_ERR_STATE, "no prototypes set (vqhip_set_prototypes)");
    if (y && c->hcls.empty()) return c->fail(VQHIP_ERR_STATE, "matched search needs vqhip_set_labels");
    if (N == 0) return VQHIP_OK;
    const size_t row = (size_t)c->D * sizeof(float);
    size_t chunk = std::max<size_t>(1024, (128ull << 20) / row);     // <= 128 MiB of samples per buffer
    chunk = (size_t)std::min<uint64_t>(chunk, N);
    int rc = ensure_staging(c, chunk, y != nullptr);
    if (rc != VQHIP_OK) return rc;
    chunk = c->st_chunk;

    const uint64_t nchunks = (N + chunk - 1) / chunk;
    auto finish = [&](uint64_t ci) -> int {
        Staging &s = c->st[ci & 1];
        HIPCHK(c, hipStreamSynchronize(s.stream));
        const uint64_t off = ci * chunk, n = std::min<uint64_t>(chunk, N - off);
        if (o0) std::memcpy(o0 + off, s.hI[0], n * sizeof(uint32_t));
        if (d0) std::memcpy(d0 + off, s.hD[0], n * sizeof(float));
        if (o1) std::memcpy(o1 + off, s.hI[1], n * sizeof(uint32_t));
        if (d1) std::memcpy(d1 + off, s.hD[1], n * sizeof(float));
        return VQHIP_OK;
    };
    for (uint64_t ci = 0; ci < nchunks; ++ci) {
        Staging &s = c->st[ci & 1];
        if (ci >= 2 && (rc = finish(ci - 2)) != VQHIP_OK) return rc;
        const uint64_t off = ci * chunk, n = std::min<uint64_t>(chunk, N - off);
        pack(s.hX, off, n);
        HIPCHK(c, hipMemcpyAsync(s.dX, s.hX, n * row, hipMemcpyHostToDevice, s.stream));
        if (y) {
            std::memcpy(s.hY, y + off, n * sizeof(uint32_t));
            HIPCHK(c, hipMemcpyAsync(s.dY, s.hY, n * sizeof(uint32_t), hipMemcpyHostToDevice, s.stream));
        }
        Outputs out{s.dI[0], s.dD[0], y ? s.dI[1] : nullptr, y ? s.dD[1] : nullptr};
        rc = run_device(c, s.dX, n, y ? s.dY : nullptr, out, s.stream);
        if (rc != VQHIP_OK) return rc;
        HIPCHK(c, hipMemcpyAsync(s.hI[0], s.dI[0], n * sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream));
        HIPCHK(c, hipMemcpyAsync(s.hD[0], s.dD[0], n * sizeof(float), hipMemcpyDeviceToHost, s.stream));
        if (y) {
            HIPCHK(c, hipMemcpyAsync(s.hI[1], s.dI[1], n * sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream));
            HIPCHK(c, hipMemcpyAsync(s.hD[1], s.dD[1], n * sizeof(float), hipMemcpyDeviceToHost, s.stream));
        }
    }
    for (uint64_t ci = nchunks >= 2 ? nchunks - 2 : 0; ci < nchunks; ++ci)
        if ((rc = finish(ci)) != VQHIP_OK) return rc;
    return VQHIP_OK;
}

void free_gradient(vqhip_ctx *c)
{
    for (int j = 0; j < 2; ++j) {
        (void)hipFree(c->gJ[j]); (void)hipFree(c->gD[j]);
        c->gJ[j] = nullptr; c->gD[j] = nullptr;
    }
    (void)hipFree(c->gScratch); (void)hipFree(c->gOut); (void)hipFree(c->gCost);
    c->gScratch = c->gOut = nullptr;
    c->gCost = nullptr;
    c->g_rows = 0;
    c->gScratch_cap = c->gOut_cap = 0;
}

// Batched GLVQ gradient of device-resident samples, enqueued on `s` (include/vqhip.h): matched search per chunk of rows,
// then the scatter-add pass of kernels_glvq.hip.  ddW: K x D on the device.  dcost: {double cost, u64 used} on the device
// (added to), or null.
constexpr uint64_t GRAD_CHUNK_ROWS = 8ull << 20;
int gradient_device(vqhip_ctx *c, const float *dX, const uint32_t *dy, uint64_t N, float *ddW, bool accumulate, double *dcost, hipStream_t s)
{
    if (c->K == 0) return c->fail(VQHIP_ERR_STATE, "no prototypes set (vqhip_set_prototypes)");
    if (c->hcls.empty()) return c->fail(VQHIP_ERR_STATE, "glvq_batch_gradient needs vqhip_set_labels");
    if (c->omega_kind != OMEGA_NONE)
        return c->fail(VQHIP_ERR_UNSUPPORTED, "glvq_batch_gradient serves Euclidean models (GLVQ, src/glvq/glvq.rs:186-205); clear Omega first");
    const size_t kd = (size_t)c->K * c->D;
    if (N == 0) {
        if (!accumulate) HIPCHK(c, hipMemsetAsync(ddW, 0, kd * sizeof(float), s));
        return VQHIP_OK;
    }
    const uint64_t chunk = std::min<uint64_t>(N, GRAD_CHUNK_ROWS);
    uint32_t nparts = 0;
    const size_t need = glvq_gradient_scratch_floats(c->K, c->D, &nparts);
    if (c->g_rows < chunk || c->gScratch_cap < need) {
        HIPCHK(c, hipStreamSynchronize(s));
        for (int j = 0; j < 2; ++j) {
            (void)hipFree(c->gJ[j]); (void)hipFree(c->gD[j]);
            c->gJ[j] = nullptr; c->gD[j] = nullptr;
        }
        (void)hipFree(c->gScratch);
        c->gScratch = nullptr;
        c->g_rows = 0;
        c->gScratch_cap = 0;
        for (int j = 0; j < 2; ++j) {
            HIPCHK(c, hipMalloc((void **)&c->gJ[j], chunk * sizeof(uint32_t)));
            HIPCHK(c, hipMalloc((void **)&c->gD[j], chunk * sizeof(float)));
        }
        HIPCHK(c, hipMalloc((void **)&c->gScratch, need * sizeof(float)));
        c->g_rows = chunk;
        c->gScratch_cap = need;
    }
    int rc = scratch_acquire(c, s);
    if (rc != VQHIP_OK) return rc;
    for (uint64_t off = 0; off < N; off += chunk) {
        const uint64_t n = std::min(chunk, N - off);
        rc = run_device(c, dX + off * c->D, n, dy + off, Outputs{c->gJ[0], c->gD[0], c->gJ[1], c->gD[1]}, s);
        if (rc != VQHIP_OK) return rc;
        HIPCHK(c, launch_glvq_gradient(dX + off * c->D, n, c->D, c->K, c->dW, c->gJ[0], c->gD[0], c->gJ[1], c->gD[1], c->gScratch, ddW,
                                       accumulate || off != 0, dcost, dcost ? (unsigned long long *)(dcost + 1) : nullptr, s));
    }
    return scratch_release(c, s);
}

int ensure_gradient_out(vqhip_ctx *c)
{
    const size_t kd = (size_t)c->K * c->D;
    if (c->gOut_cap < kd) {
        (void)hipFree(c->gOut);
        c->gOut = nullptr;
        c->gOut_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->gOut, kd * sizeof(float)));
        c->gOut_cap = kd;
    }
    if (!c->gCost) HIPCHK(c, hipMalloc((void **)&c->gCost, 16));
    return VQHIP_OK;
}

uint32_t project_width(const vqhip_ctx *c, int which) { return which == VQHIP_PROJECT_OMEGA ? c->M : c->D; }

int check_project(vqhip_ctx *c, int which)
{
    if (c->K == 0) return c->fail(VQHIP_ERR_STATE, "no prototypes set (vqhip_set_prototypes)");
    if (c->omega_kind == OMEGA_NONE) return c->fail(VQHIP_ERR_STATE, "project needs vqhip_set_omega or vqhip_set_local_omegas");
    if (which != VQHIP_PROJECT_OMEGA && which != VQHIP_PROJECT_LAMBDA) return c->fail(VQHIP_ERR_INVALID, "unknown projection");
    if (c->omega_kind == OMEGA_LOCAL && which == VQHIP_PROJECT_OMEGA)
        return c->fail(VQHIP_ERR_UNSUPPORTED, "local Omegas project with Lambda_j only (src/lgmlvq/lgmlvq.rs:372-393)");
    return VQHIP_OK;
}

// Images of Omega and of Lambda = Omega^T Omega (f64, rounded once: the reference builds Lambda first and
// multiplies every sample by it, gmlvq.rs:357-366) for the matrix-core projection kernel.
int prepare_project(vqhip_ctx *c)
{
    if (c->proj_tried || c->policy == VQHIP_KERNEL_SCALAR) return VQHIP_OK;
    c->proj_tried = true;
    hipError_t herr = hipSuccess;
    if (c->omega_kind == OMEGA_LOCAL) {                    // Lambda_j of every prototype (lgmlvq.rs:383-384)
        c->proj_local = local_project_build(c->hOm, c->K, c->M, c->D, herr);
        if (herr != hipSuccess) return c->fail_hip(herr, "local_project_build");
        return VQHIP_OK;
    }
    if (c->D > 256) {
        c->proj_long = long_project_build(c->hOm, c->M, c->D, herr);
        if (herr != hipSuccess) return c->fail_hip(herr, "long_project_build");
        if (c->proj_long && !long_project_has_lambda(c->proj_long)) {
            std::vector<float> OmT((size_t)c->D * c->M);
            for (uint32_t m = 0; m < c->M; ++m)
                for (uint32_t d = 0; d < c->D; ++d) OmT[(size_t)d * c->M + m] = c->hOm[(size_t)m * c->D + d];
            if (c->M <= 256)
                c->proj_back = project_build(OmT, c->D, c->M, build_options(c), herr);
            else
                c->proj_back_long = long_project_build(OmT, c->D, c->M, herr);
            if (herr != hipSuccess) return c->fail_hip(herr, "project_build (Omega^T)");
        }
        return VQHIP_OK;
    }
    c->proj_omega = project_build(c->hOm, c->M, c->D, build_options(c), herr);
    if (herr != hipSuccess) return c->fail_hip(herr, "project_build");
    if (!c->proj_omega) return VQHIP_OK;
    const uint32_t D = c->D, M = c->M;
    std::vector<float> lam((size_t)D * D);
    for (uint32_t i = 0; i < D; ++i)
        for (uint32_t j = i; j < D; ++j) {
            double t = 0.0;
            for (uint32_t m = 0; m < M; ++m) t += (double)c->hOm[(size_t)m * D + i] * (double)c->hOm[(size_t)m * D + j];
            lam[(size_t)i * D + j] = lam[(size_t)j * D + i] = (float)t;
        }
    c->proj_lambda = project_build(lam, D, D, build_options(c), herr);
    if (herr != hipSuccess) return c->fail_hip(herr, "project_build");
    return VQHIP_OK;
}

// The projection scratch holds at least `need` 4-byte words (grown after the stream has drained: kernels in
// flight may still use the old buffer).
int ensure_proj_scratch(vqhip_ctx *c, size_t need, hipStream_t s)
{
    if (c->dProj_cap >= need) return VQHIP_OK;
    HIPCHK(c, hipStreamSynchronize(s));
    (void)hipFree(c->dProj);
    c->dProj = nullptr;
    c->dProj_cap = 0;
    HIPCHK(c, hipMalloc((void **)&c->dProj, need * sizeof(float)));
    c->dProj_cap = need;
    return VQHIP_OK;
}

// One device-resident batch of project / project_omega, enqueued on `s`.
int project_device(vqhip_ctx *c, const float *dX, uint64_t N, int which, float *dY, hipStream_t s)
{
    int rc = check_project(c, which);
    if (rc != VQHIP_OK || N == 0) return rc;
    if ((rc = prepare_project(c)) != VQHIP_OK) return rc;
    if (c->omega_kind == OMEGA_GLOBAL && c->policy != VQHIP_KERNEL_SCALAR) {
        const ProjectModel *pm = which == VQHIP_PROJECT_OMEGA ? c->proj_omega : c->proj_lambda;
        const uint32_t ldy = project_width(c, which);
        if (pm && ((uintptr_t)dY & 3) == 0 && ((uintptr_t)dX & 3) == 0) {
            c->last_kernel = "mfma_project";
            HIPCHK(c, project_run(pm, dX, N, dY, ldy, s));
            return VQHIP_OK;
        }
        if (c->proj_long && (which == VQHIP_PROJECT_OMEGA || long_project_has_lambda(c->proj_long)) && ((uintptr_t)dY & 3) == 0 &&
            ((uintptr_t)dX & 3) == 0) {
            c->last_kernel = "mfma_project_long";
            HIPCHK(c, long_project_run(c->proj_long, dX, N, which == VQHIP_PROJECT_LAMBDA, dY, s));
            return VQHIP_OK;
        }
        if (c->proj_long && (c->proj_back || c->proj_back_long) && ((uintptr_t)dY & 3) == 0 && ((uintptr_t)dX & 3) == 0) {
            // Lambda x = Omega^T (Omega x), chunks of rows through an N x M intermediate
            const uint64_t chunk = std::min<uint64_t>(N, std::max<uint64_t>(1, (1024ull << 20) / (c->M * 4ull)));
            if ((rc = ensure_proj_scratch(c, (size_t)chunk * c->M, s)) != VQHIP_OK) return rc;
            if ((rc = scratch_acquire(c, s)) != VQHIP_OK) return rc;
            for (uint64_t off = 0; off < N; off += chunk) {
                const uint64_t n = std::min(chunk, N - off);
                HIPCHK(c, long_project_run(c->proj_long, dX + off * c->D, n, false, c->dProj, s));
                if (c->proj_back)
                    HIPCHK(c, project_run(c->proj_back, c->dProj, n, dY + off * c->D, c->D, s));
                else
                    HIPCHK(c, long_project_run(c->proj_back_long, c->dProj, n, false, dY + off * c->D, s));
            }
            c->last_kernel = "mfma_project_two_pass";
            return scratch_release(c, s);
        }
    }
    if (c->omega_kind == OMEGA_LOCAL && c->policy != VQHIP_KERNEL_SCALAR && c->proj_local && ((uintptr_t)dY & 3) == 0 &&
        ((uintptr_t)dX & 3) == 0) {
        // winners of a chunk of rows, then the chunk grouped by winner and projected on the matrix cores
        const LocalProjectModel *pm = c->proj_local;
        const uint64_t chunk = std::min<uint64_t>(N, std::min<uint64_t>(local_project_max_rows(pm), 4ull << 20));
        const size_t need = (size_t)chunk + local_project_scratch_words(pm, chunk);
        if ((rc = ensure_proj_scratch(c, need, s)) != VQHIP_OK) return rc;
        uint32_t *dwin = (uint32_t *)c->dProj;
        if ((rc = scratch_acquire(c, s)) != VQHIP_OK) return rc;
        for (uint64_t off = 0; off < N; off += chunk) {
            const uint64_t n = std::min(chunk, N - off);
            rc = run_device(c, dX + off * c->D, n, nullptr, Outputs{dwin, nullptr, nullptr, nullptr}, s);
            if (rc != VQHIP_OK) return rc;
            HIPCHK(c, local_project_run(pm, dX + off * c->D, n, dwin, dwin + chunk, dY + off * c->D, s));
        }
        c->last_kernel = "mfma_project_local";
        return scratch_release(c, s);
    }
    c->last_kernel = "scalar_project";
    if (c->omega_kind == OMEGA_GLOBAL && which == VQHIP_PROJECT_OMEGA) {          // liramlvq.rs:429-446
        HIPCHK(c, launch_project_scalar(dX, N, c->D, c->dOm, c->M, dY, s));
        return VQHIP_OK;
    }
    // Lambda x = Omega^T (Omega x): chunks of rows through an N x M intermediate (+ the winners, local case)
    const uint64_t chunk = std::min<uint64_t>(N, std::max<uint64_t>(1, (64ull << 20) / ((c->M + 1) * 4ull)));
    const size_t need = (size_t)chunk * (c->M + 1);
    if ((rc = ensure_proj_scratch(c, need, s)) != VQHIP_OK) return rc;
    float *dT = c->dProj;
    uint32_t *dwin = (uint32_t *)(c->dProj + (size_t)chunk * c->M);
    if ((rc = scratch_acquire(c, s)) != VQHIP_OK) return rc;
    for (uint64_t off = 0; off < N; off += chunk) {
        const uint64_t n = std::min(chunk, N - off);
        const float *x = dX + off * c->D;
        float *y = dY + off * c->D;
        if (c->omega_kind == OMEGA_GLOBAL) {                                        // gmlvq.rs:354-372
            HIPCHK(c, launch_project_scalar(x, n, c->D, c->dOm, c->M, dT, s));
            HIPCHK(c, launch_project_scalar(dT, n, c->M, c->dOmT, c->D, y, s));
        } else {                                                                    // lgmlvq.rs:372-393
            rc = run_device(c, x, n, nullptr, Outputs{dwin, nullptr, nullptr, nullptr}, s);
            if (rc != VQHIP_OK) return rc;
            HIPCHK(c, launch_project_local(x, n, c->D, c->dOm, c->M, dwin, dT, y, s));
        }
    }
    c->last_kernel = "scalar_project";                    // (the winner search in between has set its own name)
    return scratch_release(c, s);
}

Packer contiguous_f32(const vqhip_ctx *c, const float *X)
{
    const uint32_t D = c->D;
    // one core copies ~10 GB/s, the host link takes 50+: split the copy into the pinned buffer over the cores
    return [X, D](float *dst, uint64_t first, uint64_t n) {
        parallel_rows(n, [=](uint64_t a, uint64_t b) { std::memcpy(dst + a * D, X + (first + a) * D, (b - a) * D * sizeof(float)); });
    };
}

void free_f64(vqhip_ctx *c)
{
    (void)hipFree(c->f64_dX); (void)hipFree(c->f64_dY);
    for (int j = 0; j < 2; ++j) { (void)hipFree(c->f64_dD[j]); (void)hipFree(c->f64_dI[j]); }
    c->f64_dX = nullptr; c->f64_dY = nullptr;
    c->f64_dD[0] = c->f64_dD[1] = nullptr;
    c->f64_dI[0] = c->f64_dI[1] = nullptr;
    c->f64_rows = 0;
    c->f64_D = 0;
}

// Vec<Array1<f64>> -> pinned contiguous f32, gathered and narrowed in one pass by the host cores
Packer rows_f64(const double *const *rows, uint32_t D)
{
    return [rows, D](float *dst, uint64_t first, uint64_t n) {
        parallel_rows(n, [=](uint64_t a, uint64_t b) {
            for (uint64_t i = a; i < b; ++i) {
                const double *src = rows[first + i];
                float *out = dst + i * D;
                for (uint32_t d = 0; d < D; ++d) out[d] = (float)src[d];
            }
        });
    };
}

// f64 Euclidean path (row f4): plain chunked copies, the scalar kernel instantiated for double.  The chunk buffers
// stay with the context between calls.
int run_host_f64(vqhip_ctx *c, const double *X, const uint32_t *y, uint64_t N, uint32_t *o0, double *d0, uint32_t *o1,
                 double *d1)
{
    if (!c->dW64) return c->fail(VQHIP_ERR_STATE, "no f64 prototypes set (vqhip_set_prototypes_f64)");
    if (y && c->hcls.empty()) return c->fail(VQHIP_ERR_STATE, "matched search needs vqhip_set_labels");
    if (N == 0) return VQHIP_OK;
    const uint32_t D = c->D;
    const uint64_t chunk = std::min<uint64_t>(N, std::max<uint64_t>(1024, (64ull << 20) / (D * sizeof(double))));
    if (c->f64_rows < chunk || c->f64_D != D) {
        free_f64(c);
        HIPCHK(c, hipMalloc((void **)&c->f64_dX, chunk * D * sizeof(double)));
        HIPCHK(c, hipMalloc((void **)&c->f64_dY, chunk * sizeof(uint32_t)));
        for (int j = 0; j < 2; ++j) {
            HIPCHK(c, hipMalloc((void **)&c->f64_dD[j], chunk * sizeof(double)));
            HIPCHK(c, hipMalloc((void **)&c->f64_dI[j], chunk * sizeof(uint32_t)));
        }
        c->f64_rows = chunk;
        c->f64_D = D;
    }
    double *dX = c->f64_dX, **dD = c->f64_dD;
    uint32_t *dY = c->f64_dY, **dI = c->f64_dI;
    for (uint64_t off = 0; off < N; off += chunk) {
        const uint64_t n = std::min(chunk, N - off);
        HIPCHK(c, hipMemcpy(dX, X + off * D, n * D * sizeof(double), hipMemcpyHostToDevice));
        if (y) HIPCHK(c, hipMemcpy(dY, y + off, n * sizeof(uint32_t), hipMemcpyHostToDevice));
        Outputs64 out{dI[0], dD[0], y ? dI[1] : nullptr, y ? dD[1] : nullptr};
        HIPCHK(c, launch_assign_scalar_f64(dX, n, D, c->dW64, c->K, c->dcls, y ? dY : nullptr, out, nullptr));
        if (o0) HIPCHK(c, hipMemcpy(o0 + off, dI[0], n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (d0) HIPCHK(c, hipMemcpy(d0 + off, dD[0], n * sizeof(double), hipMemcpyDeviceToHost));
        if (y && o1) HIPCHK(c, hipMemcpy(o1 + off, dI[1], n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (y && d1) HIPCHK(c, hipMemcpy(d1 + off, dD[1], n * sizeof(double), hipMemcpyDeviceToHost));
    }
    c->last_kernel = "scalar_f64";
    return VQHIP_OK;
}

// project / project_omega from host pointers: the pinned, double-buffered pipeline of run_host (pack | H2D | kernel |
// D2H | unpack on two streams), the projected rows in place of the winners.
// where the projected rows of a chunk go: (pinned source, first row, rows, width)
using Unpacker = std::function<void(const float *, uint64_t, uint64_t, uint32_t)>;

int run_host_project(vqhip_ctx *c, const Packer &pack, uint64_t N, int which, const Unpacker &unpack)
{
    int rc = check_project(c, which);
    if (rc != VQHIP_OK || N == 0) return rc;
    const uint32_t D = c->D, Wd = project_width(c, which);
    const size_t row = (size_t)D * sizeof(float);
    size_t chunk = std::max<size_t>(1024, (128ull << 20) / (std::max(D, Wd) * sizeof(float)));
    chunk = (size_t)std::min<uint64_t>(chunk, N);
    if ((rc = ensure_staging(c, chunk, false)) != VQHIP_OK) return rc;
    chunk = std::min<size_t>(c->st_chunk, (size_t)std::max<uint64_t>(1, std::min<uint64_t>(N, (128ull << 20) / (std::max(D, Wd) * sizeof(float)))));
    for (Staging &s : c->st)
        if (s.p_cap < chunk * Wd) {
            HIPCHK(c, hipStreamSynchronize(s.stream));
            (void)hipHostFree(s.hP); (void)hipFree(s.dP);
            s.hP = nullptr; s.dP = nullptr; s.p_cap = 0;
            HIPCHK(c, hipHostMalloc((void **)&s.hP, chunk * Wd * sizeof(float), hipHostMallocDefault));
            HIPCHK(c, hipMalloc((void **)&s.dP, chunk * Wd * sizeof(float)));
            s.p_cap = chunk * Wd;
        }
    const uint64_t nchunks = (N + chunk - 1) / chunk;
    auto finish = [&](uint64_t ci) -> int {
        Staging &s = c->st[ci & 1];
        HIPCHK(c, hipStreamSynchronize(s.stream));
        const uint64_t off = ci * chunk, n = std::min<uint64_t>(chunk, N - off);
        unpack(s.hP, off, n, Wd);
        return VQHIP_OK;
    };
    for (uint64_t ci = 0; ci < nchunks; ++ci) {
        Staging &s = c->st[ci & 1];
        if (ci >= 2 && (rc = finish(ci - 2)) != VQHIP_OK) return rc;
        const uint64_t off = ci * chunk, n = std::min<uint64_t>(chunk, N - off);
        pack(s.hX, off, n);
        HIPCHK(c, hipMemcpyAsync(s.dX, s.hX, n * row, hipMemcpyHostToDevice, s.stream));
        if ((rc = project_device(c, s.dX, n, which, s.dP, s.stream)) != VQHIP_OK) return rc;
        HIPCHK(c, hipMemcpyAsync(s.hP, s.dP, n * Wd * sizeof(float), hipMemcpyDeviceToHost, s.stream));
    }
    for (uint64_t ci = nchunks >= 2 ? nchunks - 2 : 0; ci < nchunks; ++ci)
        if ((rc = finish(ci)) != VQHIP_OK) return rc;
    return VQHIP_OK;
}

}  // namespace

extern "C" {

int vqhip_create(int device, vqhip_ctx **out)
try {
    if (!out) {
        g_create_error = "out is NULL";
        return VQHIP_ERR_INVALID;
    }
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_error = std::string("no HIP device available: ") + hipGetErrorString(e);
        return VQHIP_ERR_HIP;
    }
    if (device < 0 || device >= count) {
        g_create_error = "device index out of range";
        return VQHIP_ERR_INVALID;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return VQHIP_ERR_HIP;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
        g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return VQHIP_ERR_HIP;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("libvqhip is built for gfx950 only; device is ") + prop.gcnArchName;
        return VQHIP_ERR_UNSUPPORTED;
    }
    vqhip_ctx *c = new (std::nothrow) vqhip_ctx();
    if (!c) {
        g_create_error = "out of host memory";
        return VQHIP_ERR_NOMEM;
    }
    c->device = device;
    // the device is current: look up this GPU's copy of the no-allocation entry point's counters
    if ((e = noalloc_scratch_setup(device, &c->noalloc_queue)) != hipSuccess) {
        g_create_error = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
        delete c;
        return VQHIP_ERR_HIP;
    }
    *out = c;
    return VQHIP_OK;
}
VQHIP_CATCH(nullptr)

int vqhip_destroy(vqhip_ctx *c)
try {
    if (!c) return VQHIP_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    free_staging(c);
    for (Staging &s : c->st)
        if (s.stream) (void)hipStreamDestroy(s.stream);
    drop_engine(c);
    free_omega(c);
    (void)hipFree(c->dW);
    (void)hipFree(c->dW64);
    free_f64(c);
    (void)hipFree(c->dcls);
    (void)hipFree(c->dYproj);
    (void)hipFree(c->dProj);
    free_gradient(c);
    if (c->scratch_done) (void)hipEventDestroy(c->scratch_done);
    delete c;
    return VQHIP_OK;
}
VQHIP_CATCH(c)

const char *vqhip_last_error(const vqhip_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

const char *vqhip_last_kernel(const vqhip_ctx *c) { return c ? c->last_kernel : "none"; }
const char *vqhip_last_ranking(const vqhip_ctx *c) { return c ? c->last_ranking : "none"; }
const char *vqhip_last_projection(const vqhip_ctx *c) { return c ? c->last_projection : "none"; }

const char *vqhip_version(void) { return "vqhip 0.1 (gfx950)"; }

int vqhip_set_prototypes(vqhip_ctx *c, const float *W, uint32_t K, uint32_t D)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (!W || K == 0 || D == 0) return c->fail(VQHIP_ERR_INVALID, "set_prototypes: W is NULL or K/D is 0");
    HIPCHK(c, hipSetDevice(c->device));
    drop_engine(c);
    free_omega(c);
    (void)hipFree(c->dcls);
    c->dcls = nullptr;
    c->hcls.clear();
    (void)hipFree(c->dW64);
    c->dW64 = nullptr;
    c->hW.assign(W, W + (size_t)K * D);
    c->K = K;
    c->D = D;
    return upload(c, &c->dW, W, (size_t)K * D);
}
VQHIP_CATCH(c)

int vqhip_set_labels(vqhip_ctx *c, const uint32_t *cls, uint32_t K)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (c->K == 0) return c->fail(VQHIP_ERR_STATE, "set_labels before set_prototypes");
    if (!cls || K != c->K) return c->fail(VQHIP_ERR_INVALID, "set_labels: cls is NULL or K differs from the prototypes");
    HIPCHK(c, hipSetDevice(c->device));
    drop_engine(c);
    c->hcls.assign(cls, cls + K);
    return upload(c, &c->dcls, cls, K);
}
VQHIP_CATCH(c)

int vqhip_set_omega(vqhip_ctx *c, const float *Om, uint32_t M, uint32_t D)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (c->K == 0) return c->fail(VQHIP_ERR_STATE, "set_omega before set_prototypes");
    HIPCHK(c, hipSetDevice(c->device));
    drop_engine(c);
    free_omega(c);
    if (!Om) return VQHIP_OK;
    if (M == 0 || D != c->D) return c->fail(VQHIP_ERR_INVALID, "set_omega: M is 0 or D differs from the prototypes");
    c->hOm.assign(Om, Om + (size_t)M * D);
    c->M = M;
    c->omega_kind = OMEGA_GLOBAL;
    int rc = upload(c, &c->dOm, Om, (size_t)M * D);
    if (rc != VQHIP_OK) return rc;
    // P = W . Omega^T in f64, rounded once to f32 (the scalar path searches in the projected space).
    std::vector<float> P((size_t)c->K * M);
    for (uint32_t k = 0; k < c->K; ++k)
        for (uint32_t m = 0; m < M; ++m) {
            double t = 0.0;
            for (uint32_t d = 0; d < D; ++d) t += (double)Om[(size_t)m * D + d] * (double)c->hW[(size_t)k * D + d];
            P[(size_t)k * M + m] = (float)t;
        }
    rc = upload(c, &c->dP, P.data(), P.size());
    if (rc != VQHIP_OK) return rc;
    c->hP = P;
    std::vector<float> OmT((size_t)D * M);
    for (uint32_t m = 0; m < M; ++m)
        for (uint32_t d = 0; d < D; ++d) OmT[(size_t)d * M + m] = Om[(size_t)m * D + d];
    return upload(c, &c->dOmT, OmT.data(), OmT.size());
}
VQHIP_CATCH(c)

int vqhip_set_local_omegas(vqhip_ctx *c, const float *Oms, uint32_t K, uint32_t M, uint32_t D)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (c->K == 0) return c->fail(VQHIP_ERR_STATE, "set_local_omegas before set_prototypes");
    HIPCHK(c, hipSetDevice(c->device));
    drop_engine(c);
    free_omega(c);
    if (!Oms) return VQHIP_OK;
    if (M == 0 || D != c->D || K != c->K)
        return c->fail(VQHIP_ERR_INVALID, "set_local_omegas: M is 0 or K/D differ from the prototypes");
    c->hOm.assign(Oms, Oms + (size_t)K * M * D);
    c->M = M;
    c->omega_kind = OMEGA_LOCAL;
    return upload(c, &c->dOm, Oms, (size_t)K * M * D);
}
VQHIP_CATCH(c)

int vqhip_set_kernel_policy(vqhip_ctx *c, int policy)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (policy < VQHIP_KERNEL_AUTO || policy > VQHIP_KERNEL_MFMA) return c->fail(VQHIP_ERR_INVALID, "unknown kernel policy");
    c->policy = policy;
    return VQHIP_OK;
}
VQHIP_CATCH(c)

int vqhip_set_option(vqhip_ctx *c, int option, int value)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (value != VQHIP_VALUE_DEFAULT && value != VQHIP_VALUE_F32 && value != VQHIP_VALUE_BF16)
        return c->fail(VQHIP_ERR_INVALID, "set_option: unknown value");
    int *slot = option == VQHIP_OPT_RANKING ? &c->opt_ranking : option == VQHIP_OPT_PROJECTION ? &c->opt_projection : nullptr;
    if (!slot) return c->fail(VQHIP_ERR_INVALID, "set_option: unknown option");
    if (*slot == value) return VQHIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    drop_engine(c);                                       // the images depend on it: rebuilt by the next call, like after any vqhip_set_*
    *slot = value;
    return VQHIP_OK;
}
VQHIP_CATCH(c)

int vqhip_assign(vqhip_ctx *c, const float *X, uint64_t N, uint32_t *idx_out, float *dist_out)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (N && !idx_out) return c->fail(VQHIP_ERR_INVALID, "assign: idx_out is NULL");
    if (N && !X) return c->fail(VQHIP_ERR_INVALID, "assign: X is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    return run_host(c, contiguous_f32(c, X), nullptr, N, idx_out, dist_out, nullptr, nullptr);
}
VQHIP_CATCH(c)

int vqhip_assign_rows_f64(vqhip_ctx *c, const double *const *rows, uint64_t N, uint32_t D, uint32_t *idx_out,
                          float *dist_out)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (N && (!rows || !idx_out)) return c->fail(VQHIP_ERR_INVALID, "assign_rows_f64: rows or idx_out is NULL");
    if (c->K && D != c->D) return c->fail(VQHIP_ERR_INVALID, "assign_rows_f64: D differs from the prototypes");
    for (uint64_t i = 0; i < N; ++i)
        if (!rows[i]) return c->fail(VQHIP_ERR_INVALID, "assign_rows_f64: a row pointer is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    return run_host(c, rows_f64(rows, D), nullptr, N, idx_out, dist_out, nullptr, nullptr);
}
VQHIP_CATCH(c)

int vqhip_assign_matched_rows_f64(vqhip_ctx *c, const double *const *rows, const uint32_t *y, uint64_t N, uint32_t D, uint32_t *jpos,
                                  float *dpos, uint32_t *jneg, float *dneg)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (N && (!rows || !y)) return c->fail(VQHIP_ERR_INVALID, "assign_matched_rows_f64: rows or y is NULL");
    if (c->K && D != c->D) return c->fail(VQHIP_ERR_INVALID, "assign_matched_rows_f64: D differs from the prototypes");
    for (uint64_t i = 0; i < N; ++i)
        if (!rows[i]) return c->fail(VQHIP_ERR_INVALID, "assign_matched_rows_f64: a row pointer is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    return run_host(c, rows_f64(rows, D), y, N, jpos, dpos, jneg, dneg);
}
VQHIP_CATCH(c)

int vqhip_assign_device(vqhip_ctx *c, const float *dX, uint64_t N, uint32_t *didx_out, float *ddist_out, void *stream)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (N && (!dX || !didx_out)) return c->fail(VQHIP_ERR_INVALID, "assign_device: dX or didx_out is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    return run_device(c, dX, N, nullptr, Outputs{didx_out, ddist_out, nullptr, nullptr}, (hipStream_t)stream);
}
VQHIP_CATCH(c)

int vqhip_prepare(vqhip_ctx *c)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (c->K == 0) return c->fail(VQHIP_ERR_STATE, "no prototypes set (vqhip_set_prototypes)");
    HIPCHK(c, hipSetDevice(c->device));
    return prepare_engine(c);
}
VQHIP_CATCH(c)

int vqhip_assign_device_noalloc(vqhip_ctx *c, const float *dX, uint64_t N, uint32_t *didx_out, float *ddist_out, void *stream)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (N > VQHIP_NOALLOC_MAX_ROWS)
        return c->fail(VQHIP_ERR_INVALID, "assign_device_noalloc: N exceeds VQHIP_NOALLOC_MAX_ROWS (2^36 rows; the batch counters are 32-bit)");
    if (N && (!dX || !didx_out)) return c->fail(VQHIP_ERR_INVALID, "assign_device_noalloc: dX or didx_out is NULL");
    if (c->K == 0) return c->fail(VQHIP_ERR_STATE, "no prototypes set (vqhip_set_prototypes)");
    // the tile images are device memory: they must exist already (vqhip_prepare, or any assign call since the last vqhip_set_*)
    if (!c->engine_tried || c->policy == VQHIP_KERNEL_SCALAR)
        return c->fail(VQHIP_ERR_STATE, "assign_device_noalloc: model not prepared (call vqhip_prepare after the last vqhip_set_*; not with VQHIP_KERNEL_SCALAR)");
    if (!c->engine)
        return c->fail(VQHIP_ERR_UNSUPPORTED, "assign_device_noalloc serves the models one MFMA kernel serves; this one needs scratch "
                                              "proportional to N (use vqhip_assign_device): " + c->engine_why);
    if (!c->noalloc_queue) return c->fail(VQHIP_ERR_STATE, "assign_device_noalloc: the counters of this GPU were not set up");
    HIPCHK(c, hipSetDevice(c->device));
    if (N == 0) return VQHIP_OK;
    c->last_kernel = engine_name(c->engine);
    c->last_ranking = engine_ranking(c->engine);
    c->last_projection = engine_projection(c->engine);
    // c->noalloc_queue is the device address of `scratch` on this GPU (never the array's name); the launch zeroes it
    HIPCHK(c, engine_assign(c->engine, dX, N, nullptr, Outputs{didx_out, ddist_out, nullptr, nullptr}, (hipStream_t)stream,
                            c->noalloc_queue));
    return VQHIP_OK;
}
VQHIP_CATCH(c)

int vqhip_assign_matched(vqhip_ctx *c, const float *X, const uint32_t *y, uint64_t N, uint32_t *jpos, float *dpos,
                         uint32_t *jneg, float *dneg)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (N && (!y || !X)) return c->fail(VQHIP_ERR_INVALID, "assign_matched: X or y is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    return run_host(c, contiguous_f32(c, X), y, N, jpos, dpos, jneg, dneg);
}
VQHIP_CATCH(c)

int vqhip_assign_matched_device(vqhip_ctx *c, const float *dX, const uint32_t *dy, uint64_t N, uint32_t *djpos,
                                float *ddpos, uint32_t *djneg, float *ddneg, void *stream)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (N && (!dX || !dy)) return c->fail(VQHIP_ERR_INVALID, "assign_matched_device: dX or dy is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    return run_device(c, dX, N, dy, Outputs{djpos, ddpos, djneg, ddneg}, (hipStream_t)stream);
}
VQHIP_CATCH(c)

int vqhip_glvq_batch_gradient_device(vqhip_ctx *c, const float *dX, const uint32_t *dy, uint64_t N, float *ddW, double *dcost, void *stream)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (!ddW || (N && (!dX || !dy))) return c->fail(VQHIP_ERR_INVALID, "glvq_batch_gradient_device: dX, dy or ddW is NULL");
    if (((uintptr_t)ddW & 3) || ((uintptr_t)dcost & 7)) return c->fail(VQHIP_ERR_INVALID, "glvq_batch_gradient_device: misaligned output");
    HIPCHK(c, hipSetDevice(c->device));
    if (dcost) HIPCHK(c, hipMemsetAsync(dcost, 0, 16, (hipStream_t)stream));
    return gradient_device(c, dX, dy, N, ddW, false, dcost, (hipStream_t)stream);
}
VQHIP_CATCH(c)

int vqhip_glvq_batch_gradient(vqhip_ctx *c, const float *X, const uint32_t *y, uint64_t N, float *dW, double *cost, uint64_t *used)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (!dW || (N && (!X || !y))) return c->fail(VQHIP_ERR_INVALID, "glvq_batch_gradient: X, y or dW is NULL");
    if (c->K == 0) return c->fail(VQHIP_ERR_STATE, "no prototypes set (vqhip_set_prototypes)");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_gradient_out(c);
    if (rc != VQHIP_OK) return rc;
    const size_t row = (size_t)c->D * sizeof(float), kd = (size_t)c->K * c->D;
    size_t chunk = std::max<size_t>(1024, (128ull << 20) / row);
    chunk = (size_t)std::min<uint64_t>(chunk, std::max<uint64_t>(N, 1));
    if ((rc = ensure_staging(c, chunk, true)) != VQHIP_OK) return rc;
    chunk = c->st_chunk;
    // the two pinned buffers alternate (the CPU packs chunk i + 1 while chunk i goes up and is worked); the device work of
    // all chunks is ordered on ONE stream: they add into the same gradient
    hipStream_t s = c->st[0].stream;
    HIPCHK(c, hipMemsetAsync(c->gCost, 0, 16, s));
    const Packer pack = contiguous_f32(c, X);
    hipEvent_t done[2] = {nullptr, nullptr};
    auto cleanup = [&]() { for (hipEvent_t e : done) if (e) (void)hipEventDestroy(e); };
    rc = gradient_device(c, nullptr, nullptr, 0, c->gOut, false, nullptr, s);                  // zero the gradient
    for (uint64_t off = 0, ci = 0; off < N && rc == VQHIP_OK; off += chunk, ++ci) {
        Staging &b = c->st[ci & 1];
        const uint64_t n = std::min<uint64_t>(chunk, N - off);
        if (done[ci & 1]) {
            if (hipEventSynchronize(done[ci & 1]) != hipSuccess) { rc = c->fail(VQHIP_ERR_HIP, "glvq_batch_gradient: event"); break; }
        } else if (hipEventCreateWithFlags(&done[ci & 1], hipEventDisableTiming) != hipSuccess) { rc = c->fail(VQHIP_ERR_HIP, "glvq_batch_gradient: event"); break; }
        pack(b.hX, off, n);
        std::memcpy(b.hY, y + off, n * sizeof(uint32_t));
        hipError_t e = hipMemcpyAsync(b.dX, b.hX, n * row, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(b.dY, b.hY, n * sizeof(uint32_t), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) { rc = c->fail_hip(e, "glvq_batch_gradient: upload"); break; }
        rc = gradient_device(c, b.dX, b.dY, n, c->gOut, true, c->gCost, s);
        if (rc == VQHIP_OK && hipEventRecord(done[ci & 1], s) != hipSuccess) rc = c->fail(VQHIP_ERR_HIP, "glvq_batch_gradient: event");
    }
    if (rc == VQHIP_OK) {
        unsigned long long tail[2] = {0, 0};
        hipError_t e = hipMemcpyAsync(dW, c->gOut, kd * sizeof(float), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(tail, c->gCost, 16, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = c->fail_hip(e, "glvq_batch_gradient: download");
        if (rc == VQHIP_OK) {
            if (cost) std::memcpy(cost, &tail[0], 8);
            if (used) *used = tail[1];
        }
    } else {
        (void)hipStreamSynchronize(s);
    }
    cleanup();
    return rc;
}
VQHIP_CATCH(c)

int vqhip_project_device(vqhip_ctx *c, const float *dX, uint64_t N, int which, float *dY, void *stream)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (N && (!dX || !dY)) return c->fail(VQHIP_ERR_INVALID, "project_device: dX or dY is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    return project_device(c, dX, N, which, dY, (hipStream_t)stream);
}
VQHIP_CATCH(c)

int vqhip_project(vqhip_ctx *c, const float *X, uint64_t N, int which, float *Y)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (N && (!X || !Y)) return c->fail(VQHIP_ERR_INVALID, "project: X or Y is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    return run_host_project(c, contiguous_f32(c, X), N, which, [Y](const float *src, uint64_t first, uint64_t n, uint32_t Wd) {
        float *dst = Y + first * Wd;
        parallel_rows(n, [=](uint64_t a, uint64_t b) { std::memcpy(dst + a * Wd, src + a * Wd, (b - a) * Wd * sizeof(float)); });
    });
}
VQHIP_CATCH(c)

int vqhip_project_rows_f64(vqhip_ctx *c, const double *const *rows, uint64_t N, uint32_t D, int which, double *const *out_rows)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (N && (!rows || !out_rows)) return c->fail(VQHIP_ERR_INVALID, "project_rows_f64: rows or out_rows is NULL");
    if (c->K && D != c->D) return c->fail(VQHIP_ERR_INVALID, "project_rows_f64: D differs from the prototypes");
    for (uint64_t i = 0; i < N; ++i)
        if (!rows[i] || !out_rows[i]) return c->fail(VQHIP_ERR_INVALID, "project_rows_f64: a row pointer is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    // gathered and narrowed into the pinned upload buffers, widened straight out of the pinned download buffers: no
    // contiguous copy of either matrix on the host
    return run_host_project(c, rows_f64(rows, D), N, which, [out_rows](const float *src, uint64_t first, uint64_t n, uint32_t Wd) {
        parallel_rows(n, [=](uint64_t a, uint64_t b) {
            for (uint64_t i = a; i < b; ++i) {
                double *dst = out_rows[first + i];
                const float *r = src + i * Wd;
                for (uint32_t d = 0; d < Wd; ++d) dst[d] = (double)r[d];
            }
        });
    });
}
VQHIP_CATCH(c)

int vqhip_set_prototypes_f64(vqhip_ctx *c, const double *W, uint32_t K, uint32_t D)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (!W || K == 0 || D == 0) return c->fail(VQHIP_ERR_INVALID, "set_prototypes_f64: W is NULL or K/D is 0");
    // the f32 model is set from the narrowed values, so both paths describe the same prototypes
    std::vector<float> narrow((size_t)K * D);
    for (size_t i = 0; i < narrow.size(); ++i) narrow[i] = (float)W[i];
    int rc = vqhip_set_prototypes(c, narrow.data(), K, D);
    if (rc != VQHIP_OK) return rc;
    return upload(c, &c->dW64, W, (size_t)K * D);
}
VQHIP_CATCH(c)

int vqhip_assign_f64(vqhip_ctx *c, const double *X, uint64_t N, uint32_t *idx_out, double *dist_out)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (N && (!X || !idx_out)) return c->fail(VQHIP_ERR_INVALID, "assign_f64: X or idx_out is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    return run_host_f64(c, X, nullptr, N, idx_out, dist_out, nullptr, nullptr);
}
VQHIP_CATCH(c)

int vqhip_assign_matched_f64(vqhip_ctx *c, const double *X, const uint32_t *y, uint64_t N, uint32_t *jpos, double *dpos,
                             uint32_t *jneg, double *dneg)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if (N && (!X || !y)) return c->fail(VQHIP_ERR_INVALID, "assign_matched_f64: X or y is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    return run_host_f64(c, X, y, N, jpos, dpos, jneg, dneg);
}
VQHIP_CATCH(c)

int vqhip_synth_fill_device(vqhip_ctx *c, float *dout, uint64_t count, uint64_t first, uint64_t seed,
                            uint32_t stream_id, float scale, const float *dbase, uint64_t base_len, void *stream)
try {
    if (!c) return VQHIP_ERR_INVALID;
    if ((count && !dout) || (dbase && base_len == 0)) return c->fail(VQHIP_ERR_INVALID, "synth_fill_device: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, launch_synth_fill(dout, count, first, seed, stream_id, scale, dbase, base_len, (hipStream_t)stream));
    return VQHIP_OK;
}
VQHIP_CATCH(c)

// ---- several GPUs behind one handle -------------------------------------------------------------------------------
// One context per device, the model replicated, the samples of a call cut into contiguous shards
// [g * ceil(N / n), (g + 1) * ceil(N / n)), one host thread per device driving its context's pinned pipeline, results
// written straight into the caller's arrays (no exchange between the GPUs: every sample is independent,
// src/glvq/glvq.rs:224-232).  The device-resident, one-process-per-GPU form with the RCCL gather is vq_amd/dist.py.
}  // extern "C"

struct vqhip_multi {
    std::vector<vqhip_ctx *> ctx;
    // device-resident shards (vqhip_multi_assign_device): one stream per context, and one RCCL communicator per device
    // of a single-process clique (ncclCommInitAll), created by the first call that needs them
    std::vector<hipStream_t> streams;
    std::vector<ncclComm_t> comms;
    std::string err;
    int fail(int code, const std::string &msg)
    {
        err = msg;
        return code;
    }
};

namespace {
thread_local std::string g_multi_error;

// RCCL is bound at first use (dlopen), not at link time: only vqhip_multi_assign_device needs it, and a process that
// already holds a copy (PyTorch ships its own librccl.so.1) must end up with that one copy -- the loader hands back the
// library that is already mapped under the same soname.
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    std::string why;
};
Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            const char *e = dlerror();
            r.why = std::string("librccl.so.1 not found: ") + (e ? e : "?");
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(r.handle, n);
            if (!p && r.why.empty()) r.why = std::string("librccl: symbol ") + n + " missing";
            return p;
        };
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
    });
    return r;
}


int on_exception_multi(vqhip_multi *m) noexcept
{
    int code = VQHIP_ERR_INVALID;
    try {
        throw;
    } catch (const std::bad_alloc &) {
        code = VQHIP_ERR_NOMEM;
    } catch (...) {
    }
    try {
        (m ? m->err : g_multi_error) = code == VQHIP_ERR_NOMEM ? "out of host memory" : "internal error (C++ exception)";
    } catch (...) {
    }
    return code;
}

// fn(context, shard number) on every context, one host thread each; the first failure's code and text are kept
int on_all(vqhip_multi *m, const std::function<int(vqhip_ctx *, size_t)> &fn, bool threads)
{
    const size_t n = m->ctx.size();
    std::vector<int> rc(n, VQHIP_OK);
    auto run = [&](size_t g) noexcept {
        try {
            rc[g] = fn(m->ctx[g], g);
        } catch (...) {
            rc[g] = on_exception(m->ctx[g]);
        }
    };
    std::vector<std::thread> pool;
    size_t started = 1;
    if (threads && n > 1) {
        pool.reserve(n);
        for (; started < n; ++started) {
            try {
                pool.emplace_back(run, started);
            } catch (...) {                               // no more threads: the rest runs on this one
                break;
            }
        }
    }
    run(0);
    for (size_t g = started; g < n; ++g) run(g);
    for (std::thread &t : pool) t.join();
    for (size_t g = 0; g < n; ++g)
        if (rc[g] != VQHIP_OK)
            return m->fail(rc[g], "device " + std::to_string(m->ctx[g]->device) + " (shard " + std::to_string(g) + "): " + m->ctx[g]->err);
    return VQHIP_OK;
}

struct Shard {
    uint64_t first, rows;
};
Shard shard_of(uint64_t N, size_t n, size_t g)
{
    const uint64_t per = (N + n - 1) / n, first = std::min<uint64_t>(N, g * per);
    return {first, std::min<uint64_t>(per, N - first)};
}
}  // namespace

#define VQHIP_CATCH_MULTI(mm) \
    catch (...) { return on_exception_multi(mm); }

extern "C" {

int vqhip_multi_create(const int *devices, int n_devices, vqhip_multi **out)
try {
    if (!out) return VQHIP_ERR_INVALID;
    *out = nullptr;
    if (!devices || n_devices <= 0 || n_devices > MAX_DEVICES) {
        g_multi_error = "multi_create: devices is NULL or n_devices out of range";
        return VQHIP_ERR_INVALID;
    }
    vqhip_multi *m = new vqhip_multi();
    for (int i = 0; i < n_devices; ++i) {
        vqhip_ctx *c = nullptr;
        const int rc = vqhip_create(devices[i], &c);
        if (rc != VQHIP_OK) {
            g_multi_error = "device " + std::to_string(devices[i]) + ": " + vqhip_last_error(nullptr);
            for (vqhip_ctx *k : m->ctx) (void)vqhip_destroy(k);
            delete m;
            return rc;
        }
        m->ctx.push_back(c);
    }
    *out = m;
    return VQHIP_OK;
}
VQHIP_CATCH_MULTI(nullptr)

int vqhip_multi_destroy(vqhip_multi *m)
try {
    if (!m) return VQHIP_OK;
    for (ncclComm_t comm : m->comms) (void)rccl().CommDestroy(comm);
    for (size_t g = 0; g < m->streams.size(); ++g) {
        (void)hipSetDevice(m->ctx[g]->device);
        (void)hipStreamDestroy(m->streams[g]);
    }
    for (vqhip_ctx *c : m->ctx) (void)vqhip_destroy(c);
    delete m;
    return VQHIP_OK;
}
VQHIP_CATCH_MULTI(nullptr)

const char *vqhip_multi_last_error(const vqhip_multi *m) { return m ? m->err.c_str() : g_multi_error.c_str(); }

int vqhip_multi_device_count(const vqhip_multi *m) { return m ? (int)m->ctx.size() : 0; }

int vqhip_multi_set_prototypes(vqhip_multi *m, const float *W, uint32_t K, uint32_t D)
try {
    if (!m) return VQHIP_ERR_INVALID;
    return on_all(m, [=](vqhip_ctx *c, size_t) { return vqhip_set_prototypes(c, W, K, D); }, false);
}
VQHIP_CATCH_MULTI(m)

int vqhip_multi_set_labels(vqhip_multi *m, const uint32_t *cls, uint32_t K)
try {
    if (!m) return VQHIP_ERR_INVALID;
    return on_all(m, [=](vqhip_ctx *c, size_t) { return vqhip_set_labels(c, cls, K); }, false);
}
VQHIP_CATCH_MULTI(m)

int vqhip_multi_set_omega(vqhip_multi *m, const float *Om, uint32_t M, uint32_t D)
try {
    if (!m) return VQHIP_ERR_INVALID;
    return on_all(m, [=](vqhip_ctx *c, size_t) { return vqhip_set_omega(c, Om, M, D); }, false);
}
VQHIP_CATCH_MULTI(m)

int vqhip_multi_set_local_omegas(vqhip_multi *m, const float *Oms, uint32_t K, uint32_t M, uint32_t D)
try {
    if (!m) return VQHIP_ERR_INVALID;
    return on_all(m, [=](vqhip_ctx *c, size_t) { return vqhip_set_local_omegas(c, Oms, K, M, D); }, false);
}
VQHIP_CATCH_MULTI(m)

int vqhip_multi_set_option(vqhip_multi *m, int option, int value)
try {
    if (!m) return VQHIP_ERR_INVALID;
    return on_all(m, [=](vqhip_ctx *c, size_t) { return vqhip_set_option(c, option, value); }, false);
}
VQHIP_CATCH_MULTI(m)

int vqhip_multi_assign(vqhip_multi *m, const float *X, uint64_t N, uint32_t *idx_out, float *dist_out)
try {
    if (!m) return VQHIP_ERR_INVALID;
    if (N && (!X || !idx_out)) return m->fail(VQHIP_ERR_INVALID, "multi_assign: X or idx_out is NULL");
    const size_t n = m->ctx.size();
    return on_all(m, [=](vqhip_ctx *c, size_t g) {
        const Shard sh = shard_of(N, n, g);
        return vqhip_assign(c, X + sh.first * c->D, sh.rows, idx_out + sh.first, dist_out ? dist_out + sh.first : nullptr);
    }, true);
}
VQHIP_CATCH_MULTI(m)

int vqhip_multi_glvq_batch_gradient(vqhip_multi *m, const float *X, const uint32_t *y, uint64_t N, float *dW, double *cost, uint64_t *used)
try {
    if (!m) return VQHIP_ERR_INVALID;
    if (!dW || (N && (!X || !y))) return m->fail(VQHIP_ERR_INVALID, "multi_glvq_batch_gradient: X, y or dW is NULL");
    const size_t n = m->ctx.size();
    const size_t kd = (size_t)m->ctx[0]->K * m->ctx[0]->D;
    // every device: the gradient of its shard; the shards' gradients are summed on the host in shard order (K x D floats per
    // device: nothing next to the rows that went up).  Device-resident shards would all-reduce over the handle's RCCL clique.
    std::vector<std::vector<float>> part(n, std::vector<float>(kd));
    std::vector<double> pc(n, 0.0);
    std::vector<uint64_t> pu(n, 0);
    const int rc = on_all(m, [&](vqhip_ctx *c, size_t g) {
        const Shard sh = shard_of(N, n, g);
        return vqhip_glvq_batch_gradient(c, X + sh.first * c->D, y + sh.first, sh.rows, part[g].data(), &pc[g], &pu[g]);
    }, true);
    if (rc != VQHIP_OK) return rc;
    for (size_t e = 0; e < kd; ++e) {
        float t = 0.0f;
        for (size_t g = 0; g < n; ++g) t += part[g][e];
        dW[e] = t;
    }
    if (cost) { *cost = 0.0; for (double v : pc) *cost += v; }
    if (used) { *used = 0; for (uint64_t v : pu) *used += v; }
    return VQHIP_OK;
}
VQHIP_CATCH_MULTI(m)

// ---- device-resident shards + the RCCL gather of the results (SURVEY.md section 8 row e) ---------------------------------
namespace {
// streams and the communicator clique of the handle's devices (once)
int multi_device_setup(vqhip_multi *m)
{
    const size_t n = m->ctx.size();
    if (m->streams.empty()) {
        std::vector<hipStream_t> st(n, nullptr);
        for (size_t g = 0; g < n; ++g) {
            hipError_t e = hipSetDevice(m->ctx[g]->device);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&st[g], hipStreamNonBlocking);
            if (e != hipSuccess) {
                for (size_t k = 0; k < g; ++k) {
                    (void)hipSetDevice(m->ctx[k]->device);
                    (void)hipStreamDestroy(st[k]);
                }
                return m->fail(VQHIP_ERR_HIP, std::string("multi_assign_device: stream: ") + hipGetErrorString(e));
            }
        }
        m->streams = st;
    }
    if (m->comms.empty()) {
        std::vector<int> devs(n);
        for (size_t g = 0; g < n; ++g) devs[g] = m->ctx[g]->device;
        for (size_t g = 0; g < n; ++g)
            for (size_t k = 0; k < g; ++k)
                if (devs[g] == devs[k])
                    return m->fail(VQHIP_ERR_UNSUPPORTED, "multi_assign_device: a device is listed twice (RCCL wants one rank per GPU); use the host-pointer calls");
        Rccl &R = rccl();
        if (!R.why.empty()) return m->fail(VQHIP_ERR_UNSUPPORTED, "multi_assign_device: " + R.why);
        std::vector<ncclComm_t> comms(n);
        const ncclResult_t r = R.CommInitAll(comms.data(), (int)n, devs.data());
        if (r != ncclSuccess) return m->fail(VQHIP_ERR_HIP, std::string("ncclCommInitAll: ") + R.GetErrorString(r));
        m->comms = comms;
    }
    return VQHIP_OK;
}
}  // namespace

int vqhip_multi_assign_device(vqhip_multi *m, const float *const *dX, const uint64_t *n_rows, uint32_t *const *didx_all,
                              float *const *ddist_all)
try {
    if (!m) return VQHIP_ERR_INVALID;
    if (!dX || !n_rows || !didx_all) return m->fail(VQHIP_ERR_INVALID, "multi_assign_device: dX, n_rows or didx_all is NULL");
    const size_t n = m->ctx.size();
    std::vector<uint64_t> first(n + 1, 0);
    bool equal = true;
    for (size_t g = 0; g < n; ++g) {
        if ((n_rows[g] && !dX[g]) || !didx_all[g]) return m->fail(VQHIP_ERR_INVALID, "multi_assign_device: a device's pointer is NULL");
        if (ddist_all && !ddist_all[g]) return m->fail(VQHIP_ERR_INVALID, "multi_assign_device: a device's distance pointer is NULL");
        first[g + 1] = first[g] + n_rows[g];
        equal = equal && n_rows[g] == n_rows[0];
    }
    int rc = multi_device_setup(m);
    if (rc != VQHIP_OK) return rc;
    // every device searches its own shard into its own slice of its own copy of the result arrays ...
    for (size_t g = 0; g < n; ++g) {
        vqhip_ctx *c = m->ctx[g];
        rc = vqhip_assign_device(c, dX[g], n_rows[g], didx_all[g] + first[g], ddist_all ? ddist_all[g] + first[g] : nullptr, m->streams[g]);
        if (rc != VQHIP_OK) return m->fail(rc, "device " + std::to_string(c->device) + " (shard " + std::to_string(g) + "): " + c->err);
    }
    // ... and the slices are exchanged in place, behind the kernels on the same streams: one grouped all-gather per array
    // (equal shards), or one broadcast per shard (the all-gather of unequal pieces).  8 bytes per sample over xGMI.
    Rccl &R = rccl();
    auto exchange = [&](auto *const *arrays, ncclDataType_t type) -> ncclResult_t {
        ncclResult_t r = R.GroupStart();
        for (size_t g = 0; g < n && r == ncclSuccess; ++g) {
            if (equal) {
                r = R.AllGather(arrays[g] + first[g], arrays[g], (size_t)n_rows[0], type, m->comms[g], m->streams[g]);
            } else {
                for (size_t root = 0; root < n && r == ncclSuccess; ++root)
                    if (n_rows[root])
                        r = R.Broadcast(arrays[g] + first[root], arrays[g] + first[root], (size_t)n_rows[root], type, (int)root,
                                        m->comms[g], m->streams[g]);
            }
        }
        const ncclResult_t e = R.GroupEnd();
        return r != ncclSuccess ? r : e;
    };
    if (first[n] != 0) {
        ncclResult_t r = exchange(didx_all, ncclUint32);
        if (r == ncclSuccess && ddist_all) r = exchange(ddist_all, ncclFloat32);
        if (r != ncclSuccess) return m->fail(VQHIP_ERR_HIP, std::string("RCCL gather: ") + R.GetErrorString(r));
    }
    for (size_t g = 0; g < n; ++g) {
        hipError_t e = hipSetDevice(m->ctx[g]->device);
        if (e == hipSuccess) e = hipStreamSynchronize(m->streams[g]);
        if (e != hipSuccess) return m->fail(VQHIP_ERR_HIP, std::string("multi_assign_device: ") + hipGetErrorString(e));
    }
    return VQHIP_OK;
}
VQHIP_CATCH_MULTI(m)

int vqhip_multi_assign_rows_f64(vqhip_multi *m, const double *const *rows, uint64_t N, uint32_t D, uint32_t *idx_out,
                                float *dist_out)
try {
    if (!m) return VQHIP_ERR_INVALID;
    if (N && (!rows || !idx_out)) return m->fail(VQHIP_ERR_INVALID, "multi_assign_rows_f64: rows or idx_out is NULL");
    const size_t n = m->ctx.size();
    return on_all(m, [=](vqhip_ctx *c, size_t g) {
        const Shard sh = shard_of(N, n, g);
        return vqhip_assign_rows_f64(c, rows + sh.first, sh.rows, D, idx_out + sh.first, dist_out ? dist_out + sh.first : nullptr);
    }, true);
}
VQHIP_CATCH_MULTI(m)

int vqhip_multi_assign_matched(vqhip_multi *m, const float *X, const uint32_t *y, uint64_t N, uint32_t *jpos, float *dpos,
                               uint32_t *jneg, float *dneg)
try {
    if (!m) return VQHIP_ERR_INVALID;
    if (N && (!X || !y)) return m->fail(VQHIP_ERR_INVALID, "multi_assign_matched: X or y is NULL");
    const size_t n = m->ctx.size();
    return on_all(m, [=](vqhip_ctx *c, size_t g) {
        const Shard sh = shard_of(N, n, g);
        auto at = [&](auto *p) { return p ? p + sh.first : p; };
        return vqhip_assign_matched(c, X + sh.first * c->D, y + sh.first, sh.rows, at(jpos), at(dpos), at(jneg), at(dneg));
    }, true);
}
VQHIP_CATCH_MULTI(m)

}  // extern "C"
