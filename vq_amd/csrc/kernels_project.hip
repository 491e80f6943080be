// project / project_omega on the matrix cores: src/gmlvq/gmlvq.rs:354-372,
// src/liramlvq/liramlvq.rs:401-446.  Kernel and its host-side model (tile image of Omega or Lambda in LDS).
#include <algorithm>
#include <cstdlib>
#include <thread>

#include "mfma_device.h"
#include "mfma_images.h"

namespace vq {

// ------------------------------------------------------------------------------ projection kernel
// Y = X . A^T for a matrix A with `ntiles` tiles of 32 rows resident in LDS: A = Omega for project_omega
// (src/liramlvq/liramlvq.rs:429-446), A = Lambda = Omega^T Omega for project (src/gmlvq/gmlvq.rs:354-372,
// src/liramlvq/liramlvq.rs:401-419: the reference builds Lambda once and multiplies every sample by it).
// The projection stage of the search kernels with a store behind it: a wave owns 32 samples per batch,
// rows prefetched a batch ahead, lane (s, h) of a tile's accumulator holds Y[s][32 t + 8 g + 4 h .. + 3],
// stored as 16 bytes (the lane pair writes 32 adjacent bytes, the 4 groups of a tile a whole 128-byte line
// per sample).  Columns at or past out_dim and rows past N are not written (predicate / buffer range check).
struct ProjectArgs {
    const float *X;           // N x D, rows 4-byte aligned
    uint64_t N;
    uint32_t D;               // <= R; values at or past D are cut inside the kernel
    const float *img;         // ntiles tiles of Geom<R> (PJ: of Geom3<R>)
    uint32_t ntiles;
    float *Y;                 // N x ldy
    uint32_t ldy, col0, out_dim;   // this launch writes columns [col0, min(col0 + 32 ntiles, out_dim))
    uint64_t nbatches;
    uint32_t *queue;
};

// CUT: rows shorter than R (any length, 4-byte aligned): one register set, cut at D in place once it has landed.
// PJ: the products on the bf16 matrix cores (mfma_device.h split3 / tile_mma_pj): A resident as Geom3 tiles (three bf16
// pieces per value), a batch's rows split into pieces once, six piece products per product -- as exact as the f32 chain
// (tools/bf16x6_probe.hip) at 2.7 times its matrix-pipe rate, which leaves these kernels to the memory pipe.
template <int R, bool CUT, int NW = 8, bool PJ = false>
__global__ void __launch_bounds__(NW * 64, NW / 4) k_project(const ProjectArgs a)
{
    using Gm = Geom<R>;
    constexpr uint32_t TILE_BYTES = PJ ? Geom3<R>::TILE_BYTES : Gm::TILE_BYTES;
    // Register sets of 32 rows each.  R <= 128: two, worked alternately -- the loads that refill a set go
    // out right behind its last MFMA and have the other set's whole batch to land, with no copies between
    // sets.  R = 256: one set is all the register file holds at two waves per SIMD; the partner wave of
    // the SIMD covers the wait.
    constexpr int SETS = R <= 128 && !CUT ? 2 : 1;
    __shared__ __attribute__((aligned(16))) char smem[RES_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    {
        const uint32_t bytes = a.ntiles * TILE_BYTES;
        const char *src = (const char *)a.img;
        for (uint32_t p = wave; p * 1024u < bytes; p += NW) {
            const uint32_t off = p * 1024u + lane * 16u;
            if (off < bytes)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                                 (__attribute__((address_space(3))) void *)(smem + p * 1024u), 16, 0, 0);
        }
        __syncthreads();
    }
    const float *img = (const float *)smem;

    // blocks of SETS batches: the first fixed, the others claimed a block ahead (BlockQueue, mfma_device.h)
    BlockQueue<SETS> queue(a.queue, lane, blockIdx.x * NW + (uint32_t)__builtin_amdgcn_readfirstlane(wave), gridDim.x * NW, a.nbatches);
    uint64_t blk = (uint64_t)(blockIdx.x * NW + (uint32_t)__builtin_amdgcn_readfirstlane(wave)) * SETS;
    queue.request();

    const uint32_t row_bytes = a.D * 4u;
    const uint32_t voff = (uint32_t)s_col * row_bytes + (uint32_t)h * 16u;
    auto records = [&](uint64_t bb, uint32_t unit) -> uint32_t {
        const uint64_t first = bb * 32;
        if (first >= a.N) return 0u;
        const uint64_t left = (a.N - first) * unit;
        return left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)left;
    };
    f32x4 x[SETS][R / 8];
    auto fetch = [&](f32x4 (&xs)[R / 8], uint64_t bb) {
        const __amdgpu_buffer_rsrc_t rx =
            __builtin_amdgcn_make_buffer_rsrc((void *)(a.X + bb * 32 * a.D), 0, (int)records(bb, row_bytes), 0x00020000);
        // issue order = use order: vmcnt counts in order, and a set's first chunk must not wait for the
        // loads issued behind it (its own later chunks, the other set's refill)
#pragma unroll
        for (int j = 0; j < R / 8; ++j) {
            xs[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff + j * 32, 0, 0));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll
    for (int q = 0; q < SETS; ++q) fetch(x[q], blk + q);
    const uint32_t ybytes = a.ldy * 4u;
    const uint32_t yoff = (uint32_t)s_col * ybytes;

    // all tiles of one batch: Y[b*32 + s][col0 + 32 t + 8 g + 4 h .. + 3] = accumulator group g of tile t
    const bool vec_store = (a.ldy & 3u) == 0u && ((uintptr_t)a.Y & 15u) == 0u;    // else 4-byte stores, column by column
    // (always_inline: called once per register set; as a real call the sets would live in scratch)
    auto project_batch = [&](f32x4 (&xs)[R / 8], uint64_t bb, bool claim) __attribute__((always_inline)) {
        if constexpr (CUT) {
#pragma unroll
            for (int j = 0; j < R / 8; ++j) xs[j] = mask_chunk(xs[j], (uint32_t)(2 * j + h), a.D);
        }
        const __amdgpu_buffer_rsrc_t ry =
            __builtin_amdgcn_make_buffer_rsrc((void *)(a.Y + bb * 32 * a.ldy), 0, (int)records(bb, ybytes), 0x00020000);
        u32x4 xp[1][PJ ? R / 16 : 1][3];                       // PJ: the batch's rows as pieces (the f32 rows are dead behind this)
        if constexpr (PJ) {
#pragma unroll
            for (int q = 0; q < R / 16; ++q) split24(xs[2 * q], xs[2 * q + 1], xp[0][q]);
        }
        for (uint32_t t = 0; t < a.ntiles; ++t) {
            f32x16 acc;
            if constexpr (PJ) {
                f32x16 accs[1];
                tile_mma_pj<R, 1>((const char *)img + (size_t)t * TILE_BYTES, xp, accs, s_col, h);   // (the tiles' init columns are zero)
                acc = accs[0];
            } else {
                const float *rowp = img + (size_t)t * Gm::TILE_FLOATS + Gm::chunk_off(s_col, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                f32x4 anext = *(const f32x4 *)rowp;
                static_for<0, R / 8>([&](auto jj) {
                    constexpr int j = decltype(jj)::value;
                    const f32x4 acur = anext;
                    if constexpr (j + 1 < R / 8) anext = *(const f32x4 *)(rowp + 8 * (j + 1));
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[e], xs[j][e], acc, 0, 0, 0);
                });
            }
            // the claim for the block after next goes out once this block's rows have been waited for
            if (claim && t == 0) {
                asm volatile("" ::: "memory");
                queue.request();
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint32_t col = a.col0 + 32u * t + 8u * g + 4u * h;
                if (vec_store && col + 4u <= a.out_dim) {
                    const f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, yoff + col * 4u, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float val = acc[4 * g + e];      // (a bit_cast of the vector element itself stores element 0)
                        if (col + e < a.out_dim)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, val), ry, yoff + (col + e) * 4u, 0, 0);
                    }
                }
            }
        }
    };

    while (blk < a.nbatches) {
        const uint64_t nblk = queue.take();
        static_for<0, SETS>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            project_batch(x[q], blk + q, q == 0);
            asm volatile("" ::: "memory");
            fetch(x[q], nblk + q);                     // this set is free: refill it for the next block
        });
        blk = nblk;
    }
}

// ------------------------------------------------------------------------------ host side
struct ProjectModel {
    uint32_t rows = 0, D = 0, R = 0, ntiles = 0, tile_floats = 0, tiles_per_launch = 0;
    bool pj = false;                                 // Geom3 tiles, products on the bf16 matrix cores
    float *img = nullptr;
    uint32_t *queue = nullptr;                       // QUEUE_SLOTS counters, as in EngineModel
    mutable uint32_t next_slot = 0;
    hipError_t (*launch)(const ProjectArgs &, hipStream_t) = nullptr;
};

namespace {
template <int R>
hipError_t launch_project_t(const ProjectArgs &a, hipStream_t st)
{
    const uint64_t wgs = (a.nbatches + 7) / 8;
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);
    const hipError_t e = hipMemsetAsync(a.queue, 0, QUEUE_BYTES, st);
    if (e != hipSuccess) return e;
    if (a.D < (uint32_t)R)
        hipLaunchKernelGGL((k_project<R, true>), dim3(grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((k_project<R, false>), dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}

template <int R>
hipError_t launch_project_pj(const ProjectArgs &a, hipStream_t st)
{
    const uint64_t wgs = (a.nbatches + 7) / 8;
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);
    const hipError_t e = hipMemsetAsync(a.queue, 0, QUEUE_BYTES, st);
    if (e != hipSuccess) return e;
    if (a.D < (uint32_t)R)
        hipLaunchKernelGGL((k_project<R, true, 8, true>), dim3(grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((k_project<R, false, 8, true>), dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}

template <int R>
bool bind_project(ProjectModel &m, std::vector<float> &img, const std::vector<float> &A, const BuildOptions &opt)
{
    if (m.D > (uint32_t)R || (R > 32 && m.D <= (uint32_t)R / 2)) return false;       // narrowest instantiation that holds a row
    uint32_t ngroups = 0;
    m.R = R;
    m.ntiles = (m.rows + 31) / 32;
    if constexpr (R <= 128) {
        if (opt.bf_rank && opt.bf_proj) {
            img = build_images3<R>(A.data(), m.rows, m.D, m.D, nullptr, nullptr, ngroups);
            m.tile_floats = Geom3<R>::TILE_BYTES / 4;
            m.tiles_per_launch = (uint32_t)(RES_BYTES / Geom3<R>::TILE_BYTES);
            m.launch = &launch_project_pj<R>;
            m.pj = true;
            return true;
        }
    }
    img = build_images<R>(A.data(), m.rows, m.D, m.D, 1.0f, nullptr, nullptr, 0.0f, ngroups);
    m.tile_floats = Geom<R>::TILE_FLOATS;
    m.tiles_per_launch = (uint32_t)(RES_BYTES / Geom<R>::TILE_BYTES);
    m.launch = &launch_project_t<R>;
    return true;
}
}  // namespace

ProjectModel *project_build(const std::vector<float> &A, uint32_t rows, uint32_t D, const BuildOptions &opt, hipError_t &herr)
{
    herr = hipSuccess;
    ProjectModel m;
    m.rows = rows;
    m.D = D;
    std::vector<float> img;
    if (!(bind_project<32>(m, img, A, opt) || bind_project<64>(m, img, A, opt) || bind_project<128>(m, img, A, opt) || bind_project<256>(m, img, A, opt)))
        return nullptr;
    ProjectModel *out = new ProjectModel(m);
    herr = to_device(&out->img, img);
    if (herr == hipSuccess) herr = hipMalloc((void **)&out->queue, QUEUE_SLOTS * QUEUE_STRIDE * sizeof(uint32_t));
    if (herr != hipSuccess) {
        project_free(out);
        return nullptr;
    }
    return out;
}

bool project_on_bf16(const ProjectModel *m) { return m && m->pj; }

void project_free(ProjectModel *m)
{
    if (!m) return;
    (void)hipFree(m->img);
    (void)hipFree(m->queue);
    delete m;
}

// dY: N x ldy, 16-byte aligned, ldy % 4 == 0 (the caller checks); column blocks of tiles_per_launch tiles
hipError_t project_run(const ProjectModel *m, const float *dX, uint64_t N, float *dY, uint32_t ldy, hipStream_t s)
{
    if (N == 0) return hipSuccess;
    for (uint32_t t0 = 0; t0 < m->ntiles; t0 += m->tiles_per_launch) {
        ProjectArgs a;
        a.X = dX; a.N = N; a.D = m->D;
        a.img = m->img + (size_t)t0 * m->tile_floats;
        a.ntiles = std::min(m->tiles_per_launch, m->ntiles - t0);
        a.Y = dY; a.ldy = ldy; a.col0 = t0 * 32u; a.out_dim = m->rows;
        a.nbatches = (N + 31) / 32;
        a.queue = m->queue + (m->next_slot++ % QUEUE_SLOTS) * QUEUE_STRIDE;
        const hipError_t e = m->launch(a, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ------------------------------------------------------------------------------ LGMLVQ::project (src/lgmlvq/lgmlvq.rs:372-393)
// Y[i] = Lambda_j x_i with j the sample's nearest prototype (the search has run; `win` holds j).  The samples
// are grouped by j on the device (count, plan, scatter: a counting sort into slots of 32, every group padded to
// whole batches), then every batch of 32 slots is one k_project batch against ITS matrix: rows gathered through
// the slot table, the A fragments of Lambda_j streamed from L2 in the 1 KiB piece layout of k_omega_pipe
// (K matrices do not fit LDS: 4 MiB at config 5), results scattered back to the samples' own rows.
// Lambda_j = Omega_j^T Omega_j is built once per model in f64 and rounded to f32, as the reference builds it
// before multiplying (lgmlvq.rs:384-387).
constexpr uint32_t GROUP_MAX_K = 4096;                 // LDS histograms of the grouping kernels
constexpr uint32_t NO_SLOT = 0xFFFFFFFFu;

__global__ void __launch_bounds__(256) k_group_count(const uint32_t *__restrict__ win, uint32_t n, uint32_t K, uint32_t *counts)
{
    __shared__ uint32_t hist[GROUP_MAX_K];
    for (uint32_t j = threadIdx.x; j < K; j += 256) hist[j] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t j = win[i];
        if (j < K) atomicAdd(&hist[j], 1u);
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < K; j += 256)
        if (hist[j]) atomicAdd(&counts[j], hist[j]);
}

// One workgroup.  starts[j] = first slot of group j (a multiple of 32), batch_j[b] = matrix of batch b or NO_SLOT
// for the unused batches up to `cap`; counts[] is cleared: the scatter uses it as the groups' cursors.
__global__ void __launch_bounds__(256) k_group_plan(uint32_t *counts, uint32_t K, uint32_t *starts, uint32_t *batch_j, uint32_t cap)
{
    __shared__ uint32_t first[GROUP_MAX_K + 1];
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t j = 0; j < K; ++j) {
            first[j] = run;
            run += (counts[j] + 31u) / 32u;
        }
        first[K] = run;
    }
    __syncthreads();
    for (uint32_t j = 0; j < K; ++j)
        for (uint32_t b = first[j] + threadIdx.x; b < first[j + 1]; b += 256) batch_j[b] = j;
    for (uint32_t b = first[K] + threadIdx.x; b < cap; b += 256) batch_j[b] = NO_SLOT;
    for (uint32_t j = threadIdx.x; j < K; j += 256) {
        starts[j] = first[j] * 32u;
        counts[j] = 0u;
    }
}

// slots[starts[j] + position] = sample, positions handed out per workgroup tile (one global atomic per tile and
// group, ranks inside the tile from the LDS histogram).  The order inside a group is whatever the atomics give;
// every sample's result goes back to its own row, so the output does not depend on it.
__global__ void __launch_bounds__(256) k_group_scatter(const uint32_t *__restrict__ win, uint32_t n, uint32_t K,
                                                       const uint32_t *__restrict__ starts, uint32_t *cursor, uint32_t *slots)
{
    constexpr uint32_t PER = 8, TILE = 256 * PER;
    __shared__ uint32_t hist[GROUP_MAX_K], base[GROUP_MAX_K];
    for (uint64_t t0 = (uint64_t)blockIdx.x * TILE; t0 < n; t0 += (uint64_t)gridDim.x * TILE) {
        for (uint32_t j = threadIdx.x; j < K; j += 256) hist[j] = 0;
        __syncthreads();
        uint32_t jv[PER], rank[PER];
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q) {
            const uint64_t i = t0 + q * 256 + threadIdx.x;
            jv[q] = i < n ? win[i] : NO_SLOT;
            rank[q] = jv[q] < K ? atomicAdd(&hist[jv[q]], 1u) : 0u;
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < K; j += 256)
            if (hist[j]) base[j] = starts[j] + atomicAdd(&cursor[j], hist[j]);
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q)
            if (jv[q] < K) slots[base[jv[q]] + rank[q]] = (uint32_t)(t0 + q * 256 + threadIdx.x);
        __syncthreads();
    }
}

struct LocalProjectArgs {
    const float *X;           // n x D, n * D * 4 < 2^32
    uint32_t n, D;
    const float *pieces;      // K matrices of (R/32) * (R/8) pieces of 1 KiB
    const uint32_t *slots;    // 32 per batch: sample or NO_SLOT
    const uint32_t *batch_j;  // matrix of the batch or NO_SLOT
    uint32_t nbatches;
    float *Y;                 // n x D
};

template <int R, int NW = 4>
__global__ void __launch_bounds__(NW * 64) k_project_local(const LocalProjectArgs a)
{
    constexpr int NT = R / 32, NJ = R / 8, NSTEP = NT * NJ;
    constexpr int DEPTH = 6;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int s_col = lane & 31, h = lane >> 5;
    const uint32_t row_bytes = a.D * 4u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)a.X, 0, (int)(a.n * row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)a.Y, 0, (int)(a.n * row_bytes), 0x00020000);
    const bool vec_store = (a.D & 3u) == 0u && ((uintptr_t)a.Y & 15u) == 0u;
    const uint32_t om_off = (uint32_t)lane * 16u;

    for (uint32_t b = blockIdx.x * NW + wave; b < a.nbatches; b += gridDim.x * NW) {
        const uint32_t mat = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.batch_j[b]);
        if (mat == NO_SLOT) continue;
        const uint32_t id = a.slots[(size_t)b * 32 + s_col];
        const bool valid = id < a.n;
        // an empty slot computes on row 0: column s of the product depends on sample s alone, and it is not stored
        const uint32_t xoff = (valid ? id : 0u) * row_bytes + (uint32_t)h * 16u;
        f32x4 x[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) x[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff + j * 32, 0, 0));
        const __amdgpu_buffer_rsrc_t rom =
            __builtin_amdgcn_make_buffer_rsrc((void *)(a.pieces + (size_t)mat * NSTEP * 256), 0, NSTEP * 1024, 0x00020000);
        auto piece = [&](int i) {
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rom, om_off, i * 1024, 0));
        };
        f32x4 ring[DEPTH];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) ring[i] = piece(i);
        // what lies behind a row's D values is the next row or the end of the buffer
#pragma unroll
        for (int j = 0; j < NJ; ++j) x[j] = mask_chunk(x[j], (uint32_t)(2 * j + h), a.D);
        const uint32_t yoff = (valid ? id : 0u) * row_bytes;
        static_for<0, NT>([&](auto tt) {
            constexpr int t = decltype(tt)::value;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            static_for<0, NJ>([&](auto jj) {
                constexpr int j = decltype(jj)::value, i = t * NJ + j;
                const f32x4 av = ring[i % DEPTH];
                if constexpr (i + DEPTH < NSTEP) ring[i % DEPTH] = piece(i + DEPTH);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], x[j][e], acc, 0, 0, 0);
            });
            if (valid) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t col = 32u * t + 8u * g + 4u * h;
                    if (vec_store && col + 4u <= a.D) {
                        const f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, yoff + col * 4u, 0, 0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float val = acc[4 * g + e];
                            if (col + e < a.D)
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, val), ry, yoff + (col + e) * 4u, 0, 0);
                        }
                    }
                }
            }
        });
    }
}

struct LocalProjectModel {
    uint32_t K = 0, D = 0, R = 0;
    float *pieces = nullptr;
    void (*launch)(const LocalProjectArgs &, unsigned, hipStream_t) = nullptr;
};

namespace {
template <int R>
void launch_project_local_t(const LocalProjectArgs &a, unsigned grid, hipStream_t st)
{
    hipLaunchKernelGGL((k_project_local<R>), dim3(grid), dim3(256), 0, st, a);
}

template <int R>
bool bind_local_project(LocalProjectModel &m, std::vector<float> &pieces, const std::vector<float> &Oms, uint32_t M)
{
    if (m.D > (uint32_t)R || (R > 32 && m.D <= (uint32_t)R / 2)) return false;
    constexpr int NT = R / 32, NJ = R / 8;
    const uint32_t D = m.D;
    const size_t per = (size_t)NT * NJ * 256;
    pieces.assign((size_t)m.K * per, 0.0f);
    const unsigned nthreads = std::max(1u, std::min(16u, std::min(m.K, std::thread::hardware_concurrency())));
    // every allocation happens here, before a thread exists (an exception inside a thread ends the process); a worker
    // that cannot be started has its share done by the caller
    std::vector<std::vector<double>> lams(nthreads, std::vector<double>((size_t)D * D));
    std::vector<std::thread> pool;
    pool.reserve(nthreads);
    auto work = [&](unsigned w) {
        {
            std::vector<double> &lam = lams[w];
            for (uint32_t k = w; k < m.K; k += nthreads) {
                const float *om = &Oms[(size_t)k * M * D];
                std::fill(lam.begin(), lam.end(), 0.0);
                for (uint32_t r = 0; r < M; ++r)
                    for (uint32_t i = 0; i < D; ++i) {
                        const double oi = om[(size_t)r * D + i];
                        for (uint32_t j = i; j < D; ++j) lam[(size_t)i * D + j] += oi * (double)om[(size_t)r * D + j];
                    }
                // piece (t, j), lane (row, h) = Lambda[32 t + row][4 (2 j + h) .. + 3]
                float *out = &pieces[(size_t)k * per];
                for (int t = 0; t < NT; ++t)
                    for (int j = 0; j < NJ; ++j)
                        for (int lane = 0; lane < 64; ++lane) {
                            const uint32_t row = 32u * t + (lane & 31), col = 4u * (2 * j + (lane >> 5));
                            for (uint32_t e = 0; e < 4; ++e)
                                if (row < D && col + e < D) {
                                    const uint32_t lo = std::min(row, col + e), hi = std::max(row, col + e);
                                    out[(((size_t)t * NJ + j) * 64 + lane) * 4 + e] = (float)lam[(size_t)lo * D + hi];
                                }
                        }
            }
        }
    };
    unsigned started = 1;                                  // share 0 is the caller's
    for (; started < nthreads; ++started) {
        try {
            pool.emplace_back(work, started);
        } catch (...) {
            break;
        }
    }
    work(0);
    for (unsigned w = started; w < nthreads; ++w) work(w);
    for (std::thread &t : pool) t.join();
    m.R = R;
    m.launch = &launch_project_local_t<R>;
    return true;
}
}  // namespace

LocalProjectModel *local_project_build(const std::vector<float> &Oms, uint32_t K, uint32_t M, uint32_t D, hipError_t &herr)
{
    herr = hipSuccess;
    if (K == 0 || K > GROUP_MAX_K || D > 256 || (size_t)K * 256 * 256 * 4 > ((size_t)1 << 30)) return nullptr;
    LocalProjectModel m;
    m.K = K;
    m.D = D;
    std::vector<float> pieces;
    if (!(bind_local_project<32>(m, pieces, Oms, M) || bind_local_project<64>(m, pieces, Oms, M) ||
          bind_local_project<128>(m, pieces, Oms, M) || bind_local_project<256>(m, pieces, Oms, M)))
        return nullptr;
    LocalProjectModel *out = new LocalProjectModel(m);
    herr = to_device(&out->pieces, pieces);
    if (herr != hipSuccess) {
        local_project_free(out);
        return nullptr;
    }
    return out;
}

void local_project_free(LocalProjectModel *m)
{
    if (!m) return;
    (void)hipFree(m->pieces);
    delete m;
}

// rows one call takes (32-bit byte offsets into X and Y) and the u32 words of scratch it needs for them
uint64_t local_project_max_rows(const LocalProjectModel *m) { return 0xFFFFFFFFull / (m->D * 4ull); }

size_t local_project_scratch_words(const LocalProjectModel *m, uint64_t n)
{
    const size_t cap = (size_t)((n + 31) / 32) + m->K;
    return cap * 32 + cap + 2 * (size_t)m->K;
}

// dwin: the n winners (already computed on `s`); scratch: local_project_scratch_words(m, n) words
hipError_t local_project_run(const LocalProjectModel *m, const float *dX, uint64_t n, const uint32_t *dwin, uint32_t *scratch,
                             float *dY, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    if (n > local_project_max_rows(m)) return hipErrorInvalidValue;
    const uint32_t cap = (uint32_t)((n + 31) / 32) + m->K;
    uint32_t *slots = scratch, *batch_j = slots + (size_t)cap * 32, *counts = batch_j + cap, *starts = counts + m->K;
    hipError_t e = hipMemsetAsync(slots, 0xFF, (size_t)cap * 32 * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(counts, 0, m->K * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)std::min<uint64_t>((n + 2047) / 2048, 1024);
    hipLaunchKernelGGL(k_group_count, dim3(grid), dim3(256), 0, s, dwin, (uint32_t)n, m->K, counts);
    hipLaunchKernelGGL(k_group_plan, dim3(1), dim3(256), 0, s, counts, m->K, starts, batch_j, cap);
    hipLaunchKernelGGL(k_group_scatter, dim3(grid), dim3(256), 0, s, dwin, (uint32_t)n, m->K, starts, counts, slots);
    LocalProjectArgs a;
    a.X = dX; a.n = (uint32_t)n; a.D = m->D;
    a.pieces = m->pieces; a.slots = slots; a.batch_j = batch_j; a.nbatches = cap;
    a.Y = dY;
    m->launch(a, (unsigned)std::min<uint32_t>((cap + 3) / 4, 2048), s);
    return hipGetLastError();
}

}  // namespace vq
