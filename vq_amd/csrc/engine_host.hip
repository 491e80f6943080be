// Model builder of the MFMA kernels (host side): tile images in the geometry of engine.h, kernel selection
// by shape, and the launch of one pass.  The kernels live in kernels_engine.hip (streamed weights),
// kernels_resident.hip (Euclidean, prototypes in LDS) and kernels_omega.hip (Omega models, prototypes in LDS).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <cstring>

#include "mfma_device.h"
#include "mfma_images.h"

namespace vq {

constexpr uint32_t CROWD_SLOTS = 4;    // finish-pass buffer sets (launches of one model in flight at once)

struct EngineModel {
    int mode = MODE_EUCLID;
    uint32_t K = 0, D = 0, M = 0, R1 = 0, R2 = 0, tiles_per_proto = 0;
    bool has_cls = false;
    float *img1 = nullptr, *img2 = nullptr, *plain = nullptr;
    uint32_t *cls = nullptr;
    // Batch counters, one per launch in flight: launches of one model on different streams (the two staging
    // streams of the host path) may overlap at their edges, and a launch that zeroed the counter of the
    // one still running would hand its batches to it.  Slots are used round-robin.
    uint32_t *queue = nullptr;
    mutable uint32_t next_slot = 0;
    float *mu = nullptr;                                    // centre subtracted from every sample (R1 floats), null: none
    // LDS-resident kernels that rank on the bf16 cores: bytes of LDS the kernel fills already, reduction length and image
    // (1 / 2) of its search stage; what is left of LDS takes the bf16 pieces of the first nbf tiles (imgbf)
    size_t lds_used = 0;
    int search_R = 0, search_img = 1;
    float *imgbf = nullptr;
    uint32_t nbf = 0;
    bool bfimg = false;               // the search image holds pieces only (every tile); exact rows from `plain`
    bool bfimg_ok = false;            // the kernel has that form (k_resident_pipe un-staged, k_omega_pipe)
    uint32_t ngroups1 = 0, ngroups2 = 0, ldp = 0;
    const char *name = "mfma";
    LaunchFn launch = nullptr;
    int samples_per_batch = 0;
    float ss_limit = 0.0f, wmax2 = 0.0f, mscale = 6.103515625e-05f;
    bool bf = false;                  // k_resident_pipe: screening products on the bf16 matrix cores (mscale is then that path's)
    bool pj = false;                  // Omega kernels: the projection on the bf16 matrix cores too (img1 = three bf16 pieces per value)
    BuildOptions opt;                 // what the caller allows on the bf16 matrix cores (vqhip_set_option)
    // The finish pass over the samples the first pass marks instead of scanning for them inside the wave (crowd_mark /
    // HitKeys, mfma_device.h): the kernel's FINISH variant, or null (such samples are then settled in their wave).  Its
    // buffers, one set per launch in flight like the batch counters: [3 mask words per tile of 32 samples][count][ids],
    // sized for the largest launch so far.
    LaunchFn finish = nullptr;
    float *img_f32 = nullptr;          // bfimg models with a finish pass: the f32 tile image all the same -- the finish pass
                                       // keeps IT in LDS and measures its hits from there (splitting the tiles as it goes)
    mutable uint32_t *crowd[CROWD_SLOTS] = {};
    mutable uint64_t crowd_rows[CROWD_SLOTS] = {};
    mutable uint32_t next_crowd = 0;
};

// Marked samples -> a packed list of (flags | id), any order: one thread per mask triple, a block-wide prefix sum of the
// counts, one atomic per block to claim the block's range of the list.
__global__ void __launch_bounds__(256) k_crowd_compact(const uint32_t *mask, uint32_t ntiles, uint32_t *list, uint32_t *count)
{
    __shared__ uint32_t wave_total[4], block_base;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    uint32_t mp = 0u, mn = 0u, ma = 0u;
    if (t < ntiles) {
        mp = mask[3 * (size_t)t];
        mn = mask[3 * (size_t)t + 1];
        ma = mask[3 * (size_t)t + 2];
    }
    uint32_t any = mp | mn | ma;
    const uint32_t c = (uint32_t)__popc(any);
    uint32_t incl = c;                                  // inclusive scan inside the wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
        block_base = total ? atomicAdd(count, total) : 0u;
    }
    __syncthreads();
    uint32_t pos = block_base + incl - c;
    for (int w = 0; w < wave; ++w) pos += wave_total[w];
    while (any) {
        const uint32_t sbit = (uint32_t)__ffs((int)any) - 1u;
        any &= any - 1u;
        list[pos++] = (t * 32u + sbit) | ((mp >> sbit) & 1u ? CROWD_P : 0u) | ((mn >> sbit) & 1u ? CROWD_N : 0u) | ((ma >> sbit) & 1u ? CROWD_ALL : 0u);
    }
}


namespace {

// tiles of 32 prototypes a resident kernel searches (the images are padded further, to whole stage groups)
uint32_t res_tiles(uint32_t K) { return (K + 31) / 32; }

int pad_pow2(uint32_t v, int lo)
{
    int r = lo;
    while ((uint32_t)r < v) r <<= 1;
    return r;
}

// Prototypes whose search-space row repeats that of an earlier prototype (-0 counted as +0) of the same class (any
// earlier one when the model has no classes): their distances equal the earlier one's bit for bit, so under strict '<'
// in index order (helpers.rs:70, :113, :122) they never win a search the earlier one is eligible for -- and it is
// eligible for every search they are.  They are left out of the SCREENING (their score column is PAD_SCORE; the exact
// scan still walks them), so a model with many copies of a prototype does not look crowded to the third-key guard
// (finish_keys, mfma_device.h).
std::vector<bool> shadowed_rows(const std::vector<float> &Q, uint32_t K, uint32_t L, const std::vector<uint32_t> &cls)
{
    std::vector<float> canon(Q.begin(), Q.begin() + (size_t)K * L);
    for (float &v : canon) v += 0.0f;
    auto row_cmp = [&](uint32_t x, uint32_t y) { return std::memcmp(&canon[(size_t)x * L], &canon[(size_t)y * L], L * sizeof(float)); };
    std::vector<uint32_t> order(K);
    for (uint32_t k = 0; k < K; ++k) order[k] = k;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        const int c = row_cmp(x, y);
        return c < 0 || (c == 0 && x < y);
    });
    std::vector<bool> drop(K, false);
    for (uint32_t i = 0; i < K;) {
        uint32_t j = i + 1;
        while (j < K && row_cmp(order[i], order[j]) == 0) ++j;
        for (uint32_t t = i + 1; t < j; ++t)             // order[i..j) ascending by index
            for (uint32_t e = i; e < t && !drop[order[t]]; ++e)
                drop[order[t]] = cls.empty() || cls[order[e]] == cls[order[t]];
        i = j;
    }
    return drop;
}

// The centre the kernels subtract from every sample before anything else: the mean prototype, when the prototypes
// sit off the origin by more than their own spread.  The screening error scales with ||x||^2 + max ||w||^2, the
// distances do not: for data around (100, 100, ...) with unit spread the uncentred keys would resolve nothing.
// x - mu is one f32 subtraction per value (exact by Sterbenz's lemma exactly where it matters, x and mu within a
// factor of two of each other, and for coarse-grained data: mu is rounded to 8 significant bits); the prototypes are shifted by the same f32 operation here, so a sample that equals a
// prototype bit for bit still has distance 0.  Everything downstream (images, norms, projections, the exact
// re-measurement) lives in the shifted coordinates.  Returns false (mu left empty) when the shift would not halve the
// mean squared norm.
bool centre_prototypes(const std::vector<float> &W, uint32_t K, uint32_t D, std::vector<float> &mu, std::vector<float> &Wc)
{
    if (getenv("VQHIP_NO_CENTRE")) return false;
    std::vector<double> mean(D, 0.0);
    double sumsq = 0.0;
    for (uint32_t k = 0; k < K; ++k)
        for (uint32_t d = 0; d < D; ++d) {
            const double v = W[(size_t)k * D + d];
            mean[d] += v;
            sumsq += v * v;
        }
    double mu2 = 0.0;
    std::vector<float> m(D);
    for (uint32_t d = 0; d < D; ++d) {
        // 8 significant bits: x - mu is then EXACT for integer, binary and other coarse-grained data as well (the shift
        // changes nothing there, exact ties stay exact ties), and where x and mu are within a factor of two anyway
        int ex = 0;
        const double fr = std::frexp(mean[d] / K, &ex);
        m[d] = (float)std::ldexp(std::nearbyint(std::ldexp(fr, 8)), ex - 8);
        mu2 += (double)m[d] * (double)m[d];
    }
    if (!(mu2 * K > 0.5 * sumsq) || !std::isfinite(mu2)) return false;   // also false for NaN / INF prototypes
    mu = std::move(m);
    Wc.resize((size_t)K * D);
    for (uint32_t k = 0; k < K; ++k)
        for (uint32_t d = 0; d < D; ++d) {
            volatile float t = W[(size_t)k * D + d] - mu[d];             // one f32 subtraction, as on the device
            Wc[(size_t)k * D + d] = t;
        }
    return true;
}

int variant_env()
{
    const char *v = getenv("VQHIP_VARIANT");     // experiments only: picks among co-compiled tilings
    return v ? atoi(v) : 0;
}

// The LDS-resident pipelined kernels rank the prototypes on the bf16 matrix cores (search_tiles_bf, mfma_device.h):
// operands split into two bf16 pieces, three piece products.  Its margin: twice (3 * 2^-16 for the dropped parts + the
// f32 accumulation of 3 R products + the key's 4 position bits), relative to ||x||^2 + max ||w||^2 like the f32 form's.
// bf16 pieces of the first tiles of a search image, in the layout search_tiles_bf reads (mfma_device.h BF_TILE_BYTES):
// rows at the f32 tile's pitch; per reduction step q and lane half h the hi pieces of the lane's 8 values (16 bytes), then
// their lo pieces.  Round to nearest even like v_cvt_pk_bf16_f32; lo = bf16(v - hi) with the difference exact in f32.
template <int R>
std::vector<float> build_bf_tiles(const std::vector<float> &img, uint32_t nbf)
{
    using Gm = Geom<R>;
    std::vector<float> out((size_t)nbf * Gm::W_FLOATS, 0.0f);
    for (uint32_t t = 0; t < nbf; ++t)
        for (int r = 0; r < TILE_ROWS; ++r) {
            const float *row = &img[(size_t)t * Gm::TILE_FLOATS + (size_t)r * Gm::ROW_FLOATS];
            uint16_t *dst = reinterpret_cast<uint16_t *>(&out[(size_t)t * Gm::W_FLOATS + (size_t)r * Gm::ROW_FLOATS]);
            for (int q = 0; q < R / 16; ++q)
                for (int h = 0; h < 2; ++h)
                    for (int e = 0; e < 8; ++e) {
                        const float v = row[16 * q + 4 * h + (e < 4 ? e : e + 4)];     // chunks 4q + h and 4q + 2 + h
                        const uint16_t hi = to_bf16(v);
                        const uint32_t hb = (uint32_t)hi << 16;
                        float hf;
                        std::memcpy(&hf, &hb, 4);
                        volatile float diff = v - hf;
                        dst[32 * q + 16 * h + e] = hi;
                        dst[32 * q + 16 * h + 8 + e] = to_bf16(diff);
                    }
        }
    return out;
}

// The search image with the f32 values of every tile replaced by their pieces (tails kept): the BFIMG kernels' image.
template <int R>
void pieces_image(std::vector<float> &img, uint32_t ntiles)
{
    using Gm = Geom<R>;
    const std::vector<float> pieces = build_bf_tiles<R>(img, ntiles);
    for (uint32_t t = 0; t < ntiles; ++t)
        std::memcpy(&img[(size_t)t * Gm::TILE_FLOATS], &pieces[(size_t)t * Gm::W_FLOATS], (size_t)Gm::W_FLOATS * 4);
}

// as many tiles as LDS has room for behind what the kernel keeps there already
template <int R>
std::vector<float> bf_tiles_for(const EngineModel &m, const std::vector<float> &img, uint32_t &nbf)
{
    const size_t room = (size_t)RES_BYTES > m.lds_used ? (size_t)RES_BYTES - m.lds_used : 0;
    nbf = (uint32_t)std::min<size_t>(res_tiles(m.K), room / ((size_t)Geom<R>::W_FLOATS * 4));
    if (getenv("VQHIP_NO_BF_TILES")) nbf = 0;
    return build_bf_tiles<R>(img, nbf);
}

void set_bf(EngineModel &m, int R)
{
    if (!m.opt.bf_rank) return;
    m.bf = true;
    // (the second form: the worst case of the f32 accumulation of 3 R products, which the first exceeds up to R = 128)
    m.mscale = (float)(6.0 / 65536.0 + std::max(4.0 * (8.0 * std::sqrt(3.0 * R) * 5.9604644775390625e-08 + 1.9073486328125e-06),
                                                2.0 * 3.0 * R * 5.9604644775390625e-08 + 3.814697265625e-06));
}

template <int R1>
bool bind_euclid(EngineModel &m, std::vector<float> &img, const std::vector<float> &W, const std::vector<float> &norms,
                 const std::vector<uint32_t> &cls)
{
    if (m.R1 != (uint32_t)R1) return false;
    img = build_images<R1>(W.data(), m.K, m.D, m.D, -2.0f, norms.data(), cls.empty() ? nullptr : cls.data(), PAD_SCORE,
                           m.ngroups1);
    m.launch = streamed_launcher(MODE_EUCLID, R1, R1, &m.samples_per_batch);
    if ((size_t)res_tiles(m.K) * Geom<R1>::TILE_BYTES <= (size_t)RES_BYTES && !getenv("VQHIP_NO_RESIDENT")) {
        const bool pipe_ok = !getenv("VQHIP_NO_PIPE");
        if (R1 < 256) {
            m.launch = resident_launcher(R1, &m.samples_per_batch);
            m.name = "mfma_euclid_lds";
        }
        const bool diag = variant_env() == 31 || variant_env() == 33 || variant_env() == 35;
        if (m.D == (uint32_t)R1 && pipe_ok && resident_pipe_launcher(R1, diag)) {   // unpadded rows: the pipelined kernel
            m.samples_per_batch = 32;
            m.launch = resident_pipe_launcher(R1, diag);
            m.name = "mfma_euclid_lds_pipe";
            // few prototypes: the search is short, the memory pipe binds, rows are staged through LDS
            size_t staging = 0;
            const LaunchFn staged = resident_staged_launcher(R1, &staging);
            if (staged && res_tiles(m.K) <= 2 && (size_t)res_tiles(m.K) * Geom<R1>::TILE_BYTES + staging <= (size_t)RES_BYTES &&
                !diag && !getenv("VQHIP_NO_STAGED")) {
                m.launch = staged;
                m.name = "mfma_euclid_lds_staged";
            }
        } else if (R1 >= 64 && m.D < (uint32_t)R1 && pipe_ok && !getenv("VQHIP_NO_CUT")) {
            // rows shorter than R1: the pipelined kernel cuts its fetches at D.  Measured against k_resident (two
            // 32-sample tiles per wave, no prefetch): +15 % at D=126, K=256, +35 % at D=100, K=37, equal at D=30,
            // slower at D=10 -- rows of up to 32 values stay on k_resident.
            m.samples_per_batch = 32;
            m.launch = resident_cut_launcher(R1);
            m.name = "mfma_euclid_lds_pipe_cut";
        }
    }
    // k_engine and the pipelined resident kernels rank on the bf16 matrix cores (k_resident, padded rows, does not)
    const bool plain_diag = variant_env() == 31 || variant_env() == 32;
    if (R1 <= 128 && std::strcmp(m.name, "mfma_euclid_lds") != 0 && !plain_diag) set_bf(m, R1);
    if (m.bf && std::strcmp(m.name, "mfma_euclid") == 0 && variant_env() == 0 && !getenv("VQHIP_NO_FINISH"))
        m.finish = streamed_finish_launcher(MODE_EUCLID, R1, R1);
    if (m.bf && std::strncmp(m.name, "mfma_euclid_lds_", 16) == 0) {      // pipe / staged / cut: the image is resident
        size_t staging = 0;
        if (std::strcmp(m.name, "mfma_euclid_lds_staged") == 0) (void)resident_staged_launcher(R1, &staging);
        else m.bfimg_ok = variant_env() != 31 && variant_env() != 33;   // the diagnostic builds keep the f32 image
        m.lds_used = (size_t)res_tiles(m.K) * Geom<R1>::TILE_BYTES + staging;
        m.search_R = R1;
        m.search_img = 1;
        if (variant_env() == 0 && !getenv("VQHIP_NO_FINISH")) m.finish = resident_finish_launcher(R1);
    }
    return true;
}

// At most 8 prototypes over rows of at most 256 values: k_direct (kernels_direct.hip), every distance in the direct
// form on the vector ALUs.  img1 = KP rows of R floats (the negated prototypes: no centring, no scaling), KP class ids.
bool bind_direct(EngineModel &m, std::vector<float> &img, const std::vector<float> &W, const std::vector<uint32_t> &cls)
{
    // measured against the matrix-core kernels (tools/shape_bench.py, D = 32 .. 256): 1.1 to 1.65 times their rate up to 8
    // prototypes, level at 12, behind at 16
    if (m.D > 256 || m.K > 8 || getenv("VQHIP_NO_DIRECT")) return false;
    const int R = pad_pow2(m.D, 32), KP = m.K <= 4 ? 4 : 8;
    const LaunchFn fn = direct_launcher(R, KP, &m.samples_per_batch);
    if (!fn) return false;
    img.assign((size_t)KP * R + KP, 0.0f);
    for (uint32_t k = 0; k < m.K; ++k)
        for (uint32_t d = 0; d < m.D; ++d) img[(size_t)k * R + d] = -W[(size_t)k * m.D + d];     // x + (-w): one packed add
    for (uint32_t k = 0; k < (uint32_t)KP; ++k) {
        const uint32_t c = k < m.K && !cls.empty() ? cls[k] : 0xFFFFFFFFu;
        std::memcpy(&img[(size_t)KP * R + k], &c, 4);
    }
    m.ngroups1 = 1;
    m.launch = fn;
    m.name = "valu_euclid_direct";
    return true;
}

// Rows above 256 values: k_long.  img1 = A-fragment pieces of -2 W, one 1 KiB piece per (group of 8 tiles when there
// are more than 8, chunk of 128 or 64 values, tile of 32 prototypes, step): lane (row, h) of step j holds values
// 4 (2 j + h) .. + 3 of the chunk; img2 = the tiles' tails (||w_k||^2 / class ids); plain rows padded to whole chunks.
bool bind_long(EngineModel &m, std::vector<float> &img1, std::vector<float> &img2, const std::vector<float> &W,
               const std::vector<float> &norms, const std::vector<uint32_t> &cls)
{
    if (m.D <= 256 || (uint64_t)(m.K + 255) * (m.D + 127) * 4 >= (1ull << 31)) return false;   // pieces: 32-bit byte offsets
    const uint32_t rt = res_tiles(m.K), tiles = rt <= 2 ? rt : (rt <= 4 ? 4 : 8), groups = (rt + tiles - 1) / tiles;
    const uint32_t steps = tiles <= 2 ? 16 : 8, chunk = steps * 8;     // as in the kernel: 128-value chunks up to 2 tiles, else 64
    const uint32_t NC = (m.D + chunk - 1) / chunk;
    const LaunchFn fn = long_launcher((int)tiles);
    if (!fn) return false;
    img1.assign((size_t)groups * NC * tiles * steps * 64 * 4, 0.0f);
    for (uint32_t g = 0; g < groups; ++g)
        for (uint32_t c = 0; c < NC; ++c)
            for (uint32_t t = 0; t < tiles; ++t)
                for (uint32_t j = 0; j < steps; ++j)
                    for (uint32_t lane = 0; lane < 64; ++lane) {
                        const uint32_t k = 32 * (g * tiles + t) + (lane & 31), d0 = chunk * c + 4 * (2 * j + (lane >> 5));
                        float *dst = &img1[(((((size_t)g * NC + c) * tiles + t) * steps + j) * 64 + lane) * 4];
                        for (uint32_t e = 0; e < 4; ++e)
                            if (k < m.K && d0 + e < m.D) dst[e] = -2.0f * W[(size_t)k * m.D + d0 + e];
                    }
    m.ngroups1 = groups;
    img2.assign((size_t)groups * tiles * TAIL_FLOATS, 0.0f);
    for (uint32_t t = 0; t < groups * tiles; ++t)
        for (uint32_t r = 0; r < 32; ++r) {
            const uint32_t k = 32 * t + r;
            img2[(size_t)t * TAIL_FLOATS + r] = k < m.K ? norms[k] : PAD_SCORE;
            const uint32_t c = k < m.K && !cls.empty() ? cls[k] : 0xFFFFFFFFu;
            std::memcpy(&img2[(size_t)t * TAIL_FLOATS + 32 + r], &c, 4);
        }
    m.ldp = NC * chunk;
    m.launch = fn;
    m.samples_per_batch = 32;
    m.name = "mfma_euclid_long";
    return true;
}

// Global Omega (at most R2 <= 128 rows) over rows above 256 values: k_omega_long.  img1 = A-fragment pieces of
// Omega per (chunk, tile of 32 Omega rows, step) as in bind_long, img2 = tile image of the projected prototypes.
template <int R2>
bool bind_omega_long(EngineModel &m, std::vector<float> &img1, std::vector<float> &img2, const std::vector<float> &Om,
                     const std::vector<float> &P, const std::vector<float> &norms, const std::vector<uint32_t> &cls)
{
    if (m.D <= 256 || (uint32_t)pad_pow2(m.M, 32) != (uint32_t)R2) return false;
    if ((size_t)res_tiles(m.K) * Geom<R2>::TILE_BYTES > (size_t)RES_BYTES) return false;
    const LaunchFn fn = omega_long_launcher(R2);
    if (!fn) return false;
    constexpr uint32_t tiles = R2 / 32;
    const uint32_t NC = (m.D + 63) / 64;                              // this kernel's chunks hold 64 values: 8 steps
    img1.assign((size_t)NC * tiles * 8 * 64 * 4, 0.0f);
    for (uint32_t c = 0; c < NC; ++c)
        for (uint32_t t = 0; t < tiles; ++t)
            for (uint32_t j = 0; j < 8; ++j)
                for (uint32_t lane = 0; lane < 64; ++lane) {
                    const uint32_t r = 32 * t + (lane & 31), d0 = 64 * c + 4 * (2 * j + (lane >> 5));
                    float *dst = &img1[((((size_t)c * tiles + t) * 8 + j) * 64 + lane) * 4];
                    for (uint32_t e = 0; e < 4; ++e)
                        if (r < m.M && d0 + e < m.D) dst[e] = Om[(size_t)r * m.D + d0 + e];
                }
    uint32_t ngroups2 = 0;
    img2 = build_images<R2>(P.data(), m.K, m.M, m.M, -2.0f, norms.data(), cls.empty() ? nullptr : cls.data(), PAD_SCORE, ngroups2);
    m.ngroups2 = ngroups2;
    m.R2 = R2;
    m.ldp = R2;
    m.launch = fn;
    m.samples_per_batch = 32;
    m.name = "mfma_omega_long";
    return true;
}

template <int R1, int R2>
bool bind_omega(EngineModel &m, std::vector<float> &img1, std::vector<float> &img2, const std::vector<float> &Om,
                const std::vector<float> &P, const std::vector<float> &norms, const std::vector<uint32_t> &cls)
{
    if (m.R1 != (uint32_t)R1 || m.R2 > (uint32_t)R2) return false;     // fewer Omega rows than R2: zero padded
    m.R2 = R2;
    m.ldp = R2;
    img1 = build_images<R1>(Om.data(), m.M, m.D, m.D, 1.0f, nullptr, nullptr, 0.0f, m.ngroups1, R2 >= 32 ? R2 : 32);
    img2 = build_images<R2>(P.data(), m.K, m.M, m.M, -2.0f, norms.data(), cls.empty() ? nullptr : cls.data(), PAD_SCORE,
                            m.ngroups2);
    m.launch = streamed_launcher(MODE_OMEGA, R1, R2, &m.samples_per_batch);
    if (R2 <= 128) set_bf(m, R2);
    return true;
}

// Omega with at most 16 rows, unpadded sample rows, both images resident in LDS: k_lowrank_pipe
template <int R1>
bool bind_lowrank(EngineModel &m, std::vector<float> &img1, std::vector<float> &img2, const std::vector<float> &Om,
                  const std::vector<float> &P, const std::vector<float> &norms, const std::vector<uint32_t> &cls)
{
    if (m.R1 != (uint32_t)R1 || m.M > 16 || getenv("VQHIP_NO_RESIDENT")) return false;    // D < R1: the kernel's CUT variant
    constexpr int ROWF = R1 + 4;
    // p_k with chunks 1 and 2 swapped (see the kernel), zero padded to 16
    std::vector<float> Pp((size_t)m.K * 16, 0.0f);
    static const int actual_chunk[4] = {0, 2, 1, 3};
    for (uint32_t k = 0; k < m.K; ++k)
        for (int ci = 0; ci < 4; ++ci)
            for (int e = 0; e < 4; ++e) {
                const uint32_t d = 4u * actual_chunk[ci] + e;
                if (d < m.M) Pp[(size_t)k * 16 + 4 * ci + e] = P[(size_t)k * m.M + d];
            }
    uint32_t ngroups2 = 0;
    std::vector<float> pimg = build_images<16>(Pp.data(), m.K, 16, 16, -2.0f, norms.data(), cls.empty() ? nullptr : cls.data(),
                                               PAD_SCORE, ngroups2);
    if ((size_t)res_tiles(m.K) * Geom<16>::TILE_BYTES + (size_t)16 * ROWF * sizeof(float) > (size_t)RES_BYTES) return false;
    img2 = std::move(pimg);
    m.ngroups2 = ngroups2;
    img1.assign((size_t)16 * ROWF, 0.0f);                              // Omega rows, row stride R1 + 4
    for (uint32_t r = 0; r < m.M; ++r) std::memcpy(&img1[(size_t)r * ROWF], &Om[(size_t)r * m.D], m.D * sizeof(float));
    m.ngroups1 = 1;
    m.R1 = R1;
    m.R2 = 16;
    m.launch = lowrank_launcher(R1);
    m.samples_per_batch = 32;
    m.name = "mfma_omega_lowrank_lds";
    set_bf(m, 16);
    m.lds_used = (size_t)res_tiles(m.K) * Geom<16>::TILE_BYTES + (size_t)16 * ROWF * sizeof(float);
    m.search_R = 16;
    m.search_img = 2;
    return true;
}

// Omega padded to R x R with D == R, projected prototypes resident in LDS, Omega from L2: k_omega_pipe
template <int R>
bool bind_omega_pipe(EngineModel &m, std::vector<float> &img1, std::vector<float> &img2, const std::vector<float> &Om,
                     const std::vector<float> &P, const std::vector<float> &norms, const std::vector<uint32_t> &cls)
{
    if (m.R1 != (uint32_t)R || m.R2 > (uint32_t)R || getenv("VQHIP_NO_RESIDENT")) return false;   // Omega zero padded to R x R
    if ((size_t)res_tiles(m.K) * Geom<R>::TILE_BYTES > (size_t)RES_BYTES) return false;
    m.R2 = R;
    m.ldp = R;
    uint32_t ngroups2 = 0;
    std::vector<float> pimg = build_images<R>(P.data(), m.K, m.M, m.M, -2.0f, norms.data(), cls.empty() ? nullptr : cls.data(),
                                              PAD_SCORE, ngroups2);
    img2 = std::move(pimg);
    m.ngroups2 = ngroups2;
    // Omega as A-fragment pieces: piece (t, j), lane (row, h) = Omega[32 t + row][4 (2 j + h) .. + 3]
    constexpr int NT = R / 32, NJ = R / 8;
    img1.assign((size_t)NT * NJ * 64 * 4, 0.0f);
    for (int t = 0; t < NT; ++t)
        for (int j = 0; j < NJ; ++j)
            for (int lane = 0; lane < 64; ++lane) {
                const uint32_t row = 32u * t + (lane & 31), col = 4u * (2 * j + (lane >> 5));
                for (uint32_t e = 0; e < 4; ++e)
                    if (row < m.M && col + e < m.D) img1[(((size_t)t * NJ + j) * 64 + lane) * 4 + e] = Om[(size_t)row * m.D + col + e];
            }
    m.ngroups1 = 1;
    m.launch = omega_pipe_launcher(R);
    m.samples_per_batch = 32;
    m.name = "mfma_omega_lds";
    set_bf(m, R);
    if (m.bf && m.opt.bf_proj) {
        // the projection on the bf16 matrix cores: piece (step q of 16 reduction values, tile t) at q * NT + t holds, for
        // lane (row, h), the three bf16 pieces of Omega[32 t + row][16 q + 4 h .. + 3] and [16 q + 8 + 4 h .. + 3]: 1 KiB of
        // first pieces (16 bytes per lane), 1 KiB of second, 1 KiB of third
        constexpr int NQ = R / 16;
        std::vector<uint16_t> pcs((size_t)NQ * NT * 3 * 64 * 8, 0);
        for (int q = 0; q < NQ; ++q)
            for (int t = 0; t < NT; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const uint32_t row = 32u * t + (lane & 31), col = 16u * q + 4u * (lane >> 5) + (e < 4 ? e : e + 4);
                        if (row >= m.M || col >= m.D) continue;
                        uint16_t p[3];
                        split3_host(Om[(size_t)row * m.D + col], p);
                        for (int c = 0; c < 3; ++c) pcs[((((size_t)q * NT + t) * 3 + c) * 64 + lane) * 8 + e] = p[c];
                    }
        img1.assign(pcs.size() / 2, 0.0f);
        std::memcpy(img1.data(), pcs.data(), pcs.size() * 2);
        m.pj = true;
    }
    m.lds_used = (size_t)res_tiles(m.K) * Geom<R>::TILE_BYTES;
    m.search_R = R;
    m.search_img = 2;
    m.bfimg_ok = true;
    if (m.bf && variant_env() == 0 && !getenv("VQHIP_NO_FINISH")) m.finish = omega_finish_launcher(R);
    return true;
}

// LGMLVQ over rows above 256 values: k_local_long.  The tiles of all prototypes stacked (every Omega_j padded to
// whole tiles), img1 = their A-fragment pieces per (group of tiles, chunk of 64 values, tile, step) as in bind_long,
// img2 = the tiles' init columns -(Omega_j w_j).
bool bind_local_long(EngineModel &m, std::vector<float> &img1, std::vector<float> &img2, const std::vector<float> &Oms,
                     const std::vector<float> &negp)
{
    if (m.D <= 256) return false;
    const uint32_t tpp = (m.M + 31) / 32, total = m.K * tpp;
    const uint32_t tiles = total <= 2 ? total : 4, groups = (total + tiles - 1) / tiles;   // (8 tiles of accumulators spill here)
    const uint32_t steps = 8, chunk = 64, NC = (m.D + chunk - 1) / chunk;
    const LaunchFn fn = local_long_launcher((int)tiles);
    if (!fn) return false;
    img1.assign((size_t)groups * NC * tiles * steps * 64 * 4, 0.0f);
    for (uint32_t g = 0; g < groups; ++g)
        for (uint32_t c = 0; c < NC; ++c)
            for (uint32_t t = 0; t < tiles; ++t) {
                const uint32_t T = g * tiles + t, proto = T / tpp;
                if (proto >= m.K) continue;
                for (uint32_t j = 0; j < steps; ++j)
                    for (uint32_t lane = 0; lane < 64; ++lane) {
                        const uint32_t r = 32 * (T % tpp) + (lane & 31), d0 = chunk * c + 4 * (2 * j + (lane >> 5));
                        float *dst = &img1[(((((size_t)g * NC + c) * tiles + t) * steps + j) * 64 + lane) * 4];
                        for (uint32_t e = 0; e < 4; ++e)
                            if (r < m.M && d0 + e < m.D) dst[e] = Oms[((size_t)proto * m.M + r) * m.D + d0 + e];
                    }
            }
    img2.assign((size_t)groups * tiles * TAIL_FLOATS, 0.0f);
    for (uint32_t T = 0; T < total; ++T)
        for (uint32_t row = 0; row < 32; ++row) {
            const uint32_t r = 32 * (T % tpp) + row;
            if (r < m.M) img2[(size_t)T * TAIL_FLOATS + row] = negp[(size_t)(T / tpp) * m.M + r];
        }
    m.ngroups1 = groups;
    m.tiles_per_proto = tpp;
    m.launch = fn;
    m.samples_per_batch = 32;
    m.name = "mfma_local_long";
    return true;
}

// LGMLVQ with the projections on the bf16 matrix cores: k_local_pj (kernels_local.hip).  Every Omega_j padded to whole
// 32-row tiles as in bind_local; the tiles are Geom3 tiles (engine.h): three bf16 pieces per weight, Geom's tail.
template <int R1>
bool bind_local_pj(EngineModel &m, std::vector<float> &img, const std::vector<float> &Oms, const std::vector<float> &negp,
                   const std::vector<uint32_t> &cls)
{
    if (m.R1 != (uint32_t)R1 || !m.opt.bf_rank || !m.opt.bf_proj) return false;
    int spb = 0;
    const LaunchFn fn = local_pj_launcher(R1, &spb);
    if (!fn) return false;
    // every Omega_j padded to whole 32-row tiles, so a tile never straddles two prototypes (as bind_local)
    const uint32_t tpp = (m.M + 31) / 32, rows_pp = tpp * 32;
    std::vector<float> rows((size_t)m.K * rows_pp * m.D, 0.0f), init((size_t)m.K * rows_pp, 0.0f);
    std::vector<uint32_t> rcls((size_t)m.K * rows_pp, 0xFFFFFFFFu);
    for (uint32_t j = 0; j < m.K; ++j)
        for (uint32_t r = 0; r < rows_pp; ++r) {
            const size_t o = (size_t)j * rows_pp + r;
            if (r < m.M) {
                std::memcpy(&rows[o * m.D], &Oms[((size_t)j * m.M + r) * m.D], m.D * sizeof(float));
                init[o] = negp[(size_t)j * m.M + r];
            }
            if (!cls.empty()) rcls[o] = cls[j];
        }
    uint32_t ngroups = 0;
    img = build_images3<R1>(rows.data(), rows.size() / m.D, m.D, m.D, init.data(), rcls.data(), ngroups);
    m.ngroups1 = ngroups;
    m.tiles_per_proto = tpp;
    m.launch = fn;
    m.samples_per_batch = spb;
    m.name = "mfma_local_bf16";
    m.pj = true;
    return true;
}

template <int R1>
bool bind_local(EngineModel &m, std::vector<float> &img, const std::vector<float> &Oms, const std::vector<float> &negp,
                const std::vector<uint32_t> &cls)
{
    if (m.R1 != (uint32_t)R1) return false;
    // every Omega_j padded to whole 32-row tiles, so a tile never straddles two prototypes
    const uint32_t tpp = (m.M + 31) / 32, rows_pp = tpp * 32;
    std::vector<float> rows((size_t)m.K * rows_pp * m.D, 0.0f), init((size_t)m.K * rows_pp, 0.0f);
    std::vector<uint32_t> rcls((size_t)m.K * rows_pp, 0xFFFFFFFFu);
    for (uint32_t j = 0; j < m.K; ++j)
        for (uint32_t r = 0; r < rows_pp; ++r) {
            const size_t o = (size_t)j * rows_pp + r;
            if (r < m.M) {
                std::memcpy(&rows[o * m.D], &Oms[((size_t)j * m.M + r) * m.D], m.D * sizeof(float));
                init[o] = negp[(size_t)j * m.M + r];
            }
            if (!cls.empty()) rcls[o] = cls[j];
        }
    img = build_images<R1>(rows.data(), rows.size() / m.D, m.D, m.D, 1.0f, init.data(), rcls.data(), 0.0f, m.ngroups1);
    m.tiles_per_proto = tpp;
    m.launch = streamed_launcher(MODE_LOCAL, R1, R1, &m.samples_per_batch);
    return true;
}

}  // namespace

EngineModel *engine_build(const std::vector<float> &Win, uint32_t K, uint32_t D, const std::vector<float> &Om,
                          int omega_kind, uint32_t M, const std::vector<uint32_t> &cls, const BuildOptions &opt,
                          std::string &why, hipError_t &herr)
{
    herr = hipSuccess;
    if (D > 256 && !(((omega_kind == OMEGA_NONE && (uint64_t)(K + 255) * (D + 127) * 4 < (1ull << 31)) || (omega_kind == OMEGA_GLOBAL && M <= 128) ||
                       (omega_kind == OMEGA_LOCAL && (uint64_t)K * ((M + 31) / 32) <= 2048)) && D <= 65536 &&
                     !getenv("VQHIP_NO_LONG"))) {
        why = "sample length above 256 (served for Euclidean models whose prototypes take less than 2 GiB, a global Omega of at most 128 rows, local Omegas of at most 2048 tiles of 32 rows)";
        return nullptr;
    }
    if (omega_kind != OMEGA_NONE && M > 256) {
        why = "Omega has more than 256 rows";
        return nullptr;
    }
    EngineModel m;
    m.opt = opt;
    m.K = K; m.D = D; m.M = M;
    m.R1 = D > 256 ? 0u : (uint32_t)pad_pow2(D, 16);
    m.has_cls = !cls.empty();
    std::vector<float> img1, img2, plain;
    bool ok = false;

    // the coordinates everything below lives in: the caller's, or shifted by the mean prototype (centre_prototypes);
    // rows above 256 values included (the long-row kernels subtract the centre from every chunk they fetch)
    std::vector<float> mu, Wc;
    if (omega_kind == OMEGA_NONE && bind_direct(m, img1, Win, cls)) {         // nothing to screen, nothing to centre
        m.mode = MODE_EUCLID;
        EngineModel *out = new EngineModel(m);
        herr = to_device(&out->img1, img1);
        if (herr == hipSuccess) herr = hipMalloc((void **)&out->queue, QUEUE_SLOTS * QUEUE_STRIDE * sizeof(uint32_t));
        if (herr != hipSuccess) {
            engine_free(out);
            return nullptr;
        }
        return out;
    }
    const bool centred = centre_prototypes(Win, K, D, mu, Wc);
    const std::vector<float> &W = centred ? Wc : Win;

    if (omega_kind == OMEGA_NONE || omega_kind == OMEGA_GLOBAL) {
        // search space: the prototypes themselves, or their projections p_k = Omega w_k (f64, rounded once)
        const uint32_t L = omega_kind == OMEGA_NONE ? D : M;
        std::vector<float> P;
        if (omega_kind == OMEGA_GLOBAL) {
            P.resize((size_t)K * M);
            for (uint32_t k = 0; k < K; ++k)
                for (uint32_t r = 0; r < M; ++r) {
                    double t = 0.0;
                    for (uint32_t d = 0; d < D; ++d) t += (double)Om[(size_t)r * D + d] * (double)W[(size_t)k * D + d];
                    P[(size_t)k * M + r] = (float)t;
                }
        }
        const std::vector<float> &Q = omega_kind == OMEGA_NONE ? W : P;
        std::vector<float> norms(K);
        double wmax2 = 0.0;
        for (uint32_t k = 0; k < K; ++k) {
            double t = 0.0;
            for (uint32_t d = 0; d < L; ++d) t += (double)Q[(size_t)k * L + d] * (double)Q[(size_t)k * L + d];
            norms[k] = (float)t;
            wmax2 = t > wmax2 ? t : wmax2;
            // The screening form needs ||p||^2 and 2 p.x inside f32 range; the scalar kernels do not.
            if (!(t < 1e37)) {
                why = "a prototype's squared norm is not finite in f32";
                return nullptr;
            }
        }
        // screening scores: ||q_k||^2, or PAD_SCORE for rows that repeat an earlier row of their class
        std::vector<float> scores = norms;
        {
            const std::vector<bool> drop = shadowed_rows(Q, K, L, cls);
            for (uint32_t k = 0; k < K; ++k)
                if (drop[k]) scores[k] = PAD_SCORE;
        }
        // ||x||^2 below this: |2 x.w| <= 2 sqrt(ss wmax2) <= 1e18 and ||w||^2 + c stay far inside f32 range
        m.ss_limit = (float)std::min(1e36, 1e36 / (4.0 * std::max(wmax2, 1e-36)));
        m.wmax2 = (float)wmax2;
        // screen_margin (mfma_device.h): twice the error of one key, reduction length L.  The last term is the worst case
        // of the f32 fma chain (L 2^-24 of ||x||^2 + ||w||^2 per key, twice for a pair of keys, + the position bits): it
        // takes over from the statistical middle term above L = 200, so the guarantee is one at every row length.
        m.mscale = (float)std::max({6.103515625e-05, 4.0 * (8.0 * std::sqrt((double)L) * 5.9604644775390625e-08 + 1.9073486328125e-06),
                                    2.0 * L * 5.9604644775390625e-08 + 3.814697265625e-06});
        if (omega_kind == OMEGA_NONE) {
            m.mode = MODE_EUCLID;
            m.name = "mfma_euclid";
            m.ldp = m.R1;
            ok = bind_long(m, img1, img2, W, scores, cls) || bind_euclid<16>(m, img1, W, scores, cls) || bind_euclid<32>(m, img1, W, scores, cls) ||
                 bind_euclid<64>(m, img1, W, scores, cls) || bind_euclid<128>(m, img1, W, scores, cls) ||
                 bind_euclid<256>(m, img1, W, scores, cls);
        } else {
            m.mode = MODE_OMEGA;
            m.name = "mfma_omega";
            m.R2 = (uint32_t)pad_pow2(M, 16);
            m.ldp = m.R2;
            ok = bind_omega_long<32>(m, img1, img2, Om, P, scores, cls) || bind_omega_long<64>(m, img1, img2, Om, P, scores, cls) ||
                 bind_omega_long<128>(m, img1, img2, Om, P, scores, cls) ||
                 bind_lowrank<256>(m, img1, img2, Om, P, scores, cls) || bind_lowrank<128>(m, img1, img2, Om, P, scores, cls) ||
                 bind_lowrank<64>(m, img1, img2, Om, P, scores, cls) || bind_lowrank<32>(m, img1, img2, Om, P, scores, cls) ||
                 bind_omega_pipe<128>(m, img1, img2, Om, P, scores, cls) || bind_omega_pipe<64>(m, img1, img2, Om, P, scores, cls) ||
                 bind_omega_pipe<32>(m, img1, img2, Om, P, scores, cls) ||
                 // streamed engine: per R1 the narrowest instantiated search width that holds Omega's rows
                 bind_omega<16, 16>(m, img1, img2, Om, P, scores, cls) || bind_omega<32, 16>(m, img1, img2, Om, P, scores, cls) ||
                 bind_omega<32, 32>(m, img1, img2, Om, P, scores, cls) || bind_omega<64, 64>(m, img1, img2, Om, P, scores, cls) ||
                 bind_omega<128, 128>(m, img1, img2, Om, P, scores, cls) || bind_omega<256, 16>(m, img1, img2, Om, P, scores, cls) ||
                 bind_omega<256, 64>(m, img1, img2, Om, P, scores, cls);
        }
        // zero-padded plain rows for the exact re-measurement
        plain.assign((size_t)K * m.ldp, 0.0f);
        for (uint32_t k = 0; k < K; ++k) std::memcpy(&plain[(size_t)k * m.ldp], &Q[(size_t)k * L], L * sizeof(float));
    } else {
        m.mode = MODE_LOCAL;
        m.name = "mfma_local";
        std::vector<float> negp((size_t)K * M);                        // -Omega_j w_j (f64, rounded once)
        for (uint32_t j = 0; j < K; ++j)
            for (uint32_t r = 0; r < M; ++r) {
                double t = 0.0;
                for (uint32_t d = 0; d < D; ++d)
                    t += (double)Om[((size_t)j * M + r) * D + d] * (double)W[(size_t)j * D + d];
                negp[(size_t)j * M + r] = (float)(-t);
            }
        ok = bind_local_long(m, img1, img2, Om, negp) || bind_local_pj<16>(m, img1, Om, negp, cls) || bind_local_pj<32>(m, img1, Om, negp, cls) || bind_local_pj<64>(m, img1, Om, negp, cls) ||
             bind_local_pj<128>(m, img1, Om, negp, cls) || bind_local_pj<256>(m, img1, Om, negp, cls) ||
             bind_local<16>(m, img1, Om, negp, cls) || bind_local<32>(m, img1, Om, negp, cls) ||
             bind_local<64>(m, img1, Om, negp, cls) || bind_local<128>(m, img1, Om, negp, cls) ||
             bind_local<256>(m, img1, Om, negp, cls);
    }
    if (!ok) {
        why = "no engine instantiation for this (D, M) pair";
        return nullptr;
    }

    // the streamed engine with the bf16 ranking reads its search tiles as READY pieces (tile_mma_ready): same tiles, same
    // tails, the f32 values replaced by their pieces -- nothing else reads them (exact distances come from `plain`)
    if (m.bf && m.lds_used == 0 && (m.mode == MODE_EUCLID || m.mode == MODE_OMEGA) &&
        (std::strcmp(m.name, "mfma_euclid") == 0 || std::strcmp(m.name, "mfma_omega") == 0)) {
        std::vector<float> &simg = m.mode == MODE_EUCLID ? img1 : img2;
        const uint32_t R = m.mode == MODE_EUCLID ? m.R1 : m.R2;
        switch (R) {
        case 16: pieces_image<16>(simg, (uint32_t)(simg.size() / Geom<16>::TILE_FLOATS)); break;
        case 32: pieces_image<32>(simg, (uint32_t)(simg.size() / Geom<32>::TILE_FLOATS)); break;
        case 64: pieces_image<64>(simg, (uint32_t)(simg.size() / Geom<64>::TILE_FLOATS)); break;
        case 128: pieces_image<128>(simg, (uint32_t)(simg.size() / Geom<128>::TILE_FLOATS)); break;
        default: why = "bf16 ranking on a search width without a pieces image"; return nullptr;
        }
    }

    std::vector<float> imgbf, img_f32;
    if (m.bf && m.lds_used != 0) {
        std::vector<float> &simg = m.search_img == 1 ? img1 : img2;
        switch (m.search_R) {
        case 16: imgbf = bf_tiles_for<16>(m, simg, m.nbf); break;
        case 32: imgbf = bf_tiles_for<32>(m, simg, m.nbf); break;
        case 64: imgbf = bf_tiles_for<64>(m, simg, m.nbf); break;
        case 128: imgbf = bf_tiles_for<128>(m, simg, m.nbf); break;
        default: m.nbf = 0; break;
        }
        // no room for every tile's pieces beside the f32 image: the image becomes the pieces, the exact rows come from L2
        if (m.nbf < res_tiles(m.K) && m.bfimg_ok && !getenv("VQHIP_NO_BFIMG")) {
            // (the Omega kernels' finish pass has no pieces-image form: it needs the f32 image)
            if (m.finish && (m.search_img == 2 || !getenv("VQHIP_FINISH_PIECES"))) img_f32 = simg;
            switch (m.search_R) {
            case 16: pieces_image<16>(simg, res_tiles(m.K)); break;
            case 32: pieces_image<32>(simg, res_tiles(m.K)); break;
            case 64: pieces_image<64>(simg, res_tiles(m.K)); break;
            default: pieces_image<128>(simg, res_tiles(m.K)); break;
            }
            m.bfimg = true;
            m.nbf = 0;
            imgbf.clear();
        }
    }

    EngineModel *out = new EngineModel(m);
    herr = to_device(&out->img1, img1);
    if (herr == hipSuccess && !imgbf.empty()) herr = to_device(&out->imgbf, imgbf);
    if (herr == hipSuccess && !img_f32.empty()) herr = to_device(&out->img_f32, img_f32);
    if (herr == hipSuccess && !img2.empty()) herr = to_device(&out->img2, img2);
    if (herr == hipSuccess && !plain.empty()) herr = to_device(&out->plain, plain);
    if (herr == hipSuccess && !cls.empty()) herr = to_device(&out->cls, cls);
    if (herr == hipSuccess && centred) {
        mu.resize(std::max<size_t>(256, ((size_t)D + 127) / 128 * 128), 0.0f);   // zero padded to the longest fragment / whole chunks a kernel reads
        herr = to_device(&out->mu, mu);
    }
    if (herr == hipSuccess) herr = hipMalloc((void **)&out->queue, QUEUE_SLOTS * QUEUE_STRIDE * sizeof(uint32_t));
    if (herr != hipSuccess) {
        engine_free(out);
        return nullptr;
    }
    return out;
}

hipError_t zero_queue(const EngineArgs &a, hipStream_t st)
{
    return a.queue_zeroed ? hipSuccess : hipMemsetAsync(a.queue, 0, QUEUE_BYTES, st);
}

void engine_free(EngineModel *m)
{
    if (!m) return;
    (void)hipFree(m->img1);
    (void)hipFree(m->img2);
    (void)hipFree(m->plain);
    (void)hipFree(m->cls);
    (void)hipFree(m->queue);
    (void)hipFree(m->mu);
    (void)hipFree(m->imgbf);
    (void)hipFree(m->img_f32);
    for (uint32_t *p : m->crowd) (void)hipFree(p);
    delete m;
}

const char *engine_name(const EngineModel *m) { return m->name; }
const char *engine_ranking(const EngineModel *m)
{
    if (m->bf) return "bf16x3";
    return m->mode == MODE_LOCAL || std::strncmp(m->name, "valu", 4) == 0 ? "none" : "f32";
}

const char *engine_projection(const EngineModel *m)
{
    if (m->pj) return "bf16x6";
    return m->mode == MODE_EUCLID ? "none" : "f32";
}

hipError_t engine_assign(const EngineModel *m, const float *dX, uint64_t N, const uint32_t *dy, Outputs out, hipStream_t s, uint32_t *queue)
{
    if (N == 0) return hipSuccess;
    EngineArgs a;
    a.X = dX; a.N = N; a.D = m->D; a.y = dy;
    a.img1 = m->img1; a.ngroups1 = m->ngroups1;
    a.img2 = m->img2; a.ngroups2 = m->ngroups2;
    a.plain = m->plain; a.ldp = m->ldp;
    a.K = m->K; a.tiles_per_proto = m->tiles_per_proto;
    a.res_tiles = res_tiles(m->K);
    a.nbatches = (N + m->samples_per_batch - 1) / m->samples_per_batch;
    a.out = out;
    a.ss_limit = m->ss_limit;
    a.wmax2 = m->wmax2;
    a.dbg = nullptr;
    static unsigned long long *dbg_buf = nullptr;                     // diagnostic build only (VQHIP_VARIANT=14)
    if ((variant_env() >= 31 && variant_env() <= 35) && !queue) {
        if (!dbg_buf && hipMalloc((void **)&dbg_buf, 4096 * 4 * 8 * 8) != hipSuccess) dbg_buf = nullptr;
        a.dbg = dbg_buf;
    }
    if (getenv("VQHIP_STATS") && !queue) {                            // experiment: event counters of the epilogue
        if (!dbg_buf && hipMalloc((void **)&dbg_buf, 4096 * 4 * 8 * 8) != hipSuccess) dbg_buf = nullptr;
        a.dbg = dbg_buf;
        if (dbg_buf) (void)hipMemsetAsync(dbg_buf, 0, 256, s);
    }
    a.queue = queue ? queue : m->queue + (m->next_slot++ % QUEUE_SLOTS) * QUEUE_STRIDE;
    a.queue_zeroed = 0u;
    a.mu = m->mu;
    a.mscale = m->mscale;
    a.bf = m->bf ? 1u : 0u;
    a.imgbf = m->imgbf;
    a.nbf = m->imgbf ? m->nbf : 0u;
    a.bfimg = m->bfimg ? 1u : 0u;
    a.pj = m->pj ? 1u : 0u;
    a.cls = m->cls;
    a.crowd_mask = nullptr;
    a.list = nullptr;
    a.list_count = nullptr;
    // The finish pass: not for the caller's own counters (the no-allocation entry point: no buffers may be made here),
    // not beyond 2^29 rows (ids and flags share a word)
    const bool finish = m->finish && !queue && N < (1ull << 29);
    const uint64_t ntiles32 = (N + 31) / 32;
    uint32_t *cbuf = nullptr;
    if (finish) {
        const uint32_t slot = m->next_crowd++ % CROWD_SLOTS;
        if (m->crowd_rows[slot] < N) {
            // every slot grows at once: one synchronisation (launches that still use the old buffers) and the allocations
            // in the first call of a size, none in the calls behind it (a caller's warm-up call warms all of it up)
            hipError_t ce = hipDeviceSynchronize();
            if (ce != hipSuccess) return ce;
            const uint64_t rows = N + N / 4 + 1024;
            for (uint32_t i = 0; i < CROWD_SLOTS; ++i) {
                if (m->crowd_rows[i] >= rows) continue;
                (void)hipFree(m->crowd[i]);
                m->crowd[i] = nullptr;
                m->crowd_rows[i] = 0;
                ce = hipMalloc((void **)&m->crowd[i], (2 * (size_t)QUEUE_STRIDE + 3 * ((rows + 31) / 32) + 4 + rows) * sizeof(uint32_t));
                if (ce != hipSuccess) return ce;
                m->crowd_rows[i] = rows;
            }
        }
        // the slot: [batch counters of the first pass][of the finish pass][3 mask words per tile][count, 3 pad][list] --
        // counters, marks and count zeroed by ONE memset (three separate ones and the launches they sit between cost a
        // small call 10 us: tools/small_batch_bench.py)
        uint32_t *const slot_base = m->crowd[slot];
        cbuf = slot_base + 2 * (size_t)QUEUE_STRIDE;
        const hipError_t ce = hipMemsetAsync(slot_base, 0, (2 * (size_t)QUEUE_STRIDE + 3 * ntiles32 + 1) * sizeof(uint32_t), s);
        if (ce != hipSuccess) return ce;
        a.crowd_mask = cbuf;
        a.queue = slot_base;
        a.queue_zeroed = 1u;
    }
    hipError_t e = m->launch(a, dy != nullptr, s);
    if (finish && e == hipSuccess) {
        uint32_t *count = cbuf + 3 * ntiles32, *list = count + 4;      // (the buffer holds 3 words per tile + 4 + one id per row of its largest launch)
        hipLaunchKernelGGL(k_crowd_compact, dim3((unsigned)((ntiles32 + 255) / 256)), dim3(256), 0, s, cbuf, (uint32_t)ntiles32, list, count);
        e = hipGetLastError();
        if (e == hipSuccess) {
            EngineArgs f = a;
            f.crowd_mask = nullptr;
            f.list = list;
            f.list_count = count;
            f.queue = a.queue + QUEUE_STRIDE;                            // (zeroed with the marks)
            if (m->img_f32) {                                            // the finish pass on the f32 image (hits measured from LDS)
                (m->search_img == 2 ? f.img2 : f.img1) = m->img_f32;
                f.bfimg = 0u;
                f.imgbf = nullptr;
                f.nbf = 0u;
            }
            e = m->finish(f, dy != nullptr, s);
            if (e == hipSuccess && getenv("VQHIP_STATS")) {              // diagnostic: how many samples the first pass marked
                (void)hipStreamSynchronize(s);
                uint32_t marked = 0;
                (void)hipMemcpy(&marked, count, 4, hipMemcpyDeviceToHost);
                fprintf(stderr, "stats: %u of %llu samples marked for the finish pass\n", marked, (unsigned long long)N);
            }
        }
    }
    if (a.dbg && e == hipSuccess && getenv("VQHIP_STATS")) {
        (void)hipStreamSynchronize(s);
        unsigned long long h[8];
        (void)hipMemcpy(h, a.dbg, sizeof h, hipMemcpyDeviceToHost);
        fprintf(stderr, "stats: scans(wave) %llu scan_lanes %llu close(wave) %llu crowded_p %llu crowded_n %llu nonfinite %llu\n", h[0], h[1], h[2], h[3], h[4], h[5]);
    } else
    if (a.dbg && e == hipSuccess && getenv("VQHIP_DUMP")) {           // diagnostic: phase sums of this launch
        (void)hipStreamSynchronize(s);
        const size_t nw = 512 * 4;
        std::vector<unsigned long long> h(nw * 8);
        (void)hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost);
        double sum[8] = {0};
        for (size_t w = 0; w < nw; ++w)
            for (int i = 0; i < 8; ++i) sum[i] += (double)h[w * 8 + i];
        double tot = 0;
        for (int i = 0; i < 7; ++i) tot += sum[i];
        if (variant_env() >= 31 && variant_env() <= 35) {
            double s4[6] = {0};
            for (size_t w = 0; w < (size_t)256 * (getenv("VQHIP_NW4") ? 4 : 8); ++w)
                for (int i = 0; i < 6; ++i) s4[i] += (double)h[w * 8 + i];
            unsigned long long b0 = ~0ull, b1 = 0, e0 = ~0ull, e1 = 0;
            for (size_t w = 0; w < (size_t)256 * (getenv("VQHIP_NW4") ? 4 : 8); ++w) {
                b0 = std::min(b0, h[w * 8 + 6]); b1 = std::max(b1, h[w * 8 + 6]);
                e0 = std::min(e0, h[w * 8 + 7]); e1 = std::max(e1, h[w * 8 + 7]);
            }
            fprintf(stderr, "loop start spread %.1f us, first start -> first end %.1f us, -> last end %.1f us\n", (b1 - b0) / 100.0,
                    (e0 - b0) / 100.0, (e1 - b0) / 100.0);
            fprintf(stderr, "in-kernel clock %.3f GHz (span %.0f cycles = %.1f us)\n", s4[4] / s4[5] * 0.1, s4[4] / (getenv("VQHIP_NW4") ? 1024 : 2048), s4[5] / (getenv("VQHIP_NW4") ? 1024 : 2048) / 100.0);
            const double nwv = getenv("VQHIP_NW4") ? 1024 : 2048;
            fprintf(stderr, "pipe phase cycles per wave: image_load(10ns ticks) %.0f prefetch_issue %.0f tiles %.0f epilogue %.0f\n", s4[0] / nwv,
                    s4[1] / nwv, s4[2] / nwv, s4[3] / nwv);
        } else
        fprintf(stderr, "phase cycles per wave: loads %.0f dma %.0f mma %.0f tile_epi %.0f barrier %.0f batch_epi %.0f tail %.0f total %.0f\n",
                sum[0] / nw, sum[1] / nw, sum[2] / nw, sum[3] / nw, sum[4] / nw, sum[5] / nw, sum[6] / nw, tot / nw);
    }
    return e;
}

}  // namespace vq
