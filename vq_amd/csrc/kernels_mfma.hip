// MFMA tile engine for gfx950: nearest-prototype search as streamed-weight f32 MFMA GEMMs.
// Geometry and the reasons behind it: engine.h.  What each kernel restates:
//
//   k_engine<MODE_EUCLID>  find_closest_prototype{,_matched}, omega = None     src/helpers.rs:55-77, 91-132
//        screening score_k = ||w_k||^2 - 2 x.w_k on the matrix cores (argmin-equivalent to
//        helpers.rs:11-24), best TWO candidates per sample re-measured in the direct
//        sum (x_d - w_kd)^2 form, rooted, compared like helpers.rs:70.
//   k_engine<MODE_OMEGA>   the same with Some(omega): y = Omega x on the matrix cores, then the
//        Euclidean machinery in the projected space against p_k = Omega w_k; distance
//        ||y - p_k||^2 = (x-w)^T Omega^T Omega (x-w), squared                  src/helpers.rs:35-45
//   k_engine<MODE_LOCAL>   LGMLVQ: z = Omega_j x - Omega_j w_j per prototype j, ||z||^2,
//        strict '<' in index order                                             src/lgmlvq/lgmlvq.rs:436-506
//
// One workgroup = 4 waves; each wave keeps S tiles of 32 samples in registers and all waves walk
// the same stream of weight-tile groups through a double-buffered LDS stage filled by LDS-DMA
// (global_load_lds_dwordx4), one barrier per group.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "engine.h"
#include "vq_common.h"

namespace vq {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f32x16 = __attribute__((ext_vector_type(16))) float;

enum { MODE_EUCLID = 0, MODE_OMEGA = 1, MODE_LOCAL = 2 };

struct EngineArgs {
    const float *X;           // N x D samples (row stride D)
    uint64_t N;
    uint32_t D;               // actual sample length (multiple of 4, <= R1)
    const uint32_t *y;        // sample class ids (matched) or null
    const float *img1;        // stage-1 groups: prototypes (EUCLID), Omega rows (OMEGA), Omega_j rows (LOCAL)
    uint32_t ngroups1;
    const float *img2;        // stage-2 groups: projected prototypes (OMEGA)
    uint32_t ngroups2;
    const float *plain;       // K x ldp rows for the exact re-measurement (W, or P = W.Omega^T), zero padded
    uint32_t ldp;
    uint32_t K;               // actual number of prototypes
    uint32_t tiles_per_proto; // LOCAL: 32-row tiles per Omega_j
    uint64_t nbatches;
    Outputs out;
    float wmax2;              // largest ||w_k||^2 (screening error bound)
    float ss_limit;           // resident_pipe: ||x||^2 below which no screening score can overflow
    uint32_t *queue;          // k_engine: batch counter, zeroed before every launch
    // Prototypes whose search-space rows are identical (null when the model has none): dup_first[k] = lowest
    // index with k's row, dup_next[k] = next higher index with it.  See settle_ties.
    const uint32_t *dup_first, *dup_next;
    const uint32_t *cls;      // K class ids (labelled models)
    unsigned long long *dbg;  // diagnostic build of k_resident_pipe: per-wave cycle sums of the loop phases
};

namespace {

// ------------------------------------------------------------------------------ small helpers
__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }
__device__ __forceinline__ uint32_t xor32(uint32_t v) { return (uint32_t)__shfl_xor((int)v, 32, 64); }

__device__ __forceinline__ bool lexless(float av, uint32_t ai, float bv, uint32_t bi)
{
    return av < bv || (av == bv && ai < bi);
}

__device__ __forceinline__ unsigned long long stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Running best two (value, index) of a lane.  Offers arrive with increasing index, so strict '<'
// keeps the lower index on ties (helpers.rs:70); NaN never enters.
struct Top2 {
    float b1, b2;
    uint32_t i1, i2;
    __device__ Top2() : b1(INFINITY), b2(INFINITY), i1(0u), i2(0u) {}
    __device__ __forceinline__ void offer(float v, uint32_t k)
    {
        const bool c1 = v < b1, c2 = v < b2;
        b2 = c1 ? b1 : (c2 ? v : b2);
        i2 = c1 ? i1 : (c2 ? k : i2);
        b1 = c1 ? v : b1;
        i1 = c1 ? k : i1;
    }
    // union with the partner lane's pair (the other 16 rows of every tile)
    __device__ __forceinline__ void merge_halves()
    {
        const float o1 = xor32(b1), o2 = xor32(b2);
        const uint32_t oi1 = xor32(i1), oi2 = xor32(i2);
        float f, s;
        uint32_t fi, si;
        if (lexless(o1, oi1, b1, i1)) {
            f = o1; fi = oi1;
            if (lexless(b1, i1, o2, oi2)) { s = b1; si = i1; } else { s = o2; si = oi2; }
        } else {
            f = b1; fi = i1;
            if (lexless(o1, oi1, b2, i2)) { s = o1; si = oi1; } else { s = b2; si = i2; }
        }
        b1 = f; i1 = fi; b2 = s; i2 = si;
    }
};

struct Best1 {
    float d;
    uint32_t i;
    __device__ Best1() : d(INFINITY), i(0u) {}                          // helpers.rs:58-59
    __device__ __forceinline__ void offer(float v, uint32_t k)
    {
        const bool c = v < d;                                            // helpers.rs:70
        d = c ? v : d;
        i = c ? k : i;
    }
    __device__ __forceinline__ void merge_halves()
    {
        const float o = xor32(d);
        const uint32_t oi = xor32(i);
        if (lexless(o, oi, d, i)) { d = o; i = oi; }
    }
};

// Exact ties between identical prototypes.  The packed keys order equal scores by position inside the
// tile before tile number, and keep only two (labelled searches: one) candidates, so among three
// identical rows the one found need not be the first.  Identical rows have bit-identical distances, so
// the winner is moved to the first ELIGIBLE prototype with its row (helpers.rs:70: strict '<' keeps the
// first; helpers.rs:113 / :122: eligibility).  Costs one null test per batch for models without duplicates.
template <bool MATCHED>
__device__ __forceinline__ void settle_ties(const EngineArgs &a, uint32_t label, Best1 &wp, Best1 &wn)
{
    if (a.dup_first == nullptr) return;
    auto lowest = [&](uint32_t k, bool want_match) {
        uint32_t i = a.dup_first[k];
        while (i != k && MATCHED && (a.cls[i] == label) != want_match) i = a.dup_next[i];
        return i;
    };
    if (wp.d < INFINITY) wp.i = lowest(wp.i, true);
    if (MATCHED && wn.d < INFINITY) wn.i = lowest(wn.i, false);
}

// ---- packed screening keys (what they are and why: the comment above k_resident_pipe)
// Keys are f32 bit patterns and are compared AS FLOATS (v_min_f32 / v_med3_f32), so a score may have
// either sign and the accumulators start at the plain ||w_k||^2 column.  Nothing that is screened may be
// a NaN pattern (v_med3_f32 returns the minimum when it sees one): pad rows and the "tile before the
// first" score PAD_SCORE, finite and above every real score (< 1e38 by ss_limit), instead of +INF,
// whose pattern with position bits set is a NaN.  NaN / INF scores come only from samples that are
// sent to the exact scan anyway.
constexpr uint32_t KEY_NONE = 0x7F800000u;           // +INF: "no candidate yet"
constexpr float PAD_SCORE = 3.0e38f;
constexpr float KEY_VALID_F = 1.0e38f;               // keys of real prototypes are below this
struct Keys2 {                       // unlabelled search: best two keys and their tiles
    uint32_t b1 = KEY_NONE, b2 = KEY_NONE, t1 = 0u, t2 = 0u;
};
struct KeysM {                       // labelled search: best key with cls == y, best with cls != y
    uint32_t p = KEY_NONE, n = KEY_NONE, tp = 0u, tn = 0u;
};
__device__ __forceinline__ bool key_valid(uint32_t key) { return __builtin_bit_cast(float, key) < KEY_VALID_F; }

__device__ __forceinline__ uint32_t pack_key(float score, uint32_t o)
{
    uint32_t key;                                   // (bits & ~15) | o in one instruction
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(score), "s"(0xFFFFFFF0u), "n"(o));
    return key;
}
__device__ __forceinline__ uint32_t kmin(uint32_t x, uint32_t y)                     // minimum as floats
{
    uint32_t r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// The 16 scores of one finished tile offered to the lane's best two keys: position bits in, median, minimum
// (3 VALU per score).  One asm statement: the compiler pads a wait state behind every inline-asm result
// that the next instruction reads, and the chain b1 -> b1 would collect one per score.  PRE_NOPS: wait
// states for accumulators that the MFMA right in front of the statement has written.
template <int PRE_NOPS = 0>
__device__ __forceinline__ void screen16(const f32x16 &sc, Keys2 &ku)
{
    uint32_t t0, t1;
    if constexpr (PRE_NOPS > 0) asm volatile("s_nop 15\n\ts_nop 1" ::: "memory");
    asm(
        "v_and_or_b32 %[t0], %[s0], %[m], 0\n\t"
        "v_and_or_b32 %[t1], %[s1], %[m], 1\n\t"
        "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t"
        "v_min_f32 %[b1], %[b1], %[t0]\n\t"
        "v_and_or_b32 %[t0], %[s2], %[m], 2\n\t"
        "v_med3_f32 %[b2], %[b1], %[b2], %[t1]\n\t"
        "v_min_f32 %[b1], %[b1], %[t1]\n\t"
        "v_and_or_b32 %[t1], %[s3], %[m], 3\n\t"
        "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t"
        "v_min_f32 %[b1], %[b1], %[t0]\n\t"
        "v_and_or_b32 %[t0], %[s4], %[m], 4\n\t"
        "v_med3_f32 %[b2], %[b1], %[b2], %[t1]\n\t"
        "v_min_f32 %[b1], %[b1], %[t1]\n\t"
        "v_and_or_b32 %[t1], %[s5], %[m], 5\n\t"
        "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t"
        "v_min_f32 %[b1], %[b1], %[t0]\n\t"
        "v_and_or_b32 %[t0], %[s6], %[m], 6\n\t"
        "v_med3_f32 %[b2], %[b1], %[b2], %[t1]\n\t"
        "v_min_f32 %[b1], %[b1], %[t1]\n\t"
        "v_and_or_b32 %[t1], %[s7], %[m], 7\n\t"
        "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t"
        "v_min_f32 %[b1], %[b1], %[t0]\n\t"
        "v_and_or_b32 %[t0], %[s8], %[m], 8\n\t"
        "v_med3_f32 %[b2], %[b1], %[b2], %[t1]\n\t"
        "v_min_f32 %[b1], %[b1], %[t1]\n\t"
        "v_and_or_b32 %[t1], %[s9], %[m], 9\n\t"
        "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t"
        "v_min_f32 %[b1], %[b1], %[t0]\n\t"
        "v_and_or_b32 %[t0], %[s10], %[m], 10\n\t"
        "v_med3_f32 %[b2], %[b1], %[b2], %[t1]\n\t"
        "v_min_f32 %[b1], %[b1], %[t1]\n\t"
        "v_and_or_b32 %[t1], %[s11], %[m], 11\n\t"
        "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t"
        "v_min_f32 %[b1], %[b1], %[t0]\n\t"
        "v_and_or_b32 %[t0], %[s12], %[m], 12\n\t"
        "v_med3_f32 %[b2], %[b1], %[b2], %[t1]\n\t"
        "v_min_f32 %[b1], %[b1], %[t1]\n\t"
        "v_and_or_b32 %[t1], %[s13], %[m], 13\n\t"
        "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t"
        "v_min_f32 %[b1], %[b1], %[t0]\n\t"
        "v_and_or_b32 %[t0], %[s14], %[m], 14\n\t"
        "v_med3_f32 %[b2], %[b1], %[b2], %[t1]\n\t"
        "v_min_f32 %[b1], %[b1], %[t1]\n\t"
        "v_and_or_b32 %[t1], %[s15], %[m], 15\n\t"
        "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t"
        "v_min_f32 %[b1], %[b1], %[t0]\n\t"
        "v_med3_f32 %[b2], %[b1], %[b2], %[t1]\n\t"
        "v_min_f32 %[b1], %[b1], %[t1]\n\t"
        : [b1] "+v"(ku.b1), [b2] "+v"(ku.b2), [t0] "=&v"(t0), [t1] "=&v"(t1)
        : [s0] "v"(sc[0]), [s1] "v"(sc[1]), [s2] "v"(sc[2]), [s3] "v"(sc[3]), [s4] "v"(sc[4]), [s5] "v"(sc[5]), [s6] "v"(sc[6]), [s7] "v"(sc[7]), [s8] "v"(sc[8]), [s9] "v"(sc[9]), [s10] "v"(sc[10]), [s11] "v"(sc[11]), [s12] "v"(sc[12]), [s13] "v"(sc[13]), [s14] "v"(sc[14]), [s15] "v"(sc[15]), [m] "s"(0xFFFFFFF0u));
}
// After a tile's scores went in: which tile do the best two keys come from now
__device__ __forceinline__ void settle_tiles(Keys2 &ku, const Keys2 &before, uint32_t tile_id)
{
    const bool c1 = ku.b1 != before.b1, c2 = ku.b2 != before.b2;
    const uint32_t demoted = ku.b2 == before.b1 ? before.t1 : tile_id;
    ku.t2 = c1 ? demoted : (c2 ? tile_id : before.t2);
    ku.t1 = c1 ? tile_id : before.t1;
}

__device__ __forceinline__ uint32_t key_row(uint32_t key, uint32_t tile, int h)   // prototype index of a key
{
    const uint32_t o = key & 15u;
    return tile * 32u + ((o >> 2) << 3) + (o & 3u) + 4u * (uint32_t)h;
}

__device__ __forceinline__ bool keyless(uint32_t ak, uint32_t ai, uint32_t bk, uint32_t bi)
{
    const float a = __builtin_bit_cast(float, ak), b = __builtin_bit_cast(float, bk);
    return a < b || (a == b && ai < bi);
}

// LDS-DMA of one stage (BYTES, multiple of 16) from its pre-swizzled global image: a linear copy,
// 1 KiB per wave-instruction, pieces dealt round-robin to the 4 waves.
template <int BYTES>
__device__ __forceinline__ void stage_group(char *lds_dst, const char *gsrc, int wave, int lane)
{
    constexpr int PIECES = (BYTES + 1023) / 1024;
#pragma unroll
    for (int i = 0; i < (PIECES + 3) / 4; ++i) {
        const int p = i * 4 + wave;
        const int off = p * 1024 + lane * 16;
        if (p < PIECES && off < BYTES)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + off),
                                             (__attribute__((address_space(3))) void *)(lds_dst + p * 1024), 16, 0, 0);
    }
}

// acc[s] (+)= tile . xf[s]^T for one 32-row tile in LDS; accumulators start at the tile's init column.
template <int R, int S>
__device__ __forceinline__ void tile_mma(const float *tile, const f32x4 (&xf)[S][R / 8], f32x16 (&acc)[S], int row, int h)
{
    using Gm = Geom<R>;
    const float *tail = tile + Gm::W_FLOATS;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 ci = *(const f32x4 *)(tail + 8 * g + 4 * h);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            acc[s][4 * g + 0] = ci[0]; acc[s][4 * g + 1] = ci[1];
            acc[s][4 * g + 2] = ci[2]; acc[s][4 * g + 3] = ci[3];
        }
    }
    const float *rowp = tile + Gm::chunk_off(row, h);
#pragma unroll
    for (int j = 0; j < R / 8; ++j) {
        const f32x4 a = *(const f32x4 *)(rowp + 8 * j);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int s = 0; s < S; ++s)
                acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], xf[s][j][e], acc[s], 0, 0, 0);
    }
}

// Exact distance of one sample (fragment xf, this lane's half of the chunks) to row `k` of the
// zero-padded plain matrix: sum over the lane's chunks, then both halves added.
template <int R>
__device__ __forceinline__ float exact_sq(const f32x4 (&xf)[R / 8], const float *plain, uint32_t ldp, uint32_t k, int h)
{
    const float *w = plain + (size_t)k * ldp;
    float p = 0.0f;
#pragma unroll
    for (int j = 0; j < R / 8; ++j) {
        const f32x4 wv = *(const f32x4 *)(w + 4 * (2 * j + h));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = wv[e] - xf[j][e];
            p = fmaf(d, d, p);
        }
    }
    return p + xor32(p);
}

// Full exact scan in index order (helpers.rs:55-77 in f32): used only when the screening scores of
// a sample are unusable (-INF from an infinite input, or a non-finite re-measured winner).
template <int R, bool ROOT>
__device__ __forceinline__ void exact_scan(const f32x4 (&xf)[R / 8], const float *plain, uint32_t ldp, uint32_t K, int h,
                                        const uint32_t *cls_plain, bool matched, uint32_t label, Best1 &pos, Best1 &neg)
{
    pos = Best1();
    neg = Best1();
    for (uint32_t k = 0; k < K; ++k) {
        float d = exact_sq<R>(xf, plain, ldp, k, h);
        if (ROOT) d = sqrtf(d);
        if (!matched || cls_plain[k] == label) pos.offer(d, k);
        else neg.offer(d, k);
    }
}

}  // namespace

// ------------------------------------------------------------------------------ the kernel
// R1: padded sample length.  R2: padded projected length (OMEGA only).  S: sample tiles per wave.
//
// Batches (4 waves x S tiles x 32 samples) are claimed by the workgroup from a global counter: the two
// workgroups of a CU do not get equal shares of the matrix pipes, so a fixed split leaves half the
// waves of the chip running alone at the end.  Thread 0 claims the next batch at the top of a batch;
// one barrier publishes it.  OMEGA: the samples are dead once they are
// projected, so the next batch's rows are fetched into the same registers while the search runs
// (every workgroup of the chip loading, then computing, in lockstep is an HBM burst followed by HBM
// silence).  The search stages screen with packed keys (4 VALU per score, see k_resident_pipe).
template <int MODE, int R1, int R2, int S, bool MATCHED, int WPS>
__global__ void __launch_bounds__(256, WPS) k_engine(const EngineArgs a, const uint32_t *__restrict__ cls_plain)
{
    using G1 = Geom<R1>;
    using G2 = Geom<MODE == MODE_OMEGA ? R2 : R1>;
    constexpr int RS = MODE == MODE_OMEGA ? R2 : R1;                 // reduction length of the search stage
    constexpr int BUF_BYTES = G1::STAGE_BYTES > G2::STAGE_BYTES ? G1::STAGE_BYTES : G2::STAGE_BYTES;
    constexpr bool ROOT = MODE == MODE_EUCLID;                       // helpers.rs:23 vs :44
    // OMEGA: Omega is zero padded to max(32, R2) rows, so the projection stages are known at compile time
    constexpr int T1 = R2 >= 32 ? R2 / 32 : 1;
    constexpr int NG1 = (T1 + G1::G - 1) / G1::G;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];
    __shared__ uint32_t claimed;                                     // the batch after this one

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    const uint32_t ngroups = MODE == MODE_OMEGA ? NG1 + a.ngroups2 : a.ngroups1;
    const uint32_t Cact = a.D >> 2;

    // DMA source of stage q of the cyclic weight stream
    auto issue = [&](uint32_t q, int buf) {
        char *dst = smem + buf * BUF_BYTES;
        if (MODE == MODE_OMEGA && q >= (uint32_t)NG1)
            stage_group<G2::STAGE_BYTES>(dst, (const char *)(a.img2 + (size_t)(q - NG1) * G2::STAGE_FLOATS), wave, lane);
        else
            stage_group<G1::STAGE_BYTES>(dst, (const char *)(a.img1 + (size_t)q * G1::STAGE_FLOATS), wave, lane);
    };

    // This wave's S tiles of 32 samples of batch bb -> B fragments, straight from HBM, 16 B per lane, through a
    // buffer descriptor rebased per tile (wave-uniform): rows past N read as zeros by the hardware range
    // check, one per-lane offset + immediates serves every chunk.  The loads are raw: nothing here waits for
    // them, so a batch can be fetched while the previous one is searched.
    f32x4 xf[S][R1 / 8];
    uint32_t label[S];
    const uint32_t row_bytes = a.D * 4u;
    const uint32_t voff = (uint32_t)s_col * row_bytes + (uint32_t)h * 16u;
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane(wave);
    auto load_samples = [&](uint64_t bb) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint64_t first = ((bb * 4 + wave_u) * S + s) * 32;
            uint32_t rec = 0u, rec_y = 0u;
            if (bb < a.nbatches && first < a.N) {
                const uint64_t left = a.N - first;
                rec = left * row_bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(left * row_bytes);
                rec_y = left * 4u > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(left * 4u);
            }
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)(a.X + first * a.D), 0, (int)rec, 0x00020000);
#pragma unroll
            for (int j = 0; j < R1 / 8; ++j)
                xf[s][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff + j * 32, 0, 0));
            if (MATCHED) {
                const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)(a.y + first), 0, (int)rec_y, 0x00020000);
                label[s] = __builtin_amdgcn_raw_buffer_load_b32(ry, (uint32_t)s_col * 4u, 0, 0);
            } else {
                label[s] = 0u;
            }
        }
    };
    // D < R1: the chunks past the row's end hold the next row's values (or range-check zeros); the weight
    // images are zero there, but 0 x INF is NaN, so they are cleared before the rows are used
    auto mask_columns = [&]() {
        if (Cact < (uint32_t)(R1 / 4)) {
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int j = 0; j < R1 / 8; ++j)
                    if ((uint32_t)(2 * j + h) >= Cact) xf[s][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    int cur = 0;
    uint64_t b = blockIdx.x;
    if (b < a.nbatches) {
        issue(0, 0);
        load_samples(b);
        __syncthreads();                                             // waits vmcnt(0) for the DMA, then barrier
    }

    // thread 0 keeps one claim in flight: the atomic issued at the top of batch k returns batch k + 2
    uint32_t pending = 0u;
    if (tid == 0) pending = gridDim.x + atomicAdd(a.queue, 1u);

    while (b < a.nbatches) {
        if (tid == 0) {
            claimed = pending;                                       // the wait for the atomic of a batch ago is here
            pending = gridDim.x + atomicAdd(a.queue, 1u);
        }
        __syncthreads();
        const uint64_t bnext = claimed;
        mask_columns();
        uint64_t sample[S];
        uint32_t lab[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            sample[s] = ((b * 4 + wave) * S + s) * 32 + s_col;
            lab[s] = label[s];
        }

        f32x4 yf[S][RS / 8];                                          // search-stage fragments
        Keys2 ku[S];
        KeysM km[S];
        Best1 pos[S], neg[S];                                         // LOCAL: exact distances, offered in index order
        float dsum[S];
        bool unsafe[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            dsum[s] = 0.0f;
            unsafe[s] = false;
        }
        if (MODE != MODE_OMEGA) {
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int j = 0; j < RS / 8; ++j) yf[s][j] = xf[s][j < R1 / 8 ? j : 0];
        } else {
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int j = 0; j < RS / 8; ++j) yf[s][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        // One step of the cyclic weight stream: prefetch the next stage, compute on this one, barrier.
        uint32_t q = 0;
        auto step = [&](auto &&compute) {
            const bool wrap = q + 1 == ngroups;
            if (!(wrap && bnext >= a.nbatches)) issue(wrap ? 0u : q + 1, cur ^ 1);   // nothing to stage after the last batch
            if (MODE == MODE_OMEGA && q == (uint32_t)NG1) {
                // x is dead: the next batch's rows are fetched now and arrive while the search runs.  vmcnt
                // counts in issue order, so they go out AFTER this step's DMA, and this step's barrier waits
                // for everything but them (a barrier that waited for vmcnt(0) would sit out an HBM round trip).
                load_samples(bnext);
                compute((const float *)(smem + cur * BUF_BYTES));
                static_assert(S * (R1 / 8) + 1 <= 63, "vmcnt is a 6-bit counter");
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(S * (R1 / 8) + (MATCHED ? S : 0)) : "memory");
            } else {
                compute((const float *)(smem + cur * BUF_BYTES));
                __syncthreads();               // next stage landed (vmcnt(0)) and everyone is done reading this one
            }
            cur ^= 1;
            ++q;
        };

        if constexpr (MODE == MODE_OMEGA) {
            // ---- projection: y = Omega x; accumulator registers become the search stage's fragment:
            //      row 8g + 4h + e of Omega tile t  ==  element e of chunk 2(4t + g) + h of y
            static_for<0, NG1>([&](auto q1) {
                step([&](const float *stage) {
                    static_for<0, G1::G>([&](auto tt) {
                        constexpr int t = decltype(q1)::value * G1::G + decltype(tt)::value;
                        f32x16 acc[S];
                        tile_mma<R1, S>(stage + decltype(tt)::value * G1::TILE_FLOATS, xf, acc, s_col, h);
                        static_for<0, 4>([&](auto g) {
                            constexpr int jj = 4 * t + decltype(g)::value;
                            if constexpr (jj < RS / 8) {
#pragma unroll
                                for (int s = 0; s < S; ++s)
                                    yf[s][jj] = f32x4{acc[s][4 * g], acc[s][4 * g + 1], acc[s][4 * g + 2], acc[s][4 * g + 3]};
                            }
                        });
                    });
                });
            });
        }

        if constexpr (MODE != MODE_LOCAL) {
            // samples whose screening scores could leave the f32 range go to the exact scan
#pragma unroll
            for (int s = 0; s < S; ++s) {
                float ss = 0.0f;
#pragma unroll
                for (int j = 0; j < RS / 8; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) ss = fmaf(yf[s][j][e], yf[s][j][e], ss);
                ss += xor32(ss);
                unsafe[s] = !(ss < a.ss_limit);
            }
        }

        const uint32_t nsearch = MODE == MODE_OMEGA ? a.ngroups2 : a.ngroups1;
        for (uint32_t qs = 0; qs < nsearch; ++qs) {
            step([&](const float *stage) {
                if constexpr (MODE == MODE_LOCAL) {
                    // ---- z = Omega_j x - Omega_j w_j (accumulator starts at -p_j), ||z||^2 summed per prototype
#pragma unroll
                    for (int tt = 0; tt < G1::G; ++tt) {
                        const float *tile = stage + tt * G1::TILE_FLOATS;
                        f32x16 acc[S];
                        tile_mma<R1, S>(tile, xf, acc, s_col, h);
                        const uint32_t ti = qs * G1::G + tt;
                        const uint32_t j = ti / a.tiles_per_proto, t = ti % a.tiles_per_proto;
#pragma unroll
                        for (int s = 0; s < S; ++s)
#pragma unroll
                            for (int r = 0; r < 16; ++r) dsum[s] = fmaf(acc[s][r], acc[s][r], dsum[s]);
                        if (t + 1 == a.tiles_per_proto) {
                            const uint32_t cj = ((const uint32_t *)(tile + G1::W_FLOATS))[32];
#pragma unroll
                            for (int s = 0; s < S; ++s) {
                                const float total = dsum[s] + xor32(dsum[s]);
                                dsum[s] = 0.0f;
                                if (j < a.K) {                        // pad tiles past the last prototype never compete
                                    if (!MATCHED || cj == lab[s]) pos[s].offer(total, j);   // lgmlvq.rs:448 / :487
                                    else neg[s].offer(total, j);                             // lgmlvq.rs:496
                                }
                            }
                        }
                    }
                } else {
                    // ---- search stage: key of ||p_k||^2 + c - 2 y.p_k for 32 prototypes per tile
#pragma unroll
                    for (int tt = 0; tt < G2::G; ++tt) {
                        const float *tile = stage + tt * G2::TILE_FLOATS;
                        f32x16 acc[S];
                        tile_mma<RS, S>(tile, yf, acc, s_col, h);
                        const uint32_t tile_id = qs * G2::G + tt;
                        const uint32_t *tcls = (const uint32_t *)(tile + G2::W_FLOATS) + 32 + 4 * h;
                        Keys2 ou[S];
                        KeysM om[S];
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            ou[s] = ku[s];
                            om[s] = km[s];
                        }
                        if constexpr (!MATCHED) {
#pragma unroll
                            for (int s = 0; s < S; ++s) {
                                screen16<1>(acc[s], ku[s]);            // acc[s] is fresh from the matrix pipe
                                settle_tiles(ku[s], ou[s], tile_id);
                            }
                        } else {
                            static_for<0, 16>([&](auto oo) {
                                constexpr int o = decltype(oo)::value;
                                const uint32_t cls = tcls[8 * (o >> 2) + (o & 3)];
#pragma unroll
                                for (int s = 0; s < S; ++s) {
                                    const uint32_t key = pack_key(acc[s][o], o);
                                    const bool match = cls == lab[s];                 // helpers.rs:113 / :122
                                    const uint32_t kp = match ? key : KEY_NONE, kn = match ? KEY_NONE : key;
                                    km[s].p = kmin(km[s].p, kp);
                                    km[s].n = kmin(km[s].n, kn);
                                }
                            });
#pragma unroll
                            for (int s = 0; s < S; ++s) {
                                km[s].tp = km[s].p != om[s].p ? tile_id : om[s].tp;
                                km[s].tn = km[s].n != om[s].n ? tile_id : om[s].tn;
                            }
                        }
                    }
                }
            });
        }

        // ---- epilogue: lane pair (s, h=0/1) agrees on the candidates, re-measures them exactly, stores
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const bool live = sample[s] < a.N;
            Best1 wp, wn;                                              // final (distance, index)
            if (MODE == MODE_LOCAL) {
                wp = pos[s];
                wn = neg[s];
            } else {
                bool rescan = unsafe[s];
                auto measure = [&](uint32_t k) {
                    const float d = exact_sq<RS>(yf[s], a.plain, a.ldp, k, h);
                    return ROOT ? sqrtf(d) : d;
                };
                if (!MATCHED) {
                    const uint32_t k1 = key_row(ku[s].b1, ku[s].t1, h), k2 = key_row(ku[s].b2, ku[s].t2, h);
                    const uint32_t pb1 = xor32(ku[s].b1), pb2 = xor32(ku[s].b2), pk1 = xor32(k1), pk2 = xor32(k2);
                    uint32_t f, fi, sd, si;                            // best two of the pair's four, by (key, index)
                    if (keyless(pb1, pk1, ku[s].b1, k1)) {
                        f = pb1; fi = pk1;
                        if (keyless(ku[s].b1, k1, pb2, pk2)) { sd = ku[s].b1; si = k1; } else { sd = pb2; si = pk2; }
                    } else {
                        f = ku[s].b1; fi = k1;
                        if (keyless(pb1, pk1, ku[s].b2, k2)) { sd = pb1; si = pk1; } else { sd = ku[s].b2; si = k2; }
                    }
                    if (key_valid(f) && !rescan) {
                        const float d1 = measure(fi);
                        wp.d = d1; wp.i = fi;
                        if (key_valid(sd)) {
                            const float d2 = measure(si);
                            if (lexless(d2, si, d1, fi)) { wp.d = d2; wp.i = si; }
                        }
                        rescan = !(wp.d < INFINITY);
                    } else {
                        rescan = true;
                    }
                } else {
                    uint32_t kp = key_row(km[s].p, km[s].tp, h), kn = key_row(km[s].n, km[s].tn, h), p = km[s].p, n = km[s].n;
                    const uint32_t pp = xor32(p), pkp = xor32(kp), pn = xor32(n), pkn = xor32(kn);
                    if (keyless(pp, pkp, p, kp)) { p = pp; kp = pkp; }
                    if (keyless(pn, pkn, n, kn)) { n = pn; kn = pkn; }
                    if (key_valid(p)) {
                        wp.d = measure(kp); wp.i = kp;
                        rescan |= !(wp.d < INFINITY);
                    }
                    if (key_valid(n)) {
                        wn.d = measure(kn); wn.i = kn;
                        rescan |= !(wn.d < INFINITY);
                    }
                }
                if (rescan) exact_scan<RS, ROOT>(yf[s], a.plain, a.ldp, a.K, h, cls_plain, MATCHED, lab[s], wp, wn);
                settle_ties<MATCHED>(a, lab[s], wp, wn);
            }
            if (live) {
                if (h == 0) {
                    if (a.out.idx) a.out.idx[sample[s]] = wp.i;
                    if (MATCHED && a.out.jneg) a.out.jneg[sample[s]] = wn.i;
                } else {
                    if (a.out.dist) a.out.dist[sample[s]] = wp.d;
                    if (MATCHED && a.out.dneg) a.out.dneg[sample[s]] = wn.d;
                }
            }
        }
        if (MODE != MODE_OMEGA) load_samples(bnext);
        b = bnext;
    }
}

// ------------------------------------------------------------------------------ resident-prototype kernel
// Euclidean models whose whole tile image fits in LDS (K * D <= ~36K floats; config 2 is 130 KiB):
// "prototypes staged in LDS" once per workgroup, then every wave is independent -- no per-tile DMA,
// no barriers, and the exact re-measurement of the candidates reads its rows from LDS as well.
// One 512-thread workgroup per CU = 2 waves per SIMD sharing one copy of the prototypes.
constexpr int RES_BYTES = 156 * 1024;

template <int R>
__device__ __forceinline__ float exact_sq_lds(const f32x4 (&xf)[R / 8], const float *img, uint32_t k, int h)
{
    using Gm = Geom<R>;
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    const uint32_t r = k & 31u;
    const float *rowp = img + (size_t)(k >> 5) * Gm::TILE_FLOATS + Gm::chunk_off((int)r, h);
    // two partial sums, v_pk_fma_f32: 4 VALU per chunk instead of 8 (VALU time is matrix-pipe time here)
    f32x2 p = {0.0f, 0.0f};
    const f32x2 half = {0.5f, 0.5f};
#pragma unroll
    for (int j = 0; j < R / 8; ++j) {
        const f32x4 v = *(const f32x4 *)(rowp + 8 * j);                          // v = -2 w
        const f32x2 d0 = __builtin_elementwise_fma(half, f32x2{v[0], v[1]}, f32x2{xf[j][0], xf[j][1]});   // x - w, exact scaling
        const f32x2 d1 = __builtin_elementwise_fma(half, f32x2{v[2], v[3]}, f32x2{xf[j][2], xf[j][3]});
        p = __builtin_elementwise_fma(d0, d0, p);
        p = __builtin_elementwise_fma(d1, d1, p);
    }
    const float t = p[0] + p[1];
    return t + xor32(t);
}

template <int R, int S, bool MATCHED>
__global__ void __launch_bounds__(512, 2) k_resident(const EngineArgs a)
{
    using Gm = Geom<R>;
    __shared__ __attribute__((aligned(16))) char smem[RES_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    const uint32_t ntiles = a.ngroups1 * Gm::G;
    const uint32_t Cact = a.D >> 2;

    {   // whole image -> LDS, linear LDS-DMA, 1 KiB per wave-instruction
        const uint32_t bytes = ntiles * Gm::TILE_BYTES;
        const char *src = (const char *)a.img1;
        for (uint32_t p = wave; p * 1024u < bytes; p += 8) {
            const uint32_t off = p * 1024u + lane * 16u;
            if (off < bytes)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                                 (__attribute__((address_space(3))) void *)(smem + p * 1024u), 16, 0, 0);
        }
        __syncthreads();
    }
    const float *img = (const float *)smem;

    for (uint64_t b = (uint64_t)blockIdx.x * 8 + wave; b < a.nbatches; b += (uint64_t)gridDim.x * 8) {
        f32x4 xf[S][R / 8];
        uint64_t sample[S];
        uint32_t label[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            sample[s] = (b * S + s) * 32 + s_col;
            const bool live = sample[s] < a.N;
            const float *xrow = a.X + sample[s] * a.D;
#pragma unroll
            for (int j = 0; j < R / 8; ++j) {
                const uint32_t c = 2 * j + h;
                const bool ok = live && c < Cact;
                const f32x4 v = *(const f32x4 *)(ok ? xrow + 4 * c : a.X);
                xf[s][j] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            label[s] = (MATCHED && live) ? a.y[sample[s]] : 0u;
        }
        Top2 top[S];
        Best1 pos[S], neg[S];

        for (uint32_t t = 0; t < ntiles; ++t) {
            const float *tile = img + (size_t)t * Gm::TILE_FLOATS;
            f32x16 acc[S];
            tile_mma<R, S>(tile, xf, acc, s_col, h);
            const uint32_t kbase = t * 32 + 4 * h;
            const uint32_t *tcls = (const uint32_t *)(tile + Gm::W_FLOATS) + 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x4 cg;
                if (MATCHED) cg = *(const u32x4 *)(tcls + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t k = kbase + 8 * g + e;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const float v = acc[s][4 * g + e];
                        if (!MATCHED) top[s].offer(v, k);
                        else if (cg[e] == label[s]) pos[s].offer(v, k);           // helpers.rs:113
                        else neg[s].offer(v, k);                                   // helpers.rs:122
                    }
                }
            }
        }

#pragma unroll
        for (int s = 0; s < S; ++s) {
            const bool live = sample[s] < a.N;
            Best1 wp, wn;
            bool rescan = false;
            if (!MATCHED) {
                top[s].merge_halves();
                const Top2 &t = top[s];
                if (t.b1 < KEY_VALID_F) {
                    const float d1 = sqrtf(exact_sq_lds<R>(xf[s], img, t.i1, h));
                    wp.d = d1; wp.i = t.i1;
                    if (t.b2 < KEY_VALID_F) {
                        const float d2 = sqrtf(exact_sq_lds<R>(xf[s], img, t.i2, h));
                        if (lexless(d2, t.i2, d1, t.i1)) { wp.d = d2; wp.i = t.i2; }
                    }
                    rescan = t.b1 == -INFINITY || !(wp.d < INFINITY);
                }
            } else {
                pos[s].merge_halves();
                neg[s].merge_halves();
                if (pos[s].d < KEY_VALID_F) {
                    wp.d = sqrtf(exact_sq_lds<R>(xf[s], img, pos[s].i, h)); wp.i = pos[s].i;
                    rescan |= pos[s].d == -INFINITY || !(wp.d < INFINITY);
                }
                if (neg[s].d < KEY_VALID_F) {
                    wn.d = sqrtf(exact_sq_lds<R>(xf[s], img, neg[s].i, h)); wn.i = neg[s].i;
                    rescan |= neg[s].d == -INFINITY || !(wn.d < INFINITY);
                }
            }
            if (rescan) {       // exact scan in index order (helpers.rs:55-77 in f32), rows from LDS
                wp = Best1();
                wn = Best1();
                for (uint32_t k = 0; k < a.K; ++k) {
                    const float d = sqrtf(exact_sq_lds<R>(xf[s], img, k, h));
                    const uint32_t ck = MATCHED ? ((const uint32_t *)(img + (size_t)(k >> 5) * Gm::TILE_FLOATS + Gm::W_FLOATS))[32 + (k & 31)] : 0u;
                    if (!MATCHED || ck == label[s]) wp.offer(d, k);
                    else wn.offer(d, k);
                }
            }
            if (live) {
                if (h == 0) {
                    if (a.out.idx) a.out.idx[sample[s]] = wp.i;
                    if (MATCHED && a.out.jneg) a.out.jneg[sample[s]] = wn.i;
                } else {
                    if (a.out.dist) a.out.dist[sample[s]] = wp.d;
                    if (MATCHED && a.out.dneg) a.out.dneg[sample[s]] = wn.d;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------ pipelined resident kernel
// Same contract as k_resident.  What it is built around (measured on MI355X):
//   * v_mfma_f32_32x32x2_f32 runs at the f32 vector rate and does not co-execute with VALU work of
//     the same SIMD: kernel time ~ 64 cycles per MFMA + 4 cycles per VALU instruction.  So the
//     screening of the scores is cut to 4 VALU per score (packed keys, below) instead of 8.
//   * every wave of the chip loading its samples and then computing, in lockstep, leaves HBM idle
//     during the search and the matrix pipe idle during the loads: the next batch's rows are
//     fetched into a second register set while the current batch is searched.
//   * the two waves of a SIMD do not share the matrix pipe evenly, so batches are claimed from a
//     queue instead of being dealt out.
//
// Packed keys.  A finished score is ||w_k||^2 - 2 x.w_k (the accumulator starts at the ||w_k||^2 column of
// the tile).  The low 4 mantissa bits of its f32 pattern are replaced by the row's position o among the
// lane's 16 rows of the tile, and the patterns are compared as floats: v_min_f32 / v_med3_f32 keep the
// best two keys of a lane (3 VALU per score, no shift of the scores to one sign), and which TILE they
// came from is settled once per tile by comparing the pair before and after.  Scores closer than the 4
// dropped bits are ordered by o, not by value or index: both candidates are re-measured exactly and
// compared like helpers.rs:70 where they are close, and exact ties between identical rows are settled by
// settle_ties.  Pad rows score PAD_SCORE; samples whose ||x||^2 is not safely finite (the only source of
// NaN / INF scores) go to the exact in-order scan.
// On entry:  acc = accumulator-init column of `tile`, a0 = this lane's chunk 0 of `tile`,
//            prev = finished scores of tile number `tprev_id` (class ids in tprev's tail).
// On exit:   acc = scores of (tile, xf), prev = init column of `next`, a0 = chunk 0 of `next`.
template <int R, bool MATCHED>
__device__ __forceinline__ void half_step(const float *tile, const float *next, const float *tprev, uint32_t tprev_id,
                                          uint32_t label, const f32x4 (&xf)[R / 8], f32x16 &acc, f32x16 &prev,
                                          f32x4 &a0, int row, int h, Keys2 &ku, KeysM &km)
{
    using Gm = Geom<R>;
    constexpr int NM = R / 2;                          // MFMAs of this half-step
    constexpr int SPAN = R >= 32 ? NM / 2 : NM;        // the 16 keys are spread over the first SPAN MFMAs
    constexpr int CI = NM / 2;                         // MFMA after which the next init column is fetched
    const float *rowp = tile + Gm::chunk_off(row, h);
    f32x4 acur = a0, anext = a0;
    f32x16 cnext;
    const Keys2 ou = ku;
    const KeysM om = km;
    // class ids of the screened tile's rows, one group of 4 rows ahead of the keys that use them
    const uint32_t *tcls = (const uint32_t *)(tprev + Gm::W_FLOATS) + 32 + 4 * h;
    u32x4 cgc = {0u, 0u, 0u, 0u}, cgn = {0u, 0u, 0u, 0u};
    if (MATCHED) cgn = *(const u32x4 *)tcls;
    static_for<0, NM>([&](auto ii) {
        constexpr int i = decltype(ii)::value, j = i / 4, e = i % 4;
        if constexpr (e == 0) {
            acur = anext;
            if constexpr (j + 1 < R / 8)
                anext = *(const f32x4 *)(rowp + 8 * (j + 1));
            else
                a0 = *(const f32x4 *)(next + Gm::chunk_off(row, h));
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[e], xf[j][e], acc, 0, 0, 0);
        if constexpr (!MATCHED) {
            if constexpr (i == 0) {                       // behind the first MFMA: prev's own MFMAs are long done
                screen16(prev, ku);
                settle_tiles(ku, ou, tprev_id);
            }
        } else {
            constexpr int o0 = i * 16 / SPAN < 16 ? i * 16 / SPAN : 16;
            constexpr int o1 = (i + 1) * 16 / SPAN < 16 ? (i + 1) * 16 / SPAN : 16;
            static_for<o0, o1>([&](auto oo) {
                constexpr int o = decltype(oo)::value;
                const uint32_t key = pack_key(prev[o], o);
                if constexpr ((o & 3) == 0) {
                    cgc = cgn;
                    if constexpr (o + 4 < 16) cgn = *(const u32x4 *)(tcls + 8 * ((o >> 2) + 1));
                }
                const bool match = cgc[o & 3] == label;           // helpers.rs:113 / :122
                const uint32_t kp = match ? key : KEY_NONE, kn = match ? KEY_NONE : key;
                km.p = kmin(km.p, kp);
                km.n = kmin(km.n, kn);
            });
            if constexpr (o0 < 16 && o1 == 16) {          // all 16 keys of the tile are in: settle the tiles
                km.tp = km.p != om.p ? tprev_id : om.tp;
                km.tn = km.n != om.n ? tprev_id : om.tn;
            }
        }
        if constexpr (i == CI) {
            const float *tail = next + Gm::W_FLOATS;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ci = *(const f32x4 *)(tail + 8 * g + 4 * h);
                cnext[4 * g + 0] = ci[0]; cnext[4 * g + 1] = ci[1];
                cnext[4 * g + 2] = ci[2]; cnext[4 * g + 3] = ci[3];
            }
        }
    });
    prev = cnext;
}

// Screening of the last tile of a batch (no MFMAs left to run beside it).
template <int R, bool MATCHED>
__device__ __forceinline__ void last_screen(const f32x16 &prev, const float *tprev, uint32_t tprev_id, uint32_t label, int h,
                                            Keys2 &ku, KeysM &km)
{
    const uint32_t *tcls = (const uint32_t *)(tprev + Geom<R>::W_FLOATS) + 32 + 4 * h;
    const Keys2 ou = ku;
    const KeysM om = km;
    if constexpr (!MATCHED) {
        screen16<1>(prev, ku);
        settle_tiles(ku, ou, tprev_id);
    } else {
        static_for<0, 16>([&](auto oo) {
            constexpr int o = decltype(oo)::value;
            const uint32_t key = pack_key(prev[o], o);
            const bool match = tcls[8 * (o >> 2) + (o & 3)] == label;
            const uint32_t kp = match ? key : KEY_NONE, kn = match ? KEY_NONE : key;
            km.p = kmin(km.p, kp);
            km.n = kmin(km.n, kn);
        });
        km.tp = km.p != om.p ? tprev_id : om.tp;
        km.tn = km.n != om.n ? tprev_id : om.tn;
    }
}

// Final step of a batch: the lane pair agrees on the candidates, re-measures them in the direct
// (x - w)^2 form from the LDS image, roots, compares like helpers.rs:70.
template <int R, bool MATCHED, bool ROOT = true>
__device__ __forceinline__ void finish_keys(const f32x4 (&xf)[R / 8], const float *img, uint32_t K, uint32_t label, int h,
                                            bool unsafe, float margin, const Keys2 &ku, const KeysM &km, Best1 &wp, Best1 &wn)
{
    using Gm = Geom<R>;
    // helpers.rs:23 (rooted) vs helpers.rs:44 (squared)
    auto dist = [&](uint32_t k) {
        const float d = exact_sq_lds<R>(xf, img, k, h);
        return ROOT ? sqrtf(d) : d;
    };
    if (!MATCHED) {
        // this lane's two candidates and the partner lane's, best two of the four by (key, index)
        const uint32_t k1 = key_row(ku.b1, ku.t1, h), k2 = key_row(ku.b2, ku.t2, h);
        const uint32_t pb1 = xor32(ku.b1), pb2 = xor32(ku.b2), pk1 = xor32(k1), pk2 = xor32(k2);
        uint32_t f, fi, sd, si;
        if (keyless(pb1, pk1, ku.b1, k1)) {
            f = pb1; fi = pk1;
            if (keyless(ku.b1, k1, pb2, pk2)) { sd = ku.b1; si = k1; } else { sd = pb2; si = pk2; }
        } else {
            f = ku.b1; fi = k1;
            if (keyless(pb1, pk1, ku.b2, k2)) { sd = pb1; si = pk1; } else { sd = ku.b2; si = k2; }
        }
        if (key_valid(f) && !unsafe) {
            const float d1 = dist(fi);
            wp.d = d1; wp.i = fi;
            // The runner-up is re-measured only where it can still win: a key differs from the true
            // ||w||^2 - 2 x.w by less than margin/2 (f32 accumulation of 2 x.w, rounding of the
            // init column, the 4 truncated bits), so a gap above `margin` decides the order already.
            // Decided per wave: the re-measurement is 80 VALU + 16 LDS reads for every lane or for none.
            const bool close = key_valid(sd) && !(__builtin_bit_cast(float, sd) - __builtin_bit_cast(float, f) > margin);
            if (__builtin_amdgcn_ballot_w64(close) != 0ull) {
                if (key_valid(sd)) {
                    const float d2 = dist(si);
                    if (lexless(d2, si, d1, fi)) { wp.d = d2; wp.i = si; }
                }
            }
            unsafe = !(wp.d < INFINITY);
        } else {
            unsafe = true;
        }
    } else {
        uint32_t kp = key_row(km.p, km.tp, h), kn = key_row(km.n, km.tn, h), p = km.p, n = km.n;
        const uint32_t pp = xor32(p), pkp = xor32(kp), pn = xor32(n), pkn = xor32(kn);
        if (keyless(pp, pkp, p, kp)) { p = pp; kp = pkp; }
        if (keyless(pn, pkn, n, kn)) { n = pn; kn = pkn; }
        if (key_valid(p)) {
            wp.d = dist(kp); wp.i = kp;
            unsafe |= !(wp.d < INFINITY);
        }
        if (key_valid(n)) {
            wn.d = dist(kn); wn.i = kn;
            unsafe |= !(wn.d < INFINITY);
        }
    }
    if (unsafe) {       // exact scan in index order (helpers.rs:55-77 in f32), rows from LDS
        wp = Best1();
        wn = Best1();
        for (uint32_t k = 0; k < K; ++k) {
            const float d = dist(k);
            const uint32_t ck = MATCHED ? ((const uint32_t *)(img + (size_t)(k >> 5) * Gm::TILE_FLOATS + Gm::W_FLOATS))[32 + (k & 31)] : 0u;
            if (!MATCHED || ck == label) wp.offer(d, k);
            else wn.offer(d, k);
        }
    }
}

// One wave = one tile of 32 samples per batch.  D == R (host checks); the image has an even number of
// tiles (the accumulators ping-pong).  a.ss_limit: ||x||^2 below which no score can overflow.
template <int R, bool MATCHED, bool DIAG = false, int NW = 8>
__global__ void __launch_bounds__(NW * 64, NW / 4) k_resident_pipe(const EngineArgs a)
{
    using Gm = Geom<R>;
    unsigned long long tm[4] = {0, 0, 0, 0}, tprev_ = 0;
    auto lap = [&](int slot) {                        // diagnostic build only
        if (DIAG) {
            const unsigned long long t = stamp();
            tm[slot] += t - tprev_;
            tprev_ = t;
        }
    };
    __shared__ __attribute__((aligned(16))) char smem[RES_BYTES];
    const unsigned long long r_entry = DIAG ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    const uint32_t ntiles = a.ngroups1 * Gm::G;       // even (host pads)

    {   // whole image -> LDS, linear LDS-DMA, 1 KiB per wave-instruction
        const uint32_t bytes = ntiles * Gm::TILE_BYTES;
        const char *src = (const char *)a.img1;
        for (uint32_t p = wave; p * 1024u < bytes; p += NW) {
            const uint32_t off = p * 1024u + lane * 16u;
            if (off < bytes)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                                 (__attribute__((address_space(3))) void *)(smem + p * 1024u), 16, 0, 0);
        }
        __syncthreads();
    }
    const float *img = (const float *)smem;
    unsigned long long t_begin = 0, r_begin = 0;
    if (DIAG) {
        tprev_ = t_begin = stamp();
        r_begin = __builtin_amdgcn_s_memrealtime();
    }

    // Batches are claimed, BLOCK at a time, by each wave from one global counter (zeroed before the
    // launch).  Static ownership fails twice here: the older wave of a SIMD wins the arbitration for the
    // matrix pipe, so equal shares leave the younger one alone with a quarter of its batches after its
    // partner has exited; and when another kernel (the RCCL gather of the previous chunk) holds some CUs
    // at launch, the workgroups that start late would still own a full share.  The returning atomic for
    // the next block is issued two batches before its value is needed.
    constexpr uint32_t BLOCK = 2;
    uint32_t pending = 0u;                             // lane 0: first batch of the next block (in flight)
    const uint32_t first_dynamic = gridDim.x * NW * BLOCK;     // every wave's first block is fixed
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wsometimes-uninitialized"
    auto request = [&]() {
        uint32_t fresh;                                // deliberately undefined off lane 0: no merge copy, so the
        if (lane == 0)                                 // compiler's wait for the atomic lands at the first use
            fresh = __hip_atomic_fetch_add(a.queue, BLOCK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pending = fresh;
    };
#pragma clang diagnostic pop
    uint64_t block0 = (uint64_t)(blockIdx.x * NW + (uint32_t)__builtin_amdgcn_readfirstlane(wave)) * BLOCK;
    uint32_t in_block = 0;                             // position of b inside the current block
    request();
    // the batch after (block0, in_block)
    auto successor = [&](uint64_t &nb0, uint32_t &nin) {
        if (in_block + 1 < BLOCK) {
            nb0 = block0;
            nin = in_block + 1;
        } else {
            uint32_t p = pending;
            asm volatile("" : "+v"(p));                // keeps the broadcast (and the wait for the atomic) here
            nb0 = first_dynamic + (uint32_t)__builtin_amdgcn_readfirstlane((int)p);   // added here: a VALU use next to the atomic would wait for it there
            nin = 0;
        }
    };
    // Rows come in through a buffer descriptor rebased per batch (wave-uniform, so the 4 GiB offset
    // range is per batch, not per matrix): rows past N read as zeros by the hardware range check, one
    // per-lane offset serves every chunk (immediate offsets), and nothing needs a 64-bit address.
    const uint32_t row_bytes = a.D * 4u;
    const uint32_t voff = (uint32_t)s_col * row_bytes + (uint32_t)h * 16u;
    f32x4 xn[R / 8];                                   // rows of the next batch, in flight
    uint32_t ln = 0u;
    auto records = [&](uint64_t bb, uint32_t unit) -> uint32_t {      // bytes from row bb*32 to the end, clamped
        const uint64_t first = bb * 32;
        if (first >= a.N) return 0u;
        const uint64_t left = (a.N - first) * unit;
        return left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)left;
    };
    auto fetch = [&](uint64_t bb) {
        const __amdgpu_buffer_rsrc_t rx =
            __builtin_amdgcn_make_buffer_rsrc((void *)(a.X + bb * 32 * a.D), 0, (int)records(bb, row_bytes), 0x00020000);
#pragma unroll
        for (int j = 0; j < R / 8; ++j)
            xn[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff + j * 32, 0, 0));
        if (MATCHED) {
            const __amdgpu_buffer_rsrc_t ry =
                __builtin_amdgcn_make_buffer_rsrc((void *)(a.y + bb * 32), 0, (int)records(bb, 4u), 0x00020000);
            ln = __builtin_amdgcn_raw_buffer_load_b32(ry, (uint32_t)s_col * 4u, 0, 0);
        }
    };
    uint64_t b = block0;
    fetch(b);

    while (b < a.nbatches) {
        uint64_t nb0;
        uint32_t nin;
        successor(nb0, nin);
        const uint64_t bnext = nb0 + nin;
        f32x4 xf[R / 8];
        using f32x2 = __attribute__((ext_vector_type(2))) float;
        f32x2 ss2 = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < R / 8; ++j) {
            xf[j] = xn[j];
            const f32x2 lo = {xf[j][0], xf[j][1]}, hi = {xf[j][2], xf[j][3]};
            ss2 = __builtin_elementwise_fma(lo, lo, ss2);
            ss2 = __builtin_elementwise_fma(hi, hi, ss2);
        }
        const uint32_t label = ln;
        float ss = ss2[0] + ss2[1];
        ss += xor32(ss);                               // ||x||^2, the same value in both lanes of the pair
        const bool unsafe = !(ss < a.ss_limit);        // NaN / INF / huge rows: exact scan at the end
        const float c = unsafe ? 0.0f : fmaf(ss, 1.00390625f, 1e-30f);
        lap(0);                                        // 0: delivery of this batch's rows
        // vmcnt counts in issue order: the atomic goes out only after this batch's rows have been waited
        // for, or that wait would include the atomic's own round trip
        asm volatile("" ::: "memory");
        if (nin == 0) request();                       // the block after the one just taken
        fetch(bnext);
        lap(1);                                        // 1: issue of the next batch's loads

        Keys2 ku;
        KeysM km;
        f32x16 acc0, acc1;
        f32x4 a0 = *(const f32x4 *)(img + Gm::chunk_off(s_col, h));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 ci = *(const f32x4 *)(img + Gm::W_FLOATS + 8 * g + 4 * h);
            acc0[4 * g + 0] = ci[0]; acc0[4 * g + 1] = ci[1]; acc0[4 * g + 2] = ci[2]; acc0[4 * g + 3] = ci[3];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = PAD_SCORE;             // "tile -1": its keys never win
        const float *tprev = img;
        uint32_t tid = 0;
        for (uint32_t t = 0; t < ntiles; t += 2) {
            const float *tile = img + (size_t)t * Gm::TILE_FLOATS;
            const float *tile1 = tile + Gm::TILE_FLOATS;
            const float *next = t + 2 < ntiles ? tile1 + Gm::TILE_FLOATS : img;
            // tile t computes into acc0 while tile t-1 (in acc1) is screened; acc1 leaves as the init column of t+1
            half_step<R, MATCHED>(tile, tile1, tprev, tid, label, xf, acc0, acc1, a0, s_col, h, ku, km);
            // tile t+1 computes into acc1 while tile t (in acc0) is screened; acc0 leaves as the init column of t+2
            half_step<R, MATCHED>(tile1, next, tile, t, label, xf, acc1, acc0, a0, s_col, h, ku, km);
            tprev = tile1;
            tid = t + 1;
        }
        last_screen<R, MATCHED>(acc1, tprev, tid, label, h, ku, km);
        lap(2);                                        // 2: tile loop

        Best1 wp, wn;
        finish_keys<R, MATCHED>(xf, img, a.K, label, h, unsafe, (c + a.wmax2) * 6.103515625e-05f, ku, km, wp, wn);
        settle_ties<MATCHED>(a, label, wp, wn);
        {   // lanes h = 0 store the index, lanes h = 1 the distance; rows past N are dropped by the range check
            const int rec = (int)records(b, 4u);
            const uint32_t so = (uint32_t)s_col * 4u;
            auto put = [&](void *base, uint32_t bits) {
                __builtin_amdgcn_raw_buffer_store_b32(bits, __builtin_amdgcn_make_buffer_rsrc(base, 0, rec, 0x00020000), so, 0, 0);
            };
            if (h == 0) {
                if (a.out.idx) put(a.out.idx + b * 32, wp.i);
                if (MATCHED && a.out.jneg) put(a.out.jneg + b * 32, wn.i);
            } else {
                if (a.out.dist) put(a.out.dist + b * 32, __builtin_bit_cast(uint32_t, wp.d));
                if (MATCHED && a.out.dneg) put(a.out.dneg + b * 32, __builtin_bit_cast(uint32_t, wn.d));
            }
        }
        lap(3);                                        // 3: batch epilogue
        b = bnext;
        block0 = nb0;
        in_block = nin;
    }
    if (DIAG && a.dbg && lane == 0) {
        unsigned long long *d = a.dbg + ((size_t)blockIdx.x * NW + wave) * 8;
        for (int i = 0; i < 4; ++i) d[i] = tm[i];
        d[0] = r_begin - r_entry;                          // image load, 100 MHz ticks
        d[4] = stamp() - t_begin;                          // shader cycles of the batch loop
        d[5] = __builtin_amdgcn_s_memrealtime() - r_begin; // the same span in 100 MHz ticks
        d[6] = r_begin;
        d[7] = r_begin + d[5];
    }
}

// ------------------------------------------------------------------------------ resident low-rank Omega kernel
// LiRaMLVQ-shaped models (Omega with at most 16 rows, src/liramlvq/liramlvq.rs:287-307 through
// helpers.rs:35-45): Omega AND the projected prototypes p_k = Omega w_k stay in LDS for the whole
// launch (config 4: 16.3 + 88 KiB), so a wave never meets a barrier after the image load.
//   * projection y = Omega x on v_mfma_f32_16x16x4_f32: 16 Omega rows on the M axis (a 32-row MFMA
//     would spend half of its rows on padding), 2 groups of 16 samples on the N axis, the 4 lane
//     groups share out the reduction index.  x is consumed straight out of the registers the loads
//     of the previous batch filled, and the next batch's loads go out as soon as it is projected.
//   * one lane <-> lane^16 exchange turns the 16x16 accumulators into the B fragment of the 32x32x2
//     search stage; lane half h then holds y[8h .. 8h+7], so the prototype image is built with
//     chunks 1 and 2 of every row swapped (image chunk 2j+h <- chunk 2h+j of p_k).
//   * the search is k_resident_pipe's (half_step / packed keys / exact re-measurement) at R = 16,
//     distance squared and not rooted (helpers.rs:44).
template <int R1, bool MATCHED, int NW = 8>
__global__ void __launch_bounds__(NW * 64, NW / 4) k_lowrank_pipe(const EngineArgs a)
{
    using Gs = Geom<16>;
    constexpr int ROWF = R1 + 4;                       // Omega row stride in LDS: rows start 4 banks apart
    constexpr int NJ = R1 / 16;                        // projection steps: 4 MFMAs per step and sample group
    __shared__ __attribute__((aligned(16))) char smem[RES_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    const int n16 = lane & 15, kg = lane >> 4;
    const uint32_t ntiles = a.ngroups2 * Gs::G;       // even (G = 8)
    const uint32_t p_bytes = ntiles * Gs::TILE_BYTES;

    {   // both images -> LDS, linear LDS-DMA, 1 KiB per wave-instruction
        auto to_lds = [&](uint32_t dst, const char *src, uint32_t bytes) {
            for (uint32_t p = wave; p * 1024u < bytes; p += NW) {
                const uint32_t off = p * 1024u + lane * 16u;
                if (off < bytes)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                                     (__attribute__((address_space(3))) void *)(smem + dst + p * 1024u), 16, 0, 0);
            }
        };
        to_lds(0u, (const char *)a.img2, p_bytes);
        to_lds(p_bytes, (const char *)a.img1, 16u * ROWF * 4u);
        __syncthreads();
    }
    const float *img = (const float *)smem;
    const float *om = (const float *)(smem + p_bytes) + n16 * ROWF + kg * 4;   // this lane's Omega row and k group

    // batch queue: as in k_resident_pipe
    constexpr uint32_t BLOCK = 2;
    uint32_t pending = 0u;
    const uint32_t first_dynamic = gridDim.x * NW * BLOCK;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wsometimes-uninitialized"
    auto request = [&]() {
        uint32_t fresh;
        if (lane == 0)
            fresh = __hip_atomic_fetch_add(a.queue, BLOCK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pending = fresh;
    };
#pragma clang diagnostic pop
    uint64_t block0 = (uint64_t)(blockIdx.x * NW + (uint32_t)__builtin_amdgcn_readfirstlane(wave)) * BLOCK;
    uint32_t in_block = 0;
    request();
    auto successor = [&](uint64_t &nb0, uint32_t &nin) {
        if (in_block + 1 < BLOCK) {
            nb0 = block0;
            nin = in_block + 1;
        } else {
            uint32_t p = pending;
            asm volatile("" : "+v"(p));
            nb0 = first_dynamic + (uint32_t)__builtin_amdgcn_readfirstlane((int)p);   // added here: a VALU use next to the atomic would wait for it there
            nin = 0;
        }
    };
    // lane (n16, kg) loads 16 B of row n16 (group 0) and row 16 + n16 (group 1) per step: chunk 4j + kg
    constexpr uint32_t row_bytes = R1 * 4u;
    const uint32_t voff0 = (uint32_t)n16 * row_bytes + (uint32_t)kg * 16u;
    const uint32_t voff1 = voff0 + 16u * row_bytes;
    f32x4 xn[2][NJ];
    uint32_t ln = 0u;
    auto records = [&](uint64_t bb, uint32_t unit) -> uint32_t {
        const uint64_t first = bb * 32;
        if (first >= a.N) return 0u;
        const uint64_t left = (a.N - first) * unit;
        return left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)left;
    };
    auto fetch = [&](uint64_t bb) {
        const __amdgpu_buffer_rsrc_t rx =
            __builtin_amdgcn_make_buffer_rsrc((void *)(a.X + bb * 32 * R1), 0, (int)records(bb, row_bytes), 0x00020000);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            xn[0][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff0 + j * 64, 0, 0));
            xn[1][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff1 + j * 64, 0, 0));
        }
        if (MATCHED) {
            const __amdgpu_buffer_rsrc_t ry =
                __builtin_amdgcn_make_buffer_rsrc((void *)(a.y + bb * 32), 0, (int)records(bb, 4u), 0x00020000);
            ln = __builtin_amdgcn_raw_buffer_load_b32(ry, (uint32_t)s_col * 4u, 0, 0);
        }
    };
    uint64_t b = block0;
    fetch(b);

    while (b < a.nbatches) {
        uint64_t nb0;
        uint32_t nin;
        successor(nb0, nin);
        const uint64_t bnext = nb0 + nin;

        // ---- projection: accA / accB = y of samples n16 / 16 + n16, rows 4 kg .. 4 kg + 3
        f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
        {
            f32x4 anext = *(const f32x4 *)om;
            static_for<0, NJ>([&](auto jj) {
                constexpr int j = decltype(jj)::value;
                const f32x4 acur = anext;
                if constexpr (j + 1 < NJ) anext = *(const f32x4 *)(om + 16 * (j + 1));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    accA = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[e], xn[0][j][e], accA, 0, 0, 0);
                    accB = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[e], xn[1][j][e], accB, 0, 0, 0);
                }
            });
        }
        const uint32_t label = ln;
        // x is dead: the next batch's rows go out now and arrive while this batch is searched.  The atomic
        // goes out after this batch's rows have been waited for (vmcnt counts in issue order).
        asm volatile("" ::: "memory");
        if (nin == 0) request();
        fetch(bnext);

        // ---- 16x16 accumulators -> B fragment of the 32x32x2 search: lane (s, h) gets y[8h .. 8h+7] of sample s
        f32x4 yf[2];
        {
            const bool up = (lane & 16) != 0;          // this lane's sample sits in group 1
            f32x4 own, recv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                own[r] = up ? accB[r] : accA[r];
                const float send = up ? accA[r] : accB[r];
                recv[r] = __shfl_xor(send, 16, 64);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                yf[0][r] = up ? recv[r] : own[r];
                yf[1][r] = up ? own[r] : recv[r];
            }
        }
        float ss = 0.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) ss = fmaf(yf[j][e], yf[j][e], ss);
        ss += xor32(ss);                               // ||y||^2, the same value in both lanes of the pair
        const bool unsafe = !(ss < a.ss_limit);        // NaN / INF / huge: exact scan at the end
        const float c = unsafe ? 0.0f : fmaf(ss, 1.00390625f, 1e-30f);

        // ---- search in the projected space (k_resident_pipe's tile loop at R = 16)
        Keys2 ku;
        KeysM km;
        f32x16 acc0, acc1;
        f32x4 a0 = *(const f32x4 *)(img + Gs::chunk_off(s_col, h));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 ci = *(const f32x4 *)(img + Gs::W_FLOATS + 8 * g + 4 * h);
            acc0[4 * g + 0] = ci[0]; acc0[4 * g + 1] = ci[1]; acc0[4 * g + 2] = ci[2]; acc0[4 * g + 3] = ci[3];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = PAD_SCORE;
        const float *tprev = img;
        uint32_t tid = 0;
        for (uint32_t t = 0; t < ntiles; t += 2) {
            const float *tile = img + (size_t)t * Gs::TILE_FLOATS;
            const float *tile1 = tile + Gs::TILE_FLOATS;
            const float *next = t + 2 < ntiles ? tile1 + Gs::TILE_FLOATS : img;
            half_step<16, MATCHED>(tile, tile1, tprev, tid, label, yf, acc0, acc1, a0, s_col, h, ku, km);
            half_step<16, MATCHED>(tile1, next, tile, t, label, yf, acc1, acc0, a0, s_col, h, ku, km);
            tprev = tile1;
            tid = t + 1;
        }
        last_screen<16, MATCHED>(acc1, tprev, tid, label, h, ku, km);

        Best1 wp, wn;
        finish_keys<16, MATCHED, false>(yf, img, a.K, label, h, unsafe, (c + a.wmax2) * 6.103515625e-05f, ku, km, wp, wn);
        settle_ties<MATCHED>(a, label, wp, wn);
        {
            const int rec = (int)records(b, 4u);
            const uint32_t so = (uint32_t)s_col * 4u;
            auto put = [&](void *base, uint32_t bits) {
                __builtin_amdgcn_raw_buffer_store_b32(bits, __builtin_amdgcn_make_buffer_rsrc(base, 0, rec, 0x00020000), so, 0, 0);
            };
            if (h == 0) {
                if (a.out.idx) put(a.out.idx + b * 32, wp.i);
                if (MATCHED && a.out.jneg) put(a.out.jneg + b * 32, wn.i);
            } else {
                if (a.out.dist) put(a.out.dist + b * 32, __builtin_bit_cast(uint32_t, wp.d));
                if (MATCHED && a.out.dneg) put(a.out.dneg + b * 32, __builtin_bit_cast(uint32_t, wn.d));
            }
        }
        b = bnext;
        block0 = nb0;
        in_block = nin;
    }
}

// ------------------------------------------------------------------------------ resident-prototype Omega kernel
// GMLVQ-shaped models (Omega with about as many rows as columns, src/gmlvq/gmlvq.rs:270-290 through
// helpers.rs:35-45): the projected prototypes p_k = Omega w_k stay in LDS for the whole launch as in
// k_resident_pipe (config 3: 134 KiB), which leaves no room for Omega (66 KiB).  Omega's A fragments
// come straight from L2 instead: the host lays Omega out as one 1 KiB piece per (tile, reduction step)
// -- 16 bytes per lane, exactly the fragment -- and every wave keeps a ring of DEPTH pieces in flight
// ahead of its MFMAs.  No LDS stage, no barrier after the image load: waves run independently and
// claim batches from the queue.  The projection's accumulators are the search stage's B fragment
// (engine.h), the search is k_resident_pipe's, distance squared and not rooted (helpers.rs:44).
template <int R, bool MATCHED, int NW = 8>
__global__ void __launch_bounds__(NW * 64, NW / 4) k_omega_pipe(const EngineArgs a)
{
    using Gs = Geom<R>;
    constexpr int NT = R / 32, NJ = R / 8, NSTEP = NT * NJ;   // Omega tiles, reduction steps per tile, pieces
    constexpr int DEPTH = 6;                                  // Omega pieces in flight per wave
    __shared__ __attribute__((aligned(16))) char smem[RES_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    const uint32_t ntiles = a.ngroups2 * Gs::G;       // even (host pads)

    {   // prototype image -> LDS, linear LDS-DMA, 1 KiB per wave-instruction
        const uint32_t bytes = ntiles * Gs::TILE_BYTES;
        const char *src = (const char *)a.img2;
        for (uint32_t p = wave; p * 1024u < bytes; p += NW) {
            const uint32_t off = p * 1024u + lane * 16u;
            if (off < bytes)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                                 (__attribute__((address_space(3))) void *)(smem + p * 1024u), 16, 0, 0);
        }
        __syncthreads();
    }
    const float *img = (const float *)smem;
    // Omega pieces: piece i = (tile i / NJ, step i % NJ), this lane's 16 bytes at i * 1024 + lane * 16
    const __amdgpu_buffer_rsrc_t rom = __builtin_amdgcn_make_buffer_rsrc((void *)a.img1, 0, NSTEP * 1024, 0x00020000);
    const uint32_t om_off = (uint32_t)lane * 16u;
    auto piece = [&](int i) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rom, om_off, i * 1024, 0));
    };

    // batch queue: as in k_resident_pipe
    constexpr uint32_t BLOCK = 2;
    uint32_t pending = 0u;
    const uint32_t first_dynamic = gridDim.x * NW * BLOCK;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wsometimes-uninitialized"
    auto request = [&]() {
        uint32_t fresh;
        if (lane == 0)
            fresh = __hip_atomic_fetch_add(a.queue, BLOCK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pending = fresh;
    };
#pragma clang diagnostic pop
    uint64_t block0 = (uint64_t)(blockIdx.x * NW + (uint32_t)__builtin_amdgcn_readfirstlane(wave)) * BLOCK;
    uint32_t in_block = 0;
    request();
    auto successor = [&](uint64_t &nb0, uint32_t &nin) {
        if (in_block + 1 < BLOCK) {
            nb0 = block0;
            nin = in_block + 1;
        } else {
            uint32_t p = pending;
            asm volatile("" : "+v"(p));
            nb0 = first_dynamic + (uint32_t)__builtin_amdgcn_readfirstlane((int)p);   // added here: a VALU use next to the atomic would wait for it there
            nin = 0;
        }
    };
    const uint32_t row_bytes = R * 4u;
    const uint32_t voff = (uint32_t)s_col * row_bytes + (uint32_t)h * 16u;
    f32x4 xn[R / 8];                                   // rows of the next batch, in flight
    uint32_t ln = 0u;
    auto records = [&](uint64_t bb, uint32_t unit) -> uint32_t {
        const uint64_t first = bb * 32;
        if (first >= a.N) return 0u;
        const uint64_t left = (a.N - first) * unit;
        return left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)left;
    };
    auto fetch = [&](uint64_t bb) {
        const __amdgpu_buffer_rsrc_t rx =
            __builtin_amdgcn_make_buffer_rsrc((void *)(a.X + bb * 32 * R), 0, (int)records(bb, row_bytes), 0x00020000);
#pragma unroll
        for (int j = 0; j < R / 8; ++j)
            xn[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff + j * 32, 0, 0));
        if (MATCHED) {
            const __amdgpu_buffer_rsrc_t ry =
                __builtin_amdgcn_make_buffer_rsrc((void *)(a.y + bb * 32), 0, (int)records(bb, 4u), 0x00020000);
            ln = __builtin_amdgcn_raw_buffer_load_b32(ry, (uint32_t)s_col * 4u, 0, 0);
        }
    };
    uint64_t b = block0;
    fetch(b);

    while (b < a.nbatches) {
        uint64_t nb0;
        uint32_t nin;
        successor(nb0, nin);
        const uint64_t bnext = nb0 + nin;

        // ---- projection y = Omega x: 32 Omega rows per tile, A fragments from the ring; the accumulators of
        //      tile t are chunks 2(4t + g) + h of y, i.e. fragment entries 4t + g of the search stage
        f32x4 yf[R / 8];
        {
            f32x4 ring[DEPTH];
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) ring[i] = piece(i);
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
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], xn[j][e], acc, 0, 0, 0);
                });
#pragma unroll
                for (int g = 0; g < 4; ++g) yf[4 * t + g] = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            });
        }
        const uint32_t label = ln;
        // x is dead: the next batch's rows go out now and arrive while this batch is searched.  The atomic
        // goes out after this batch's rows have been waited for (vmcnt counts in issue order).
        asm volatile("" ::: "memory");
        if (nin == 0) request();
        fetch(bnext);

        float ss = 0.0f;
#pragma unroll
        for (int j = 0; j < R / 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) ss = fmaf(yf[j][e], yf[j][e], ss);
        ss += xor32(ss);                               // ||y||^2, the same value in both lanes of the pair
        const bool unsafe = !(ss < a.ss_limit);        // NaN / INF / huge: exact scan at the end
        const float c = unsafe ? 0.0f : fmaf(ss, 1.00390625f, 1e-30f);

        // ---- search in the projected space (k_resident_pipe's tile loop)
        Keys2 ku;
        KeysM km;
        f32x16 acc0, acc1;
        f32x4 a0 = *(const f32x4 *)(img + Gs::chunk_off(s_col, h));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 ci = *(const f32x4 *)(img + Gs::W_FLOATS + 8 * g + 4 * h);
            acc0[4 * g + 0] = ci[0]; acc0[4 * g + 1] = ci[1]; acc0[4 * g + 2] = ci[2]; acc0[4 * g + 3] = ci[3];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = PAD_SCORE;
        const float *tprev = img;
        uint32_t tid = 0;
        for (uint32_t t = 0; t < ntiles; t += 2) {
            const float *tile = img + (size_t)t * Gs::TILE_FLOATS;
            const float *tile1 = tile + Gs::TILE_FLOATS;
            const float *next = t + 2 < ntiles ? tile1 + Gs::TILE_FLOATS : img;
            half_step<R, MATCHED>(tile, tile1, tprev, tid, label, yf, acc0, acc1, a0, s_col, h, ku, km);
            half_step<R, MATCHED>(tile1, next, tile, t, label, yf, acc1, acc0, a0, s_col, h, ku, km);
            tprev = tile1;
            tid = t + 1;
        }
        last_screen<R, MATCHED>(acc1, tprev, tid, label, h, ku, km);

        Best1 wp, wn;
        finish_keys<R, MATCHED, false>(yf, img, a.K, label, h, unsafe, (c + a.wmax2) * 6.103515625e-05f, ku, km, wp, wn);
        settle_ties<MATCHED>(a, label, wp, wn);
        {
            const int rec = (int)records(b, 4u);
            const uint32_t so = (uint32_t)s_col * 4u;
            auto put = [&](void *base, uint32_t bits) {
                __builtin_amdgcn_raw_buffer_store_b32(bits, __builtin_amdgcn_make_buffer_rsrc(base, 0, rec, 0x00020000), so, 0, 0);
            };
            if (h == 0) {
                if (a.out.idx) put(a.out.idx + b * 32, wp.i);
                if (MATCHED && a.out.jneg) put(a.out.jneg + b * 32, wn.i);
            } else {
                if (a.out.dist) put(a.out.dist + b * 32, __builtin_bit_cast(uint32_t, wp.d));
                if (MATCHED && a.out.dneg) put(a.out.dneg + b * 32, __builtin_bit_cast(uint32_t, wn.d));
            }
        }
        b = bnext;
        block0 = nb0;
        in_block = nin;
    }
}

// ------------------------------------------------------------------------------ projection kernel
// Y = X . A^T for a matrix A with `ntiles` tiles of 32 rows resident in LDS: A = Omega for project_omega
// (src/liramlvq/liramlvq.rs:429-446), A = Lambda = Omega^T Omega for project (src/gmlvq/gmlvq.rs:354-372,
// src/liramlvq/liramlvq.rs:401-419: the reference builds Lambda once and multiplies every sample by it).
// The projection stage of the search kernels with a store behind it: a wave owns 32 samples per batch,
// rows prefetched a batch ahead, lane (s, h) of a tile's accumulator holds Y[s][32 t + 8 g + 4 h .. + 3],
// stored as 16 bytes (the lane pair writes 32 adjacent bytes, the 4 groups of a tile a whole 128-byte line
// per sample).  Columns at or past out_dim and rows past N are not written (predicate / buffer range check).
struct ProjectArgs {
    const float *X;
    uint64_t N;
    const float *img;         // ntiles tiles of Geom<R>
    uint32_t ntiles;
    float *Y;                 // N x ldy
    uint32_t ldy, col0, out_dim;   // this launch writes columns [col0, min(col0 + 32 ntiles, out_dim))
    uint64_t nbatches;
    uint32_t *queue;
};

template <int R, int NW = 8>
__global__ void __launch_bounds__(NW * 64, NW / 4) k_project(const ProjectArgs a)
{
    using Gm = Geom<R>;
    // Register sets of 32 rows each.  R <= 128: two, worked alternately -- the loads that refill a set go
    // out right behind its last MFMA and have the other set's whole batch to land, with no copies between
    // sets.  R = 256: one set is all the register file holds at two waves per SIMD; the partner wave of
    // the SIMD covers the wait.
    constexpr int SETS = R <= 128 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) char smem[RES_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    {
        const uint32_t bytes = a.ntiles * Gm::TILE_BYTES;
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

    // blocks of SETS batches: a wave's first block is fixed, the others are claimed from the queue one
    // block ahead of use (the returning atomic is issued a block before its value is needed)
    uint32_t pending = 0u;
    const uint32_t first_dynamic = gridDim.x * NW * SETS;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wsometimes-uninitialized"
    auto request = [&]() {
        uint32_t fresh;                                // deliberately undefined off lane 0 (see k_resident_pipe)
        if (lane == 0)
            fresh = __hip_atomic_fetch_add(a.queue, (uint32_t)SETS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pending = fresh;
    };
#pragma clang diagnostic pop
    uint64_t blk = (uint64_t)(blockIdx.x * NW + (uint32_t)__builtin_amdgcn_readfirstlane(wave)) * SETS;
    request();

    const uint32_t row_bytes = R * 4u;
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
            __builtin_amdgcn_make_buffer_rsrc((void *)(a.X + bb * 32 * R), 0, (int)records(bb, row_bytes), 0x00020000);
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
    auto project_batch = [&](const f32x4 (&xs)[R / 8], uint64_t bb, bool claim) {
        const __amdgpu_buffer_rsrc_t ry =
            __builtin_amdgcn_make_buffer_rsrc((void *)(a.Y + bb * 32 * a.ldy), 0, (int)records(bb, ybytes), 0x00020000);
        for (uint32_t t = 0; t < a.ntiles; ++t) {
            const float *rowp = img + (size_t)t * Gm::TILE_FLOATS + Gm::chunk_off(s_col, h);
            f32x16 acc;
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
            // the claim for the block after next goes out once this block's rows have been waited for
            if (claim && t == 0) {
                asm volatile("" ::: "memory");
                request();
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint32_t col = a.col0 + 32u * t + 8u * g + 4u * h;
                if (col < a.out_dim) {
                    const f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, yoff + col * 4u, 0, 0);
                }
            }
        }
    };

    while (blk < a.nbatches) {
        uint32_t p = pending;
        asm volatile("" : "+v"(p));                    // keeps the broadcast (and the wait for the atomic) here
        const uint64_t nblk = first_dynamic + (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
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
constexpr uint32_t QUEUE_SLOTS = 16, QUEUE_STRIDE = 16;     // 16 counters, 64 bytes apart

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
    uint32_t *dup_first = nullptr, *dup_next = nullptr;     // identical search-space rows (null: none)
    uint32_t ngroups1 = 0, ngroups2 = 0, ldp = 0;
    const char *name = "mfma";
    hipError_t (*launch)(const EngineModel &, const EngineArgs &, bool matched, hipStream_t) = nullptr;
    int samples_per_batch = 0;
    float ss_limit = 0.0f, wmax2 = 0.0f;
};

namespace {

int pad_pow2(uint32_t v, int lo)
{
    int r = lo;
    while ((uint32_t)r < v) r <<= 1;
    return r;
}

// Tile images for `nrows` weight rows of length `len` (zero padded to R), scaled by `scale`;
// init[r] / cls[r] fill the tail; pad rows get (0 weights, pad_init, cls 0xFFFFFFFF).
template <int R>
std::vector<float> build_images(const float *rows, size_t nrows, size_t len, size_t ld, float scale, const float *init,
                                const uint32_t *cls, float pad_init, uint32_t &ngroups, size_t min_rows = 0)
{
    using Gm = Geom<R>;
    const size_t tiles = ((nrows > min_rows ? nrows : min_rows) + 31) / 32;
    ngroups = (uint32_t)((tiles + Gm::G - 1) / Gm::G);
    std::vector<float> img((size_t)ngroups * Gm::STAGE_FLOATS, 0.0f);
    for (size_t t = 0; t < (size_t)ngroups * Gm::G; ++t) {
        float *tile = img.data() + t * Gm::TILE_FLOATS;
        float *tail = tile + Gm::W_FLOATS;
        for (int r = 0; r < 32; ++r) {
            const size_t gr = t * 32 + r;
            uint32_t c = 0xFFFFFFFFu;
            if (gr < nrows) {
                for (size_t d = 0; d < len; ++d)
                    tile[Gm::chunk_off(r, (int)(d >> 2)) + (d & 3)] = scale * rows[gr * ld + d];
                tail[r] = init ? init[gr] : 0.0f;
                if (cls) c = cls[gr];
            } else {
                tail[r] = pad_init;
            }
            std::memcpy(&tail[32 + r], &c, 4);
        }
    }
    return img;
}

// Groups of prototypes with identical rows (-0 counted as +0): first / next tables for settle_ties, left
// empty when every row is unique.
void find_duplicates(const std::vector<float> &Q, uint32_t K, uint32_t L, std::vector<uint32_t> &first, std::vector<uint32_t> &next)
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
    bool any = false;
    std::vector<uint32_t> f(K), n(K);
    for (uint32_t i = 0; i < K;) {
        uint32_t j = i + 1;
        while (j < K && row_cmp(order[i], order[j]) == 0) ++j;
        for (uint32_t t = i; t < j; ++t) {
            f[order[t]] = order[i];
            n[order[t]] = t + 1 < j ? order[t + 1] : order[t];
        }
        any |= j - i > 1;
        i = j;
    }
    if (any) {
        first = std::move(f);
        next = std::move(n);
    }
}

template <class T>
hipError_t to_device(T **dst, const std::vector<T> &src)
{
    hipError_t e = hipMalloc((void **)dst, src.size() * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
}

// WPS = workgroups per CU = waves per SIMD the register budget is cut for (256-thread workgroups).
template <int MODE, int R1, int R2, int S, int WPS>
hipError_t launch_t(const EngineModel &m, const EngineArgs &a, bool matched, hipStream_t st)
{
    const uint64_t cap = 256ull * WPS;                                // 256 CUs x workgroups per CU
    const unsigned grid = (unsigned)(a.nbatches < cap ? a.nbatches : cap);
    const hipError_t e = hipMemsetAsync(a.queue, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    if (matched)
        hipLaunchKernelGGL((k_engine<MODE, R1, R2, S, true, WPS>), dim3(grid), dim3(256), 0, st, a, m.cls);
    else
        hipLaunchKernelGGL((k_engine<MODE, R1, R2, S, false, WPS>), dim3(grid), dim3(256), 0, st, a, m.cls);
    return hipGetLastError();
}

template <int MODE, int R1, int R2, int S, int WPS = (R1 >= 256 ? 1 : 2)>
void bind(EngineModel &m)
{
    m.launch = &launch_t<MODE, R1, R2, S, WPS>;
    m.samples_per_batch = 4 * S * 32;
}

template <int R, int S>
hipError_t launch_resident(const EngineModel &, const EngineArgs &a, bool matched, hipStream_t st)
{
    const uint64_t wgs = (a.nbatches + 7) / 8;                        // one batch of S*32 samples per wave
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);          // one 512-thread workgroup per CU
    if (matched)
        hipLaunchKernelGGL((k_resident<R, S, true>), dim3(grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((k_resident<R, S, false>), dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}

template <int R, bool DIAG = false>
hipError_t launch_resident_pipe(const EngineModel &m, const EngineArgs &a, bool matched, hipStream_t st)
{
    const uint64_t wgs = (a.nbatches + 7) / 8;
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);
    const hipError_t e = hipMemsetAsync(a.queue, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    if (DIAG && getenv("VQHIP_NW4"))
        hipLaunchKernelGGL((k_resident_pipe<R, false, true, 4>), dim3(grid), dim3(256), 0, st, a);
    else if (DIAG)
        hipLaunchKernelGGL((k_resident_pipe<R, false, true>), dim3(grid), dim3(512), 0, st, a);
    else if (matched)
        hipLaunchKernelGGL((k_resident_pipe<R, true>), dim3(grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((k_resident_pipe<R, false>), dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}

template <int R1>
hipError_t launch_lowrank(const EngineModel &m, const EngineArgs &a, bool matched, hipStream_t st)
{
    const uint64_t wgs = (a.nbatches + 7) / 8;
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);
    const hipError_t e = hipMemsetAsync(a.queue, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    if (matched)
        hipLaunchKernelGGL((k_lowrank_pipe<R1, true>), dim3(grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((k_lowrank_pipe<R1, false>), dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}

template <int R>
hipError_t launch_omega_pipe(const EngineModel &m, const EngineArgs &a, bool matched, hipStream_t st)
{
    const uint64_t wgs = (a.nbatches + 7) / 8;
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);
    const hipError_t e = hipMemsetAsync(a.queue, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    if (matched)
        hipLaunchKernelGGL((k_omega_pipe<R, true>), dim3(grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((k_omega_pipe<R, false>), dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}

int variant_env()
{
    const char *v = getenv("VQHIP_VARIANT");     // experiments only: picks among co-compiled tilings
    return v ? atoi(v) : 0;
}

template <int R1>
bool bind_euclid(EngineModel &m, std::vector<float> &img, const std::vector<float> &W, const std::vector<float> &norms,
                 const std::vector<uint32_t> &cls)
{
    if (m.R1 != (uint32_t)R1) return false;
    // an even number of tiles: the pipelined kernel ping-pongs two accumulators (pad rows score +INF)
    img = build_images<R1>(W.data(), m.K, m.D, m.D, -2.0f, norms.data(), cls.empty() ? nullptr : cls.data(), PAD_SCORE,
                           m.ngroups1, ((size_t)m.K + 63) / 64 * 64);
    bind<MODE_EUCLID, R1, R1, (R1 >= 128 ? 1 : 2)>(m);
    if constexpr (R1 < 256) {
        if ((size_t)m.ngroups1 * Geom<R1>::STAGE_BYTES <= (size_t)RES_BYTES && !getenv("VQHIP_NO_RESIDENT")) {
            constexpr int S = 2;
            m.launch = &launch_resident<R1, S>;
            m.samples_per_batch = S * 32;
            m.name = "mfma_euclid_lds";
            if (m.D == (uint32_t)R1 && !getenv("VQHIP_NO_PIPE")) {       // unpadded rows: the pipelined kernel
                m.samples_per_batch = 32;
                m.launch = variant_env() == 31 ? &launch_resident_pipe<R1, true> : &launch_resident_pipe<R1>;
                m.name = "mfma_euclid_lds_pipe";
            }
        }
    }
    return true;
}

template <int R1, int R2>
bool bind_omega(EngineModel &m, std::vector<float> &img1, std::vector<float> &img2, const std::vector<float> &Om,
                const std::vector<float> &P, const std::vector<float> &norms, const std::vector<uint32_t> &cls)
{
    if (m.R1 != (uint32_t)R1 || m.R2 != (uint32_t)R2) return false;
    img1 = build_images<R1>(Om.data(), m.M, m.D, m.D, 1.0f, nullptr, nullptr, 0.0f, m.ngroups1, R2 >= 32 ? R2 : 32);
    img2 = build_images<R2>(P.data(), m.K, m.M, m.M, -2.0f, norms.data(), cls.empty() ? nullptr : cls.data(), PAD_SCORE,
                            m.ngroups2);
    bind<MODE_OMEGA, R1, R2, 1>(m);
    return true;
}

// Omega with at most 16 rows, unpadded sample rows, both images resident in LDS: k_lowrank_pipe
template <int R1>
bool bind_lowrank(EngineModel &m, std::vector<float> &img1, std::vector<float> &img2, const std::vector<float> &Om,
                  const std::vector<float> &P, const std::vector<float> &norms, const std::vector<uint32_t> &cls)
{
    if (m.D != (uint32_t)R1 || m.M > 16 || getenv("VQHIP_NO_RESIDENT")) return false;
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
    if (pimg.size() * sizeof(float) + (size_t)16 * ROWF * sizeof(float) > (size_t)RES_BYTES) return false;
    img2 = std::move(pimg);
    m.ngroups2 = ngroups2;
    img1.assign((size_t)16 * ROWF, 0.0f);                              // Omega rows, row stride R1 + 4
    for (uint32_t r = 0; r < m.M; ++r) std::memcpy(&img1[(size_t)r * ROWF], &Om[(size_t)r * m.D], m.D * sizeof(float));
    m.ngroups1 = 1;
    m.R1 = R1;
    m.R2 = 16;
    m.launch = &launch_lowrank<R1>;
    m.samples_per_batch = 32;
    m.name = "mfma_omega_lowrank_lds";
    return true;
}

// Omega padded to R x R with D == R, projected prototypes resident in LDS, Omega from L2: k_omega_pipe
template <int R>
bool bind_omega_pipe(EngineModel &m, std::vector<float> &img1, std::vector<float> &img2, const std::vector<float> &Om,
                     const std::vector<float> &P, const std::vector<float> &norms, const std::vector<uint32_t> &cls)
{
    if (m.D != (uint32_t)R || m.R2 != (uint32_t)R || getenv("VQHIP_NO_RESIDENT")) return false;
    uint32_t ngroups2 = 0;
    std::vector<float> pimg = build_images<R>(P.data(), m.K, m.M, m.M, -2.0f, norms.data(), cls.empty() ? nullptr : cls.data(),
                                              PAD_SCORE, ngroups2, ((size_t)m.K + 63) / 64 * 64);
    if (pimg.size() * sizeof(float) > (size_t)RES_BYTES) return false;
    img2 = std::move(pimg);
    m.ngroups2 = ngroups2;
    // Omega as A-fragment pieces: piece (t, j), lane (row, h) = Omega[32 t + row][4 (2 j + h) .. + 3]
    constexpr int NT = R / 32, NJ = R / 8;
    img1.assign((size_t)NT * NJ * 64 * 4, 0.0f);
    for (int t = 0; t < NT; ++t)
        for (int j = 0; j < NJ; ++j)
            for (int lane = 0; lane < 64; ++lane) {
                const uint32_t row = 32u * t + (lane & 31), col = 4u * (2 * j + (lane >> 5));
                if (row < m.M) std::memcpy(&img1[(((size_t)t * NJ + j) * 64 + lane) * 4], &Om[(size_t)row * m.D + col], 4 * sizeof(float));
            }
    m.ngroups1 = 1;
    m.launch = &launch_omega_pipe<R>;
    m.samples_per_batch = 32;
    m.name = "mfma_omega_lds";
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
    bind<MODE_LOCAL, R1, R1, (R1 >= 256 ? 1 : 2)>(m);
    return true;
}

}  // namespace

EngineModel *engine_build(const std::vector<float> &W, uint32_t K, uint32_t D, const std::vector<float> &Om,
                          int omega_kind, uint32_t M, const std::vector<uint32_t> &cls, std::string &why,
                          hipError_t &herr)
{
    herr = hipSuccess;
    if (D % 4 != 0 || D > 256) {
        why = "sample length must be a multiple of 4 and at most 256";
        return nullptr;
    }
    if (omega_kind != OMEGA_NONE && M > 256) {
        why = "Omega has more than 256 rows";
        return nullptr;
    }
    EngineModel m;
    m.K = K; m.D = D; m.M = M;
    m.R1 = (uint32_t)pad_pow2(D, 16);
    m.has_cls = !cls.empty();
    std::vector<float> img1, img2, plain;
    std::vector<uint32_t> dup_first, dup_next;
    bool ok = false;

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
        // ||x||^2 below this: |2 x.w| <= 2 sqrt(ss wmax2) <= 1e18 and ||w||^2 + c stay far inside f32 range
        m.ss_limit = (float)std::min(1e36, 1e36 / (4.0 * std::max(wmax2, 1e-36)));
        m.wmax2 = (float)wmax2;
        if (omega_kind == OMEGA_NONE) {
            m.mode = MODE_EUCLID;
            m.name = "mfma_euclid";
            m.ldp = m.R1;
            ok = bind_euclid<16>(m, img1, W, norms, cls) || bind_euclid<32>(m, img1, W, norms, cls) ||
                 bind_euclid<64>(m, img1, W, norms, cls) || bind_euclid<128>(m, img1, W, norms, cls) ||
                 bind_euclid<256>(m, img1, W, norms, cls);
        } else {
            m.mode = MODE_OMEGA;
            m.name = "mfma_omega";
            m.R2 = (uint32_t)pad_pow2(M, 16);
            m.ldp = m.R2;
            ok = bind_lowrank<256>(m, img1, img2, Om, P, norms, cls) || bind_lowrank<128>(m, img1, img2, Om, P, norms, cls) ||
                 bind_lowrank<64>(m, img1, img2, Om, P, norms, cls) || bind_lowrank<32>(m, img1, img2, Om, P, norms, cls) ||
                 bind_omega_pipe<128>(m, img1, img2, Om, P, norms, cls) || bind_omega_pipe<64>(m, img1, img2, Om, P, norms, cls) ||
                 bind_omega_pipe<32>(m, img1, img2, Om, P, norms, cls) ||
                 bind_omega<128, 128>(m, img1, img2, Om, P, norms, cls) || bind_omega<256, 16>(m, img1, img2, Om, P, norms, cls) ||
                 bind_omega<16, 16>(m, img1, img2, Om, P, norms, cls) || bind_omega<32, 16>(m, img1, img2, Om, P, norms, cls) ||
                 bind_omega<32, 32>(m, img1, img2, Om, P, norms, cls) || bind_omega<64, 64>(m, img1, img2, Om, P, norms, cls);
        }
        // zero-padded plain rows for the exact re-measurement
        plain.assign((size_t)K * m.ldp, 0.0f);
        for (uint32_t k = 0; k < K; ++k) std::memcpy(&plain[(size_t)k * m.ldp], &Q[(size_t)k * L], L * sizeof(float));
        find_duplicates(Q, K, L, dup_first, dup_next);
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
        ok = bind_local<16>(m, img1, Om, negp, cls) || bind_local<32>(m, img1, Om, negp, cls) ||
             bind_local<64>(m, img1, Om, negp, cls) || bind_local<128>(m, img1, Om, negp, cls) ||
             bind_local<256>(m, img1, Om, negp, cls);
    }
    if (!ok) {
        why = "no engine instantiation for this (D, M) pair";
        return nullptr;
    }

    EngineModel *out = new EngineModel(m);
    herr = to_device(&out->img1, img1);
    if (herr == hipSuccess && !img2.empty()) herr = to_device(&out->img2, img2);
    if (herr == hipSuccess && !plain.empty()) herr = to_device(&out->plain, plain);
    if (herr == hipSuccess && !cls.empty()) herr = to_device(&out->cls, cls);
    if (herr == hipSuccess && !dup_first.empty()) herr = to_device(&out->dup_first, dup_first);
    if (herr == hipSuccess && !dup_next.empty()) herr = to_device(&out->dup_next, dup_next);
    if (herr == hipSuccess) herr = hipMalloc((void **)&out->queue, QUEUE_SLOTS * QUEUE_STRIDE * sizeof(uint32_t));
    if (herr != hipSuccess) {
        engine_free(out);
        return nullptr;
    }
    return out;
}

void engine_free(EngineModel *m)
{
    if (!m) return;
    (void)hipFree(m->img1);
    (void)hipFree(m->img2);
    (void)hipFree(m->plain);
    (void)hipFree(m->cls);
    (void)hipFree(m->queue);
    (void)hipFree(m->dup_first);
    (void)hipFree(m->dup_next);
    delete m;
}

const char *engine_name(const EngineModel *m) { return m->name; }

hipError_t engine_assign(const EngineModel *m, const float *dX, uint64_t N, const uint32_t *dy, Outputs out, hipStream_t s)
{
    if (N == 0) return hipSuccess;
    EngineArgs a;
    a.X = dX; a.N = N; a.D = m->D; a.y = dy;
    a.img1 = m->img1; a.ngroups1 = m->ngroups1;
    a.img2 = m->img2; a.ngroups2 = m->ngroups2;
    a.plain = m->plain; a.ldp = m->ldp;
    a.K = m->K; a.tiles_per_proto = m->tiles_per_proto;
    a.nbatches = (N + m->samples_per_batch - 1) / m->samples_per_batch;
    a.out = out;
    a.ss_limit = m->ss_limit;
    a.wmax2 = m->wmax2;
    a.dbg = nullptr;
    static unsigned long long *dbg_buf = nullptr;                     // diagnostic build only (VQHIP_VARIANT=14)
    if (variant_env() == 31) {
        if (!dbg_buf && hipMalloc((void **)&dbg_buf, 4096 * 4 * 8 * 8) != hipSuccess) dbg_buf = nullptr;
        a.dbg = dbg_buf;
    }
    a.queue = m->queue + (m->next_slot++ % QUEUE_SLOTS) * QUEUE_STRIDE;
    a.dup_first = m->dup_first;
    a.dup_next = m->dup_next;
    a.cls = m->cls;
    hipError_t e = m->launch(*m, a, dy != nullptr, s);
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
        if (variant_env() == 31) {
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

// ------------------------------------------------------------------------------ projection (host side)
struct ProjectModel {
    uint32_t rows = 0, D = 0, R = 0, ntiles = 0, tile_floats = 0, tiles_per_launch = 0;
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
    const hipError_t e = hipMemsetAsync(a.queue, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_project<R>), dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}

template <int R>
bool bind_project(ProjectModel &m, std::vector<float> &img, const std::vector<float> &A)
{
    if (m.D != (uint32_t)R) return false;
    uint32_t ngroups = 0;
    img = build_images<R>(A.data(), m.rows, m.D, m.D, 1.0f, nullptr, nullptr, 0.0f, ngroups);
    m.R = R;
    m.ntiles = (m.rows + 31) / 32;
    m.tile_floats = Geom<R>::TILE_FLOATS;
    m.tiles_per_launch = (uint32_t)(RES_BYTES / Geom<R>::TILE_BYTES);
    m.launch = &launch_project_t<R>;
    return true;
}
}  // namespace

ProjectModel *project_build(const std::vector<float> &A, uint32_t rows, uint32_t D, hipError_t &herr)
{
    herr = hipSuccess;
    ProjectModel m;
    m.rows = rows;
    m.D = D;
    std::vector<float> img;
    if (!(bind_project<32>(m, img, A) || bind_project<64>(m, img, A) || bind_project<128>(m, img, A) || bind_project<256>(m, img, A)))
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
        a.X = dX; a.N = N;
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

}  // namespace vq
