// Device-side building blocks shared by the MFMA kernels of libvqhip (gfx950): fragment types, the kernel
// argument block, running minima, packed screening keys, the tile MFMA loop, exact re-measurement, and
// the pipelined tile step of the LDS-resident kernels.  Geometry: engine.h.  Included by the kernels_*.hip
// translation units only; everything here is __device__ __forceinline__ (or constexpr) in an unnamed namespace.
#pragma once
#include <cmath>
#include <cstdint>
#include <type_traits>

#include "engine.h"
#include "mfma_host.h"
#include "vq_common.h"

namespace vq {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f32x16 = __attribute__((ext_vector_type(16))) float;

struct EngineArgs {
    const float *X;           // N x D samples (row stride D)
    uint64_t N;
    uint32_t D;               // actual sample length (<= R1; any value: rows need only 4-byte alignment)
    const uint32_t *y;        // sample class ids (matched) or null
    const float *img1;        // stage-1 groups: prototypes (EUCLID), Omega rows (OMEGA), Omega_j rows (LOCAL)
    uint32_t ngroups1;
    const float *img2;        // stage-2 groups: projected prototypes (OMEGA)
    uint32_t ngroups2;
    const float *plain;       // K x ldp rows for the exact re-measurement (W, or P = W.Omega^T), zero padded
    uint32_t ldp;
    uint32_t K;               // actual number of prototypes
    uint32_t tiles_per_proto; // LOCAL: 32-row tiles per Omega_j
    uint32_t res_tiles;       // resident kernels: tiles of 32 prototypes actually searched, ceil(K / 32)
    uint64_t nbatches;
    Outputs out;
    float wmax2;              // largest ||w_k||^2 (screening error bound), in the centred coordinates when mu is set
    float mscale;             // screening margin = mscale * (||x||^2 + wmax2): see screen_margin
    const float *mu;          // centre subtracted from every sample before anything else (R1 floats, zero padded), or null
    float ss_limit;           // resident_pipe: ||x||^2 below which no screening score can overflow
    uint32_t bf;              // resident_pipe: screening products on the bf16 matrix cores (search_tiles_bf); mscale is then its margin
    const float *imgbf;       // bf16 pieces of the first nbf tiles of the search image (as many as LDS has room for), or null
    uint32_t nbf;
    uint32_t bfimg;           // the search image itself holds bf16 pieces (every tile); exact rows come from `plain`
    uint32_t pj;              // Omega kernels: img1 holds Omega as three bf16 pieces per value, the projection runs on the bf16 matrix cores
    uint32_t *queue;          // k_engine: batch counter, zeroed before every launch
    uint32_t queue_zeroed;    // the caller's memset has covered the counters already (zero_queue, mfma_host.h)
    const uint32_t *cls;      // K class ids (labelled models)
    unsigned long long *dbg;  // diagnostic build of k_resident_pipe: per-wave cycle sums of the loop phases
    // Samples the screening cannot settle (crowd_mark below).  First pass: crowd_mask = three bit planes per tile of 32
    // samples (zeroed before the launch), or null: such samples take the exact scan inside their wave.  Finish pass
    // (FINISH kernels): list = their ids, packed by k_crowd_compact, list_count = how many.
    uint32_t *crowd_mask;
    const uint32_t *list;
    const uint32_t *list_count;
};

namespace {

// ------------------------------------------------------------------------------ small helpers
// The value of lane ^ 32 (the other lane of a sample's pair).  v_permlane32_swap_b32 (gfx950) swaps the upper half of one
// register with the lower half of another: of two copies of v one ends up holding the lower half twice, the other the
// upper half twice -- two vector instructions and no trip through the LDS queue (ds_bpermute_b32: the epilogues' twelve
// exchanges per batch each waited for it).
__device__ __forceinline__ uint32_t xor32(uint32_t v)
{
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // r[0]: v of lanes 0..31 in both halves, r[1]: of 32..63
    return (threadIdx.x & 32u) ? (uint32_t)r[0] : (uint32_t)r[1];
}
__device__ __forceinline__ float xor32(float v) { return __builtin_bit_cast(float, xor32(__builtin_bit_cast(uint32_t, v))); }

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
    float b1, b2, b3;                                   // b3: third best value (the crowding guard, see finish_keys)
    uint32_t i1, i2;
    __device__ Top2() : b1(INFINITY), b2(INFINITY), b3(INFINITY), i1(0u), i2(0u) {}
    __device__ __forceinline__ void offer(float v, uint32_t k)
    {
        const bool c1 = v < b1, c2 = v < b2;
        b3 = c2 ? b2 : (v < b3 ? v : b3);
        b2 = c1 ? b1 : (c2 ? v : b2);
        i2 = c1 ? i1 : (c2 ? k : i2);
        b1 = c1 ? v : b1;
        i1 = c1 ? k : i1;
    }
    // union with the partner lane's pair (the other 16 rows of every tile)
    __device__ __forceinline__ void merge_halves()
    {
        const float o1 = xor32(b1), o2 = xor32(b2), o3 = xor32(b3);
        const uint32_t oi1 = xor32(i1), oi2 = xor32(i2);
        // third smallest of the two sorted triples
        b3 = fminf(fminf(b3, o3), fminf(fmaxf(b1, o2), fmaxf(b2, o1)));
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

// ------------------------------------------------------------------------------ batch queue of the wave-per-batch kernels
// Every wave works through blocks of BLOCK consecutive batches of 32 samples.  Its first block is fixed (its
// wave number); the others are claimed from counters in global memory, zeroed before the launch.
//   * Dynamic, because static ownership fails twice: the older wave of a SIMD wins the arbitration for the
//     matrix pipe, so equal shares leave the younger one alone at the end; and when another kernel (the RCCL
//     gather of the previous chunk) holds some CUs at launch, late workgroups would still own a full share.
//   * QUEUE_GROUPS counters, not one: returning device-scope atomics on one address complete at about 80 per
//     microsecond on MI355X, and short batches (few prototypes) need more -- one counter capped every such
//     launch at 2.7 TB/s, whatever its loads looked like.  Wave w starts on counter w % GROUPS; claim v of
//     counter g is block nwaves + v * GROUPS + g.  A wave whose counter runs dry walks the others.
//   * One block ahead: the atomic for the block after next is issued when a block is taken, and its result is
//     first touched by the broadcast that needs it (nothing next to the atomic may use it: the compiler would
//     wait for the round trip there).
template <uint32_t BLOCK>
struct BlockQueue {
    uint32_t *ctr;
    uint32_t group, first_dynamic, pending;
    uint64_t nblocks;
    int lane;
    __device__ __forceinline__ BlockQueue(uint32_t *counters, int lane_, uint32_t wave_id, uint32_t nwaves, uint64_t nbatches)
        : ctr(counters), group(wave_id % QUEUE_GROUPS), first_dynamic(nwaves), pending(0u),
          nblocks((nbatches + BLOCK - 1) / BLOCK), lane(lane_) {}
    __device__ __forceinline__ void request()
    {
        // Off lane 0 the value is never looked at (take() broadcasts lane 0), but it has to be SOMETHING: an empty asm
        // defines it without an instruction.  Lane 0's atomic then overwrites its own lane of the same register, so no
        // copy that would have to wait for the atomic's round trip sits between here and the first use.
        uint32_t fresh;
        asm volatile("" : "=v"(fresh));
        if (lane == 0)
            fresh = __hip_atomic_fetch_add(ctr + group * QUEUE_GROUP_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pending = fresh;
    }
    __device__ __forceinline__ uint64_t block_of(uint32_t claim) const { return first_dynamic + (uint64_t)claim * QUEUE_GROUPS + group; }
    // First batch of the block claimed by the last request() (wave-uniform); >= nbatches when nothing is left.
    __device__ __forceinline__ uint64_t take()
    {
        uint32_t p = pending;
        asm volatile("" : "+v"(p));                    // keeps the broadcast (and the wait for the atomic) here
        uint64_t id = block_of((uint32_t)__builtin_amdgcn_readfirstlane((int)p));
        for (uint32_t tries = 1; tries < QUEUE_GROUPS && id >= nblocks; ++tries) {     // the end of a launch only
            group = (group + 1) % QUEUE_GROUPS;
            uint32_t v = 0u;
            if (lane == 0) v = __hip_atomic_fetch_add(ctr + group * QUEUE_GROUP_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            id = block_of((uint32_t)__builtin_amdgcn_readfirstlane((int)v));
        }
        return id * BLOCK;
    }
};

// ---- packed screening keys (what they are and why: the comment above k_resident_pipe)
// Keys are f32 bit patterns and are compared AS FLOATS (v_min_f32 / v_med3_f32), so a score may have
// either sign and the accumulators start at the plain ||w_k||^2 column.  Nothing that is screened may be
// a NaN pattern (v_med3_f32 returns the minimum when it sees one): pad rows and the "tile before the
// first" score PAD_SCORE, finite and above every real score (< 1e38 by ss_limit), instead of +INF,
// whose pattern with position bits set is a NaN.  NaN / INF scores come only from samples that are
// sent to the exact scan anyway.
constexpr uint32_t KEY_NONE = 0x7F800000u;           // +INF: "no candidate yet"
constexpr float KEY_VALID_F = 1.0e38f;               // keys of real prototypes are below this
// The THIRD best key (value only) is the crowding guard: the two best are re-measured exactly, and that settles the
// sample only if no other prototype's key lies within the screening error of the best one (finish_keys).
struct Keys2 {                       // unlabelled search: best two keys and their tiles, third best key
    uint32_t b1 = KEY_NONE, b2 = KEY_NONE, t1 = 0u, t2 = 0u;
    uint32_t b3 = KEY_NONE;
};
struct KeysM {                       // labelled search: best two keys with cls == y, best two with cls != y, their tiles; third best keys
    uint32_t p = KEY_NONE, n = KEY_NONE, tp = 0u, tn = 0u;
    uint32_t p2 = KEY_NONE, n2 = KEY_NONE, tp2 = 0u, tn2 = 0u;
    uint32_t p3 = KEY_NONE, n3 = KEY_NONE;
};
__device__ __forceinline__ bool key_valid(uint32_t key) { return __builtin_bit_cast(float, key) < KEY_VALID_F; }

// Wait states between an MFMA that has just written `acc` and inline asm that reads it (the compiler pads hazards only
// for instructions it knows): tied to the accumulator, so the statement stays behind the MFMA and in front of the readers.
__device__ __forceinline__ void settle_mfma(f32x16 &acc) { asm volatile("s_nop 15\n\ts_nop 1" : "+v"(acc)); }

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
__device__ __forceinline__ uint32_t kmed3(uint32_t x, uint32_t y, uint32_t z)        // median as floats
{
    uint32_t r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}
// One key offered to the labelled search: kp is the key where the row's class matches the sample's label (else
// KEY_NONE), kn the other way round; each group keeps its best two (the runner-up is re-measured where it is close).
__device__ __forceinline__ void offer_m(KeysM &km, uint32_t kp, uint32_t kn)
{
    km.p3 = kmed3(km.p2, km.p3, kp);
    km.p2 = kmed3(km.p, km.p2, kp);
    km.p = kmin(km.p, kp);
    km.n3 = kmed3(km.n2, km.n3, kn);
    km.n2 = kmed3(km.n, km.n2, kn);
    km.n = kmin(km.n, kn);
}
// The 16 scores of one finished tile offered to the lane's best two keys: position bits in, median, minimum
// and the third best key (4 VALU per score).  One asm statement: the compiler pads a wait state behind every inline-asm result
// that the next instruction reads, and the chain b1 -> b1 would collect one per score.  PRE_NOPS: wait
// states for accumulators that the MFMA right in front of the statement has written.
// The nops of the PRE_NOPS form are the wait states between an MFMA that has just written the scores and their first
// reader, and they are part of the SAME asm statement as the readers: the compiler does not count inline asm as a
// reader of MFMA results (no hazard padding of its own), and a separate nop statement is free to drift away from them.
#define VQ_SCREEN16_TEXT \
    "v_and_or_b32 %[t0], %[s0], %[m], 0\n\t" \
    "v_and_or_b32 %[t1], %[s1], %[m], 1\n\t" \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t0]\n\t" \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t" \
    "v_min_f32 %[b1], %[b1], %[t0]\n\t" \
    "v_and_or_b32 %[t0], %[s2], %[m], 2\n\t" \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t1]\n\t" \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t1]\n\t" \
    "v_min_f32 %[b1], %[b1], %[t1]\n\t" \
    "v_and_or_b32 %[t1], %[s3], %[m], 3\n\t" \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t0]\n\t" \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t" \
    "v_min_f32 %[b1], %[b1], %[t0]\n\t" \
    "v_and_or_b32 %[t0], %[s4], %[m], 4\n\t" \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t1]\n\t" \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t1]\n\t" \
    "v_min_f32 %[b1], %[b1], %[t1]\n\t" \
    "v_and_or_b32 %[t1], %[s5], %[m], 5\n\t" \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t0]\n\t" \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t" \
    "v_min_f32 %[b1], %[b1], %[t0]\n\t" \
    "v_and_or_b32 %[t0], %[s6], %[m], 6\n\t" \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t1]\n\t" \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t1]\n\t" \
    "v_min_f32 %[b1], %[b1], %[t1]\n\t" \
    "v_and_or_b32 %[t1], %[s7], %[m], 7\n\t" \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t0]\n\t" \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t" \
    "v_min_f32 %[b1], %[b1], %[t0]\n\t" \
    "v_and_or_b32 %[t0], %[s8], %[m], 8\n\t" \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t1]\n\t" \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t1]\n\t" \
    "v_min_f32 %[b1], %[b1], %[t1]\n\t" \
    "v_and_or_b32 %[t1], %[s9], %[m], 9\n\t" \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t0]\n\t" \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t" \
    "v_min_f32 %[b1], %[b1], %[t0]\n\t" \
    "v_and_or_b32 %[t0], %[s10], %[m], 10\n\t" \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t1]\n\t" \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t1]\n\t" \
    "v_min_f32 %[b1], %[b1], %[t1]\n\t" \
    "v_and_or_b32 %[t1], %[s11], %[m], 11\n\t" \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t0]\n\t" \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t" \
    "v_min_f32 %[b1], %[b1], %[t0]\n\t" \
    "v_and_or_b32 %[t0], %[s12], %[m], 12\n\t" \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t1]\n\t" \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t1]\n\t" \
    "v_min_f32 %[b1], %[b1], %[t1]\n\t" \
    "v_and_or_b32 %[t1], %[s13], %[m], 13\n\t" \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t0]\n\t" \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t" \
    "v_min_f32 %[b1], %[b1], %[t0]\n\t" \
    "v_and_or_b32 %[t0], %[s14], %[m], 14\n\t" \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t1]\n\t" \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t1]\n\t" \
    "v_min_f32 %[b1], %[b1], %[t1]\n\t" \
    "v_and_or_b32 %[t1], %[s15], %[m], 15\n\t" \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t0]\n\t" \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t" \
    "v_min_f32 %[b1], %[b1], %[t0]\n\t" \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t1]\n\t" \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t1]\n\t" \
    "v_min_f32 %[b1], %[b1], %[t1]\n\t" \
    ""
template <int PRE_NOPS = 0>
__device__ __forceinline__ void screen16(const f32x16 &sc, Keys2 &ku)
{
    uint32_t t0, t1;
    if constexpr (PRE_NOPS > 0)
        asm("s_nop 15\n\ts_nop 1\n\t" VQ_SCREEN16_TEXT
        : [b1] "+v"(ku.b1), [b2] "+v"(ku.b2), [b3] "+v"(ku.b3), [t0] "=&v"(t0), [t1] "=&v"(t1)
        : [s0] "v"(sc[0]), [s1] "v"(sc[1]), [s2] "v"(sc[2]), [s3] "v"(sc[3]), [s4] "v"(sc[4]), [s5] "v"(sc[5]), [s6] "v"(sc[6]), [s7] "v"(sc[7]), [s8] "v"(sc[8]), [s9] "v"(sc[9]), [s10] "v"(sc[10]), [s11] "v"(sc[11]), [s12] "v"(sc[12]), [s13] "v"(sc[13]), [s14] "v"(sc[14]), [s15] "v"(sc[15]), [m] "s"(0xFFFFFFF0u));
    else
        asm(VQ_SCREEN16_TEXT
        : [b1] "+v"(ku.b1), [b2] "+v"(ku.b2), [b3] "+v"(ku.b3), [t0] "=&v"(t0), [t1] "=&v"(t1)
        : [s0] "v"(sc[0]), [s1] "v"(sc[1]), [s2] "v"(sc[2]), [s3] "v"(sc[3]), [s4] "v"(sc[4]), [s5] "v"(sc[5]), [s6] "v"(sc[6]), [s7] "v"(sc[7]), [s8] "v"(sc[8]), [s9] "v"(sc[9]), [s10] "v"(sc[10]), [s11] "v"(sc[11]), [s12] "v"(sc[12]), [s13] "v"(sc[13]), [s14] "v"(sc[14]), [s15] "v"(sc[15]), [m] "s"(0xFFFFFFF0u));
}
// ONE of a tile's 16 scores (register O) offered the same way: screen16 cut into pieces that sit between the MFMAs of the
// next tile (pieces_step).  The keys of a tile are distinct patterns (position bits), so the best three do not depend on
// the order they are offered in.
template <int O, int PRE_NOPS>
__device__ __forceinline__ void screen1(const f32x16 &sc, Keys2 &ku)
{
    uint32_t t0;
#define VQ_SCREEN1_TEXT                                \
    "v_and_or_b32 %[t0], %[s0], %[m], %[o0]\n\t"       \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t0]\n\t"        \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t"        \
    "v_min_f32 %[b1], %[b1], %[t0]\n\t"
    if constexpr (PRE_NOPS > 0)
        asm volatile("s_nop 15\n\ts_nop 1\n\t" VQ_SCREEN1_TEXT
                     : [b1] "+v"(ku.b1), [b2] "+v"(ku.b2), [b3] "+v"(ku.b3), [t0] "=&v"(t0)
                     : [s0] "v"(sc[O]), [m] "s"(0xFFFFFFF0u), [o0] "n"(O));
    else
        asm volatile(VQ_SCREEN1_TEXT
                     : [b1] "+v"(ku.b1), [b2] "+v"(ku.b2), [b3] "+v"(ku.b3), [t0] "=&v"(t0)
                     : [s0] "v"(sc[O]), [m] "s"(0xFFFFFFF0u), [o0] "n"(O));
#undef VQ_SCREEN1_TEXT
}
// After a tile's scores went in: which tile do the best two keys come from now
__device__ __forceinline__ void settle_tiles(Keys2 &ku, const Keys2 &before, uint32_t tile_id)
{
    const bool c1 = ku.b1 != before.b1, c2 = ku.b2 != before.b2;
    const uint32_t demoted = ku.b2 == before.b1 ? before.t1 : tile_id;
    ku.t2 = c1 ? demoted : (c2 ? tile_id : before.t2);
    ku.t1 = c1 ? tile_id : before.t1;
}

__device__ __forceinline__ void settle_tiles_m(KeysM &km, const KeysM &before, uint32_t tile_id)
{
    {
        const bool c1 = km.p != before.p, c2 = km.p2 != before.p2;
        const uint32_t demoted = km.p2 == before.p ? before.tp : tile_id;
        km.tp2 = c1 ? demoted : (c2 ? tile_id : before.tp2);
        km.tp = c1 ? tile_id : before.tp;
    }
    {
        const bool c1 = km.n != before.n, c2 = km.n2 != before.n2;
        const uint32_t demoted = km.n2 == before.n ? before.tn : tile_id;
        km.tn2 = c1 ? demoted : (c2 ? tile_id : before.tn2);
        km.tn = c1 ? tile_id : before.tn;
    }
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

// The lane pair (h = 0 / 1) of a sample agrees on the best two of its four candidates, by (key, prototype index):
// (f, fi) the best key and its prototype, (sd, si) the runner-up.
__device__ __forceinline__ void merge_pair(uint32_t b1, uint32_t t1, uint32_t b2, uint32_t t2, int h, uint32_t &f, uint32_t &fi,
                                           uint32_t &sd, uint32_t &si)
{
    const uint32_t k1 = key_row(b1, t1, h), k2 = key_row(b2, t2, h);
    const uint32_t pb1 = xor32(b1), pb2 = xor32(b2), pk1 = xor32(k1), pk2 = xor32(k2);
    if (keyless(pb1, pk1, b1, k1)) {
        f = pb1; fi = pk1;
        if (keyless(b1, k1, pb2, pk2)) { sd = b1; si = k1; } else { sd = pb2; si = pk2; }
    } else {
        f = b1; fi = k1;
        if (keyless(pb1, pk1, b2, k2)) { sd = pb1; si = pk1; } else { sd = b2; si = k2; }
    }
}

// Third smallest key of the lane pair's two sorted triples (value only; +INF while fewer than three were offered)
__device__ __forceinline__ float pair_third(uint32_t b1, uint32_t b2, uint32_t b3)
{
    const float a1 = __builtin_bit_cast(float, b1), a2 = __builtin_bit_cast(float, b2), a3 = __builtin_bit_cast(float, b3);
    const float c1 = xor32(a1), c2 = xor32(a2), c3 = xor32(a3);
    return fminf(fminf(a3, c3), fminf(fmaxf(a1, c2), fmaxf(a2, c1)));
}

// How far two screening keys may lie apart while the order of the true scores ||w||^2 - 2 x.w is still open:
// twice the error of one key, which is the f32 accumulation of the fma chain (bounded by gamma_L * (||w||^2 +
// 2 ||w|| ||x||) <= gamma_L * (2 wmax2 + ss)) plus the 4 mantissa bits the position replaces (2^-19 of the same).
// mscale (host, engine_build) = max(2^-14, 4 (8 sqrt(L) 2^-24 + 2^-19)) for a reduction of length L: above the
// worst-case bound up to L = 200, a many-sigma bound on the accumulated rounding beyond.
__device__ __forceinline__ float screen_margin(float ss, const EngineArgs &a)
{
    return (fmaf(ss, 1.00390625f, 1e-30f) + a.wmax2) * a.mscale;
}

// Chunk c = 2j + h of a row of D floats fetched as 16 bytes: values at or past D belong to the next row (or are
// range-check zeros).  The weight images are zero there, but 0 x INF is NaN, so they are cleared.  D need not
// be a multiple of 4 (rows are then only 4-byte aligned, which buffer loads accept).
__device__ __forceinline__ f32x4 mask_chunk(f32x4 v, uint32_t c, uint32_t D)
{
    const uint32_t first = 4u * c;
    if (first + 4u <= D) return v;
    return f32x4{first + 0u < D ? v[0] : 0.f, first + 1u < D ? v[1] : 0.f, first + 2u < D ? v[2] : 0.f, 0.f};
}

// 16 bytes of a row that is only known to be 4-byte aligned (gathered rows of the finish pass)
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ f32x4 load16(const float *p) { return *(const f32x4_a4 *)p; }
// chunk c (16 bytes) of a gathered row of D floats through a plain pointer: nothing past the row's end is touched (the
// row may be the last of its allocation; the buffer loads of the first pass are range-checked by the hardware instead)
__device__ __forceinline__ f32x4 gather_chunk(const float *row, uint32_t c, uint32_t D)
{
    const uint32_t first = 4u * c;
    if (first + 4u <= D) return load16(row + first);
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (first + 0u < D) v[0] = row[first + 0u];
    if (first + 1u < D) v[1] = row[first + 1u];
    if (first + 2u < D) v[2] = row[first + 2u];
    return v;
}

// Centre of the model (EngineArgs::mu) -> LDS at kernel start, NF floats; the caller's barrier publishes it.
template <int NF>
__device__ __forceinline__ void stage_centre(float *mu_lds, const float *mu, int tid, int nthreads)
{
    if (mu != nullptr)
        for (int i = tid; i < NF; i += nthreads) mu_lds[i] = mu[i];
}
// x -= mu for a fragment whose entry j holds chunk j * CSTEP + c0 of the row (engine.h: CSTEP = 2, c0 = h).  One
// v_sub per value: the prototypes were shifted by the same f32 operation on the host (centre_prototypes).
template <int NJ, int CSTEP>
__device__ __forceinline__ void centre_fragment(f32x4 (&xf)[NJ], const float *mu_lds, int c0)
{
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const f32x4 m = *(const f32x4 *)(mu_lds + 4 * (j * CSTEP + c0));
        xf[j] = xf[j] - m;
    }
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

// ------------------------------------------------------------------------------ LDS-resident images
// Euclidean models whose whole tile image fits in LDS (K * D <= ~36K floats; config 2 is 130 KiB):
// "prototypes staged in LDS" once per workgroup, then every wave is independent -- no per-tile DMA,
// no barriers, and the exact re-measurement of the candidates reads its rows from LDS as well.
// One 512-thread workgroup per CU = 2 waves per SIMD sharing one copy of the prototypes.

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

// The same for two rows at once (two independent accumulation chains).
template <int R>
__device__ __forceinline__ void exact_sq_lds2(const f32x4 (&xf)[R / 8], const float *img, uint32_t ka, uint32_t kb, int h, float &da, float &db)
{
    using Gm = Geom<R>;
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    const float *ra = img + (size_t)(ka >> 5) * Gm::TILE_FLOATS + Gm::chunk_off((int)(ka & 31u), h);
    const float *rb = img + (size_t)(kb >> 5) * Gm::TILE_FLOATS + Gm::chunk_off((int)(kb & 31u), h);
    f32x2 pa = {0.0f, 0.0f}, pb = {0.0f, 0.0f};
    const f32x2 half = {0.5f, 0.5f};
#pragma unroll
    for (int j = 0; j < R / 8; ++j) {
        const f32x4 va = *(const f32x4 *)(ra + 8 * j), vb = *(const f32x4 *)(rb + 8 * j);      // -2 w
        const f32x2 xl = {xf[j][0], xf[j][1]}, xh = {xf[j][2], xf[j][3]};
        const f32x2 a0 = __builtin_elementwise_fma(half, f32x2{va[0], va[1]}, xl), a1 = __builtin_elementwise_fma(half, f32x2{va[2], va[3]}, xh);
        const f32x2 b0 = __builtin_elementwise_fma(half, f32x2{vb[0], vb[1]}, xl), b1 = __builtin_elementwise_fma(half, f32x2{vb[2], vb[3]}, xh);
        pa = __builtin_elementwise_fma(a0, a0, pa);
        pb = __builtin_elementwise_fma(b0, b0, pb);
        pa = __builtin_elementwise_fma(a1, a1, pa);
        pb = __builtin_elementwise_fma(b1, b1, pb);
    }
    const float ta = pa[0] + pa[1], tb = pb[0] + pb[1];
    da = ta + xor32(ta);
    db = tb + xor32(tb);
}

// helpers.rs:55-77 / 91-132 in f32 against an LDS-resident image: every prototype in index order, direct (x - w)^2
// form, strict '<' from (0, +INF).  The answer for samples the screening cannot settle (finish_keys), and the whole
// search where the model has so few prototypes that walking them costs less than the matrix cores' 32-row tile
// (EngineArgs::direct).  Two prototypes per step; same sums per prototype as exact_sq_lds, so the result does not
// depend on which of the two paths measured it.
template <int R, bool MATCHED, bool ROOT>
__device__ __forceinline__ void scan_exact_lds(const f32x4 (&xf)[R / 8], const float *img, uint32_t K, uint32_t label, int h, Best1 &wp,
                                               Best1 &wn)
{
    using Gm = Geom<R>;
    wp = Best1();
    wn = Best1();
    auto cls_of = [&](uint32_t k) { return ((const uint32_t *)(img + (size_t)(k >> 5) * Gm::TILE_FLOATS + Gm::W_FLOATS))[32 + (k & 31)]; };
    for (uint32_t k = 0; k < K; k += 2) {
        const uint32_t kb = k + 1 < K ? k + 1 : k;
        float da, db;
        exact_sq_lds2<R>(xf, img, k, kb, h, da, db);
        if (ROOT) {
            da = sqrtf(da);
            db = sqrtf(db);
        }
        if (!MATCHED || cls_of(k) == label) wp.offer(da, k);
        else wn.offer(da, k);
        if (kb != k) {
            if (!MATCHED || cls_of(kb) == label) wp.offer(db, kb);
            else wn.offer(db, kb);
        }
    }
}

// ------------------------------------------------------------------------------ pipelined tile step of the resident kernels
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
                offer_m(km, kp, kn);
            });
            if constexpr (o0 < 16 && o1 == 16) settle_tiles_m(km, om, tprev_id);   // all 16 keys of the tile are in
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
            offer_m(km, kp, kn);
        });
        settle_tiles_m(km, om, tprev_id);
    }
}

// The search of one batch over `ntiles` resident tiles (any count >= 1): tile t computes into one accumulator
// while the finished scores of tile t-1 in the other are screened; an odd last tile runs on its own.
template <int R, bool MATCHED>
__device__ __forceinline__ void search_tiles(const float *img, uint32_t ntiles, uint32_t label, const f32x4 (&xf)[R / 8],
                                             int s_col, int h, Keys2 &ku, KeysM &km)
{
    using Gm = Geom<R>;
    f32x16 acc0, acc1;
    f32x4 a0 = *(const f32x4 *)(img + Gm::chunk_off(s_col, h));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 ci = *(const f32x4 *)(img + Gm::W_FLOATS + 8 * g + 4 * h);
        acc0[4 * g + 0] = ci[0]; acc0[4 * g + 1] = ci[1]; acc0[4 * g + 2] = ci[2]; acc0[4 * g + 3] = ci[3];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = PAD_SCORE;                 // "tile -1": its keys never win
    const float *tprev = img;
    uint32_t tid = 0, t = 0;
    for (; t + 1 < ntiles; t += 2) {
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
    if (t < ntiles) {                                                 // odd count: acc0 holds the last tile's init column
        const float *tile = img + (size_t)t * Gm::TILE_FLOATS;
        half_step<R, MATCHED>(tile, img, tprev, tid, label, xf, acc0, acc1, a0, s_col, h, ku, km);
        settle_mfma(acc0);
        last_screen<R, MATCHED>(acc0, tile, t, label, h, ku, km);
    } else {
        settle_mfma(acc1);
        last_screen<R, MATCHED>(acc1, tprev, tid, label, h, ku, km);
    }
}

// ------------------------------------------------------------------------------ the search on the bf16 matrix cores
// Screening only ranks candidates: what leaves the kernel is re-measured in f32 in the direct form, and the third-key
// guard (finish_keys) sends every sample the ranking could have got wrong to the exact scan.  So the ranking may run
// wherever it is cheapest as long as its error is inside the margin -- and v_mfma_f32_32x32x16_bf16 does in 32 cycles
// what sixteen v_mfma_f32_32x32x2_f32 do in 1 024, and leaves the vector ALUs free while it runs (the f32 form does
// not: DESIGN.md 4.0).  Every f32 operand is split into two bf16 pieces, v = hi + lo + e with hi = bf16(v),
// lo = bf16(v - hi), |e| <= 2^-16 |v| (round to nearest, 8 significant bits each), and three of the four piece
// products are accumulated in f32: lo.hi + hi.lo + hi.hi.  The dropped parts (lo.lo, e.w, x.e) are below
// 3 * 2^-16 |x_d| |a_d| per term, so a score differs from the f32-exact ||w||^2 - 2 x.w by less than
// 3 * 2^-16 (||x||^2 + ||w||^2) plus the f32 accumulation: the host's margin for this path (engine_build, mscale_bf;
// measured: tools/bf16x3_probe.hip).  The image in LDS stays the f32 one (the exact re-measurement reads it, and a
// second image does not fit beside it at config 2's size), so the prototype fragments are split on the fly: 6 VALU
// per pair of values, hidden under nothing -- this search is bound by the vector ALUs, at about a third of the f32
// form's cycles per tile.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2(float v0, float v1, uint32_t &hp, uint32_t &lp)
{
    hp = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){v0, v1}, bf16x2));          // v_cvt_pk_bf16_f32
    const float h0 = __builtin_bit_cast(float, hp << 16), h1 = __builtin_bit_cast(float, hp & 0xFFFF0000u);
    lp = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){v0 - h0, v1 - h1}, bf16x2));
}
// the 8 values a lane contributes to one 32x32x16 step: its chunks 2q and 2q + 1 (any order, as long as both operands agree)
__device__ __forceinline__ void split8(const f32x4 &a0, const f32x4 &a1, u32x4 &hi, u32x4 &lo)
{
    uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
    split2(a0[0], a0[1], h0, l0);
    split2(a0[2], a0[3], h1, l1);
    split2(a1[0], a1[1], h2, l2);
    split2(a1[2], a1[3], h3, l3);
    hi = u32x4{h0, h1, h2, h3};
    lo = u32x4{l0, l1, l2, l3};
}
template <int R>
__device__ __forceinline__ void split_fragment(const f32x4 (&xf)[R / 8], u32x4 (&bh)[R / 16], u32x4 (&bl)[R / 16])
{
#pragma unroll
    for (int q = 0; q < R / 16; ++q) split8(xf[2 * q], xf[2 * q + 1], bh[q], bl[q]);
}

// Three pieces: v = p1 + p2 + p3 EXACTLY for a normal f32 (3 x 8 significant bits; the remainders v - p1 and
// v - p1 - p2 are exact in f32).  The operand form of the projections y = Omega x on the bf16 matrix cores: the six piece
// products down to 2^-16 of the leading one (p1.p1 | p1.p2, p2.p1 | p1.p3, p2.p2, p3.p1), smallest class first, land
// within 2^-24-ish of the exact dot product -- measured at or below the error of the v_mfma_f32_32x32x2_f32 chain they
// replace on every data set tried (tools/bf16x6_probe.hip: 1.7e-7 against 1.7e-7 of |x||w| on uniform data, 8e-7 against
// 1.1e-6 on one-signed data), at 6 x 32 cycles per 16 reduction values instead of 8 x 64.
__device__ __forceinline__ void split3(float v0, float v1, uint32_t &p1, uint32_t &p2, uint32_t &p3)
{
    p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){v0, v1}, bf16x2));
    const float r0 = v0 - __builtin_bit_cast(float, p1 << 16), r1 = v1 - __builtin_bit_cast(float, p1 & 0xFFFF0000u);
    p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){r0, r1}, bf16x2));
    const float s0 = r0 - __builtin_bit_cast(float, p2 << 16), s1 = r1 - __builtin_bit_cast(float, p2 & 0xFFFF0000u);
    p3 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2v){s0, s1}, bf16x2));
}
// the 8 values a lane contributes to one 32x32x16 step (its chunks 2q and 2q + 1, as split8)
__device__ __forceinline__ void split24(const f32x4 &a0, const f32x4 &a1, u32x4 (&p)[3])
{
    uint32_t a[4], b[4], c[4];
    split3(a0[0], a0[1], a[0], b[0], c[0]);
    split3(a0[2], a0[3], a[1], b[1], c[1]);
    split3(a1[0], a1[1], a[2], b[2], c[2]);
    split3(a1[2], a1[3], a[3], b[3], c[3]);
    p[0] = u32x4{a[0], a[1], a[2], a[3]};
    p[1] = u32x4{b[0], b[1], b[2], b[3]};
    p[2] = u32x4{c[0], c[1], c[2], c[3]};
}
__device__ __forceinline__ f32x16 mma_bf(const u32x4 &a, const u32x4 &b, const f32x16 &c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the six products of one step for TG tiles at once (independent chains side by side); a[u] = the weight pieces of tile u
template <int TG>
__device__ __forceinline__ void mma_bf6(const u32x4 (&a)[TG][3], const u32x4 (&x)[3], f32x16 *acc)
{
#pragma unroll
    for (int u = 0; u < TG; ++u) acc[u] = mma_bf(a[u][2], x[0], acc[u]);
#pragma unroll
    for (int u = 0; u < TG; ++u) acc[u] = mma_bf(a[u][1], x[1], acc[u]);
#pragma unroll
    for (int u = 0; u < TG; ++u) acc[u] = mma_bf(a[u][0], x[2], acc[u]);
#pragma unroll
    for (int u = 0; u < TG; ++u) acc[u] = mma_bf(a[u][1], x[0], acc[u]);
#pragma unroll
    for (int u = 0; u < TG; ++u) acc[u] = mma_bf(a[u][0], x[1], acc[u]);
#pragma unroll
    for (int u = 0; u < TG; ++u) acc[u] = mma_bf(a[u][0], x[0], acc[u]);
}

// acc[s] = init column + tile . x[s]^T for one Geom3 tile in LDS (rows = the tile's 32 Omega_j rows, column = sample)
template <int R, int S>
__device__ __forceinline__ void tile_mma_pj(const char *tile, const u32x4 (&xp)[S][R / 16][3], f32x16 (&acc)[S], int row, int h)
{
    using G3 = Geom3<R>;
    const float *tail = (const float *)(tile + G3::W_BYTES);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 ci = *(const f32x4 *)(tail + 8 * g + 4 * h);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            acc[s][4 * g + 0] = ci[0]; acc[s][4 * g + 1] = ci[1];
            acc[s][4 * g + 2] = ci[2]; acc[s][4 * g + 3] = ci[3];
        }
    }
    const char *rowp = tile + G3::piece_off(row, 0, h, 0);
    u32x4 an[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) an[c] = *(const u32x4 *)(rowp + 16 * c);
    static_for<0, R / 16>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        u32x4 av[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) av[c] = an[c];
        if constexpr (q + 1 < R / 16) {
#pragma unroll
            for (int c = 0; c < 3; ++c) an[c] = *(const u32x4 *)(rowp + 96 * (q + 1) + 16 * c);
        }
        // smallest class of products first; the S chains side by side
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = mma_bf(av[2], xp[s][q][0], acc[s]);
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = mma_bf(av[1], xp[s][q][1], acc[s]);
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = mma_bf(av[0], xp[s][q][2], acc[s]);
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = mma_bf(av[1], xp[s][q][0], acc[s]);
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = mma_bf(av[0], xp[s][q][1], acc[s]);
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = mma_bf(av[0], xp[s][q][0], acc[s]);
    });
}

// tile_mma with the products on the bf16 matrix cores: every split prototype fragment serves the S sample tiles of the wave
template <int R, int S>
__device__ __forceinline__ void tile_mma_bf(const float *tile, const u32x4 (&bh)[S][R / 16], const u32x4 (&bl)[S][R / 16], f32x16 (&acc)[S],
                                            int row, int h)
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
    for (int q = 0; q < R / 16; ++q) {
        u32x4 ah, al;
        split8(*(const f32x4 *)(rowp + 16 * q), *(const f32x4 *)(rowp + 16 * q + 8), ah, al);
        const bf16x8 vah = __builtin_bit_cast(bf16x8, ah), val = __builtin_bit_cast(bf16x8, al);
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(val, __builtin_bit_cast(bf16x8, bh[s][q]), acc[s], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah, __builtin_bit_cast(bf16x8, bl[s][q]), acc[s], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah, __builtin_bit_cast(bf16x8, bh[s][q]), acc[s], 0, 0, 0);
    }
}

// tile_mma_bf over a tile of READY pieces (host: pieces_image / build_bf_tiles: per row, step q and lane half h 16 bytes
// of hi pieces and 16 of lo at byte 64 q + 32 h of the f32 row's pitch; the tail is the f32 tile's): nothing is split,
// every pair of LDS reads serves the S sample tiles of the wave.  The streamed engine's search stage.
template <int R, int S>
__device__ __forceinline__ void tile_mma_ready(const float *tile, const u32x4 (&bh)[S][R / 16], const u32x4 (&bl)[S][R / 16], f32x16 (&acc)[S],
                                               int row, int h)
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
    const char *rowp = (const char *)tile + row * (Gm::ROW_FLOATS * 4) + 32 * h;
#pragma unroll
    for (int q = 0; q < R / 16; ++q) {
        const bf16x8 vah = __builtin_bit_cast(bf16x8, *(const u32x4 *)(rowp + 64 * q));
        const bf16x8 val = __builtin_bit_cast(bf16x8, *(const u32x4 *)(rowp + 64 * q + 16));
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(val, __builtin_bit_cast(bf16x8, bh[s][q]), acc[s], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah, __builtin_bit_cast(bf16x8, bl[s][q]), acc[s], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < S; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah, __builtin_bit_cast(bf16x8, bh[s][q]), acc[s], 0, 0, 0);
    }
}

// tile_mma_ready for one sample tile with the screening of the PREVIOUS tile's scores between its MFMAs (pieces_step's
// interleave for the streamed engine): score o of `prev` behind MFMA o + 1 of this tile, pinned there.  `prev` comes in
// finished (its last MFMA was the last one of the previous call, or it is the pad tile) and is not written here.
template <int R, class SCR>
__device__ __forceinline__ void tile_mma_ready_spread(const float *tile, const u32x4 (&bh)[R / 16], const u32x4 (&bl)[R / 16], f32x16 &acc,
                                                      f32x16 &prev, uint32_t tprev_id, SCR &scr, int row, int h)
{
    using Gm = Geom<R>;
    constexpr int NQ = R / 16;
    static_assert(NQ >= 6, "16 scores need 17 MFMAs of the tile");
    const float *tail = tile + Gm::W_FLOATS;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 ci = *(const f32x4 *)(tail + 8 * g + 4 * h);
        acc[4 * g + 0] = ci[0]; acc[4 * g + 1] = ci[1]; acc[4 * g + 2] = ci[2]; acc[4 * g + 3] = ci[3];
    }
    const char *rowp = (const char *)tile + row * (Gm::ROW_FLOATS * 4) + 32 * h;
    static_for<0, NQ>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        const bf16x8 vah = __builtin_bit_cast(bf16x8, *(const u32x4 *)(rowp + 64 * q));
        const bf16x8 val = __builtin_bit_cast(bf16x8, *(const u32x4 *)(rowp + 64 * q + 16));
        auto slot = [&](auto ii) {
            constexpr int sidx = 3 * q + decltype(ii)::value;           // MFMA number in the tile
            if constexpr (sidx >= 1 && sidx <= 16) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (sidx == 1) scr.begin();
                scr.template one<sidx - 1, (sidx == 1 ? 1 : 0)>(prev, nullptr, 0u, h);
                if constexpr (sidx == 16) scr.end(tprev_id);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(val, __builtin_bit_cast(bf16x8, bh[q]), acc, 0, 0, 0);
        slot(std::integral_constant<int, 0>{});
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah, __builtin_bit_cast(bf16x8, bl[q]), acc, 0, 0, 0);
        slot(std::integral_constant<int, 1>{});
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah, __builtin_bit_cast(bf16x8, bh[q]), acc, 0, 0, 0);
        slot(std::integral_constant<int, 2>{});
    });
}

// Bytes of one tile of bf16 pieces (host: build_bf_tiles): the rows of the f32 tile at the same pitch, each row holding,
// per reduction step q and lane half h, the 8 hi pieces and then the 8 lo pieces of that lane's values (16 + 16 bytes).
template <int R>
constexpr int BF_TILE_BYTES = Geom<R>::W_FLOATS * 4;

// ------------------------------------------------------------------------------ samples the screening cannot settle
// A sample whose third best key lies within the margin of its best one (or whose scores mean nothing: NaN / INF / huge
// rows) needs every close prototype measured exactly.  Doing that inside the wave makes 63 lanes wait for one: at 4 096
// random prototypes 0.7 % of the samples are crowded and one wave-batch in five walked all 4 096 rows on their behalf.
// Instead the first pass MARKS them -- one bit per sample in a mask word per tile of 32 samples, no atomics -- and
// carries on; k_crowd_compact packs the marked samples into a list and the kernel's FINISH variant works the list, 32
// crowded samples per wave (HitKeys below).  Bit planes of tile t: mask[3 t] the unlabelled / matching search is
// crowded, mask[3 t + 1] the non-matching search is, mask[3 t + 2] take every row (unusable scores).
constexpr uint32_t CROWD_P = 1u << 29, CROWD_N = 1u << 30, CROWD_ALL = 1u << 31, CROWD_ID = CROWD_P - 1u;   // list entry: flags | sample
// Wave-uniform call.  Returns true when the marks were taken (the finish pass answers for the marked samples).  Only rows
// that exist are marked: a batch's lanes past row N (and whole batches past the last one, which the kernels run on
// zeros) must leave no bit behind -- their tile may not even have mask words.
__device__ __forceinline__ bool crowd_mark(uint32_t *mask, uint64_t tile32, uint64_t N, bool cp, bool cn, bool all, int h)
{
    if (mask == nullptr) return false;
    const bool live = h == 0 && tile32 * 32 + (uint64_t)(threadIdx.x & 31) < N;
    const uint32_t mp = (uint32_t)__builtin_amdgcn_ballot_w64(cp && live), mn = (uint32_t)__builtin_amdgcn_ballot_w64(cn && live);
    const uint32_t ma = (uint32_t)__builtin_amdgcn_ballot_w64(all && live);
    if ((mp | mn | ma) != 0u && (threadIdx.x & 63) == 0) {
        uint32_t *w = mask + 3 * tile32;
        if (mp) w[0] = mp;
        if (mn) w[1] = mn;
        if (ma) w[2] = ma;
    }
    return true;
}

// This lane's part of the exact distance of row k (its chunks of the row; the lane pair's two parts add up to exact_sq_*'s
// value in exact_sq_*'s order, so a distance does not depend on which pass measured it)
template <int R>
__device__ __forceinline__ float exact_part_lds(const f32x4 (&xf)[R / 8], const float *img, uint32_t k, int h)
{
    using Gm = Geom<R>;
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    const float *rowp = img + (size_t)(k >> 5) * Gm::TILE_FLOATS + Gm::chunk_off((int)(k & 31u), h);
    f32x2 p = {0.0f, 0.0f};
    const f32x2 half = {0.5f, 0.5f};
#pragma unroll
    for (int j = 0; j < R / 8; ++j) {
        const f32x4 v = *(const f32x4 *)(rowp + 8 * j);                          // v = -2 w
        const f32x2 d0 = __builtin_elementwise_fma(half, f32x2{v[0], v[1]}, f32x2{xf[j][0], xf[j][1]});
        const f32x2 d1 = __builtin_elementwise_fma(half, f32x2{v[2], v[3]}, f32x2{xf[j][2], xf[j][3]});
        p = __builtin_elementwise_fma(d0, d0, p);
        p = __builtin_elementwise_fma(d1, d1, p);
    }
    return p[0] + p[1];
}
template <int R>
__device__ __forceinline__ float exact_part_g(const f32x4 (&xf)[R / 8], const float *plain, uint32_t ldp, uint32_t k, int h)
{
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    const float *w = plain + (size_t)k * ldp + 4 * h;
    f32x2 p = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < R / 8; ++j) {
        const f32x4 v = *(const f32x4 *)(w + 8 * j);
        const f32x2 d0 = f32x2{xf[j][0], xf[j][1]} - f32x2{v[0], v[1]};
        const f32x2 d1 = f32x2{xf[j][2], xf[j][3]} - f32x2{v[2], v[3]};
        p = __builtin_elementwise_fma(d0, d0, p);
        p = __builtin_elementwise_fma(d1, d1, p);
    }
    return p[0] + p[1];
}
// (exact_sq's order: one chain -- the streamed engine's re-measurement)
template <int R>
__device__ __forceinline__ float exact_part_chain(const f32x4 (&xf)[R / 8], const float *plain, uint32_t ldp, uint32_t k, int h)
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
    return p;
}

// ------------------------------------------------------------------------------ screening policies of the bf16 tile loops
// What happens to the 16 finished scores a lane holds for one tile (rows 8 g + 4 h + e of the tile, register 4 g + e) is a
// policy object handed to the loops below:
//   scr.template tile<PRE>(sc, tile_id, tail, label, h)
//     sc    the scores; PRE != 0: the MFMA that wrote them is right in front (wait states are the policy's business: the
//           compiler pads hazards only for instructions it knows, not for inline asm)
//     tail  the tile's tail: 32 accumulator-init values, 32 class ids
// TopKeys   the best two keys with their tiles and the third best key (4 VALU per score): finish_keys and its relatives
// GroupKeys the same over GROUPS OF FOUR ROWS (the lane's rows 8 g + 4 h .. + 3 of a tile are four consecutive prototypes):
//           1.5 VALU per score; finish_groups re-measures all four rows of the best group (and of the runner-up where close)
// HitKeys   the finishing pass over samples the first pass could not settle: every row whose score lies within the margin
//           of the running minimum is re-measured exactly on the spot (finish pass of the kernels, FINISH)
template <bool MATCHED>
struct TopKeys {
    Keys2 ku;
    KeysM km;
    // The screening can be spread over the MFMAs of the next tile (pieces_step): begin(), one<O>() for each of the 16
    // scores, end() -- together exactly tile<>().  Measured on one box: config 2 4.25 -> 4.40e9 samples/s, config 3
    // 2.53 -> 2.57e9; the labelled form (10 vector instructions per score, bound by their issue either way) lost 1.7 %
    // to the pinning and keeps its block.
    static constexpr bool SPREAD = !MATCHED;
    Keys2 before;
    KeysM before_m;
    __device__ __forceinline__ void begin()
    {
        if constexpr (MATCHED) before_m = km;
        else before = ku;
    }
    template <int O, int PRE>
    __device__ __forceinline__ void one(f32x16 &sc, const float *tail, uint32_t label, int h)
    {
        if constexpr (!MATCHED) {
            screen1<O, PRE>(sc, ku);
        } else {
            if constexpr (PRE != 0) settle_mfma(sc);
            const uint32_t *tcls = (const uint32_t *)tail + 32 + 4 * h;
            const uint32_t key = pack_key(sc[O], O);
            const bool match = tcls[8 * (O >> 2) + (O & 3)] == label;              // helpers.rs:113 / :122
            offer_m(km, match ? key : KEY_NONE, match ? KEY_NONE : key);
        }
    }
    __device__ __forceinline__ void end(uint32_t tile_id)
    {
        if constexpr (MATCHED) settle_tiles_m(km, before_m, tile_id);
        else settle_tiles(ku, before, tile_id);
    }
    template <int PRE>
    __device__ __forceinline__ void tile(f32x16 &sc, uint32_t tile_id, const float *tail, uint32_t label, int h)
    {
        if constexpr (!MATCHED) {
            const Keys2 ou = ku;
            screen16<PRE>(sc, ku);
            settle_tiles(ku, ou, tile_id);
        } else {
            const KeysM om = km;
            if constexpr (PRE != 0) settle_mfma(sc);
            const uint32_t *tcls = (const uint32_t *)tail + 32 + 4 * h;
            static_for<0, 16>([&](auto oo) {
                constexpr int o = decltype(oo)::value;
                const uint32_t key = pack_key(sc[o], o);
                const bool match = tcls[8 * (o >> 2) + (o & 3)] == label;          // helpers.rs:113 / :122
                offer_m(km, match ? key : KEY_NONE, match ? KEY_NONE : key);
            });
            settle_tiles_m(km, om, tile_id);
        }
    }
};

// Keys of GROUPS: the minimum of a group's four scores with the group number g in the position bits.  A key stands for the
// rows tile * 32 + 8 g + 4 h .. + 3 (h: the lane half that offered it).  Everything around the keys -- best two with
// their tiles, third best as the witness -- is TopKeys'.  Why the guarantee survives: every row's key is at least its
// group's, so when the third best GROUP key lies more than the margin above the best one, no row outside the best two
// groups can be the reference's winner; all rows of those two groups are re-measured exactly (finish_groups).
#define VQ_GROUP2_TEXT(A, B, PA, PB)                          \
    "v_min3_f32 %[t0], %[s" #A "0], %[s" #A "1], %[s" #A "2]\n\t" \
    "v_min3_f32 %[t1], %[s" #B "0], %[s" #B "1], %[s" #B "2]\n\t" \
    "v_min_f32 %[t0], %[t0], %[s" #A "3]\n\t"                  \
    "v_min_f32 %[t1], %[t1], %[s" #B "3]\n\t"                  \
    "v_and_or_b32 %[t0], %[t0], %[m], " #PA "\n\t"             \
    "v_and_or_b32 %[t1], %[t1], %[m], " #PB "\n\t"             \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t0]\n\t"                \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t0]\n\t"                \
    "v_min_f32 %[b1], %[b1], %[t0]\n\t"                        \
    "v_med3_f32 %[b3], %[b2], %[b3], %[t1]\n\t"                \
    "v_med3_f32 %[b2], %[b1], %[b2], %[t1]\n\t"                \
    "v_min_f32 %[b1], %[b1], %[t1]\n\t"
template <int PRE_NOPS>
__device__ __forceinline__ void screen4g(const f32x16 &sc, Keys2 &ku)
{
    uint32_t t0, t1;
#define VQ_GROUP_OPERANDS                                                                                                          \
    : [b1] "+v"(ku.b1), [b2] "+v"(ku.b2), [b3] "+v"(ku.b3), [t0] "=&v"(t0), [t1] "=&v"(t1)                                          \
    : [sa0] "v"(sc[0]), [sa1] "v"(sc[1]), [sa2] "v"(sc[2]), [sa3] "v"(sc[3]), [sb0] "v"(sc[4]), [sb1] "v"(sc[5]), [sb2] "v"(sc[6]),   \
      [sb3] "v"(sc[7]), [sc0] "v"(sc[8]), [sc1] "v"(sc[9]), [sc2] "v"(sc[10]), [sc3] "v"(sc[11]), [sd0] "v"(sc[12]),                 \
      [sd1] "v"(sc[13]), [sd2] "v"(sc[14]), [sd3] "v"(sc[15]), [m] "s"(0xFFFFFFF0u)
    if constexpr (PRE_NOPS > 0)
        asm("s_nop 15\n\ts_nop 1\n\t" VQ_GROUP2_TEXT(a, b, 0, 1) VQ_GROUP2_TEXT(c, d, 2, 3) VQ_GROUP_OPERANDS);
    else
        asm(VQ_GROUP2_TEXT(a, b, 0, 1) VQ_GROUP2_TEXT(c, d, 2, 3) VQ_GROUP_OPERANDS);
#undef VQ_GROUP_OPERANDS
}
// minimum of a group's four scores (plain floats: nothing here is a NaN pattern, see PAD_SCORE)
__device__ __forceinline__ float group_min(float s0, float s1, float s2, float s3)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3\n\tv_min_f32 %0, %0, %4" : "=&v"(r) : "v"(s0), "v"(s1), "v"(s2), "v"(s3));
    return r;
}

template <bool MATCHED>
struct GroupKeys {
    static constexpr bool SPREAD = false;
    __device__ __forceinline__ void begin() {}                       // (the spread form is TopKeys' alone)
    template <int O, int PRE>
    __device__ __forceinline__ void one(f32x16 &, const float *, uint32_t, int) {}
    __device__ __forceinline__ void end(uint32_t) {}
    Keys2 ku;
    KeysM km;
    template <int PRE>
    __device__ __forceinline__ void tile(f32x16 &sc, uint32_t tile_id, const float *tail, uint32_t label, int h)
    {
        if constexpr (!MATCHED) {
            const Keys2 ou = ku;
            screen4g<PRE>(sc, ku);
            settle_tiles(ku, ou, tile_id);
        } else {
            const KeysM om = km;
            if constexpr (PRE != 0) settle_mfma(sc);
            const uint32_t *tcls = (const uint32_t *)tail + 32 + 4 * h;
            static_for<0, 4>([&](auto gg) {
                constexpr int g = decltype(gg)::value;
                const u32x4 cg = *(const u32x4 *)(tcls + 8 * g);
                // rows on the wrong side of the label score PAD_SCORE (not +INF: its pattern with position bits is a NaN)
                const bool m0 = cg[0] == label, m1 = cg[1] == label, m2 = cg[2] == label, m3 = cg[3] == label;   // helpers.rs:113 / :122
                const float v0 = sc[4 * g + 0], v1 = sc[4 * g + 1], v2 = sc[4 * g + 2], v3 = sc[4 * g + 3];
                const uint32_t kp = pack_key(group_min(m0 ? v0 : PAD_SCORE, m1 ? v1 : PAD_SCORE, m2 ? v2 : PAD_SCORE, m3 ? v3 : PAD_SCORE), g);
                const uint32_t kn = pack_key(group_min(m0 ? PAD_SCORE : v0, m1 ? PAD_SCORE : v1, m2 ? PAD_SCORE : v2, m3 ? PAD_SCORE : v3), g);
                offer_m(km, kp, kn);
            });
            settle_tiles_m(km, om, tile_id);
        }
    }
};
// The finish pass: every row whose score is within the margin of the lane's RUNNING minimum is measured exactly at once
// and offered in index order (strict '<': helpers.rs:70).  Why that is enough: a key differs from the true score by less
// than margin / 2, so the reference's winner scores at most `margin` above the least score of all rows -- and the
// running minimum, taken over a subset, is never below that.  Early tiles measure a few rows more than the final
// minimum would ask for; the lane pair's two lists are merged by (distance, index) at the end.  A row is measured by
// BOTH lanes of its pair (each holds half of the sample's chunks), so hits are worked one score register at a time with
// the partner's row number exchanged.  margin = +INF (unusable scores) makes every row a hit: the exact scan.
//   ROWS  where the exact rows come from, and in which order a row is summed (the first pass's: a distance must not depend
//         on the pass that measured it): 1 the f32 image in LDS (rows = the image, exact_sq_lds), 0 the plain matrix
//         (K x ldp) as exact_sq_g sums it, 2 the plain matrix as exact_sq does
//   CAP   0: a tile's hits are measured on the spot, one round trip to the exact rows per hit of the tile's busiest lane.
//         16: hits go to a per-lane list in LDS (row | side << 31, score; `lst` + i * 2 * stride) and are measured when a
//         list could overflow and at the end -- as many round trips as the busiest lane has hits in ALL, not per tile (with
//         thousands of prototypes nearly every tile holds some sample's hit), and by then the threshold has come down: an
//         entry whose score is above the running minimum + margin of the moment it is read is dropped unmeasured.
//         8: the same with the list in eight registers of the lane (no scores: nothing is dropped), for kernels whose LDS
//         is taken; a tile that brings a lane more than 8 hits (the exact scan: all 16) is measured on the spot.
template <int R, bool MATCHED, bool ROOT, int ROWS, int CAP = 0>
struct HitKeys {
    static_assert(CAP == 0 || CAP == 8 || CAP == 16, "0: on the spot, 8: register list, 16: LDS list");
    uint32_t ids[CAP == 8 ? 8 : 1] = {};   // CAP == 8: the list (row | side << 31)
    float listed_at = INFINITY;            // CAP == 8: the running minimum when the list last grew (no entry scores below it)
    static constexpr bool SPREAD = false;
    __device__ __forceinline__ void begin() {}                       // (the spread form is TopKeys' alone)
    template <int O, int PRE>
    __device__ __forceinline__ void one(f32x16 &, const float *, uint32_t, int) {}
    __device__ __forceinline__ void end(uint32_t) {}
    static constexpr uint32_t NONE = 0xFFFFFFFFu;
    const f32x4 (&xf)[R / 8];
    const float *rows;
    uint32_t ldp, K;
    float margin_p, margin_n;          // -INF: the search is not this sample's problem (no hits, nothing stored)
    float run_p = INFINITY, run_n = INFINITY;
    Best1 wp, wn;
    uint32_t *lst;                     // CAP != 0: this lane's list
    uint32_t stride, cnt = 0u;
    int hh = 0;
    __device__ __forceinline__ HitKeys(const f32x4 (&xf_)[R / 8], const float *rows_, uint32_t ldp_, uint32_t K_, float mp, float mn,
                                       uint32_t *lst_ = nullptr, uint32_t stride_ = 0u)
        : xf(xf_), rows(rows_), ldp(ldp_), K(K_), margin_p(mp), margin_n(mn), lst(lst_), stride(stride_) {}
    __device__ __forceinline__ float part(uint32_t k, int h) const
    {
        if constexpr (ROWS == 1) return exact_part_lds<R>(xf, rows, k, h);
        else if constexpr (ROWS == 2) return exact_part_chain<R>(xf, rows, ldp, k, h);
        else return exact_part_g<R>(xf, rows, ldp, k, h);
    }
    // One step for the whole wave: every lane's row `mine` (< K, or NONE) measured exactly and offered.  A row is measured
    // by BOTH lanes of its pair (each holds half of the sample's chunks): row numbers and partial sums are exchanged.
    // No branch around the loads -- a lane without a row reads row 0 and drops the sum -- so the two rows' loads go out
    // together: one round trip per step, not two.
    __device__ __forceinline__ void measure(uint32_t mine, bool positive, int h)
    {
        const uint32_t theirs = xor32(mine);
        const float pm = part(mine != NONE ? mine : 0u, h);
        const float pt = part(theirs != NONE ? theirs : 0u, h);
        const float got = xor32(pt);                                                   // the partner's part of MY row
        if (mine != NONE) {
            const float total = pm + got;
            const float d = ROOT ? sqrtf(total) : total;
            if (positive) wp.offer(d, mine);
            else wn.offer(d, mine);
        }
    }
    // The list against the thresholds as they stand NOW: an entry whose score lies above the running minimum + margin of
    // this moment can never qualify (the minimum only falls) and leaves; the others move up, order kept (index order inside
    // a lane: strict '<').  What it buys: the steps of a drain are the busiest lane's entries, and before the sample's
    // true neighbourhood turns up every tile with a new running minimum lists a row -- different lanes hold their live
    // entries at different places of the list, so without the closing-up nearly every place is some lane's.
    __device__ __forceinline__ void filter()
    {
        if constexpr (CAP == 16) {               // (the register list keeps no scores: tile() has its own, coarser rule)
            const float thr_p = run_p + margin_p, thr_n = run_n + margin_n;
            uint32_t w = 0u;
            for (uint32_t i = 0; i < cnt; ++i) {     // (a lane's own list: lanes run out at different i)
                const uint32_t e = lst[(2u * i) * stride], scb = lst[(2u * i + 1u) * stride];
                if (!(__builtin_bit_cast(float, scb) > ((e >> 31) == 0u ? thr_p : thr_n))) {
                    lst[(2u * w) * stride] = e;      // (w <= i)
                    lst[(2u * w + 1u) * stride] = scb;
                    ++w;
                }
            }
            cnt = w;
        }
    }
    // the listed hits that are still live, oldest first (index order inside a lane)
    __device__ __forceinline__ void drain()
    {
        if constexpr (CAP == 0) return;
        filter();
        if constexpr (CAP == 8) {
            static_for<0, 8>([&](auto ii) {
                constexpr uint32_t i = (uint32_t)decltype(ii)::value;
                if (__builtin_amdgcn_ballot_w64(i < cnt) != 0ull) {
                    const uint32_t e = ids[i];
                    measure(i < cnt ? (e & 0x7FFFFFFFu) : NONE, (e >> 31) == 0u, hh);
                }
            });
        } else {
            for (uint32_t i = 0; __builtin_amdgcn_ballot_w64(i < cnt) != 0ull; ++i) {
                const bool have = i < cnt;
                const uint32_t e = have ? lst[(2u * i) * stride] : 0u;
                measure(have ? (e & 0x7FFFFFFFu) : NONE, (e >> 31) == 0u, hh);
            }
        }
        cnt = 0u;
    }
    template <int PRE>
    __device__ __forceinline__ void tile(f32x16 &sc, uint32_t tile_id, const float *tail, uint32_t label, int h)
    {
        if constexpr (PRE != 0) settle_mfma(sc);
        hh = h;
        const uint32_t *tcls = (const uint32_t *)tail + 32 + 4 * h;
        bool match[16];
        float val[16];                                 // (plain floats: bit casts of vector elements have been miscompiled, 4.11)
        // running minima first (they include this tile), then the thresholds
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            match[o] = !MATCHED || tcls[8 * (o >> 2) + (o & 3)] == label;             // helpers.rs:113 / :122
            const float v = sc[o];
            val[o] = v;
            if (match[o]) run_p = fminf(run_p, v);
            else run_n = fminf(run_n, v);
        }
        const bool en_p = margin_p > -INFINITY, en_n = margin_n > -INFINITY;
        const float thr_p = run_p + margin_p, thr_n = run_n + margin_n;
        bool hit[16];
        uint32_t hm = 0u, mm = 0u;
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            const uint32_t k = tile_id * 32u + 8u * (uint32_t)(o >> 2) + 4u * (uint32_t)h + (uint32_t)(o & 3);
            // NaN scores count as hits: measured, never win.  Rows past K (pad rows of the last tile) are nobody's hit.
            hit[o] = (match[o] ? en_p : en_n) && !(val[o] > (match[o] ? thr_p : thr_n)) && k < K;
            hm |= hit[o] ? 1u << o : 0u;
            mm |= match[o] ? 1u << o : 0u;
        }
        if (__builtin_amdgcn_ballot_w64(hm != 0u) == 0ull) return;
        // one step per hit of the tile's busiest lane: measured on the spot (CAP == 0), or put on the lane's register list
        auto each_hit = [&](auto &&what) {
            while (__builtin_amdgcn_ballot_w64(hm != 0u) != 0ull) {
                const uint32_t o = hm != 0u ? (uint32_t)__builtin_ctz(hm) : 0u;
                const uint32_t k = tile_id * 32u + 8u * (o >> 2) + 4u * (uint32_t)h + (o & 3u);
                const uint32_t mine = hm != 0u ? k : NONE;
                const bool positive = ((mm >> o) & 1u) != 0u;
                hm &= hm - 1u;
                what(mine, positive);
            }
        };
        if constexpr (CAP == 8) {
            const uint32_t n = (uint32_t)__builtin_popcount(hm);
            // The register list keeps no scores, but every entry scores at least the running minimum of the moment the
            // list last grew (`listed_at`: an entry is never below the minimum of its own moment, and the minimum only
            // falls).  Once the minimum has come down by more than the margin since then, ALL entries lie above minimum +
            // margin and can never qualify: the lane's list starts over.  Before the sample's true neighbourhood turns up,
            // every tile with a new minimum lists a row -- this is what keeps those out of the drain's steps.  (A lane that
            // serves both searches keeps two minima for one list and does not do this.)
            if (en_p != en_n) {
                if ((en_p ? thr_p : thr_n) < listed_at) cnt = 0u;
                if (hm != 0u) listed_at = en_p ? run_p : run_n;
            }
            if (__builtin_amdgcn_ballot_w64(cnt + n > 8u) != 0ull) drain();
            if (__builtin_amdgcn_ballot_w64(n > 8u) != 0ull) {
                each_hit([&](uint32_t mine, bool positive) { measure(mine, positive, h); });
            } else {
                each_hit([&](uint32_t mine, bool positive) {
                    const uint32_t e = mine | (positive ? 0u : 0x80000000u);
#pragma unroll
                    for (uint32_t j = 0; j < 8; ++j) ids[j] = (mine != NONE && cnt == j) ? e : ids[j];
                    cnt += mine != NONE ? 1u : 0u;
                });
            }
        } else if constexpr (CAP != 0) {
            if (__builtin_amdgcn_ballot_w64(cnt + (uint32_t)__builtin_popcount(hm) > (uint32_t)CAP) != 0ull) filter();
            if (__builtin_amdgcn_ballot_w64(cnt + (uint32_t)__builtin_popcount(hm) > (uint32_t)CAP) != 0ull) drain();
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                if (hit[o]) {
                    const uint32_t k = tile_id * 32u + 8u * (uint32_t)(o >> 2) + 4u * (uint32_t)h + (uint32_t)(o & 3);
                    lst[(2u * cnt) * stride] = k | (match[o] ? 0u : 0x80000000u);
                    lst[(2u * cnt + 1u) * stride] = __float_as_uint(val[o]);
                    ++cnt;
                }
            }
        } else {
            each_hit([&](uint32_t mine, bool positive) { measure(mine, positive, h); });
        }
    }
    // the lane pair's answer
    __device__ __forceinline__ void merge()
    {
        drain();
        wp.merge_halves();
        wn.merge_halves();
    }
};

// first row of the group a GroupKeys key stands for
__device__ __forceinline__ uint32_t group_row(uint32_t key, uint32_t tile, int h) { return tile * 32u + ((key & 3u) << 3) + 4u * (uint32_t)h; }

// One tile of READY pieces into `acc` while the finished scores of the tile before (`prev`) are screened: half_step
// for the bf16 cores.  Nothing is split here, so the tile is three MFMAs per step against two LDS reads; the screening
// of the previous tile (one asm block) goes out behind the first step and runs while the partner wave owns the pipe.
// On entry:  acc = accumulator-init column of the tile, (ph, pl) = this lane's pieces of its step 0,
//            prev = finished scores of tile number tprev_id (class ids in tail_prev).
// On exit:   acc = scores of the tile, prev = init column of the next tile (tail_next), (ph, pl) = its step-0 pieces.
template <int R, class SCR>
__device__ __forceinline__ void pieces_step(const char *ptile, const char *pnext, const float *tail_next, const float *tail_prev,
                                            uint32_t tprev_id, uint32_t label, const u32x4 (&bh)[R / 16], const u32x4 (&bl)[R / 16],
                                            f32x16 &acc, f32x16 &prev, u32x4 &ph, u32x4 &pl, int row, int h, SCR &scr)
{
    using Gm = Geom<R>;
    constexpr int NQ = R / 16, ROWB = Gm::ROW_FLOATS * 4;
    const char *rowp = ptile + row * ROWB + h * 32;
    u32x4 ch = ph, cl = pl, nh = ph, nl = pl;
    f32x16 cnext;
    static_for<0, NQ>([&](auto qq) {
        constexpr int q = decltype(qq)::value;
        if constexpr (q + 1 < NQ) {
            nh = *(const u32x4 *)(rowp + 64 * (q + 1));
            nl = *(const u32x4 *)(rowp + 64 * (q + 1) + 16);
        } else {
            ph = *(const u32x4 *)(pnext + row * ROWB + h * 32);
            pl = *(const u32x4 *)(pnext + row * ROWB + h * 32 + 16);
        }
        const bf16x8 vh = __builtin_bit_cast(bf16x8, ch), vl = __builtin_bit_cast(bf16x8, cl);
        // The screening of the previous tile: one asm block behind the first step -- or (SPREAD) PER scores behind every MFMA
        // from the second of the tile on, pinned there (sched_barrier): the MFMAs of a tile are one dependent chain, the
        // wave issues in order, so only vector work that stands BETWEEN two of them runs under the first one's 32 cycles.
        // prev's last MFMA sits one MFMA and 18 wait states in front of its first reader (8 passes ask for 11; nothing pads
        // hazards for inline asm).
        constexpr bool SPREAD = SCR::SPREAD && NQ >= 4;
        constexpr int PER = NQ >= 8 ? 1 : 2;           // scores per MFMA slot: 16 or 8 slots of the tile's 3 NQ
        auto slot = [&](auto ii) {
            constexpr int sidx = 3 * q + decltype(ii)::value;           // MFMA number in the tile
            if constexpr (SPREAD && sidx >= 1 && (sidx - 1) * PER < 16) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (sidx == 1) scr.begin();
                static_for<0, PER>([&](auto jj) {
                    scr.template one<(sidx - 1) * PER + decltype(jj)::value, (sidx == 1 && decltype(jj)::value == 0) ? 1 : 0>(prev, tail_prev, label, h);
                });
                if constexpr (sidx * PER == 16) scr.end(tprev_id);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, __builtin_bit_cast(bf16x8, bh[q]), acc, 0, 0, 0);
        slot(std::integral_constant<int, 0>{});
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, __builtin_bit_cast(bf16x8, bl[q]), acc, 0, 0, 0);
        slot(std::integral_constant<int, 1>{});
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, __builtin_bit_cast(bf16x8, bh[q]), acc, 0, 0, 0);
        slot(std::integral_constant<int, 2>{});
        if constexpr (!SPREAD && q == 0) scr.template tile<1>(prev, tprev_id, tail_prev, label, h);
        // the next tile's init column; SPREAD: once the last score of prev is in (its registers are free then)
        if constexpr (q == (SPREAD ? (16 / PER) / 3 + 1 : NQ / 2)) {   // NQ == 1: q == 0
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ci = *(const f32x4 *)(tail_next + 8 * g + 4 * h);
                cnext[4 * g + 0] = ci[0]; cnext[4 * g + 1] = ci[1]; cnext[4 * g + 2] = ci[2]; cnext[4 * g + 3] = ci[3];
            }
        }
        ch = nh;
        cl = nl;
    });
    prev = cnext;
}

// The search over `ntiles` tiles whose pieces are ready in LDS: tile t's pieces at pbase + t * pstride (layout:
// BF_TILE_BYTES), its tail (init column, class ids) at tails + t * TILE_FLOATS.  Two accumulators, as search_tiles.
template <int R, class SCR>
__device__ __forceinline__ void search_pieces(const char *pbase, uint32_t pstride, const float *tails, uint32_t ntiles, uint32_t label,
                                              const u32x4 (&bh)[R / 16], const u32x4 (&bl)[R / 16], int s_col, int h, SCR &scr)
{
    using Gm = Geom<R>;
    constexpr int ROWB = Gm::ROW_FLOATS * 4;
    f32x16 acc0, acc1;
    u32x4 ph = *(const u32x4 *)(pbase + s_col * ROWB + h * 32), pl = *(const u32x4 *)(pbase + s_col * ROWB + h * 32 + 16);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 ci = *(const f32x4 *)(tails + 8 * g + 4 * h);
        acc0[4 * g + 0] = ci[0]; acc0[4 * g + 1] = ci[1]; acc0[4 * g + 2] = ci[2]; acc0[4 * g + 3] = ci[3];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = PAD_SCORE;                 // "tile -1": its keys never win
    const float *tail_prev = tails;
    uint32_t tid = 0, t = 0;
    for (; t + 1 < ntiles; t += 2) {
        const char *p0 = pbase + (size_t)t * pstride, *p1 = p0 + pstride, *pn = t + 2 < ntiles ? p1 + pstride : pbase;
        const float *tl0 = tails + (size_t)t * Gm::TILE_FLOATS, *tl1 = tl0 + Gm::TILE_FLOATS;
        const float *tln = t + 2 < ntiles ? tl1 + Gm::TILE_FLOATS : tails;
        pieces_step<R>(p0, p1, tl1, tail_prev, tid, label, bh, bl, acc0, acc1, ph, pl, s_col, h, scr);
        pieces_step<R>(p1, pn, tln, tl0, t, label, bh, bl, acc1, acc0, ph, pl, s_col, h, scr);
        tail_prev = tl1;
        tid = t + 1;
    }
    if (t < ntiles) {                                                 // odd count: acc0 holds the last tile's init column
        const float *tl = tails + (size_t)t * Gm::TILE_FLOATS;
        pieces_step<R>(pbase + (size_t)t * pstride, pbase, tails, tail_prev, tid, label, bh, bl, acc0, acc1, ph, pl, s_col, h, scr);
        scr.template tile<1>(acc0, t, tl, label, h);
    } else if (ntiles != 0) {
        scr.template tile<1>(acc1, tid, tail_prev, label, h);
    }
}

// search_tiles with the products on the bf16 matrix cores; same contract, the screening a policy (above).  The first nbf
// tiles have their pieces ready in LDS at bfimg (search_pieces), the others are split on the fly.
template <int R, class SCR>
__device__ __forceinline__ void search_tiles_bf(const float *img, uint32_t ntiles, uint32_t label, const u32x4 (&bh)[R / 16],
                                                const u32x4 (&bl)[R / 16], int s_col, int h, SCR &scr,
                                                const char *bfimg = nullptr, uint32_t nbf = 0)
{
    using Gm = Geom<R>;
    constexpr int NQ = R / 16;
    if (nbf != 0) search_pieces<R>(bfimg, (uint32_t)BF_TILE_BYTES<R>, img + Gm::W_FLOATS, nbf, label, bh, bl, s_col, h, scr);
    uint32_t t = nbf;
    for (; t < ntiles; ++t) {
        const float *tile = img + (size_t)t * Gm::TILE_FLOATS;
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 ci = *(const f32x4 *)(tile + Gm::W_FLOATS + 8 * g + 4 * h);
            acc[4 * g + 0] = ci[0]; acc[4 * g + 1] = ci[1]; acc[4 * g + 2] = ci[2]; acc[4 * g + 3] = ci[3];
        }
        const float *rowp = tile + Gm::chunk_off(s_col, h);
        static_for<0, NQ>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            const f32x4 a0 = *(const f32x4 *)(rowp + 16 * q), a1 = *(const f32x4 *)(rowp + 16 * q + 8);
            u32x4 ah, al;
            split8(a0, a1, ah, al);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh[q]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl[q]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh[q]), acc, 0, 0, 0);
        });
        // (the wait states between the last MFMA and the first read of its result are the policy's: PRE = 1)
        scr.template tile<1>(acc, t, tile + Gm::W_FLOATS, label, h);
    }
}

// Final step of a batch: the lane pair agrees on the candidates, re-measures them in the direct
// (x - w)^2 form from the LDS image, roots, compares like helpers.rs:70.
//
// The guarantee.  Screening orders the prototypes by keys that differ from the true scores ||w||^2 - 2 x.w by less
// than margin / 2 each (screen_margin).  The best TWO keys are re-measured exactly, so the answer is the reference's
// as long as every other prototype's key lies more than `margin` above the best key -- none of them can then be
// closer than the best candidate.  The third best key says whether that holds: where it does not (three or more
// prototypes closer to each other, seen from this sample, than the f32 screening resolves: dense models, exact
// multi-way ties, duplicates) the sample takes the exact in-order scan below, which IS helpers.rs:55-77 / 91-132 in
// f32 (strict '<' in index order, so the lowest index wins ties).
template <int R, bool MATCHED, bool ROOT = true>
__device__ __forceinline__ void finish_keys(const f32x4 (&xf)[R / 8], const float *img, uint32_t K, uint32_t label, int h,
                                            bool unsafe, float margin, const Keys2 &ku, const KeysM &km, Best1 &wp, Best1 &wn,
                                            uint32_t *crowd_mask = nullptr, uint64_t tile32 = 0, uint64_t N = 0)
{
    // helpers.rs:23 (rooted) vs helpers.rs:44 (squared)
    auto dist = [&](uint32_t k) {
        const float d = exact_sq_lds<R>(xf, img, k, h);
        return ROOT ? sqrtf(d) : d;
    };
    // best two of the pair's four by (key, index), re-measured; the runner-up only where its key is within the margin
    // of the winner's (decided per wave: the re-measurement is 80 VALU + 16 LDS reads for every lane or for none)
    const bool skip = unsafe;                          // NaN / INF / huge rows: their keys mean nothing
    bool all = unsafe;                                 // take every row: unusable scores, or a winner that is not finite
    auto group = [&](uint32_t b1, uint32_t t1, uint32_t b2, uint32_t t2, uint32_t b3, Best1 &w, bool &crowded) {
        uint32_t f, fi, sd, si;
        merge_pair(b1, t1, b2, t2, h, f, fi, sd, si);
        const float third = pair_third(b1, b2, b3);
        const bool have = key_valid(f) && !skip;
        const float fk = __builtin_bit_cast(float, f);
        float d1 = INFINITY;
        if (have) {
            d1 = dist(fi);
            w.d = d1; w.i = fi;
        }
        const bool close = have && key_valid(sd) && !(__builtin_bit_cast(float, sd) - fk > margin);
        if (__builtin_amdgcn_ballot_w64(close) != 0ull) {
            if (have && key_valid(sd)) {
                const float d2 = dist(si);
                if (lexless(d2, si, d1, fi)) { w.d = d2; w.i = si; }
            }
        }
        // crowded: a third key within the margin; a non-finite re-measured winner cannot be trusted either
        crowded = have && !(third - fk > margin);
        if (have) all |= !(w.d < INFINITY);
        return have;
    };
    bool cp = false, cn = false;
    if (!MATCHED) {
        if (!group(ku.b1, ku.t1, ku.b2, ku.t2, ku.b3, wp, cp)) all = true;      // no usable key at all: the scan answers (0, +INF)
    } else {
        group(km.p, km.tp, km.p2, km.tp2, km.p3, wp, cp);
        group(km.n, km.tn, km.n2, km.tn2, km.n3, wn, cn);
    }
    // marked for the finish pass (crowd_mark), or settled here: helpers.rs:55-77 in f32, rows from LDS
    if (!crowd_mark(crowd_mask, tile32, N, cp, cn, all, h) && (cp || cn || all)) scan_exact_lds<R, MATCHED, ROOT>(xf, img, K, label, h, wp, wn);
}

// finish_keys for GroupKeys: the lane pair agrees on the best two GROUPS of its four by (key, first row); all four rows of
// the best group are re-measured exactly (dist(k): the pair's complete distance of row k, rooted where the model roots),
// those of the runner-up where its key is within the margin (decided per wave, as in finish_keys); the third best group key
// is the crowding witness.  Rows are offered in index order with strict '<' inside a group and merged by (distance, index)
// across the two, so the lowest index wins ties (helpers.rs:70).  eligible(k): row k may win this search (k < K, and for the
// labelled searches its class on the right side of the label); scan(): the exact in-order scan for samples that stay unsafe.
template <bool MATCHED, class Dist, class Eligible, class Scan>
__device__ __forceinline__ void finish_groups(int h, bool unsafe, float margin, const Keys2 &ku, const KeysM &km, Best1 &wp, Best1 &wn,
                                              Dist &&dist, Eligible &&eligible, Scan &&scan, unsigned long long *dbg = nullptr,
                                              uint32_t *crowd_mask = nullptr, uint64_t tile32 = 0, uint64_t N = 0)
{
    const bool skip = unsafe;
    bool all = unsafe, cp = false, cn = false;
    auto group = [&](uint32_t b1, uint32_t t1, uint32_t b2, uint32_t t2, uint32_t b3, Best1 &w, bool positive) {
        // merge_pair with the groups' first rows in place of the rows
        const uint32_t k1 = group_row(b1, t1, h), k2 = group_row(b2, t2, h);
        const uint32_t pb1 = xor32(b1), pb2 = xor32(b2), pk1 = xor32(k1), pk2 = xor32(k2);
        uint32_t f, fi, sd, si;
        if (keyless(pb1, pk1, b1, k1)) {
            f = pb1; fi = pk1;
            if (keyless(b1, k1, pb2, pk2)) { sd = b1; si = k1; } else { sd = pb2; si = pk2; }
        } else {
            f = b1; fi = k1;
            if (keyless(pb1, pk1, b2, k2)) { sd = pb1; si = pk1; } else { sd = b2; si = k2; }
        }
        const float third = pair_third(b1, b2, b3);
        const bool have = key_valid(f) && !skip;
        const float fk = __builtin_bit_cast(float, f);
        auto measure4 = [&](uint32_t base, Best1 &best) {           // rows base .. base + 3, index order
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e) {
                const uint32_t k = base + e;
                if (eligible(k, positive)) best.offer(dist(k), k);
            }
        };
        if (have) measure4(fi, w);
        const bool close = have && key_valid(sd) && !(__builtin_bit_cast(float, sd) - fk > margin);
        if (__builtin_amdgcn_ballot_w64(close) != 0ull) {
            if (have && key_valid(sd)) {
                Best1 w2;
                measure4(si, w2);
                if (lexless(w2.d, w2.i, w.d, w.i)) w = w2;
            }
        }
        if (dbg) {
            if (__builtin_amdgcn_ballot_w64(close) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(dbg + 2, 1ull);
            if (have && h == 0 && !(third - fk > margin)) atomicAdd(dbg + (positive ? 3 : 4), 1ull);
            if (have && h == 0 && !(w.d < INFINITY)) atomicAdd(dbg + 5, 1ull);
        }
        if (have) {
            (positive ? cp : cn) = !(third - fk > margin);
            all |= !(w.d < INFINITY);
        }
        return have;
    };
    if (!MATCHED) {
        if (!group(ku.b1, ku.t1, ku.b2, ku.t2, ku.b3, wp, true)) all = true;
    } else {
        group(km.p, km.tp, km.p2, km.tp2, km.p3, wp, true);
        group(km.n, km.tn, km.n2, km.tn2, km.n3, wn, false);
    }
    unsafe = cp || cn || all;
    if (dbg) {
        if (__builtin_amdgcn_ballot_w64(unsafe) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(dbg, 1ull);
        if (unsafe && h == 0) atomicAdd(dbg + 1, 1ull);
    }
    if (!crowd_mark(crowd_mask, tile32, N, cp, cn, all, h) && unsafe) scan();
}

// ------------------------------------------------------------------------------ the same with the exact rows in global memory
// exact_sq_lds / scan_exact_lds / finish_keys for kernels whose LDS holds no f32 image: the rows come from the model's
// plain matrix (K x ldp, zero padded; L2-resident: at most a few MiB), same sums in the same order, so a winner's
// distance does not depend on where its row was read.
template <int R>
__device__ __forceinline__ float exact_sq_g(const f32x4 (&xf)[R / 8], const float *plain, uint32_t ldp, uint32_t k, int h)
{
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    const float *w = plain + (size_t)k * ldp + 4 * h;
    f32x2 p = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < R / 8; ++j) {
        const f32x4 v = *(const f32x4 *)(w + 8 * j);
        const f32x2 d0 = f32x2{xf[j][0], xf[j][1]} - f32x2{v[0], v[1]};
        const f32x2 d1 = f32x2{xf[j][2], xf[j][3]} - f32x2{v[2], v[3]};
        p = __builtin_elementwise_fma(d0, d0, p);
        p = __builtin_elementwise_fma(d1, d1, p);
    }
    const float t = p[0] + p[1];
    return t + xor32(t);
}

template <int R, bool MATCHED, bool ROOT = true>
__device__ __forceinline__ void finish_keys_g(const f32x4 (&xf)[R / 8], const float *plain, uint32_t ldp, const uint32_t *cls, uint32_t K,
                                              uint32_t label, int h, bool unsafe, float margin, const Keys2 &ku, const KeysM &km, Best1 &wp,
                                              Best1 &wn, uint32_t *crowd_mask = nullptr, uint64_t tile32 = 0, uint64_t N = 0)
{
    auto dist = [&](uint32_t k) {
        const float d = exact_sq_g<R>(xf, plain, ldp, k, h);
        return ROOT ? sqrtf(d) : d;
    };
    const bool skip = unsafe;
    bool all = unsafe;
    auto group = [&](uint32_t b1, uint32_t t1, uint32_t b2, uint32_t t2, uint32_t b3, Best1 &w, bool &crowded) {
        uint32_t f, fi, sd, si;
        merge_pair(b1, t1, b2, t2, h, f, fi, sd, si);
        const float third = pair_third(b1, b2, b3);
        const bool have = key_valid(f) && !skip;
        const float fk = __builtin_bit_cast(float, f);
        float d1 = INFINITY;
        if (have) {
            d1 = dist(fi);
            w.d = d1; w.i = fi;
        }
        const bool close = have && key_valid(sd) && !(__builtin_bit_cast(float, sd) - fk > margin);
        if (__builtin_amdgcn_ballot_w64(close) != 0ull) {
            if (have && key_valid(sd)) {
                const float d2 = dist(si);
                if (lexless(d2, si, d1, fi)) { w.d = d2; w.i = si; }
            }
        }
        crowded = have && !(third - fk > margin);
        if (have) all |= !(w.d < INFINITY);
        return have;
    };
    bool cp = false, cn = false;
    if (!MATCHED) {
        if (!group(ku.b1, ku.t1, ku.b2, ku.t2, ku.b3, wp, cp)) all = true;
    } else {
        group(km.p, km.tp, km.p2, km.tp2, km.p3, wp, cp);
        group(km.n, km.tn, km.n2, km.tn2, km.n3, wn, cn);
    }
    if (!crowd_mark(crowd_mask, tile32, N, cp, cn, all, h) && (cp || cn || all)) {     // helpers.rs:55-77 / 91-132 in f32, in index order
        wp = Best1();
        wn = Best1();
        for (uint32_t k = 0; k < K; ++k) {
            const float d = dist(k);
            if (!MATCHED || cls[k] == label) wp.offer(d, k);
            else wn.offer(d, k);
        }
    }
}

// A prototype's row for this lane's chunks, requested now and used later (rows_sq).  Through a buffer descriptor over
// the whole plain matrix: an index that is not a row of it reads zeros instead of faulting.
template <int R>
__device__ __forceinline__ void load_row_g(f32x4 (&w)[R / 8], __amdgpu_buffer_rsrc_t rplain, uint32_t ldp, uint32_t k, int h)
{
    const uint32_t off = (k * ldp + 4u * (uint32_t)h) * 4u;
#pragma unroll
    for (int j = 0; j < R / 8; ++j) w[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rplain, off, j * 32, 0));
}
template <int R>
__device__ __forceinline__ float rows_sq(const f32x4 (&xf)[R / 8], const f32x4 (&w)[R / 8])
{
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    f32x2 p = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < R / 8; ++j) {
        const f32x2 d0 = f32x2{xf[j][0], xf[j][1]} - f32x2{w[j][0], w[j][1]};
        const f32x2 d1 = f32x2{xf[j][2], xf[j][3]} - f32x2{w[j][2], w[j][3]};
        p = __builtin_elementwise_fma(d0, d0, p);
        p = __builtin_elementwise_fma(d1, d1, p);
    }
    const float t = p[0] + p[1];
    return t + xor32(t);
}

// finish_keys_g for the unlabelled search, with the best candidate's row REQUESTED before `between` runs and used after
// it: the caller issues the next batch's sample loads in `between` -- vmcnt completes in order, so loads issued ahead
// of the candidate's row would have to land first, and those come from HBM (measured on config 2: 15.4k cycles of
// epilogue per batch with the samples requested first, 5.8k this way).  Same sums as exact_sq_g.
template <int R, bool ROOT = true, class Between>
__device__ __forceinline__ void finish_keys_ahead(const f32x4 (&xf)[R / 8], const float *plain, uint32_t ldp, uint32_t K, int h, bool unsafe,
                                                  float margin, const Keys2 &ku, Best1 &wp, Between &&between,
                                                  uint32_t *crowd_mask = nullptr, uint64_t tile32 = 0, uint64_t N = 0)
{
    const __amdgpu_buffer_rsrc_t rplain = __builtin_amdgcn_make_buffer_rsrc((void *)plain, 0, (int)(K * ldp * 4u), 0x00020000);
    auto dist = [&](uint32_t k) {
        f32x4 w[R / 8];
        load_row_g<R>(w, rplain, ldp, k, h);
        const float d = rows_sq<R>(xf, w);
        return ROOT ? sqrtf(d) : d;
    };
    uint32_t f, fi, sd, si;
    merge_pair(ku.b1, ku.t1, ku.b2, ku.t2, h, f, fi, sd, si);
    const float third = pair_third(ku.b1, ku.b2, ku.b3);
    const bool have = key_valid(f) && !unsafe;
    const float fk = __builtin_bit_cast(float, f);
    // Whether any sample of the wave needs its runner-up measured as well is known from the keys alone.  Then BOTH rows
    // are requested up front and `between` runs behind their use: a second row requested behind `between`'s loads would
    // queue behind an HBM round trip (vmcnt completes in order) -- on uniform random data one wave-batch in three.
    const bool close = have && key_valid(sd) && !(__builtin_bit_cast(float, sd) - fk > margin);
    const bool any_close = __builtin_amdgcn_ballot_w64(close) != 0ull;
    f32x4 row[R / 8];
    load_row_g<R>(row, rplain, ldp, have ? fi : 0u, h);
    float d1 = INFINITY;
    if (any_close) {
        f32x4 row2[R / 8];
        load_row_g<R>(row2, rplain, ldp, close ? si : 0u, h);           // (lanes that are not close read row 0: one line for all)
        if (have) {
            d1 = rows_sq<R>(xf, row);
            if (ROOT) d1 = sqrtf(d1);
            wp.d = d1; wp.i = fi;
        }
        float d2 = rows_sq<R>(xf, row2);               // (wave-uniform call: the lane pair exchanges halves inside)
        if (ROOT) d2 = sqrtf(d2);
        if (close && lexless(d2, si, d1, fi)) { wp.d = d2; wp.i = si; }
        asm volatile("" ::: "memory");
        between();
        asm volatile("" ::: "memory");
    } else {
        asm volatile("" ::: "memory");
        between();
        asm volatile("" ::: "memory");
        if (have) {
            d1 = rows_sq<R>(xf, row);
            if (ROOT) d1 = sqrtf(d1);
            wp.d = d1; wp.i = fi;
        }
    }
    // crowded (a third key within the margin), a non-finite winner, or no usable key at all: marked for the finish pass
    // (crowd_mark), or the exact scan answers here
    const bool crowded = have && !(third - fk > margin), all = !have || !(wp.d < INFINITY);
    if (!crowd_mark(crowd_mask, tile32, N, crowded, false, all, h) && (crowded || all)) {
        wp = Best1();                                   // helpers.rs:55-77 in f32, in index order
        for (uint32_t k = 0; k < K; ++k) wp.offer(dist(k), k);
    }
}

}  // namespace

}  // namespace vq
