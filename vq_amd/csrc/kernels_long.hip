// Euclidean models with rows longer than 256 values (VQ / LVQ / GLVQ: src/helpers.rs:11-24, 55-77, 91-132) on the
// matrix cores.  Up to 256 prototypes the reduction runs once, over chunks of 128 (more than 64 prototypes: 64) values;
// more prototypes are worked in groups of 256 (8 tiles), the rows fetched once more per group:
//   pass 1  every chunk of the wave's 32 rows is fetched once and multiplied into ALL the tiles' accumulators
//           (NT x 16 registers stay live across the chunks); the prototypes' A fragments stream from L2 through a
//           register ring, one 1 KiB piece per (chunk, tile, step), so there is no LDS stage and no barrier;
//   screen  packed keys over the finished scores ||w_k||^2 - 2 x.w_k, best two per lane (labelled: best per class group);
//   pass 2  the rows are fetched again and the candidates re-measured in the direct sum (x_d - w_kd)^2 form,
//           rooted, compared like helpers.rs:70.  The second fetch is the price of rows that fit neither the
//           register file nor LDS; the kernel is HBM-bound at about twice the algorithmic bytes.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "mfma_device.h"

namespace vq {

// The rows of a batch of 32 samples, walked in chunks of NJ * 8 values: lane (s, h) holds the 16-byte pieces 2 j + h
// of its sample's chunk.  Rows past N and bytes past the last row read as zeros (buffer range check); values at or
// past D inside the last chunk -- they belong to the next row -- are cleared by cut().
template <int NJ>
struct RowWalker {
    static constexpr uint32_t CHUNK = NJ * 8;
    const float *X;
    uint64_t N;
    uint32_t D, row_bytes, voff;
    int h;
    const float *mu = nullptr;            // the model's centre (EngineArgs::mu), zero padded to whole chunks; null: none
    __device__ __forceinline__ RowWalker(const float *X_, uint64_t N_, uint32_t D_, int s_col, int h_)
        : X(X_), N(N_), D(D_), row_bytes(D_ * 4u), voff((uint32_t)s_col * D_ * 4u + (uint32_t)h_ * 16u), h(h_)
    {
    }
    // x -= mu for chunk c, after cut() (mu is zero past D, so cleared values stay zero): one v_sub per value, the same
    // f32 operation the host applied to the prototypes (centre_prototypes).  The centre's pieces come from L1 / L2 -- the
    // 32 lanes of a half read the same 16 bytes -- as the chunk is worked: rows of this length leave neither registers
    // to keep it nor (k_omega_long) LDS.
    __device__ __forceinline__ void centre(f32x4 (&xs)[NJ], uint32_t c) const
    {
        if (mu == nullptr) return;
        const uint32_t padded = (D + CHUNK - 1) / CHUNK * CHUNK;
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void *)mu, 0, (int)(padded * 4u), 0x00020000);
        // two pieces at a time (the memory clobber keeps the next loads behind them): all NJ in flight at once would
        // take NJ * 4 registers these kernels do not have
#pragma unroll
        for (int j0 = 0; j0 < NJ; j0 += 2) {
            f32x4 m[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
                m[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, (uint32_t)h * 16u + (j0 + j) * 32, (int)(c * CHUNK * 4u), 0));
#pragma unroll
            for (int j = 0; j < 2; ++j) xs[j0 + j] = xs[j0 + j] - m[j];
            asm volatile("" : "+v"(xs[j0]), "+v"(xs[j0 + 1])::"memory");
        }
    }
    // num_records of a descriptor that starts at batch bb of an array with `unit` bytes per sample
    __device__ __forceinline__ uint32_t records(uint64_t bb, uint32_t unit) const
    {
        const uint64_t first = bb * 32;
        if (first >= N) return 0u;
        const uint64_t left = (N - first) * unit;
        return left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)left;
    }
    // chunk c of batch bb
    __device__ __forceinline__ void fetch(f32x4 (&xs)[NJ], uint64_t bb, uint32_t c) const
    {
        const __amdgpu_buffer_rsrc_t rx =
            __builtin_amdgcn_make_buffer_rsrc((void *)(X + bb * 32 * D), 0, (int)records(bb, row_bytes), 0x00020000);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            xs[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff + j * 32, (int)(c * CHUNK * 4u), 0));
    }
    __device__ __forceinline__ void cut(f32x4 (&xs)[NJ], uint32_t c) const
    {
        if ((c + 1u) * CHUNK > D) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) xs[j] = mask_chunk(xs[j], c * (CHUNK / 4) + (uint32_t)(2 * j + h), D);
        }
    }
    // class id of this lane's sample in batch bb
    __device__ __forceinline__ uint32_t label(const uint32_t *y, uint64_t bb, int s_col) const
    {
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)(y + bb * 32), 0, (int)records(bb, 4u), 0x00020000);
        return __builtin_amdgcn_raw_buffer_load_b32(ry, (uint32_t)s_col * 4u, 0, 0);
    }
};

template <int NT, bool MATCHED, int NW = 8>
__global__ void __launch_bounds__(NW * 64, NW / 4) k_long(const EngineArgs a)
{
    // chunks of 128 values (16 steps of 4 MFMAs per tile) while the accumulators of up to 2 tiles leave room for
    // two chunk sets of that size, chunks of 64 values for 4 and 8 tiles; DEPTH pieces in flight
    constexpr int NJ = NT <= 2 ? 16 : 8, DEPTH = 8;
    constexpr uint32_t CHUNK = NJ * 8;
    static_assert(NJ % DEPTH == 0, "ring slots are compile-time");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    const uint32_t NC = (a.D + CHUNK - 1) / CHUNK, NG = a.ngroups1, npieces = NG * NC * NT * NJ;

    // A fragments: piece p = ((group * NC + chunk) * NT + tile) * NJ + step, this lane's 16 bytes at p * 1024 + lane * 16
    const __amdgpu_buffer_rsrc_t rom = __builtin_amdgcn_make_buffer_rsrc((void *)a.img1, 0, (int)(npieces * 1024u), 0x00020000);
    const uint32_t om_off = (uint32_t)lane * 16u;
    auto piece = [&](uint32_t p) {                     // past the last piece: range-check zeros, never used
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rom, om_off, (int)(p * 1024u), 0));
    };
    const float *tails = a.img2;                       // per tile: 32 x ||w_k||^2 (pad rows PAD_SCORE), 32 x class id

    constexpr uint32_t BLOCK = 2;
    BlockQueue<BLOCK> queue(a.queue, lane, blockIdx.x * NW + (uint32_t)__builtin_amdgcn_readfirstlane(wave), gridDim.x * NW, a.nbatches);
    uint64_t block0 = (uint64_t)(blockIdx.x * NW + (uint32_t)__builtin_amdgcn_readfirstlane(wave)) * BLOCK;
    queue.request();

    RowWalker<NJ> rows(a.X, a.N, a.D, s_col, h);
    rows.mu = a.mu;                                    // centred model: both passes work on x - mu
    f32x4 xa[NJ], xb[NJ];                              // two chunks of the rows, worked alternately
    // fn(chunk registers, chunk number) over all chunks of batch bb; xa must hold (or be receiving) chunk 0
    auto for_chunks = [&](uint64_t bb, auto &&fn) {
        for (uint32_t c = 0; c < NC; c += 2) {
            if (c + 1 < NC) rows.fetch(xb, bb, c + 1);
            rows.cut(xa, c);
            rows.centre(xa, c);
            fn(xa, c);
            if (c + 1 < NC) {
                if (c + 2 < NC) rows.fetch(xa, bb, c + 2);
                rows.cut(xb, c + 1);
                rows.centre(xb, c + 1);
                fn(xb, c + 1);
            }
        }
    };
    // sum over this lane's part of (x - w_k)^2 for two prototypes, all chunks (rows fetched again)
    auto exact_pair = [&](uint64_t bb, uint32_t k0, uint32_t k1, float &e0, float &e1) {
        const float *w0 = a.plain + (size_t)k0 * a.ldp + 4 * h, *w1 = a.plain + (size_t)k1 * a.ldp + 4 * h;
        float p0 = 0.0f, p1 = 0.0f;
        rows.fetch(xa, bb, 0);
        for_chunks(bb, [&](const f32x4 (&xs)[NJ], uint32_t c) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f32x4 v0 = *(const f32x4 *)(w0 + c * CHUNK + 8 * j), v1 = *(const f32x4 *)(w1 + c * CHUNK + 8 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d0 = xs[j][e] - v0[e], d1 = xs[j][e] - v1[e];
                    p0 = fmaf(d0, d0, p0);
                    p1 = fmaf(d1, d1, p1);
                }
            }
        });
        e0 = p0 + xor32(p0);
        e1 = p1 + xor32(p1);
    };

    uint32_t ln[BLOCK];
#pragma unroll
    for (uint32_t q = 0; q < BLOCK; ++q) ln[q] = 0u;
    auto fetch_label = [&](uint32_t q, uint64_t bb) {
        if (MATCHED) ln[q] = rows.label(a.y, bb, s_col);
    };
    rows.fetch(xa, block0, 0);
    fetch_label(0, block0);
    fetch_label(1, block0 + 1);

    while (block0 < a.nbatches) {
        const uint64_t nb0 = queue.take();
        queue.request();
#pragma unroll
        for (uint32_t q = 0; q < BLOCK; ++q) {
            const uint64_t b = block0 + q;
            const uint64_t bnext = q + 1 < BLOCK ? b + 1 : nb0;
            const uint32_t label = ln[q];

            f32x4 ring[DEPTH];
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) ring[i] = piece((uint32_t)i);
            uint32_t pnext = DEPTH;                    // next piece to request: the ring runs on across the groups
            float ss = 0.0f;
            Keys2 ku;
            KeysM km;
            for (uint32_t grp = 0; grp < NG; ++grp) {
            // ---- pass 1: all tiles' scores of this group, accumulators start at the ||w_k||^2 column
            const float *gtails = tails + (size_t)grp * NT * TAIL_FLOATS;
            f32x16 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 ci = *(const f32x4 *)(gtails + t * TAIL_FLOATS + 8 * g + 4 * h);
                    acc[t][4 * g + 0] = ci[0]; acc[t][4 * g + 1] = ci[1]; acc[t][4 * g + 2] = ci[2]; acc[t][4 * g + 3] = ci[3];
                }
            for_chunks(b, [&](const f32x4 (&xs)[NJ], uint32_t) {
                if (grp == 0) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) ss = fmaf(xs[j][e], xs[j][e], ss);
                }
                static_for<0, NT>([&](auto tt) {
                    static_for<0, NJ>([&](auto jj) {
                        constexpr int j = decltype(jj)::value;
                        const f32x4 av = ring[j % DEPTH];
                        ring[j % DEPTH] = piece(pnext++);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[decltype(tt)::value] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], xs[j][e], acc[decltype(tt)::value], 0, 0, 0);
                    });
                });
            });
            if (grp + 1 < NG) rows.fetch(xa, b, 0);       // the next group walks the rows again

            // ---- screening
            static_for<0, NT>([&](auto tt) {
                constexpr int t = decltype(tt)::value;
                const uint32_t tile = grp * NT + (uint32_t)t;
                if constexpr (!MATCHED) {
                    const Keys2 before = ku;
                    screen16<1>(acc[t], ku);
                    settle_tiles(ku, before, tile);
                } else {
                    const KeysM before = km;
                    const uint32_t *tcls = (const uint32_t *)(gtails + t * TAIL_FLOATS) + 32 + 4 * h;
                    settle_mfma(acc[t]);                       // asm readers of fresh MFMA results: wait states by hand
                    static_for<0, 16>([&](auto oo) {
                        constexpr int o = decltype(oo)::value;
                        const uint32_t key = pack_key(acc[t][o], o);
                        const bool match = tcls[8 * (o >> 2) + (o & 3)] == label;     // helpers.rs:113 / :122
                        const uint32_t kp = match ? key : KEY_NONE, kn = match ? KEY_NONE : key;
                        offer_m(km, kp, kn);
                    });
                    settle_tiles_m(km, before, tile);
                }
            });
            }
            ss += xor32(ss);
            bool unsafe = !(ss < a.ss_limit);          // NaN / INF / huge rows: exact scan below

            // ---- the lane pair agrees on two candidates; pass 2 re-measures them
            uint32_t c0, c1;                           // unlabelled: best two; labelled: best matching, best non-matching
            bool v0, v1;
            uint32_t r0 = 0u, r1 = 0u;                 // labelled: the two groups' runner-ups, where their keys are close
            bool w0 = false, w1 = false;
            // a third key within the margin of a group's best: the exact scan below (the guarantee: finish_keys, mfma_device.h)
            const float margin = screen_margin(ss, a);
            if (!MATCHED) {
                uint32_t f, sd;
                merge_pair(ku.b1, ku.t1, ku.b2, ku.t2, h, f, c0, sd, c1);
                v0 = key_valid(f);
                v1 = key_valid(sd);
                if (v0) unsafe |= !(pair_third(ku.b1, ku.b2, ku.b3) - __builtin_bit_cast(float, f) > margin);
            } else {
                uint32_t fp, sp, fn, sn;
                merge_pair(km.p, km.tp, km.p2, km.tp2, h, fp, c0, sp, r0);
                merge_pair(km.n, km.tn, km.n2, km.tn2, h, fn, c1, sn, r1);
                v0 = key_valid(fp);
                v1 = key_valid(fn);
                w0 = v0 && key_valid(sp) && !(__builtin_bit_cast(float, sp) - __builtin_bit_cast(float, fp) > margin);
                w1 = v1 && key_valid(sn) && !(__builtin_bit_cast(float, sn) - __builtin_bit_cast(float, fn) > margin);
                if (v0) unsafe |= !(pair_third(km.p, km.p2, km.p3) - __builtin_bit_cast(float, fp) > margin);
                if (v1) unsafe |= !(pair_third(km.n, km.n2, km.n3) - __builtin_bit_cast(float, fn) > margin);
            }
            float e0, e1;
            exact_pair(b, v0 ? c0 : 0u, v1 ? c1 : 0u, e0, e1);
            Best1 wp, wn;
            if (!MATCHED) {
                if (v0) {
                    const float d1 = sqrtf(e0);
                    wp.d = d1; wp.i = c0;
                    if (v1) {
                        const float d2 = sqrtf(e1);
                        if (lexless(d2, c1, d1, c0)) { wp.d = d2; wp.i = c1; }
                    }
                    unsafe |= !(wp.d < INFINITY);
                } else {
                    unsafe = true;
                }
            } else {
                if (v0) { wp.d = sqrtf(e0); wp.i = c0; }
                if (v1) { wn.d = sqrtf(e1); wn.i = c1; }
                if (__builtin_amdgcn_ballot_w64(w0 || w1) != 0ull) {      // a runner-up within the margin somewhere in the wave
                    float e2, e3;
                    exact_pair(b, w0 ? r0 : 0u, w1 ? r1 : 0u, e2, e3);
                    if (w0) {
                        const float d2 = sqrtf(e2);
                        if (lexless(d2, r0, wp.d, wp.i)) { wp.d = d2; wp.i = r0; }
                    }
                    if (w1) {
                        const float d3 = sqrtf(e3);
                        if (lexless(d3, r1, wn.d, wn.i)) { wn.d = d3; wn.i = r1; }
                    }
                }
                if (v0) unsafe |= !(wp.d < INFINITY);
                if (v1) unsafe |= !(wn.d < INFINITY);
            }
            if (__builtin_amdgcn_ballot_w64(unsafe) != 0ull) {
                // exact scan in index order (helpers.rs:55-77 in f32), two prototypes per sweep over the rows; the
                // whole wave walks it, lanes whose screening was usable keep their result
                Best1 sp, sn;
                for (uint32_t k = 0; k < a.K; k += 2) {
                    const uint32_t kb = k + 1 < a.K ? k + 1 : k;
                    float s0, s1;
                    exact_pair(b, k, kb, s0, s1);
                    const float d0 = sqrtf(s0), d1 = sqrtf(s1);
                    if (!MATCHED || a.cls[k] == label) sp.offer(d0, k); else sn.offer(d0, k);
                    if (kb != k) {
                        if (!MATCHED || a.cls[kb] == label) sp.offer(d1, kb); else sn.offer(d1, kb);
                    }
                }
                if (unsafe) { wp = sp; wn = sn; }
            }
            {
                const int rec = (int)rows.records(b, 4u);
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
            rows.fetch(xa, bnext, 0);                     // the next batch's first chunk
            fetch_label(q, nb0 + q);
        }
        block0 = nb0;
    }
}

// ------------------------------------------------------------------------------ Omega models with long rows
// GMLVQ / LiRaMLVQ with rows above 256 values and an Omega of at most 128 rows (src/helpers.rs:35-45 through
// 55-77, 91-132).  The projection y = Omega x runs over chunks of 64 values like pass 1 of k_long -- Omega's A
// fragments from L2 through the ring, all R2 / 32 tiles of Omega rows accumulating at once -- and leaves y in the
// accumulators, which are the search stage's B fragment (engine.h).  From there on it is k_omega_pipe: projected
// prototypes resident in LDS, pipelined packed-key search, exact re-measurement ||y - p_k||^2 (squared) from the LDS
// image.  The search space is y, which stays in registers, so the rows are fetched once.
template <int R2, bool MATCHED, int NW = 8>
__global__ void __launch_bounds__(NW * 64, NW / 4) k_omega_long(const EngineArgs a)
{
    using Gs = Geom<R2>;
    constexpr int NT = R2 / 32, NJ = 8, DEPTH = 8;        // chunks of 64 values here: two chunk sets, Omega's tiles and y share the registers
    constexpr uint32_t CHUNK = NJ * 8;
    __shared__ __attribute__((aligned(16))) char smem[RES_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    const uint32_t ntiles = a.res_tiles;
    const uint32_t NC = (a.D + CHUNK - 1) / CHUNK, npieces = NC * NT * NJ;
    {   // projected prototypes -> LDS
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
    const __amdgpu_buffer_rsrc_t rom = __builtin_amdgcn_make_buffer_rsrc((void *)a.img1, 0, (int)(npieces * 1024u), 0x00020000);
    const uint32_t om_off = (uint32_t)lane * 16u;
    auto piece = [&](uint32_t p) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rom, om_off, (int)(p * 1024u), 0));
    };

    constexpr uint32_t BLOCK = 2;
    BlockQueue<BLOCK> queue(a.queue, lane, blockIdx.x * NW + (uint32_t)__builtin_amdgcn_readfirstlane(wave), gridDim.x * NW, a.nbatches);
    uint64_t block0 = (uint64_t)(blockIdx.x * NW + (uint32_t)__builtin_amdgcn_readfirstlane(wave)) * BLOCK;
    queue.request();

    RowWalker<NJ> rows(a.X, a.N, a.D, s_col, h);
    rows.mu = a.mu;                                    // centred model: y = Omega (x - mu), p_k = Omega (w_k - mu)
    f32x4 xa[NJ], xb[NJ];
    uint32_t ln[BLOCK];
#pragma unroll
    for (uint32_t q = 0; q < BLOCK; ++q) ln[q] = 0u;
    auto fetch_label = [&](uint32_t q, uint64_t bb) {
        if (MATCHED) ln[q] = rows.label(a.y, bb, s_col);
    };
    rows.fetch(xa, block0, 0);
    fetch_label(0, block0);
    fetch_label(1, block0 + 1);

    while (block0 < a.nbatches) {
        const uint64_t nb0 = queue.take();
        queue.request();
#pragma unroll
        for (uint32_t q = 0; q < BLOCK; ++q) {
            const uint64_t b = block0 + q;
            const uint64_t bnext = q + 1 < BLOCK ? b + 1 : nb0;
            const uint32_t label = ln[q];

            // ---- projection over the chunks: acc[t] = rows 32 t .. 32 t + 31 of Omega x
            f32x16 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
            f32x4 ring[DEPTH];
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) ring[i] = piece((uint32_t)i);
            uint32_t pnext = DEPTH;
            auto mma = [&](const f32x4 (&xs)[NJ]) {
                static_for<0, NT>([&](auto tt) {
                    static_for<0, NJ>([&](auto jj) {
                        constexpr int j = decltype(jj)::value;
                        const f32x4 av = ring[j % DEPTH];
                        ring[j % DEPTH] = piece(pnext++);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[decltype(tt)::value] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], xs[j][e], acc[decltype(tt)::value], 0, 0, 0);
                    });
                });
            };
            for (uint32_t c = 0; c < NC; c += 2) {
                if (c + 1 < NC) rows.fetch(xb, b, c + 1);
                rows.cut(xa, c);
                rows.centre(xa, c);
                mma(xa);
                if (c + 1 < NC) {
                    if (c + 2 < NC) rows.fetch(xa, b, c + 2);
                    rows.cut(xb, c + 1);
                    rows.centre(xb, c + 1);
                    mma(xb);
                }
            }
            f32x4 yf[R2 / 8];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) yf[4 * t + g] = f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
            asm volatile("" ::: "memory");
            rows.fetch(xa, bnext, 0);                     // x is dead: the next batch's first chunk lands during the search
            fetch_label(q, nb0 + q);

            float ss = 0.0f;
#pragma unroll
            for (int j = 0; j < R2 / 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) ss = fmaf(yf[j][e], yf[j][e], ss);
            ss += xor32(ss);
            const bool unsafe = !(ss < a.ss_limit);
            const float margin = unsafe ? 0.0f : screen_margin(ss, a);

            Keys2 ku;
            KeysM km;
            search_tiles<R2, MATCHED>(img, ntiles, label, yf, s_col, h, ku, km);
            Best1 wp, wn;
            finish_keys<R2, MATCHED, false>(yf, img, a.K, label, h, unsafe, margin, ku, km, wp, wn);
            {
                const int rec = (int)rows.records(b, 4u);
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
        }
        block0 = nb0;
    }
}

// ------------------------------------------------------------------------------ LGMLVQ with long rows
// Local Omegas over rows above 256 values (src/lgmlvq/lgmlvq.rs:436-455, 468-506).  Every Omega_j is padded to
// whole tiles of 32 rows and the tiles of all prototypes are stacked: pass 1 of k_long over groups of NT tiles with
// the accumulators starting at -Omega_j w_j, so a finished tile holds 32 rows of z = Omega_j (x - w_j).  The
// epilogue sums z^2 over the tile's rows and the tiles of a prototype and offers the total in index order
// (strict '<', lgmlvq.rs:448); there is no screening and no second pass, the sum IS the direct form.
template <int NT, bool MATCHED, int NW = 8>
__global__ void __launch_bounds__(NW * 64, NW / 4) k_local_long(const EngineArgs a)
{
    constexpr int NJ = 8, DEPTH = 8;
    constexpr uint32_t CHUNK = NJ * 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    const uint32_t NC = (a.D + CHUNK - 1) / CHUNK, NG = a.ngroups1, npieces = NG * NC * NT * NJ;
    const __amdgpu_buffer_rsrc_t rom = __builtin_amdgcn_make_buffer_rsrc((void *)a.img1, 0, (int)(npieces * 1024u), 0x00020000);
    const uint32_t om_off = (uint32_t)lane * 16u;
    auto piece = [&](uint32_t p) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rom, om_off, (int)(p * 1024u), 0));
    };
    const float *tails = a.img2;                       // per tile: 32 x -(Omega_j w_j) of its rows (pad rows 0)

    constexpr uint32_t BLOCK = 2;
    BlockQueue<BLOCK> queue(a.queue, lane, blockIdx.x * NW + (uint32_t)__builtin_amdgcn_readfirstlane(wave), gridDim.x * NW, a.nbatches);
    uint64_t block0 = (uint64_t)(blockIdx.x * NW + (uint32_t)__builtin_amdgcn_readfirstlane(wave)) * BLOCK;
    queue.request();

    RowWalker<NJ> rows(a.X, a.N, a.D, s_col, h);
    rows.mu = a.mu;                                    // centred model: z = Omega_j (x - mu) - Omega_j (w_j - mu)
    f32x4 xa[NJ], xb[NJ];
    uint32_t ln[BLOCK];
#pragma unroll
    for (uint32_t q = 0; q < BLOCK; ++q) ln[q] = 0u;
    auto fetch_label = [&](uint32_t q, uint64_t bb) {
        if (MATCHED) ln[q] = rows.label(a.y, bb, s_col);
    };
    rows.fetch(xa, block0, 0);
    fetch_label(0, block0);
    fetch_label(1, block0 + 1);

    while (block0 < a.nbatches) {
        const uint64_t nb0 = queue.take();
        queue.request();
#pragma unroll
        for (uint32_t q = 0; q < BLOCK; ++q) {
            const uint64_t b = block0 + q;
            const uint64_t bnext = q + 1 < BLOCK ? b + 1 : nb0;
            const uint32_t label = ln[q];
            f32x4 ring[DEPTH];
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) ring[i] = piece((uint32_t)i);
            uint32_t pnext = DEPTH;                    // the ring runs on across the groups
            Best1 wp, wn;
            float dsum = 0.0f;
            uint32_t proto = 0, tin = 0;               // prototype of the current tile, its tile inside the prototype
            for (uint32_t grp = 0; grp < NG; ++grp) {
                const float *gtails = tails + (size_t)grp * NT * TAIL_FLOATS;
                f32x16 acc[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 ci = *(const f32x4 *)(gtails + t * TAIL_FLOATS + 8 * g + 4 * h);
                        acc[t][4 * g + 0] = ci[0]; acc[t][4 * g + 1] = ci[1]; acc[t][4 * g + 2] = ci[2]; acc[t][4 * g + 3] = ci[3];
                    }
                auto mma = [&](const f32x4 (&xs)[NJ]) {
                    static_for<0, NT>([&](auto tt) {
                        static_for<0, NJ>([&](auto jj) {
                            constexpr int j = decltype(jj)::value;
                            const f32x4 av = ring[j % DEPTH];
                            ring[j % DEPTH] = piece(pnext++);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc[decltype(tt)::value] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], xs[j][e], acc[decltype(tt)::value], 0, 0, 0);
                        });
                    });
                };
                for (uint32_t c = 0; c < NC; c += 2) {
                    if (c + 1 < NC) rows.fetch(xb, b, c + 1);
                    rows.cut(xa, c);
                    rows.centre(xa, c);
                    mma(xa);
                    if (c + 1 < NC) {
                        if (c + 2 < NC) rows.fetch(xa, b, c + 2);
                        rows.cut(xb, c + 1);
                        rows.centre(xb, c + 1);
                        mma(xb);
                    }
                }
                asm volatile("" ::: "memory");
                rows.fetch(xa, grp + 1 < NG ? b : bnext, 0);     // the next group walks the rows again; after the last, the next batch
                // ---- ||z||^2 per prototype, offered in index order
                static_for<0, NT>([&](auto tt) {
                    constexpr int t = decltype(tt)::value;
#pragma unroll
                    for (int r = 0; r < 16; ++r) dsum = fmaf(acc[t][r], acc[t][r], dsum);
                    if (++tin == a.tiles_per_proto) {
                        const float total = dsum + xor32(dsum);
                        dsum = 0.0f;
                        if (proto < a.K) {                         // pad tiles past the last prototype never compete
                            if (!MATCHED || a.cls[proto] == label) wp.offer(total, proto);     // lgmlvq.rs:448 / :487
                            else wn.offer(total, proto);                                        // lgmlvq.rs:496
                        }
                        tin = 0;
                        ++proto;
                    }
                });
            }
            {
                const int rec = (int)rows.records(b, 4u);
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
            fetch_label(q, nb0 + q);
        }
        block0 = nb0;
    }
}

namespace {
template <int NT>
hipError_t launch_long(const EngineArgs &a, bool matched, hipStream_t st)
{
    const uint64_t wgs = (a.nbatches + 7) / 8;
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);
    const hipError_t e = zero_queue(a, st);
    if (e != hipSuccess) return e;
    if (matched)
        hipLaunchKernelGGL((k_long<NT, true>), dim3(grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((k_long<NT, false>), dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}
}  // namespace

namespace {
template <int R2>
hipError_t launch_omega_long(const EngineArgs &a, bool matched, hipStream_t st)
{
    const uint64_t wgs = (a.nbatches + 7) / 8;
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);
    const hipError_t e = zero_queue(a, st);
    if (e != hipSuccess) return e;
    if (matched)
        hipLaunchKernelGGL((k_omega_long<R2, true>), dim3(grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((k_omega_long<R2, false>), dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}
}  // namespace

LaunchFn omega_long_launcher(int R2)
{
    switch (R2) {
    case 32: return &launch_omega_long<32>;
    case 64: return &launch_omega_long<64>;
    case 128: return &launch_omega_long<128>;
    default: return nullptr;
    }
}

namespace {
template <int NT>
hipError_t launch_local_long(const EngineArgs &a, bool matched, hipStream_t st)
{
    const uint64_t wgs = (a.nbatches + 7) / 8;
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);
    const hipError_t e = zero_queue(a, st);
    if (e != hipSuccess) return e;
    if (matched)
        hipLaunchKernelGGL((k_local_long<NT, true>), dim3(grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((k_local_long<NT, false>), dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}
}  // namespace

LaunchFn local_long_launcher(int tiles)
{
    switch (tiles) {
    case 1: return &launch_local_long<1>;
    case 2: return &launch_local_long<2>;
    case 4: return &launch_local_long<4>;
    default: return nullptr;
    }
}

// ------------------------------------------------------------------------------ project / project_omega over long rows
// A global Omega of at most 128 rows over rows above 256 values (src/gmlvq/gmlvq.rs:354-372,
// src/liramlvq/liramlvq.rs:401-446).  Stage 1 is the projection of k_omega_long: y = Omega x over chunks of 64
// values, Omega's A fragments from L2.  project_omega stores y.  project goes on to Lambda x = Omega^T y with y as
// the B fragment straight out of the accumulators: one output tile of 32 columns per pass over y, the A fragments
// of Omega^T (pieces2) through a second ring, 16-byte stores as in k_project.
struct LongProjectArgs {
    const float *X;
    uint64_t N;
    uint32_t D, M;
    const float *pieces1;     // Omega: NC x NT x 8 pieces
    const float *pieces2;     // Omega^T: ceil(D / 32) x (R2 / 8) pieces
    float *Y;                 // N x ldy
    uint32_t ldy;
    uint64_t nbatches;
    uint32_t *queue;
};

template <int R2, bool LAMBDA, int NW = 8>
__global__ void __launch_bounds__(NW * 64, NW / 4) k_project_long(const LongProjectArgs a)
{
    constexpr int NT = R2 / 32, NJ = 8, DEPTH = 8;
    constexpr int NJ2 = R2 / 8, DEPTH2 = NJ2 < 8 ? NJ2 : 8;
    constexpr uint32_t CHUNK = NJ * 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    const uint32_t NC = (a.D + CHUNK - 1) / CHUNK, npieces = NC * NT * NJ, DT = (a.D + 31) / 32;
    const __amdgpu_buffer_rsrc_t rom = __builtin_amdgcn_make_buffer_rsrc((void *)a.pieces1, 0, (int)(npieces * 1024u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rot =
        __builtin_amdgcn_make_buffer_rsrc((void *)a.pieces2, 0, LAMBDA ? (int)(DT * NJ2 * 1024u) : 0, 0x00020000);
    const uint32_t om_off = (uint32_t)lane * 16u;
    auto piece = [&](uint32_t p) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rom, om_off, (int)(p * 1024u), 0));
    };
    auto piece2 = [&](uint32_t p) {                    // past the last piece: range-check zeros, never used
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rot, om_off, (int)(p * 1024u), 0));
    };

    constexpr uint32_t BLOCK = 2;
    BlockQueue<BLOCK> queue(a.queue, lane, blockIdx.x * NW + (uint32_t)__builtin_amdgcn_readfirstlane(wave), gridDim.x * NW, a.nbatches);
    uint64_t block0 = (uint64_t)(blockIdx.x * NW + (uint32_t)__builtin_amdgcn_readfirstlane(wave)) * BLOCK;
    queue.request();

    const uint32_t ybytes = a.ldy * 4u, out_dim = LAMBDA ? a.D : a.M;
    const RowWalker<NJ> rows(a.X, a.N, a.D, s_col, h);
    const uint32_t yoff = (uint32_t)s_col * ybytes;
    const bool vec_store = (a.ldy & 3u) == 0u && ((uintptr_t)a.Y & 15u) == 0u;
    f32x4 xa[NJ], xb[NJ];
    rows.fetch(xa, block0, 0);

    while (block0 < a.nbatches) {
        const uint64_t nb0 = queue.take();
        queue.request();
#pragma unroll
        for (uint32_t q = 0; q < BLOCK; ++q) {
            const uint64_t b = block0 + q;
            const uint64_t bnext = q + 1 < BLOCK ? b + 1 : nb0;
            f32x16 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
            f32x4 ring[DEPTH];
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) ring[i] = piece((uint32_t)i);
            uint32_t pnext = DEPTH;
            auto mma = [&](const f32x4 (&xs)[NJ]) {
                static_for<0, NT>([&](auto tt) {
                    static_for<0, NJ>([&](auto jj) {
                        constexpr int j = decltype(jj)::value;
                        const f32x4 av = ring[j % DEPTH];
                        ring[j % DEPTH] = piece(pnext++);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[decltype(tt)::value] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], xs[j][e], acc[decltype(tt)::value], 0, 0, 0);
                    });
                });
            };
            for (uint32_t c = 0; c < NC; c += 2) {
                if (c + 1 < NC) rows.fetch(xb, b, c + 1);
                rows.cut(xa, c);
                mma(xa);
                if (c + 1 < NC) {
                    if (c + 2 < NC) rows.fetch(xa, b, c + 2);
                    rows.cut(xb, c + 1);
                    mma(xb);
                }
            }
            asm volatile("" ::: "memory");
            rows.fetch(xa, bnext, 0);                     // x is dead: the next batch's first chunk
            const __amdgpu_buffer_rsrc_t ry =
                __builtin_amdgcn_make_buffer_rsrc((void *)(a.Y + b * 32 * a.ldy), 0, (int)rows.records(b, ybytes), 0x00020000);
            // accumulator group g of output tile T = columns 32 T + 8 g + 4 h .. + 3 of this lane's sample
            auto store_tile = [&](const f32x16 &v, uint32_t T) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t col = 32u * T + 8u * g + 4u * h;
                    if (vec_store && col + 4u <= out_dim) {
                        const f32x4 o = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, yoff + col * 4u, 0, 0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float val = v[4 * g + e];
                            if (col + e < out_dim)
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, val), ry, yoff + (col + e) * 4u, 0, 0);
                        }
                    }
                }
            };
            if constexpr (!LAMBDA) {
                static_for<0, NT>([&](auto tt) { store_tile(acc[decltype(tt)::value], (uint32_t)decltype(tt)::value); });
            } else {
                f32x4 yf[NJ2];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) yf[4 * t + g] = f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
                f32x4 ring2[DEPTH2];
#pragma unroll
                for (int i = 0; i < DEPTH2; ++i) ring2[i] = piece2((uint32_t)i);
                uint32_t p2 = DEPTH2;
                for (uint32_t T = 0; T < DT; ++T) {
                    f32x16 o;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] = 0.0f;
                    static_for<0, NJ2>([&](auto jj) {
                        constexpr int j = decltype(jj)::value;
                        const f32x4 av = ring2[j % DEPTH2];
                        ring2[j % DEPTH2] = piece2(p2++);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], yf[j][e], o, 0, 0, 0);
                    });
                    store_tile(o, T);
                }
            }
        }
        block0 = nb0;
    }
}

// Omega rows are worked in blocks of at most 128 (the accumulators of one pass); project (Lambda) needs all of
// Omega x in registers and is served for one block only.
struct LongProjectModel {
    uint32_t D = 0, M = 0, nblocks = 0;
    size_t block_floats = 0;                                      // pieces of one full block
    float *pieces1 = nullptr, *pieces2 = nullptr;
    uint32_t *queue = nullptr;
    mutable uint32_t next_slot = 0;
};

namespace {
template <int R2>
void launch_project_long(const LongProjectArgs &a, bool lambda, unsigned grid, hipStream_t st)
{
    if (lambda)
        hipLaunchKernelGGL((k_project_long<R2, true>), dim3(grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((k_project_long<R2, false>), dim3(grid), dim3(512), 0, st, a);
}
}  // namespace

namespace {
uint32_t block_width(uint32_t rows) { return rows <= 32 ? 32 : (rows <= 64 ? 64 : 128); }
}  // namespace

LongProjectModel *long_project_build(const std::vector<float> &Om, uint32_t M, uint32_t D, hipError_t &herr)
{
    herr = hipSuccess;
    if (D <= 256 || D > 65536 || M == 0 || M > 4096) return nullptr;
    LongProjectModel m;
    m.D = D;
    m.M = M;
    m.nblocks = (M + 127) / 128;
    const uint32_t steps = 8, chunk = 64, NC = (D + chunk - 1) / chunk, DT = (D + 31) / 32;
    const uint32_t R2 = block_width(M), NJ2 = R2 / 8;                     // (pieces2: one block only)
    m.block_floats = (size_t)NC * 4 * steps * 256;
    // Omega, per block of 128 rows: piece (c, t, j), lane (row, h) = Omega[128 b + 32 t + row][64 c + 4 (2 j + h) .. + 3];
    // the last block has NT = its own width / 32 tiles
    std::vector<float> p1((size_t)m.nblocks * m.block_floats, 0.0f), p2((size_t)DT * NJ2 * 256, 0.0f);
    for (uint32_t b = 0; b < m.nblocks; ++b) {
        const uint32_t rows = std::min(128u, M - 128 * b), NT = block_width(rows) / 32;
        for (uint32_t c = 0; c < NC; ++c)
            for (uint32_t t = 0; t < NT; ++t)
                for (uint32_t j = 0; j < steps; ++j)
                    for (uint32_t lane = 0; lane < 64; ++lane) {
                        const uint32_t r = 128 * b + 32 * t + (lane & 31), d0 = chunk * c + 4 * (2 * j + (lane >> 5));
                        float *dst = &p1[b * m.block_floats + ((((size_t)c * NT + t) * steps + j) * 64 + lane) * 4];
                        for (uint32_t e = 0; e < 4; ++e)
                            if (r < M && d0 + e < D) dst[e] = Om[(size_t)r * D + d0 + e];
                    }
    }
    // Omega^T: piece (T, j), lane (row, h) = Omega^T[32 T + row][4 (2 j + h) .. + 3] = Omega[4 (2 j + h) + e][32 T + row]
    for (uint32_t T = 0; T < DT; ++T)
        for (uint32_t j = 0; j < NJ2; ++j)
            for (uint32_t lane = 0; lane < 64; ++lane) {
                const uint32_t d = 32 * T + (lane & 31), r0 = 4 * (2 * j + (lane >> 5));
                float *dst = &p2[(((size_t)T * NJ2 + j) * 64 + lane) * 4];
                for (uint32_t e = 0; e < 4; ++e)
                    if (m.nblocks == 1 && d < D && r0 + e < M) dst[e] = Om[(size_t)(r0 + e) * D + d];
            }
    LongProjectModel *out = new LongProjectModel(m);
    auto up = [&](float **dst, const std::vector<float> &v) {
        hipError_t e = hipMalloc((void **)dst, v.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice);
        return e;
    };
    herr = up(&out->pieces1, p1);
    if (herr == hipSuccess) herr = up(&out->pieces2, p2);
    if (herr == hipSuccess) herr = hipMalloc((void **)&out->queue, 16 * QUEUE_BYTES);
    if (herr != hipSuccess) {
        long_project_free(out);
        return nullptr;
    }
    return out;
}

void long_project_free(LongProjectModel *m)
{
    if (!m) return;
    (void)hipFree(m->pieces1);
    (void)hipFree(m->pieces2);
    (void)hipFree(m->queue);
    delete m;
}

bool long_project_has_lambda(const LongProjectModel *m) { return m->nblocks == 1; }

// dY: N x ldy with ldy = M (project_omega) or D (project; one block of Omega rows only)
hipError_t long_project_run(const LongProjectModel *m, const float *dX, uint64_t N, bool lambda, float *dY, hipStream_t s)
{
    if (N == 0) return hipSuccess;
    if (lambda && m->nblocks != 1) return hipErrorInvalidValue;
    for (uint32_t b = 0; b < m->nblocks; ++b) {
        const uint32_t rows = std::min(128u, m->M - 128 * b), R2 = block_width(rows);
        LongProjectArgs a;
        a.X = dX; a.N = N; a.D = m->D; a.M = rows;
        a.pieces1 = m->pieces1 + b * m->block_floats; a.pieces2 = m->pieces2;
        a.Y = dY + 128 * b; a.ldy = lambda ? m->D : m->M;
        a.nbatches = (N + 31) / 32;
        a.queue = m->queue + (size_t)(m->next_slot++ % 16) * (QUEUE_BYTES / sizeof(uint32_t));
        const hipError_t e = hipMemsetAsync(a.queue, 0, QUEUE_BYTES, s);
        if (e != hipSuccess) return e;
        const uint64_t wgs = (a.nbatches + 7) / 8;
        const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);
        if (R2 == 32) launch_project_long<32>(a, lambda, grid, s);
        else if (R2 == 64) launch_project_long<64>(a, lambda, grid, s);
        else launch_project_long<128>(a, lambda, grid, s);
    }
    return hipGetLastError();
}

LaunchFn long_launcher(int tiles)
{
    switch (tiles) {
    case 1: return &launch_long<1>;
    case 2: return &launch_long<2>;
    case 4: return &launch_long<4>;
    case 8: return &launch_long<8>;
    default: return nullptr;
    }
}

}  // namespace vq
