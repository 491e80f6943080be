// LDS-resident Euclidean kernels for gfx950 (VQ / LVQ / GLVQ: src/helpers.rs:11-24, 55-77, 91-132): the whole
// prototype image is staged in LDS once per workgroup, then every wave runs on its own -- no per-tile DMA, no
// barriers.  k_resident_pipe (unpadded rows, the measured path of config 2) and k_resident (padded rows).
#include <cstdlib>

#include "mfma_device.h"

namespace vq {

// ------------------------------------------------------------------------------ resident-prototype kernel (padded rows)
template <int R, int S, bool MATCHED>
__global__ void __launch_bounds__(512, 2) k_resident(const EngineArgs a)
{
    using Gm = Geom<R>;
    __shared__ __attribute__((aligned(16))) char smem[RES_BYTES];
    __shared__ __attribute__((aligned(16))) float mu_lds[R];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    const uint32_t ntiles = a.res_tiles;
    const uint32_t row_bytes = a.D * 4u;
    const uint32_t voff = (uint32_t)s_col * row_bytes + (uint32_t)h * 16u;

    {   // whole image -> LDS, linear LDS-DMA, 1 KiB per wave-instruction
        stage_centre<R>(mu_lds, a.mu, threadIdx.x, 512);
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
            // rows through a buffer descriptor rebased per tile: rows past N read as zeros, rows of any length
            // (4-byte aligned) are fetched 16 bytes at a time and cut at D
            const uint64_t first = (b * S + s) * 32;
            sample[s] = first + s_col;
            uint32_t rec = 0u, rec_y = 0u;
            if (first < a.N) {
                const uint64_t left = a.N - first;
                rec = left * row_bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(left * row_bytes);
                rec_y = left * 4u > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(left * 4u);
            }
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)(a.X + first * a.D), 0, (int)rec, 0x00020000);
#pragma unroll
            for (int j = 0; j < R / 8; ++j)
                xf[s][j] = mask_chunk(__builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff + j * 32, 0, 0)),
                                      (uint32_t)(2 * j + h), a.D);
            if (a.mu != nullptr) centre_fragment<R / 8, 2>(xf[s], mu_lds, h);
            label[s] = 0u;
            if (MATCHED) {
                const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)(a.y + first), 0, (int)rec_y, 0x00020000);
                label[s] = __builtin_amdgcn_raw_buffer_load_b32(ry, (uint32_t)s_col * 4u, 0, 0);
            }
        }
        Top2 top[S], neg[S];                                          // labelled: top = matching group, neg = the other

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
                        if (!MATCHED || cg[e] == label[s]) top[s].offer(v, k);     // helpers.rs:113
                        else neg[s].offer(v, k);                                   // helpers.rs:122
                    }
                }
            }
        }

#pragma unroll
        for (int s = 0; s < S; ++s) {
            const bool live = sample[s] < a.N;
            Best1 wp, wn;
            float ss = 0.0f;
#pragma unroll
            for (int j = 0; j < R / 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) ss = fmaf(xf[s][j][e], xf[s][j][e], ss);
            ss += xor32(ss);
            bool rescan = !(ss < a.ss_limit);                          // NaN / INF / huge rows: their scores mean nothing
            const float margin = screen_margin(ss, a);
            // the group's best two scores re-measured exactly; a third score within the margin of the best: exact scan
            // (the guarantee: finish_keys, mfma_device.h)
            auto settle = [&](Top2 &t, Best1 &w) {
                t.merge_halves();
                if (t.b1 < KEY_VALID_F && !rescan) {
                    const float d1 = sqrtf(exact_sq_lds<R>(xf[s], img, t.i1, h));
                    w.d = d1; w.i = t.i1;
                    if (t.b2 < KEY_VALID_F) {
                        const float d2 = sqrtf(exact_sq_lds<R>(xf[s], img, t.i2, h));
                        if (lexless(d2, t.i2, d1, t.i1)) { w.d = d2; w.i = t.i2; }
                    }
                    return !(t.b3 - t.b1 > margin) || !(w.d < INFINITY);
                }
                return false;
            };
            const bool crowded_p = settle(top[s], wp);
            const bool crowded_n = MATCHED ? settle(neg[s], wn) : false;
            rescan |= crowded_p || crowded_n;
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
// One wave = one tile of 32 samples per batch.  D == R (host checks); a.res_tiles tiles of 32
// prototypes are searched (search_tiles).  a.ss_limit: ||x||^2 below which no score can overflow.
//
// STAGED (few prototypes: the image leaves room in LDS for 32 rows per wave).  With the search this short
// the kernel is bound by the memory pipe, and the fragment-shaped loads are what it chokes on: every
// buffer_load_dwordx4 takes 32 bytes out of each of 32 rows, 32 cache lines per instruction and every line
// four times (measured: ~230 cycles per instruction and CU, 2.7 TB/s).  Here the next batch's rows come in
// as whole 1 KiB pieces of the matrix (8 full lines per instruction) by LDS-DMA into the wave's staging
// buffer, a batch ahead, and are read out as fragments from there at the top of their batch.
template <int R>
struct Staging {
    static constexpr int ROWS_PER_PIECE = 256 / R;                    // rows in 1 KiB
    static constexpr int PIECES = 32 / ROWS_PER_PIECE;
    static constexpr int PIECE_STRIDE = 1024 + 32;                    // pad: consecutive pieces start 2 bank slots apart
    static constexpr int BYTES = PIECES * PIECE_STRIDE;               // per wave
    __host__ __device__ static constexpr int row_off(int r) { return (r / ROWS_PER_PIECE) * PIECE_STRIDE + (r % ROWS_PER_PIECE) * R * 4; }
};

//
// CUT: rows shorter than R (any length, 4-byte aligned; the image is zero padded): the 16-byte fetches are cut at
// D in place once a set has landed (mask_chunk).
//
// BF: the screening products run on the bf16 matrix cores (search_tiles_bf, mfma_device.h).  The sample fragment is
// split into bf16 pieces once per batch; the pieces die with the last tile, and without STAGED the next batch's rows
// are requested into their registers right there (the f32 rows are still needed by the exact re-measurement):
// two register sets that swap roles from batch to batch, the prefetch running over the epilogue and the partner wave.
//
// BFIMG (with BF): the image in LDS holds the prototypes' bf16 PIECES instead of their f32 values (same tiles, same
// tails; host: pieces_image) -- nothing is split in the tile loop (search_pieces), and the exact re-measurement reads
// its one or two rows per sample from the model's plain matrix in L2 (finish_keys_g).  For models whose f32 image
// leaves no room for the pieces beside it (config 2): the L2 reads cost the launch 6 %, the split they replace 20 %.
//
// FINISH (with BF): the finish pass over the samples the first pass marked (crowd_mark, mfma_device.h): batches of 32 ids
// from a.list instead of 32 consecutive rows, rows gathered by id, every row within the margin of the running minimum
// measured exactly (HitKeys), results scattered to the samples' own slots -- for the searches their flags name.
template <int R, bool MATCHED, bool DIAG = false, int NW = 8, bool STAGED = false, bool CUT = false, bool BF = false, bool BFIMG = false,
          bool FINISH = false>
__global__ void __launch_bounds__(NW * 64, NW / 4) k_resident_pipe(const EngineArgs a)
{
    static_assert(!BF || R <= 128, "the bf16 pieces of a 256-value fragment do not fit beside the fragment");
    constexpr bool PRIO = BF && !FINISH && !DIAG && !STAGED;   // (measured: config 2 +1.3 %; the staged variant: no change)
    static_assert(!FINISH || (BF && !STAGED && !DIAG), "the finish pass ranks on the bf16 cores and gathers its rows");
    uint32_t count = 0;                               // FINISH: entries of the list
    if constexpr (FINISH) {
        count = *a.list_count;
        if (count > a.N) count = (uint32_t)a.N;       // (never more entries than rows)
        if (count == 0) return;                        // nothing was marked: not even the image is loaded
    }
    // FINISH: a list of fewer than 32 samples per wave is spread THIN -- spb samples to a wave (lanes s_col < spb), one
    // batch per wave, no queue: a batch's time is its hits' round trips to the exact rows one after the other, whoever's
    // they are, and then the walk over the dry counters (191 marked samples of 10M as 6 full batches on 3 waves: 241 us).
    // Thin: wave w works entries [w spb, (w + 1) spb) as batch 2 w; the odd batches are empty.
    uint32_t spb = 32;
    bool thin = false;
    if constexpr (FINISH) {
        const uint32_t waves = gridDim.x * NW;
        if (count < 32u * waves) {
            thin = true;
            spb = (count + waves - 1) / waves;         // 1 .. 32
        }
    }
    const uint64_t nbatches = !FINISH ? a.nbatches : thin ? 2 * (((uint64_t)count + spb - 1) / spb) : ((uint64_t)count + 31) / 32;
    // FINISH: the batches are dealt across the workgroups first (wave w of workgroup g starts on block w * groups + g): a
    // short list keeps one wave per CU busy instead of eight on a few CUs, and a workgroup whose first wave finds
    // nothing returns before it loads the image
    if constexpr (FINISH) {
        if ((uint64_t)blockIdx.x * 2 >= nbatches) return;
    }
    using Gm = Geom<R>;
    using St = Staging<R>;
    unsigned long long tm[4] = {0, 0, 0, 0}, tprev_ = 0;
    auto lap = [&](int slot) {                        // diagnostic build only
        if (DIAG) {
            const unsigned long long t = stamp();
            tm[slot] += t - tprev_;
            tprev_ = t;
        }
    };
    __shared__ __attribute__((aligned(16))) char smem[RES_BYTES];
    __shared__ __attribute__((aligned(16))) float mu_lds[R];
    const unsigned long long r_entry = DIAG ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    const uint32_t ntiles = a.res_tiles;
    const bool centred = a.mu != nullptr;

    {   // whole image -> LDS, linear LDS-DMA, 1 KiB per wave-instruction
        stage_centre<R>(mu_lds, a.mu, threadIdx.x, NW * 64);
        const uint32_t bytes = ntiles * Gm::TILE_BYTES;
        const char *src = (const char *)a.img1;
        for (uint32_t p = wave; p * 1024u < bytes; p += NW) {
            const uint32_t off = p * 1024u + lane * 16u;
            if (off < bytes)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                                 (__attribute__((address_space(3))) void *)(smem + p * 1024u), 16, 0, 0);
        }
        if constexpr (BF && !BFIMG) {     // the pieces of the first a.nbf tiles behind the image
            const uint32_t bbytes = a.nbf * (uint32_t)BF_TILE_BYTES<R>;
            const char *bsrc = (const char *)a.imgbf;
            for (uint32_t p = wave; p * 1024u < bbytes; p += NW) {
                const uint32_t off = p * 1024u + lane * 16u;
                if (off < bbytes)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bsrc + off),
                                                     (__attribute__((address_space(3))) void *)(smem + bytes + p * 1024u), 16, 0, 0);
            }
        }
        __syncthreads();
    }
    const float *img = (const float *)smem;
    const uint32_t nbf = BF && !BFIMG ? a.nbf : 0u;
    const char *bfimg = smem + ntiles * Gm::TILE_BYTES;
    unsigned long long t_begin = 0, r_begin = 0;
    if (DIAG) {
        tprev_ = t_begin = stamp();
        r_begin = __builtin_amdgcn_s_memrealtime();
    }

    // blocks of BLOCK batches per wave: the first fixed, the others claimed a block ahead (BlockQueue)
    constexpr uint32_t BLOCK = 2;
    static_assert(BLOCK == 2, "the early return of the finish pass above counts two batches per block");
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane(wave);
    const uint32_t wave_id = FINISH ? wave_u * gridDim.x + blockIdx.x : blockIdx.x * NW + wave_u;
    BlockQueue<BLOCK> queue(a.queue, lane, wave_id, gridDim.x * NW, nbatches);
    uint64_t block0 = (uint64_t)wave_id * BLOCK;
    if (!thin) queue.request();                        // (thin: every batch has its wave, nothing is claimed)
    // Rows come in through a buffer descriptor rebased per batch (wave-uniform, so the 4 GiB offset
    // range is per batch, not per matrix): rows past N read as zeros by the hardware range check, one
    // per-lane offset serves every chunk (immediate offsets), and nothing needs a 64-bit address.
    // One register set per batch of the block: a set is refilled (for the same position of the next block)
    // right after its batch's epilogue and has the other batch's whole search to land -- no copies between sets.
    const uint32_t row_bytes = a.D * 4u;
    const uint32_t voff = (uint32_t)s_col * row_bytes + (uint32_t)h * 16u;
    // STAGED: the rows wait in LDS, one register set serves.  R = 256: one set is all the register file holds at two
    // waves per SIMD; it is refilled behind the epilogue for the very next batch and the partner wave covers the wait.
    constexpr bool SINGLE = !STAGED && R > 128;
    constexpr int SETS = STAGED || SINGLE ? 1 : (int)BLOCK;
    f32x4 x[SETS][R / 8];
    uint32_t ln[BLOCK], sid[BLOCK];                    // labels; FINISH: list entries (flags | sample id) of the batches in flight
#pragma unroll
    for (uint32_t q = 0; q < BLOCK; ++q) ln[q] = sid[q] = 0u;
    auto records = [&](uint64_t bb, uint32_t unit) -> uint32_t {      // bytes from row bb*32 to the end, clamped
        const uint64_t first = bb * 32;
        if (first >= a.N) return 0u;
        const uint64_t left = (a.N - first) * unit;
        return left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)left;
    };
    // STAGED: this wave's 32-row staging buffer behind the image
    char *stage = smem + ntiles * Gm::TILE_BYTES + nbf * (uint32_t)BF_TILE_BYTES<R> + (uint32_t)__builtin_amdgcn_readfirstlane(wave) * St::BYTES;
    // rows (and labels) of batch bb -> register set / label slot q, or -> the staging buffer
    auto fetch = [&](auto qq, uint64_t bb) {           // qq: label slot = position of bb in its block
        constexpr int q = decltype(qq)::value;
        constexpr int fset = SETS == 1 ? 0 : q;
        if constexpr (FINISH) {
            // entry of this lane pair's sample; slots past the list read entry 0 with no flags (searched, never stored)
            const uint64_t slot = (thin ? bb / 2 : bb) * spb + (uint64_t)s_col;
            uint32_t e = (uint32_t)s_col < spb && !(thin && (bb & 1)) && slot < count ? a.list[slot] : 0u;
            if ((e & CROWD_ID) >= a.N) e = 0u;         // (an id that is no row: searched as row 0, never stored)
            sid[q] = e;
            const float *row = a.X + (size_t)(e & CROWD_ID) * a.D;
#pragma unroll
            for (int j = 0; j < R / 8; ++j) x[fset][j] = gather_chunk(row, (uint32_t)(2 * j + h), a.D);
            if (MATCHED) ln[q] = a.y[e & CROWD_ID];
            return;
        }
        const __amdgpu_buffer_rsrc_t rx =
            __builtin_amdgcn_make_buffer_rsrc((void *)(a.X + bb * 32 * a.D), 0, (int)records(bb, row_bytes), 0x00020000);
        if constexpr (STAGED) {
            // lane l of piece p: bytes [1024 p + 16 l, + 16) of the batch; out-of-range lanes deliver zeros
#pragma unroll
            for (int p = 0; p < St::PIECES; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void *)(stage + p * St::PIECE_STRIDE), 16,
                                                         lane * 16, p * 1024, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < R / 8; ++j)
                x[fset][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff + j * 32, 0, 0));
        }
        if (MATCHED) {
            const __amdgpu_buffer_rsrc_t ry =
                __builtin_amdgcn_make_buffer_rsrc((void *)(a.y + bb * 32), 0, (int)records(bb, 4u), 0x00020000);
            ln[q] = __builtin_amdgcn_raw_buffer_load_b32(ry, (uint32_t)s_col * 4u, 0, 0);
        }
    };
    constexpr bool SWAP = BF && !STAGED;               // two sets, one batch in flight: see BF above
    if constexpr (SETS == 1 || SWAP) {
        fetch(std::integral_constant<int, 0>{}, block0);
    } else {
        static_for<0, (int)BLOCK>([&](auto qq) { fetch(qq, block0 + decltype(qq)::value); });
    }

    // store instructions a batch issues (wave-uniform): idx and dist by different lane halves, the labelled search two more
    const uint32_t stores_per_batch = (a.out.idx ? 1u : 0u) + (a.out.dist ? 1u : 0u) +
                                      (MATCHED ? (a.out.jneg ? 1u : 0u) + (a.out.dneg ? 1u : 0u) : 0u);
    uint32_t stores_behind = 0u;                       // STAGED: store instructions issued after the pieces now in flight
    while (block0 < nbatches) {                       // the last block may end past nbatches: such a batch reads zeros, stores nothing
        const uint64_t nb0 = thin ? nbatches : queue.take();   // first batch of the next block (the wait for its atomic is here)
        static_for<0, (int)BLOCK>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            if (FINISH && thin && q != 0) return;      // (the odd batches of a thin list are empty)
            constexpr int set = SETS == 1 ? 0 : q;
            const uint64_t b = block0 + q;
            f32x4(&xf)[R / 8] = x[set];
            using f32x2 = __attribute__((ext_vector_type(2))) float;
            f32x2 ss2 = {0.0f, 0.0f};
            if constexpr (CUT && !FINISH) {            // (gathered rows come in cut: gather_chunk)
#pragma unroll
                for (int j = 0; j < R / 8; ++j) xf[j] = mask_chunk(xf[j], (uint32_t)(2 * j + h), a.D);
            }
            if constexpr (STAGED) {
                // This batch's pieces have landed: vmcnt counts in issue order and the only requests younger than the
                // pieces are the result stores of the previous batch, which need not be waited for (their round trip to
                // L2, every batch, costs the streaming rate 13 %: tools/hbm_stream.hip).  The fragments are out of the
                // buffer (lgkmcnt) before the next batch's pieces are sent into it.
                switch (stores_behind) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                }
                const char *rowp = stage + St::row_off(s_col) + h * 16;
#pragma unroll
                for (int j = 0; j < R / 8; ++j) xf[j] = *(const f32x4 *)(rowp + 32 * j);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (centred) centre_fragment<R / 8, 2>(xf, mu_lds, h);
            const uint32_t label = ln[q];
#pragma unroll
            for (int j = 0; j < R / 8; ++j) {
                const f32x2 lo = {xf[j][0], xf[j][1]}, hi = {xf[j][2], xf[j][3]};
                ss2 = __builtin_elementwise_fma(lo, lo, ss2);
                ss2 = __builtin_elementwise_fma(hi, hi, ss2);
            }
            float ss = ss2[0] + ss2[1];
            ss += xor32(ss);                           // ||x||^2, the same value in both lanes of the pair
            const bool unsafe = !(ss < a.ss_limit);    // NaN / INF / huge rows: exact scan at the end
            const float margin = unsafe ? 0.0f : screen_margin(ss, a);
            lap(0);                                    // 0: delivery of this batch's rows
            // vmcnt counts in issue order: the atomic goes out only after this batch's rows have been waited
            // for, or that wait would include the atomic's own round trip
            asm volatile("" ::: "memory");
            if constexpr (q == 0) {
                if (!thin) queue.request();            // the block after the one just taken
            }
            // the staging buffer is free again; the label goes to the slot of the batch it belongs to
            if constexpr (STAGED) fetch(std::integral_constant<int, (q + 1) % (int)BLOCK>{}, q + 1 < (int)BLOCK ? b + 1 : nb0);
            lap(1);                                    // 1: issue of the claim / of the staged variant's next pieces

            if constexpr (FINISH) {
                // the searches this sample was marked for; margin +INF: every row is measured (the exact scan)
                const uint32_t flags = sid[q];
                const float mall = (flags & CROWD_ALL) || unsafe ? INFINITY : margin;
                HitKeys<R, MATCHED, true, BFIMG ? 0 : 1, 8> hits(xf, BFIMG ? a.plain : img, a.ldp, a.K, flags & (CROWD_P | CROWD_ALL) ? mall : -INFINITY,
                                                      flags & (CROWD_N | CROWD_ALL) ? mall : -INFINITY);
                {
                    u32x4 bh[R / 16], bl[R / 16];
                    split_fragment<R>(xf, bh, bl);
                    if constexpr (BFIMG)
                        search_pieces<R>((const char *)img, (uint32_t)Gm::TILE_BYTES, img + Gm::W_FLOATS, ntiles, label, bh, bl, s_col, h, hits);
                    else
                        search_tiles_bf<R>(img, ntiles, label, bh, bl, s_col, h, hits, bfimg, nbf);
                }
                asm volatile("" ::: "memory");
                fetch(std::integral_constant<int, (q + 1) % (int)BLOCK>{}, q + 1 < (int)BLOCK ? b + 1 : nb0);   // the pieces are dead
                hits.merge();
                const uint32_t id = flags & CROWD_ID;
                if (flags & (CROWD_P | CROWD_ALL)) {
                    if (h == 0) {
                        if (a.out.idx) a.out.idx[id] = hits.wp.i;
                    } else if (a.out.dist) a.out.dist[id] = hits.wp.d;
                }
                if (MATCHED && (flags & (CROWD_N | CROWD_ALL))) {
                    if (h == 0) {
                        if (a.out.jneg) a.out.jneg[id] = hits.wn.i;
                    } else if (a.out.dneg) a.out.dneg[id] = hits.wn.d;
                }
                return;
            }
            TopKeys<MATCHED> scr;
            Keys2 &ku = scr.ku;
            KeysM &km = scr.km;
            if constexpr (BF) {
                u32x4 bh[R / 16], bl[R / 16];
                split_fragment<R>(xf, bh, bl);
                // Outside the tile loop a wave is a chain of short vector sequences and waits; inside it the matrix pipe
                // paces it.  The wave in its epilogue / delivery goes first at the issue port (s_setprio), its partner's
                // screening fills what is left.
                if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
                if constexpr (BFIMG)
                    search_pieces<R>((const char *)img, (uint32_t)Gm::TILE_BYTES, img + Gm::W_FLOATS, ntiles, label, bh, bl, s_col, h, scr);
                else
                    search_tiles_bf<R>(img, ntiles, label, bh, bl, s_col, h, scr, bfimg, nbf);
            } else {
                search_tiles<R, MATCHED>(img, ntiles, label, xf, s_col, h, ku, km);
            }
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(3);
            auto fetch_next = [&]() { fetch(std::integral_constant<int, (q + 1) % (int)BLOCK>{}, q + 1 < (int)BLOCK ? b + 1 : nb0); };
            constexpr bool AHEAD = BFIMG && SWAP && !MATCHED;   // the fetch goes out inside finish_keys_ahead instead
            // The labelled search re-measures up to four rows from L2 (two class groups): vmcnt completes in order, so with the
            // samples requested first every one of those rows would queue behind an HBM round trip.  Its fetch goes out
            // BEHIND the re-measurement instead (the rows of two groups in flight beside the samples do not fit the registers).
            constexpr bool LATE = BFIMG && SWAP && MATCHED;
            if constexpr (SWAP && !AHEAD && !LATE) {   // the pieces are dead: the next batch's rows take their registers
                asm volatile("" ::: "memory");
                fetch_next();
            }
            lap(2);                                    // 2: tile loop

            Best1 wp, wn;
            if constexpr (AHEAD)       // the next batch's rows go out behind the request for the candidate's row
                finish_keys_ahead<R>(xf, a.plain, a.ldp, a.K, h, unsafe, margin, ku, wp, [&]() { fetch_next(); }, a.crowd_mask, b, a.N);
            else if constexpr (BFIMG) {
                finish_keys_g<R, MATCHED>(xf, a.plain, a.ldp, a.cls, a.K, label, h, unsafe, margin, ku, km, wp, wn, a.crowd_mask, b, a.N);
                if constexpr (LATE) {
                    asm volatile("" ::: "memory");
                    fetch_next();
                }
            } else finish_keys<R, MATCHED>(xf, img, a.K, label, h, unsafe, margin, ku, km, wp, wn, a.crowd_mask, b, a.N);
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
                stores_behind = stores_per_batch;
            }
            // this set's rows are dead: refill it for the same position of the next block
            if constexpr (SINGLE) {
                asm volatile("" ::: "memory");
                fetch(std::integral_constant<int, (q + 1) % (int)BLOCK>{}, q + 1 < (int)BLOCK ? b + 1 : nb0);
            } else if constexpr (!STAGED && !SWAP) {
                asm volatile("" ::: "memory");
                fetch(qq, nb0 + q);
            }
            lap(3);                                    // 3: batch epilogue (+ issue of the refill)
        });
        block0 = nb0;
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

namespace {
template <int R, int S>
hipError_t launch_resident(const EngineArgs &a, bool matched, hipStream_t st)
{
    const uint64_t wgs = (a.nbatches + 7) / 8;                        // one batch of S*32 samples per wave
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);          // one 512-thread workgroup per CU
    if (matched)
        hipLaunchKernelGGL((k_resident<R, S, true>), dim3(grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((k_resident<R, S, false>), dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}

template <int R>
hipError_t launch_resident_staged(const EngineArgs &a, bool matched, hipStream_t st)
{
    const uint64_t wgs = (a.nbatches + 7) / 8;
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);
    const hipError_t e = zero_queue(a, st);
    if (e != hipSuccess) return e;
    if (R == 128 && a.dbg && !matched)                               // diagnostic build (VQHIP_VARIANT=32): phase stamps
        hipLaunchKernelGGL((k_resident_pipe<128, false, true, 8, true>), dim3(grid), dim3(512), 0, st, a);
    else if (a.bf && matched)
        hipLaunchKernelGGL((k_resident_pipe<R, true, false, 8, true, false, true>), dim3(grid), dim3(512), 0, st, a);
    else if (a.bf)
        hipLaunchKernelGGL((k_resident_pipe<R, false, false, 8, true, false, true>), dim3(grid), dim3(512), 0, st, a);
    else if (matched)
        hipLaunchKernelGGL((k_resident_pipe<R, true, false, 8, true>), dim3(grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((k_resident_pipe<R, false, false, 8, true>), dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}

template <int R>
hipError_t launch_resident_cut(const EngineArgs &a, bool matched, hipStream_t st)
{
    const uint64_t wgs = (a.nbatches + 7) / 8;
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);
    const hipError_t e = zero_queue(a, st);
    if (e != hipSuccess) return e;
    if constexpr (R <= 128) {
        if (a.bf && a.bfimg) {
            if (matched)
                hipLaunchKernelGGL((k_resident_pipe<R, true, false, 8, false, true, true, true>), dim3(grid), dim3(512), 0, st, a);
            else
                hipLaunchKernelGGL((k_resident_pipe<R, false, false, 8, false, true, true, true>), dim3(grid), dim3(512), 0, st, a);
            return hipGetLastError();
        }
        if (a.bf) {
            if (matched)
                hipLaunchKernelGGL((k_resident_pipe<R, true, false, 8, false, true, true>), dim3(grid), dim3(512), 0, st, a);
            else
                hipLaunchKernelGGL((k_resident_pipe<R, false, false, 8, false, true, true>), dim3(grid), dim3(512), 0, st, a);
            return hipGetLastError();
        }
    }
    if (matched)
        hipLaunchKernelGGL((k_resident_pipe<R, true, false, 8, false, true>), dim3(grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((k_resident_pipe<R, false, false, 8, false, true>), dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}

template <int R, bool DIAG>
hipError_t launch_resident_pipe(const EngineArgs &a, bool matched, hipStream_t st)
{
    const uint64_t wgs = (a.nbatches + 7) / 8;
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);
    const hipError_t e = zero_queue(a, st);
    if (e != hipSuccess) return e;
    if constexpr (R <= 128 && !DIAG) {
        if (a.bf) {
            if (a.bfimg) {
                if (matched)
                    hipLaunchKernelGGL((k_resident_pipe<R, true, false, 8, false, false, true, true>), dim3(grid), dim3(512), 0, st, a);
                else
                    hipLaunchKernelGGL((k_resident_pipe<R, false, false, 8, false, false, true, true>), dim3(grid), dim3(512), 0, st, a);
                return hipGetLastError();
            }
            if (matched)
                hipLaunchKernelGGL((k_resident_pipe<R, true, false, 8, false, false, true>), dim3(grid), dim3(512), 0, st, a);
            else
                hipLaunchKernelGGL((k_resident_pipe<R, false, false, 8, false, false, true>), dim3(grid), dim3(512), 0, st, a);
            return hipGetLastError();
        }
    }
    if constexpr (DIAG && R == 128) {
        if (a.bf && a.bfimg) {                                        // VQHIP_VARIANT=35: the same with the pieces image
            hipLaunchKernelGGL((k_resident_pipe<R, false, true, 8, false, false, true, true>), dim3(grid), dim3(512), 0, st, a);
            return hipGetLastError();
        }
        if (a.bf) {                                                   // VQHIP_VARIANT=33: phase stamps of the bf16 search
            hipLaunchKernelGGL((k_resident_pipe<R, false, true, 8, false, false, true>), dim3(grid), dim3(512), 0, st, a);
            return hipGetLastError();
        }
    }
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
}  // namespace

namespace {
// the finish pass of every pipelined variant (pipe / staged / cut): rows gathered and cut at D, bf16 ranking, the
// image as the first pass left it (pieces only: a.bfimg, or f32 with a.nbf ready tiles behind)
template <int R>
hipError_t launch_resident_finish(const EngineArgs &a, bool matched, hipStream_t st)
{
    const hipError_t e = zero_queue(a, st);
    if (e != hipSuccess) return e;
    const dim3 grid(256), block(512);
    if (a.bfimg) {
        if (matched)
            hipLaunchKernelGGL((k_resident_pipe<R, true, false, 8, false, true, true, true, true>), grid, block, 0, st, a);
        else
            hipLaunchKernelGGL((k_resident_pipe<R, false, false, 8, false, true, true, true, true>), grid, block, 0, st, a);
    } else if (matched)
        hipLaunchKernelGGL((k_resident_pipe<R, true, false, 8, false, true, true, false, true>), grid, block, 0, st, a);
    else
        hipLaunchKernelGGL((k_resident_pipe<R, false, false, 8, false, true, true, false, true>), grid, block, 0, st, a);
    return hipGetLastError();
}
}  // namespace

LaunchFn resident_finish_launcher(int R)
{
    switch (R) {
    case 16: return &launch_resident_finish<16>;
    case 32: return &launch_resident_finish<32>;
    case 64: return &launch_resident_finish<64>;
    case 128: return &launch_resident_finish<128>;
    default: return nullptr;
    }
}

LaunchFn resident_launcher(int R, int *samples_per_batch)
{
    constexpr int S = 2;
    *samples_per_batch = S * 32;
    switch (R) {
    case 16: return &launch_resident<16, S>;
    case 32: return &launch_resident<32, S>;
    case 64: return &launch_resident<64, S>;
    case 128: return &launch_resident<128, S>;
    default: return nullptr;
    }
}

LaunchFn resident_staged_launcher(int R, size_t *staging_bytes)
{
    switch (R) {
    case 32: *staging_bytes = 8 * Staging<32>::BYTES; return &launch_resident_staged<32>;
    case 64: *staging_bytes = 8 * Staging<64>::BYTES; return &launch_resident_staged<64>;
    case 128: *staging_bytes = 8 * Staging<128>::BYTES; return &launch_resident_staged<128>;
    default: return nullptr;
    }
}

LaunchFn resident_cut_launcher(int R)
{
    switch (R) {
    case 16: return &launch_resident_cut<16>;
    case 32: return &launch_resident_cut<32>;
    case 64: return &launch_resident_cut<64>;
    case 128: return &launch_resident_cut<128>;
    case 256: return &launch_resident_cut<256>;
    default: return nullptr;
    }
}

LaunchFn resident_pipe_launcher(int R, bool diag)
{
    switch (R) {
    case 16: return diag ? &launch_resident_pipe<16, true> : &launch_resident_pipe<16, false>;
    case 32: return diag ? &launch_resident_pipe<32, true> : &launch_resident_pipe<32, false>;
    case 64: return diag ? &launch_resident_pipe<64, true> : &launch_resident_pipe<64, false>;
    case 128: return diag ? &launch_resident_pipe<128, true> : &launch_resident_pipe<128, false>;
    case 256: return diag ? nullptr : &launch_resident_pipe<256, false>;
    default: return nullptr;
    }
}

}  // namespace vq
