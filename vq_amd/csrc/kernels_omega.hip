// LDS-resident kernels for the Omega models on gfx950 (GMLVQ / LiRaMLVQ predict and matched search:
// src/helpers.rs:35-45 through 55-77, 91-132): k_lowrank_pipe (Omega with at most 16 rows, Omega and the projected
// prototypes resident) and k_omega_pipe (Omega about square, projected prototypes resident, Omega from L2).
#include <cstdlib>

#include "mfma_device.h"

namespace vq {

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
// CUT (both kernels of this file): rows shorter than the padded length (any length, 4-byte aligned): the fetches are
// cut at D in place before the projection reads them; the Omega images are zero there.
template <int R1, bool MATCHED, bool CUT = false, int NW = 8, bool BF = false>
__global__ void __launch_bounds__(NW * 64, NW / 4) k_lowrank_pipe(const EngineArgs a)
{
    using Gs = Geom<16>;
    constexpr int ROWF = R1 + 4;                       // Omega row stride in LDS: rows start 4 banks apart
    constexpr int NJ = R1 / 16;                        // projection steps: 4 MFMAs per step and sample group
    __shared__ __attribute__((aligned(16))) char smem[RES_BYTES];
    __shared__ __attribute__((aligned(16))) float mu_lds[R1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    const int n16 = lane & 15, kg = lane >> 4;
    const uint32_t ntiles = a.res_tiles;
    const uint32_t p_bytes = ntiles * Gs::TILE_BYTES;
    const bool centred = a.mu != nullptr;

    {   // both images -> LDS, linear LDS-DMA, 1 KiB per wave-instruction
        stage_centre<R1>(mu_lds, a.mu, threadIdx.x, NW * 64);
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
        if constexpr (BF) to_lds(p_bytes + 16u * ROWF * 4u, (const char *)a.imgbf, a.nbf * (uint32_t)BF_TILE_BYTES<16>);
        __syncthreads();
    }
    const float *img = (const float *)smem;
    const char *bfimg = smem + p_bytes + 16u * ROWF * 4u;             // bf16 pieces of the first a.nbf tiles
    const float *om = (const float *)(smem + p_bytes) + n16 * ROWF + kg * 4;   // this lane's Omega row and k group

    // blocks of BLOCK batches per wave: the first fixed, the others claimed a block ahead (BlockQueue)
    constexpr uint32_t BLOCK = 2;
    BlockQueue<BLOCK> queue(a.queue, lane, blockIdx.x * NW + (uint32_t)__builtin_amdgcn_readfirstlane(wave), gridDim.x * NW, a.nbatches);
    uint64_t block0 = (uint64_t)(blockIdx.x * NW + (uint32_t)__builtin_amdgcn_readfirstlane(wave)) * BLOCK;
    uint32_t in_block = 0;                             // position of b inside the current block
    queue.request();
    // the batch after (block0, in_block)
    auto successor = [&](uint64_t &nb0, uint32_t &nin) {
        if (in_block + 1 < BLOCK) {
            nb0 = block0;
            nin = in_block + 1;
        } else {
            nb0 = queue.take();
            nin = 0;
        }
    };
    // lane (n16, kg) loads 16 B of row n16 (group 0) and row 16 + n16 (group 1) per step: chunk 4j + kg
    const uint32_t row_bytes = a.D * 4u;
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
            __builtin_amdgcn_make_buffer_rsrc((void *)(a.X + bb * 32 * a.D), 0, (int)records(bb, row_bytes), 0x00020000);
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

    while (block0 < a.nbatches) {                     // the last block may end past nbatches: such a batch reads zeros, stores nothing
        uint64_t nb0;
        uint32_t nin;
        successor(nb0, nin);
        const uint64_t bnext = nb0 + nin;

        // ---- projection: accA / accB = y of samples n16 / 16 + n16, rows 4 kg .. 4 kg + 3
        f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
        if constexpr (CUT) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                xn[0][j] = mask_chunk(xn[0][j], (uint32_t)(4 * j + kg), a.D);
                xn[1][j] = mask_chunk(xn[1][j], (uint32_t)(4 * j + kg), a.D);
            }
        }
        if (centred) {                                 // entry j of both groups holds chunk 4 j + kg of its row
            centre_fragment<NJ, 4>(xn[0], mu_lds, kg);
            centre_fragment<NJ, 4>(xn[1], mu_lds, kg);
        }
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
        if (nin == 0) queue.request();
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
        const float margin = unsafe ? 0.0f : screen_margin(ss, a);

        // ---- search in the projected space (k_resident_pipe's tile loop at R = 16)
        Best1 wp, wn;
        if constexpr (BF) {
            // The ranking on the bf16 matrix cores (search_tiles_bf; the projection above stays f32), the keys over groups
            // of four prototypes: with 16 reduction values a tile is three MFMAs, and 4 VALU per score were three quarters
            // of its instructions; re-measuring a group's four rows of 16 values costs the epilogue next to nothing.
            GroupKeys<MATCHED> scr;
            u32x4 bh[1], bl[1];
            split_fragment<16>(yf, bh, bl);
            search_tiles_bf<16>(img, ntiles, label, bh, bl, s_col, h, scr, bfimg, a.nbf);
            auto cls_of = [&](uint32_t k) { return ((const uint32_t *)(img + (size_t)(k >> 5) * Gs::TILE_FLOATS + Gs::W_FLOATS))[32 + (k & 31)]; };
            finish_groups<MATCHED>(
                h, unsafe, margin, scr.ku, scr.km, wp, wn, [&](uint32_t k) { return exact_sq_lds<16>(yf, img, k, h); },      // squared: helpers.rs:44
                [&](uint32_t k, bool positive) { return k < a.K && (!MATCHED || (cls_of(k) == label) == positive); },
                [&]() { scan_exact_lds<16, MATCHED, false>(yf, img, a.K, label, h, wp, wn); }, a.dbg);
        } else {
            Keys2 ku;
            KeysM km;
            search_tiles<16, MATCHED>(img, ntiles, label, yf, s_col, h, ku, km);
            finish_keys<16, MATCHED, false>(yf, img, a.K, label, h, unsafe, margin, ku, km, wp, wn);
        }
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
// BFIMG: the image of the projected prototypes holds their bf16 pieces (k_resident_pipe's BFIMG): search_pieces, exact
// rows p_k from the plain matrix in L2.
// PJ: the projection on the bf16 matrix cores as well (mfma_device.h split3 / mma_bf6): Omega comes as three bf16 pieces
// per value, one 3 KiB piece per (step of 16 reduction values, tile) -- step-major, so a step's sample pieces are split
// once, serve every tile and are never stored; two tiles' chains run side by side.  y is at least as exact as from the
// f32 chain (tools/bf16x6_probe.hip) and everything behind it (search, exact re-measurement) works on that y as before.
// FINISH (with BF and CUT, not BFIMG): the finish pass over the samples the first pass marked (crowd_mark, mfma_device.h;
// k_resident_pipe's FINISH): ids from a.list, rows gathered by id and cut at D, the same projection, then every row
// within the margin of the running minimum measured exactly from the f32 image in LDS (HitKeys), results scattered.
template <int R, bool MATCHED, bool CUT = false, int NW = 8, bool BF = false, bool BFIMG = false, bool PJ = false, bool FINISH = false>
__global__ void __launch_bounds__(NW * 64, NW / 4) k_omega_pipe(const EngineArgs a)
{
    static_assert(!FINISH || (BF && CUT && !BFIMG), "the finish pass ranks on the bf16 cores, gathers cut rows and keeps the f32 image");
    // FINISH: entries of the list; a list of fewer than 32 samples per wave is dealt thin, spb to a wave, one batch each
    uint32_t count = 0, spb = 32;
    bool thin = false;
    if constexpr (FINISH) {
        count = *a.list_count;
        if (count > a.N) count = (uint32_t)a.N;       // (never more entries than rows)
        if (count == 0) return;                        // nothing was marked: not even the image is loaded
        const uint32_t waves = gridDim.x * NW;
        if (count < 32u * waves) {
            thin = true;
            spb = (count + waves - 1) / waves;         // 1 .. 32
        }
    }
    const uint64_t nbatches = FINISH ? ((uint64_t)count + spb - 1) / spb : a.nbatches;
    if constexpr (FINISH) {                            // wave w of workgroup g is wave number w * groups + g: see wave_id
        if ((uint64_t)blockIdx.x * (thin ? 1u : 2u) >= nbatches) return;
    }
    using Gs = Geom<R>;
    constexpr int NT = R / 32, NJ = R / 8, NSTEP = NT * NJ;   // Omega tiles, reduction steps per tile, pieces
    constexpr int DEPTH = 6;                                  // Omega pieces in flight per wave
    constexpr int NQ = R / 16, TG = NT >= 2 ? 2 : 1, NP = NQ * NT, PD = 2 * TG;   // PJ: steps, tiles side by side, pieces, in flight
    __shared__ __attribute__((aligned(16))) char smem[RES_BYTES];
    __shared__ __attribute__((aligned(16))) float mu_lds[R];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    const uint32_t ntiles = a.res_tiles;
    const bool centred = a.mu != nullptr;

    {   // prototype image -> LDS, linear LDS-DMA, 1 KiB per wave-instruction
        stage_centre<R>(mu_lds, a.mu, threadIdx.x, NW * 64);
        const uint32_t bytes = ntiles * Gs::TILE_BYTES;
        const char *src = (const char *)a.img2;
        for (uint32_t p = wave; p * 1024u < bytes; p += NW) {
            const uint32_t off = p * 1024u + lane * 16u;
            if (off < bytes)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                                 (__attribute__((address_space(3))) void *)(smem + p * 1024u), 16, 0, 0);
        }
        if constexpr (BF && !BFIMG) {     // the bf16 pieces of the first a.nbf tiles behind the image
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
    const char *bfimg = smem + ntiles * Gs::TILE_BYTES;
    // Omega pieces: piece i = (tile i / NJ, step i % NJ), this lane's 16 bytes at i * 1024 + lane * 16
    // PJ: piece i = (step i / NT, tile i % NT), 3 KiB: the lane's 16 bytes of p1 at i * 3072 + lane * 16, p2 and p3 1 KiB further each
    const __amdgpu_buffer_rsrc_t rom = __builtin_amdgcn_make_buffer_rsrc((void *)a.img1, 0, PJ ? NP * 3072 : NSTEP * 1024, 0x00020000);
    const uint32_t om_off = (uint32_t)lane * 16u;
    auto piece = [&](int i) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rom, om_off, i * 1024, 0));
    };
    auto piece3 = [&](int i, u32x4 (&p)[3]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) p[c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rom, om_off, i * 3072 + c * 1024, 0));
    };

    // blocks of BLOCK batches per wave: the first fixed, the others claimed a block ahead (BlockQueue)
    constexpr uint32_t BLOCK = 2;
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane(wave);
    const uint32_t wave_id = FINISH ? wave_u * gridDim.x + blockIdx.x : blockIdx.x * NW + wave_u;
    BlockQueue<BLOCK> queue(a.queue, lane, wave_id, gridDim.x * NW, nbatches);
    uint64_t block0 = thin ? (uint64_t)wave_id : (uint64_t)wave_id * BLOCK;   // (thin: batch w is wave w's, the only one)
    uint32_t in_block = 0;                             // position of b inside the current block
    if (!thin) queue.request();
    // the batch after (block0, in_block)
    auto successor = [&](uint64_t &nb0, uint32_t &nin) {
        if (thin) {
            nb0 = nbatches;
            nin = 0;
        } else if (in_block + 1 < BLOCK) {
            nb0 = block0;
            nin = in_block + 1;
        } else {
            nb0 = queue.take();
            nin = 0;
        }
    };
    const uint32_t row_bytes = a.D * 4u;
    const uint32_t voff = (uint32_t)s_col * row_bytes + (uint32_t)h * 16u;
    f32x4 xn[R / 8];                                   // rows of the next batch, in flight
    uint32_t ln = 0u, sidn = 0u;                       // its labels; FINISH: its list entries (flags | sample id)
    auto records = [&](uint64_t bb, uint32_t unit) -> uint32_t {
        const uint64_t first = bb * 32;
        if (first >= a.N) return 0u;
        const uint64_t left = (a.N - first) * unit;
        return left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)left;
    };
    auto fetch = [&](uint64_t bb) {
        if constexpr (FINISH) {
            // entry of this lane pair's sample; slots past the list read entry 0 with no flags (searched, never stored)
            const uint64_t slot = bb * spb + (uint64_t)s_col;
            uint32_t e = (uint32_t)s_col < spb && slot < count ? a.list[slot] : 0u;
            if ((e & CROWD_ID) >= a.N) e = 0u;         // (an id that is no row: searched as row 0, never stored)
            sidn = e;
            const float *row = a.X + (size_t)(e & CROWD_ID) * a.D;
#pragma unroll
            for (int j = 0; j < R / 8; ++j) xn[j] = gather_chunk(row, (uint32_t)(2 * j + h), a.D);
            if (MATCHED) ln = a.y[e & CROWD_ID];
            return;
        }
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
#ifdef VQ_OMEGA_DIAG
    unsigned long long tm[4] = {0, 0, 0, 0}, tprev_ = stamp();
    const unsigned long long t_begin = tprev_, r_begin = __builtin_amdgcn_s_memrealtime();
    auto lap = [&](int slot) { const unsigned long long t = stamp(); tm[slot] += t - tprev_; tprev_ = t; };
#endif

    while (block0 < nbatches) {                       // the last block may end past nbatches: such a batch reads zeros, stores nothing
        uint64_t nb0;
        uint32_t nin;
        successor(nb0, nin);
        const uint64_t bnext = nb0 + nin;
#ifdef VQ_OMEGA_DIAG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lap(0);
#endif

        // ---- projection y = Omega x: 32 Omega rows per tile, A fragments from the ring; the accumulators of
        //      tile t are chunks 2(4t + g) + h of y, i.e. fragment entries 4t + g of the search stage
        f32x4 yf[R / 8];
        if constexpr (CUT && !FINISH) {                // (gathered rows come in cut: gather_chunk)
#pragma unroll
            for (int j = 0; j < R / 8; ++j) xn[j] = mask_chunk(xn[j], (uint32_t)(2 * j + h), a.D);
        }
        if (centred) centre_fragment<R / 8, 2>(xn, mu_lds, h);
        if constexpr (PJ) {
            f32x16 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
            // The ring: PD pieces requested ahead.  (Deeper rings with the loads pinned behind their group -- 2 and 3 groups of
            // lookahead -- ran at the same 3.95-3.98 ms per 10M rows on one box: the stage is not waiting for L2.)
            u32x4 ring[PD][3];
            static_for<0, PD>([&](auto ii) { piece3(decltype(ii)::value, ring[decltype(ii)::value]); });
            static_for<0, NQ>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                u32x4 xp[3];
                split24(xn[2 * q], xn[2 * q + 1], xp);
                static_for<0, NT / TG>([&](auto gg) {
                    constexpr int i0 = q * NT + decltype(gg)::value * TG;
                    u32x4 av[TG][3];
#pragma unroll
                    for (int u = 0; u < TG; ++u)
#pragma unroll
                        for (int c = 0; c < 3; ++c) av[u][c] = ring[(i0 + u) % PD][c];
                    mma_bf6<TG>(av, xp, &acc[decltype(gg)::value * TG]);
                    static_for<0, TG>([&](auto uu) {
                        constexpr int i = i0 + decltype(uu)::value;
                        if constexpr (i + PD < NP) piece3(i + PD, ring[i % PD]);
                    });
                });
            });
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) yf[4 * t + g] = f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
        } else {
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
        const uint32_t label = ln, flags = sidn;
#ifdef VQ_OMEGA_DIAG
        asm volatile("" : "+v"(yf[0][0]), "+v"(yf[R / 8 - 1][3]));
        lap(1);
#endif
        // x is dead: the next batch's rows go out now and arrive while this batch is searched.  The atomic
        // goes out after this batch's rows have been waited for (vmcnt counts in issue order).
        asm volatile("" ::: "memory");
        if (nin == 0 && !thin) queue.request();
        fetch(bnext);

        float ss = 0.0f;
#pragma unroll
        for (int j = 0; j < R / 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) ss = fmaf(yf[j][e], yf[j][e], ss);
        ss += xor32(ss);                               // ||y||^2, the same value in both lanes of the pair
        const bool unsafe = !(ss < a.ss_limit);        // NaN / INF / huge: exact scan at the end
        const float margin = unsafe ? 0.0f : screen_margin(ss, a);

        if constexpr (FINISH) {
            // the searches this sample was marked for; margin +INF: every row is measured (the exact scan)
            const float mall = (flags & CROWD_ALL) || unsafe ? INFINITY : margin;
            HitKeys<R, MATCHED, false, 1, 8> hits(yf, img, a.ldp, a.K, flags & (CROWD_P | CROWD_ALL) ? mall : -INFINITY,
                                                  flags & (CROWD_N | CROWD_ALL) ? mall : -INFINITY);
            {
                u32x4 bh[R / 16], bl[R / 16];
                split_fragment<R>(yf, bh, bl);
                search_tiles_bf<R>(img, ntiles, label, bh, bl, s_col, h, hits, bfimg, a.nbf);
            }
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
            b = bnext;
            block0 = nb0;
            in_block = nin;
            continue;
        }
        // ---- search in the projected space (k_resident_pipe's tile loop)
        TopKeys<MATCHED> scr;
        Keys2 &ku = scr.ku;
        KeysM &km = scr.km;
        if constexpr (BF) {        // the ranking on the bf16 matrix cores (search_tiles_bf); the projection above: f32, or bf16x6 with PJ
            u32x4 bh[R / 16], bl[R / 16];
            split_fragment<R>(yf, bh, bl);
            __builtin_amdgcn_s_setprio(0);             // (k_resident_pipe: the wave outside its tile loop goes first)
            if constexpr (BFIMG)
                search_pieces<R>((const char *)img, (uint32_t)Gs::TILE_BYTES, img + Gs::W_FLOATS, ntiles, label, bh, bl, s_col, h, scr);
            else
                search_tiles_bf<R>(img, ntiles, label, bh, bl, s_col, h, scr, bfimg, a.nbf);
        } else {
            search_tiles<R, MATCHED>(img, ntiles, label, yf, s_col, h, ku, km);
        }

#ifdef VQ_OMEGA_DIAG
        lap(2);
#endif
        Best1 wp, wn;
        if constexpr (BF) __builtin_amdgcn_s_setprio(3);
        if constexpr (BFIMG) finish_keys_g<R, MATCHED, false>(yf, a.plain, a.ldp, a.cls, a.K, label, h, unsafe, margin, ku, km, wp, wn, a.crowd_mask, b, a.N);
        else finish_keys<R, MATCHED, false>(yf, img, a.K, label, h, unsafe, margin, ku, km, wp, wn, a.crowd_mask, b, a.N);
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
#ifdef VQ_OMEGA_DIAG
        lap(3);
#endif
        b = bnext;
        block0 = nb0;
        in_block = nin;
    }
#ifdef VQ_OMEGA_DIAG
    if (a.dbg && lane == 0) {
        unsigned long long *d = a.dbg + ((size_t)blockIdx.x * NW + wave) * 8;
        for (int i = 0; i < 4; ++i) d[i] = tm[i];
        d[4] = stamp() - t_begin;
        d[5] = __builtin_amdgcn_s_memrealtime() - r_begin;
        d[6] = r_begin;
        d[7] = r_begin + d[5];
    }
#endif
}

namespace {
template <int R1>
hipError_t launch_lowrank(const EngineArgs &a, bool matched, hipStream_t st)
{
    const uint64_t wgs = (a.nbatches + 7) / 8;
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);
    const hipError_t e = zero_queue(a, st);
    if (e != hipSuccess) return e;
    if (a.bf) {
        if (a.D < (uint32_t)R1) {
            if (matched)
                hipLaunchKernelGGL((k_lowrank_pipe<R1, true, true, 8, true>), dim3(grid), dim3(512), 0, st, a);
            else
                hipLaunchKernelGGL((k_lowrank_pipe<R1, false, true, 8, true>), dim3(grid), dim3(512), 0, st, a);
        } else if (matched)
            hipLaunchKernelGGL((k_lowrank_pipe<R1, true, false, 8, true>), dim3(grid), dim3(512), 0, st, a);
        else
            hipLaunchKernelGGL((k_lowrank_pipe<R1, false, false, 8, true>), dim3(grid), dim3(512), 0, st, a);
    } else if (a.D < (uint32_t)R1) {
        if (matched)
            hipLaunchKernelGGL((k_lowrank_pipe<R1, true, true>), dim3(grid), dim3(512), 0, st, a);
        else
            hipLaunchKernelGGL((k_lowrank_pipe<R1, false, true>), dim3(grid), dim3(512), 0, st, a);
    } else if (matched)
        hipLaunchKernelGGL((k_lowrank_pipe<R1, true>), dim3(grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((k_lowrank_pipe<R1, false>), dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}

template <int R>
hipError_t launch_omega_pipe(const EngineArgs &a, bool matched, hipStream_t st)
{
    const uint64_t wgs = (a.nbatches + 7) / 8;
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);
    const hipError_t e = zero_queue(a, st);
    if (e != hipSuccess) return e;
    if (a.pj && !a.bf) return hipErrorInvalidValue;                     // built together (bind_omega_pipe)
    if (a.pj && a.bfimg) {
        if (a.D < (uint32_t)R) {
            if (matched)
                hipLaunchKernelGGL((k_omega_pipe<R, true, true, 8, true, true, true>), dim3(grid), dim3(512), 0, st, a);
            else
                hipLaunchKernelGGL((k_omega_pipe<R, false, true, 8, true, true, true>), dim3(grid), dim3(512), 0, st, a);
        } else if (matched)
            hipLaunchKernelGGL((k_omega_pipe<R, true, false, 8, true, true, true>), dim3(grid), dim3(512), 0, st, a);
        else
            hipLaunchKernelGGL((k_omega_pipe<R, false, false, 8, true, true, true>), dim3(grid), dim3(512), 0, st, a);
    } else if (a.pj) {
        if (a.D < (uint32_t)R) {
            if (matched)
                hipLaunchKernelGGL((k_omega_pipe<R, true, true, 8, true, false, true>), dim3(grid), dim3(512), 0, st, a);
            else
                hipLaunchKernelGGL((k_omega_pipe<R, false, true, 8, true, false, true>), dim3(grid), dim3(512), 0, st, a);
        } else if (matched)
            hipLaunchKernelGGL((k_omega_pipe<R, true, false, 8, true, false, true>), dim3(grid), dim3(512), 0, st, a);
        else
            hipLaunchKernelGGL((k_omega_pipe<R, false, false, 8, true, false, true>), dim3(grid), dim3(512), 0, st, a);
    } else if (a.bf && a.bfimg) {
        if (a.D < (uint32_t)R) {
            if (matched)
                hipLaunchKernelGGL((k_omega_pipe<R, true, true, 8, true, true>), dim3(grid), dim3(512), 0, st, a);
            else
                hipLaunchKernelGGL((k_omega_pipe<R, false, true, 8, true, true>), dim3(grid), dim3(512), 0, st, a);
        } else if (matched)
            hipLaunchKernelGGL((k_omega_pipe<R, true, false, 8, true, true>), dim3(grid), dim3(512), 0, st, a);
        else
            hipLaunchKernelGGL((k_omega_pipe<R, false, false, 8, true, true>), dim3(grid), dim3(512), 0, st, a);
    } else if (a.bf) {
        if (a.D < (uint32_t)R) {
            if (matched)
                hipLaunchKernelGGL((k_omega_pipe<R, true, true, 8, true>), dim3(grid), dim3(512), 0, st, a);
            else
                hipLaunchKernelGGL((k_omega_pipe<R, false, true, 8, true>), dim3(grid), dim3(512), 0, st, a);
        } else if (matched)
            hipLaunchKernelGGL((k_omega_pipe<R, true, false, 8, true>), dim3(grid), dim3(512), 0, st, a);
        else
            hipLaunchKernelGGL((k_omega_pipe<R, false, false, 8, true>), dim3(grid), dim3(512), 0, st, a);
    } else if (a.D < (uint32_t)R) {
        if (matched)
            hipLaunchKernelGGL((k_omega_pipe<R, true, true>), dim3(grid), dim3(512), 0, st, a);
        else
            hipLaunchKernelGGL((k_omega_pipe<R, false, true>), dim3(grid), dim3(512), 0, st, a);
    } else if (matched)
        hipLaunchKernelGGL((k_omega_pipe<R, true>), dim3(grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((k_omega_pipe<R, false>), dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}
}  // namespace

namespace {
// the finish pass of k_omega_pipe: the projection as the first pass ran it (a.pj), the f32 image (a.img2; a.nbf ready tiles)
template <int R>
hipError_t launch_omega_finish(const EngineArgs &a, bool matched, hipStream_t st)
{
    if (!a.bf || a.bfimg) return hipErrorInvalidValue;
    const hipError_t e = zero_queue(a, st);
    if (e != hipSuccess) return e;
    const dim3 grid(256), block(512);
    if (a.pj) {
        if (matched)
            hipLaunchKernelGGL((k_omega_pipe<R, true, true, 8, true, false, true, true>), grid, block, 0, st, a);
        else
            hipLaunchKernelGGL((k_omega_pipe<R, false, true, 8, true, false, true, true>), grid, block, 0, st, a);
    } else if (matched)
        hipLaunchKernelGGL((k_omega_pipe<R, true, true, 8, true, false, false, true>), grid, block, 0, st, a);
    else
        hipLaunchKernelGGL((k_omega_pipe<R, false, true, 8, true, false, false, true>), grid, block, 0, st, a);
    return hipGetLastError();
}
}  // namespace

LaunchFn omega_finish_launcher(int R)
{
    switch (R) {
    case 32: return &launch_omega_finish<32>;
    case 64: return &launch_omega_finish<64>;
    case 128: return &launch_omega_finish<128>;
    default: return nullptr;
    }
}

LaunchFn lowrank_launcher(int R1)
{
    switch (R1) {
    case 32: return &launch_lowrank<32>;
    case 64: return &launch_lowrank<64>;
    case 128: return &launch_lowrank<128>;
    case 256: return &launch_lowrank<256>;
    default: return nullptr;
    }
}

LaunchFn omega_pipe_launcher(int R)
{
    switch (R) {
    case 32: return &launch_omega_pipe<32>;
    case 64: return &launch_omega_pipe<64>;
    case 128: return &launch_omega_pipe<128>;
    default: return nullptr;
    }
}

}  // namespace vq
