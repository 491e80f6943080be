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
#include <cstdlib>

#include "mfma_device.h"

namespace vq {

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
// BF: the search stage's products on the bf16 matrix cores (tile_mma_ready, mfma_device.h): the search image then holds
// the prototypes' bf16 pieces (nothing else reads its values: the exact re-measurement and scan go to a.plain); projections stay f32.
// GRP (with BF): the keys stand for groups of four prototypes (GroupKeys / finish_groups, mfma_device.h): 1.5 VALU per score
// instead of 4, four exact rows per candidate in the epilogue -- for models of many tiles, where the screening is most of
// the instruction stream and the epilogue a small part of a batch.
// FINISH (with BF, not GRP): the finish pass over the samples the first pass marked (crowd_mark / HitKeys, mfma_device.h):
// batches of ids from a.list, rows gathered by id, results scattered -- k_resident_pipe's FINISH for the streamed engine.
template <int MODE, int R1, int R2, int S, bool MATCHED, int WPS, bool BF = false, bool GRP = false, bool FINISH = false>
__global__ void __launch_bounds__(256, WPS) k_engine(const EngineArgs a)
{
    static_assert(!BF || MODE != MODE_LOCAL, "LOCAL has no ranking stage");
    static_assert(!GRP || BF, "group keys come with the bf16 ranking");
    static_assert(!FINISH || (BF && !GRP), "the finish pass ranks on the bf16 cores");
    uint32_t count = 0;                               // FINISH: entries of the list
    if constexpr (FINISH) {
        count = *a.list_count;
        if (count > a.N) count = (uint32_t)a.N;
        if (count == 0) return;
    }
    const uint64_t nbatches = FINISH ? ((uint64_t)count + 4 * S * 32 - 1) / (4 * S * 32) : a.nbatches;
    if (FINISH && blockIdx.x >= nbatches) return;      // (before the first barrier: the whole workgroup leaves)
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
    // FINISH: the lanes' deferred hit lists (HitKeys CAP = 16: 32 words per lane and sample tile, lane-interleaved)
    constexpr int HIT_CAP = 16;
    __shared__ uint32_t hit_lds[FINISH ? S * 2 * HIT_CAP * 256 : 1];
    __shared__ __attribute__((aligned(16))) float mu_lds[R1];
    stage_centre<R1>(mu_lds, a.mu, threadIdx.x, 256);                // published by the first barrier below

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    const uint32_t ngroups = MODE == MODE_OMEGA ? NG1 + a.ngroups2 : a.ngroups1;

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
    uint32_t label[S], sid[S];                                        // sid: FINISH: list entries (flags | sample id)
    const uint32_t row_bytes = a.D * 4u;
    const uint32_t voff = (uint32_t)s_col * row_bytes + (uint32_t)h * 16u;
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane(wave);
    auto load_samples = [&](uint64_t bb) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint64_t first = ((bb * 4 + wave_u) * S + s) * 32;
            if constexpr (FINISH) {
                // entry of this lane pair's sample; slots past the list: row 0 with no flags (searched, never stored)
                const uint64_t slot = first + (uint64_t)s_col;
                uint32_t e = slot < count ? a.list[slot] : 0u;
                if ((e & CROWD_ID) >= a.N) e = 0u;
                sid[s] = e;
                const float *row = a.X + (size_t)(e & CROWD_ID) * a.D;
#pragma unroll
                for (int j = 0; j < R1 / 8; ++j) xf[s][j] = gather_chunk(row, (uint32_t)(2 * j + h), a.D);
                label[s] = MATCHED ? a.y[e & CROWD_ID] : 0u;
                continue;
            }
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
        if (!FINISH && a.D < (uint32_t)R1) {           // (gathered rows come in cut: gather_chunk)
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int j = 0; j < R1 / 8; ++j) xf[s][j] = mask_chunk(xf[s][j], (uint32_t)(2 * j + h), a.D);
        }
    };

#ifdef VQ_ENGINE_DIAG      // phase stamps and the in-kernel clock (engine_assign prints them: VQHIP_VARIANT=35 VQHIP_DUMP=1 [VQHIP_NW4=1 for WPS = 1])
    const unsigned long long t_begin = stamp(), r_begin = __builtin_amdgcn_s_memrealtime();
    unsigned long long tm[4] = {0, 0, 0, 0}, tprev_ = t_begin;
    auto lap = [&](int slot) { const unsigned long long t = stamp(); tm[slot] += t - tprev_; tprev_ = t; };
#define VQ_LAP(i) lap(i)
#else
#define VQ_LAP(i)
#endif
    int cur = 0;
    uint64_t b = blockIdx.x;
    if (b < nbatches) {
        issue(0, 0);
        load_samples(b);
        __syncthreads();                                             // waits vmcnt(0) for the DMA, then barrier
    }

    // thread 0 keeps one claim in flight: the atomic issued at the top of batch k returns batch k + 2
    uint32_t pending = 0u;
    if (tid == 0) pending = gridDim.x + atomicAdd(a.queue, 1u);

    while (b < nbatches) {
        if (tid == 0) {
            claimed = pending;                                       // the wait for the atomic of a batch ago is here
            pending = gridDim.x + atomicAdd(a.queue, 1u);
        }
        __syncthreads();
        const uint64_t bnext = claimed;
        mask_columns();
        if (a.mu != nullptr) {
#pragma unroll
            for (int s = 0; s < S; ++s) centre_fragment<R1 / 8, 2>(xf[s], mu_lds, h);
        }
        uint64_t sample[S];
        uint32_t lab[S], flags[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            sample[s] = FINISH ? (uint64_t)(sid[s] & CROWD_ID) : ((b * 4 + wave) * S + s) * 32 + s_col;
            lab[s] = label[s];
            flags[s] = FINISH ? sid[s] : 0u;
        }

        f32x4 yf[S][RS / 8];                                          // search-stage fragments
        Keys2 ku[S];
        KeysM km[S];
        Best1 pos[S], neg[S];                                         // LOCAL: exact distances, offered in index order
        float dsum[S], margin[S];
        bool unsafe[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            dsum[s] = 0.0f;
            margin[s] = 0.0f;
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
            VQ_LAP(3);                     // 3: everything outside the three phases of a stage
            if (!(wrap && bnext >= nbatches)) issue(wrap ? 0u : q + 1, cur ^ 1);   // nothing to stage after the last batch
            VQ_LAP(0);                     // 0: issue of the next stage's DMA
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
                VQ_LAP(1);                 // 1: the stage's tiles
                __syncthreads();               // next stage landed (vmcnt(0)) and everyone is done reading this one
                VQ_LAP(2);                 // 2: the barrier (and the wait for the DMA in it)
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
                margin[s] = screen_margin(ss, a);
            }
        }

        u32x4 bh[BF ? S : 1][RS / 16], bl[BF ? S : 1][RS / 16];       // search fragments as bf16 pieces
        if constexpr (BF) {
#pragma unroll
            for (int s = 0; s < S; ++s) split_fragment<RS>(yf[s], bh[s], bl[s]);
        }
        // FINISH: the searches every sample was marked for; margin +INF: every row is measured (the exact scan)
        auto hit_margin = [&](int s, uint32_t which) {
            if (!FINISH || !(flags[s] & (which | CROWD_ALL))) return -INFINITY;
            return (flags[s] & CROWD_ALL) || unsafe[s] ? INFINITY : margin[s];
        };
        HitKeys<RS, MATCHED, ROOT, 2, HIT_CAP> hits0(yf[0], a.plain, a.ldp, a.K, hit_margin(0, CROWD_P), hit_margin(0, CROWD_N),
                                                     hit_lds + (FINISH ? tid : 0), 256u);
        HitKeys<RS, MATCHED, ROOT, 2, HIT_CAP> hits1(yf[S - 1], a.plain, a.ldp, a.K, hit_margin(S - 1, CROWD_P), hit_margin(S - 1, CROWD_N),
                                                     hit_lds + (FINISH ? (S - 1) * 2 * HIT_CAP * 256 + tid : 0), 256u);
        static_assert(S <= 2, "one HitKeys per sample tile of the wave");
        // One sample tile per wave, unlabelled, on the bf16 cores: the scores of a tile are screened between the MFMAs of
        // the next one (tile_mma_ready_spread); the last tile's after the stage loop.
        constexpr bool SPREAD = BF && !GRP && !FINISH && !MATCHED && S == 1 && RS >= 96 && MODE == MODE_EUCLID;
        TopKeys<false> spread_scr;
        f32x16 held;                                                  // SPREAD: the finished scores of the tile before
        uint32_t held_id = 0u;
        if constexpr (SPREAD) {
#pragma unroll
            for (int r = 0; r < 16; ++r) held[r] = PAD_SCORE;         // "tile -1": its keys never win
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
                        const uint32_t tile_id = qs * G2::G + tt;
                        if constexpr (SPREAD) {
                            spread_scr.ku = ku[0];
                            tile_mma_ready_spread<RS>(tile, bh[0], bl[0], acc[0], held, held_id, spread_scr, s_col, h);
                            ku[0] = spread_scr.ku;
                            held = acc[0];
                            held_id = tile_id;
                            continue;
                        }
                        if constexpr (BF) tile_mma_ready<RS, S>(tile, bh, bl, acc, s_col, h);   // (the stage holds pieces: engine_build)
                        else tile_mma<RS, S>(tile, yf, acc, s_col, h);
                        const uint32_t *tcls = (const uint32_t *)(tile + G2::W_FLOATS) + 32 + 4 * h;
                        Keys2 ou[S];
                        KeysM om[S];
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            ou[s] = ku[s];
                            om[s] = km[s];
                        }
                        if constexpr (FINISH) {
                            hits0.template tile<1>(acc[0], tile_id, tile + G2::W_FLOATS, lab[0], h);
                            if constexpr (S == 2) hits1.template tile<1>(acc[S - 1], tile_id, tile + G2::W_FLOATS, lab[S - 1], h);
                        } else if constexpr (GRP) {
#pragma unroll
                            for (int s = 0; s < S; ++s) {
                                GroupKeys<MATCHED> scr{ku[s], km[s]};
                                scr.template tile<1>(acc[s], tile_id, tile + G2::W_FLOATS, lab[s], h);
                                ku[s] = scr.ku;
                                km[s] = scr.km;
                            }
                        } else if constexpr (!MATCHED) {
#pragma unroll
                            for (int s = 0; s < S; ++s) {
                                screen16<1>(acc[s], ku[s]);            // acc[s] is fresh from the matrix pipe
                                settle_tiles(ku[s], ou[s], tile_id);
                            }
                        } else {
                            // the keys are packed by inline asm, which the compiler does not count as a reader of
                            // the accumulators the MFMA right in front has written: wait states by hand
#pragma unroll
                            for (int s = 0; s < S; ++s) settle_mfma(acc[s]);
                            static_for<0, 16>([&](auto oo) {
                                constexpr int o = decltype(oo)::value;
                                const uint32_t cls = tcls[8 * (o >> 2) + (o & 3)];
#pragma unroll
                                for (int s = 0; s < S; ++s) {
                                    const uint32_t key = pack_key(acc[s][o], o);
                                    const bool match = cls == lab[s];                 // helpers.rs:113 / :122
                                    const uint32_t kp = match ? key : KEY_NONE, kn = match ? KEY_NONE : key;
                                    offer_m(km[s], kp, kn);
                                }
                            });
#pragma unroll
                            for (int s = 0; s < S; ++s) settle_tiles_m(km[s], om[s], tile_id);
                        }
                    }
                }
            });
        }

        if constexpr (SPREAD) {                                       // the last tile's scores
            const Keys2 before = ku[0];
            screen16<1>(held, ku[0]);
            settle_tiles(ku[0], before, held_id);
        }
        // ---- epilogue: lane pair (s, h=0/1) agrees on the candidates, re-measures them exactly, stores
        if constexpr (FINISH) {
            hits0.merge();
            hits1.merge();
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const Best1 &wp = s == 0 ? hits0.wp : hits1.wp, &wn = s == 0 ? hits0.wn : hits1.wn;
                if (flags[s] & (CROWD_P | CROWD_ALL)) {
                    if (h == 0) {
                        if (a.out.idx) a.out.idx[sample[s]] = wp.i;
                    } else if (a.out.dist) a.out.dist[sample[s]] = wp.d;
                }
                if (MATCHED && (flags[s] & (CROWD_N | CROWD_ALL))) {
                    if (h == 0) {
                        if (a.out.jneg) a.out.jneg[sample[s]] = wn.i;
                    } else if (a.out.dneg) a.out.dneg[sample[s]] = wn.d;
                }
            }
        } else {
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
                // each group: best two of the pair's four by (key, index), both re-measured; a third key within the
                // margin of the best sends the sample to the exact scan (the guarantee: finish_keys, mfma_device.h)
                const bool skip = rescan;
                bool all = rescan, cp = false, cn = false;             // what the sample is marked for (crowd_mark) or scanned for
                auto group = [&](uint32_t b1, uint32_t t1, uint32_t b2, uint32_t t2, uint32_t b3, Best1 &w, bool &crowded) {
                    uint32_t f, fi, sd, si;
                    merge_pair(b1, t1, b2, t2, h, f, fi, sd, si);
                    const float third = pair_third(b1, b2, b3);
                    const bool have = key_valid(f) && !skip;
                    if (have) {
                        const float d1 = measure(fi);
                        w.d = d1; w.i = fi;
                        if (key_valid(sd)) {
                            const float d2 = measure(si);
                            if (lexless(d2, si, d1, fi)) { w.d = d2; w.i = si; }
                        }
                        crowded = !(third - __builtin_bit_cast(float, f) > margin[s]);
                        all |= !(w.d < INFINITY);
                    }
                    return have;
                };
                const uint64_t tile32 = (b * 4 + wave) * S + s;        // this tile of 32 samples in the launch
                if constexpr (GRP) {
                    finish_groups<MATCHED>(
                        h, unsafe[s], margin[s], ku[s], km[s], wp, wn, measure,
                        [&](uint32_t k, bool positive) { return k < a.K && (!MATCHED || (a.cls[k] == lab[s]) == positive); },
                        [&]() { exact_scan<RS, ROOT>(yf[s], a.plain, a.ldp, a.K, h, a.cls, MATCHED, lab[s], wp, wn); }, nullptr,
                        a.crowd_mask, tile32, a.N);
                } else {
                if (!MATCHED) {
                    if (!group(ku[s].b1, ku[s].t1, ku[s].b2, ku[s].t2, ku[s].b3, wp, cp)) all = true;
                } else {
                    group(km[s].p, km[s].tp, km[s].p2, km[s].tp2, km[s].p3, wp, cp);
                    group(km[s].n, km[s].tn, km[s].n2, km[s].tn2, km[s].n3, wn, cn);
                }
                if (!crowd_mark(a.crowd_mask, tile32, a.N, cp, cn, all, h) && (cp || cn || all))
                    exact_scan<RS, ROOT>(yf[s], a.plain, a.ldp, a.K, h, a.cls, MATCHED, lab[s], wp, wn);
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
        if (MODE != MODE_OMEGA) load_samples(bnext);
        b = bnext;
    }
#ifdef VQ_ENGINE_DIAG
    if (a.dbg && lane == 0) {
        unsigned long long *d = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 4; ++i) d[i] = tm[i];
        d[4] = stamp() - t_begin;
        d[5] = __builtin_amdgcn_s_memrealtime() - r_begin;
        d[6] = r_begin;
        d[7] = r_begin + d[5];
    }
#endif
}

namespace {
// WPS = workgroups per CU = waves per SIMD the register budget is cut for (256-thread workgroups).
template <int MODE, int R1, int R2, int S, int WPS = (R1 >= 256 ? 1 : 2)>
hipError_t launch_t(const EngineArgs &a, bool matched, hipStream_t st)
{
    const uint64_t cap = 256ull * WPS;                                // 256 CUs x workgroups per CU
    const unsigned grid = (unsigned)(a.nbatches < cap ? a.nbatches : cap);
    const hipError_t e = zero_queue(a, st);
    if (e != hipSuccess) return e;
    if constexpr (MODE != MODE_LOCAL && (MODE == MODE_OMEGA ? R2 : R1) <= 128) {
        // Group keys where a candidate's four rows are one or two fetches each (the exact rows come from L2 here, a round
        // trip per row: at 128 values per row four of them cost a batch more than the cheaper screening returns --
        // K = 1024: 6.6e8 samples/s against 7.3e8, K = 4096: 0.93e8 against 1.09e8)
        if constexpr ((MODE == MODE_OMEGA ? R2 : R1) <= 32) {
            if (a.bf && a.K > 15u * 32u) {
                if (matched)
                    hipLaunchKernelGGL((k_engine<MODE, R1, R2, S, true, WPS, true, true>), dim3(grid), dim3(256), 0, st, a);
                else
                    hipLaunchKernelGGL((k_engine<MODE, R1, R2, S, false, WPS, true, true>), dim3(grid), dim3(256), 0, st, a);
                return hipGetLastError();
            }
        }
        if (a.bf) {
            if (matched)
                hipLaunchKernelGGL((k_engine<MODE, R1, R2, S, true, WPS, true>), dim3(grid), dim3(256), 0, st, a);
            else
                hipLaunchKernelGGL((k_engine<MODE, R1, R2, S, false, WPS, true>), dim3(grid), dim3(256), 0, st, a);
            return hipGetLastError();
        }
    }
    if (matched)
        hipLaunchKernelGGL((k_engine<MODE, R1, R2, S, true, WPS>), dim3(grid), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((k_engine<MODE, R1, R2, S, false, WPS>), dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}

// the finish pass (k_engine FINISH): one workgroup per batch of the list, at most the first pass's grid
template <int MODE, int R1, int R2, int S, int WPS = (R1 >= 256 ? 1 : 2)>
hipError_t launch_finish_t(const EngineArgs &a, bool matched, hipStream_t st)
{
    if constexpr (MODE != MODE_LOCAL && (MODE == MODE_OMEGA ? R2 : R1) <= 128) {
        const hipError_t e = zero_queue(a, st);
        if (e != hipSuccess) return e;
        const unsigned grid = 256u * WPS;
        if (matched)
            hipLaunchKernelGGL((k_engine<MODE, R1, R2, S, true, WPS, true, false, true>), dim3(grid), dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((k_engine<MODE, R1, R2, S, false, WPS, true, false, true>), dim3(grid), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

template <int MODE, int R1, int R2, int S>
bool pick(int mode, int r1, int r2, LaunchFn &fn, int *samples_per_batch)
{
    if (mode != MODE || r1 != R1 || (MODE == MODE_OMEGA && r2 != R2)) return false;
    fn = &launch_t<MODE, R1, R2, S>;
    *samples_per_batch = 4 * S * 32;                                  // 4 waves x S tiles x 32 samples
    return true;
}
template <int MODE, int R1, int R2, int S>
bool pick_finish(int mode, int r1, int r2, LaunchFn &fn)
{
    if (mode != MODE || r1 != R1 || (MODE == MODE_OMEGA && r2 != R2)) return false;
    fn = &launch_finish_t<MODE, R1, R2, S>;
    return true;
}
}  // namespace

// the Euclidean engines whose search rows are 64 or 128 values long (shorter ones run on group keys, which settle most
// crowded samples in the first pass; the Omega engines scan in their waves)
LaunchFn streamed_finish_launcher(int mode, int R1, int R2)
{
    LaunchFn fn = nullptr;
    (void)(pick_finish<MODE_EUCLID, 64, 64, 2>(mode, R1, R2, fn) || pick_finish<MODE_EUCLID, 128, 128, 1>(mode, R1, R2, fn));
    return fn;
}

LaunchFn streamed_launcher(int mode, int R1, int R2, int *spb)
{
    LaunchFn fn = nullptr;
    (void)(pick<MODE_EUCLID, 16, 16, 2>(mode, R1, R2, fn, spb) || pick<MODE_EUCLID, 32, 32, 2>(mode, R1, R2, fn, spb) ||
           pick<MODE_EUCLID, 64, 64, 2>(mode, R1, R2, fn, spb) || pick<MODE_EUCLID, 128, 128, 1>(mode, R1, R2, fn, spb) ||
           pick<MODE_EUCLID, 256, 256, 1>(mode, R1, R2, fn, spb) ||
           pick<MODE_OMEGA, 128, 128, 1>(mode, R1, R2, fn, spb) || pick<MODE_OMEGA, 256, 16, 1>(mode, R1, R2, fn, spb) ||
           pick<MODE_OMEGA, 16, 16, 1>(mode, R1, R2, fn, spb) || pick<MODE_OMEGA, 32, 16, 1>(mode, R1, R2, fn, spb) ||
           pick<MODE_OMEGA, 32, 32, 1>(mode, R1, R2, fn, spb) || pick<MODE_OMEGA, 64, 64, 1>(mode, R1, R2, fn, spb) ||
           pick<MODE_OMEGA, 256, 64, 1>(mode, R1, R2, fn, spb) ||
           pick<MODE_LOCAL, 16, 16, 2>(mode, R1, R2, fn, spb) || pick<MODE_LOCAL, 32, 32, 2>(mode, R1, R2, fn, spb) ||
           pick<MODE_LOCAL, 64, 64, 2>(mode, R1, R2, fn, spb) || pick<MODE_LOCAL, 128, 128, 2>(mode, R1, R2, fn, spb) ||
           pick<MODE_LOCAL, 256, 256, 1>(mode, R1, R2, fn, spb));
    return fn;
}

}  // namespace vq
