// LGMLVQ (src/lgmlvq/lgmlvq.rs:436-506) with the projections on the bf16 matrix cores: k_local_pj.
//
// The distance of sample x to prototype j is ||Omega_j x - Omega_j w_j||^2 with a matrix of its own per prototype, so
// the whole job is projections -- k_engine<MODE_LOCAL> runs them on v_mfma_f32_32x32x2_f32 at 0.9 of that pipe's peak
// and that is where it ends.  Here Omega_j and x are each split into three bf16 pieces that add up to the f32 value
// exactly, and the six piece products down to 2^-16 of the leading one run on v_mfma_f32_32x32x16_bf16 with f32
// accumulation (mfma_device.h split3): 6 x 32 cycles per 16 reduction values instead of 8 x 64, and a result measured
// at or below the f32 chain's error (tools/bf16x6_probe.hip).
//
// Same skeleton as k_engine: a workgroup of 4 waves walks the stream of Omega_j tiles through a double-buffered LDS
// stage filled by LDS-DMA, one barrier per stage; batches are claimed from a counter.  What differs:
//   * the tiles are Geom3 tiles (engine.h): 6 bytes per weight, read as three ds_read_b128 per step;
//   * a wave holds S tiles of 32 samples as PIECES (96 registers per tile at 128 values), one workgroup per CU: two waves
//     per SIMD from 64 values on (S = 2; S = 1 at 256 values), one wave with S = 3 and the whole register file below;
//     the tiles' chains are independent, which is what keeps the matrix pipe fed -- so a staged tile (25 KiB for 128
//     values) serves 512 samples, and the stream of all Omega_j (6.5 MB for config 5, more than an XCD's L2 holds) stays
//     below 2 TB/s chip-wide;
//   * the sample rows are dead once split, and are fetched again for the next batch behind the last tile.
#include <cstdlib>

#include "mfma_device.h"

namespace vq {

// stage_group (mfma_device.h) for NW waves
template <int BYTES, int NW>
__device__ __forceinline__ void stage_group_nw(char *lds_dst, const char *gsrc, int wave, int lane)
{
    constexpr int PIECES = (BYTES + 1023) / 1024;
#pragma unroll
    for (int i = 0; i < (PIECES + NW - 1) / NW; ++i) {
        const int p = i * NW + wave;
        const int off = p * 1024 + lane * 16;
        if (p < PIECES && off < BYTES)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + off),
                                             (__attribute__((address_space(3))) void *)(lds_dst + p * 1024), 16, 0, 0);
    }
}

template <int R1, int S, bool MATCHED, int NW = 4>
__global__ void __launch_bounds__(NW * 64, 1) k_local_pj(const EngineArgs a)
{
    using G3 = Geom3<R1>;
    constexpr int NQ = R1 / 16;
    __shared__ __attribute__((aligned(16))) char smem[2 * G3::STAGE_BYTES];
    __shared__ uint32_t claimed;                                     // the batch after this one
    __shared__ __attribute__((aligned(16))) float mu_lds[R1];
    stage_centre<R1>(mu_lds, a.mu, threadIdx.x, NW * 64);                // published by the first barrier below

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    const uint32_t ngroups = a.ngroups1;

    auto issue = [&](uint32_t q, int buf) {                          // stage q of the cyclic stream of Omega_j tiles
        stage_group_nw<G3::STAGE_BYTES, NW>(smem + buf * G3::STAGE_BYTES, (const char *)a.img1 + (size_t)q * G3::STAGE_BYTES, wave, lane);
    };

    // this wave's S tiles of 32 samples of batch bb, straight from HBM through a range-checked buffer descriptor
    // (k_engine's load_samples)
    f32x4 xf[S][R1 / 8];
    uint32_t label[S];
    const uint32_t row_bytes = a.D * 4u;
    const uint32_t voff = (uint32_t)s_col * row_bytes + (uint32_t)h * 16u;
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane(wave);
    auto load_samples = [&](uint64_t bb) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint64_t first = ((bb * NW + wave_u) * S + s) * 32;
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

#ifdef VQ_LOCAL_DIAG      // phase stamps and the in-kernel clock (engine_assign prints them: VQHIP_VARIANT=35 VQHIP_DUMP=1 VQHIP_NW4=1)
    const unsigned long long t_begin = stamp(), r_begin = __builtin_amdgcn_s_memrealtime();
    unsigned long long tm[4] = {0, 0, 0, 0}, tprev_ = t_begin;
    auto lap = [&](int slot) { const unsigned long long t = stamp(); tm[slot] += t - tprev_; tprev_ = t; };
#define VQ_LAP(i) lap(i)
#else
#define VQ_LAP(i)
#endif
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
        // rows -> pieces: D < R1 cut (0 x INF is NaN), the model's centre subtracted, three bf16 pieces per value
        u32x4 xp[S][NQ][3];
        uint64_t sample[S];
        uint32_t lab[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (a.D < (uint32_t)R1) {
#pragma unroll
                for (int j = 0; j < R1 / 8; ++j) xf[s][j] = mask_chunk(xf[s][j], (uint32_t)(2 * j + h), a.D);
            }
            if (a.mu != nullptr) centre_fragment<R1 / 8, 2>(xf[s], mu_lds, h);
#pragma unroll
            for (int q = 0; q < NQ; ++q) split24(xf[s][2 * q], xf[s][2 * q + 1], xp[s][q]);
            sample[s] = ((b * NW + wave) * S + s) * 32 + s_col;
            lab[s] = label[s];
        }

        Best1 pos[S], neg[S];                                         // exact distances, offered in index order
        float dsum[S];
#pragma unroll
        for (int s = 0; s < S; ++s) dsum[s] = 0.0f;

        for (uint32_t q = 0; q < ngroups; ++q) {
            // prefetch the next stage (nothing to stage after the last batch), compute on this one, barrier
            const bool wrap = q + 1 == ngroups;
            VQ_LAP(3);                     // 3: everything outside the stage loop's three phases
            if (!(wrap && bnext >= a.nbatches)) issue(wrap ? 0u : q + 1, cur ^ 1);
            VQ_LAP(0);                     // 0: issue of the next stage's DMA
            const char *stage = smem + cur * G3::STAGE_BYTES;
            // ---- z = Omega_j x - Omega_j w_j (accumulator starts at -p_j), ||z||^2 summed per prototype
#pragma unroll
            for (int tt = 0; tt < G3::G; ++tt) {
                const char *tile = stage + tt * G3::TILE_BYTES;
                f32x16 acc[S];
                tile_mma_pj<R1, S>(tile, xp, acc, s_col, h);
#ifdef VQ_LOCAL_DIAG
                asm volatile("" : "+v"(acc[0]), "+v"(acc[S - 1]));
#endif
                VQ_LAP(1);                 // 1: the tile's MFMAs
                const uint32_t ti = q * G3::G + tt;
                const uint32_t j = ti / a.tiles_per_proto, t = ti % a.tiles_per_proto;
#pragma unroll
                for (int s = 0; s < S; ++s)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dsum[s] = fmaf(acc[s][r], acc[s][r], dsum[s]);
                if (t + 1 == a.tiles_per_proto) {
                    const uint32_t cj = ((const uint32_t *)(tile + G3::W_BYTES))[32];
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const float total = dsum[s] + xor32(dsum[s]);
                        dsum[s] = 0.0f;
                        if (j < a.K) {                                // pad tiles past the last prototype never compete
                            if (!MATCHED || cj == lab[s]) pos[s].offer(total, j);   // lgmlvq.rs:448 / :487
                            else neg[s].offer(total, j);                             // lgmlvq.rs:496
                        }
                    }
                }
            }
            VQ_LAP(2);                     // 2: squared norms, offers
            __syncthreads();               // next stage landed (vmcnt(0)) and everyone is done reading this one
            cur ^= 1;
        }

#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (sample[s] < a.N) {
                if (h == 0) {
                    if (a.out.idx) a.out.idx[sample[s]] = pos[s].i;
                    if (MATCHED && a.out.jneg) a.out.jneg[sample[s]] = neg[s].i;
                } else {
                    if (a.out.dist) a.out.dist[sample[s]] = pos[s].d;
                    if (MATCHED && a.out.dneg) a.out.dneg[sample[s]] = neg[s].d;
                }
            }
        }
        load_samples(bnext);
        b = bnext;
    }
#ifdef VQ_LOCAL_DIAG
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
template <int R1, int S, int NW = 4>
hipError_t launch_local_pj(const EngineArgs &a, bool matched, hipStream_t st)
{
    const unsigned grid = (unsigned)(a.nbatches < 256 ? a.nbatches : 256);      // one workgroup per CU
    const hipError_t e = zero_queue(a, st);
    if (e != hipSuccess) return e;
    if (matched)
        hipLaunchKernelGGL((k_local_pj<R1, S, true, NW>), dim3(grid), dim3(NW * 64), 0, st, a);
    else
        hipLaunchKernelGGL((k_local_pj<R1, S, false, NW>), dim3(grid), dim3(NW * 64), 0, st, a);
    return hipGetLastError();
}
}  // namespace

LaunchFn local_pj_launcher(int R1, int *samples_per_batch)
{
    // S sample tiles per wave as pieces (12 S registers per 16 values of a row), NW waves per workgroup, one workgroup per CU.
    // 128 values: two waves per SIMD with two tiles each measured 6.8 % faster than one wave with three (config 5, one box:
    // 31.2 against 33.2 ms per 4M rows; 8 x 1: 34.4, 12 x 1: 32.0); 64 values the same way +4 %, 256 values as 8 x 1
    // against 4 x 2 +7.5 % (D = 200: +14 %); rows of up to 32 values keep four waves.
    int S = 3, NW = 4;
    LaunchFn fn = nullptr;
    switch (R1) {
    case 16: fn = &launch_local_pj<16, 3>; break;
    case 32: fn = &launch_local_pj<32, 3>; break;
    case 64: fn = &launch_local_pj<64, 2, 8>; S = 2; NW = 8; break;
    case 128: fn = &launch_local_pj<128, 2, 8>; S = 2; NW = 8; break;
    case 256: fn = &launch_local_pj<256, 1, 8>; S = 1; NW = 8; break;
    default: return nullptr;
    }
    *samples_per_batch = NW * S * 32;                                 // NW waves x S tiles x 32 samples
    return fn;
}

}  // namespace vq
