// Euclidean models with a handful of prototypes (VQ / LVQ / GLVQ with K <= 8: src/helpers.rs:11-24, 55-77, 91-132) on
// gfx950: the HBM-bound member of the path.
//
// With so few prototypes one 32-row MFMA tile is mostly padding, and everything around it (screening, exact
// re-measurement, the guard) costs more than the search itself; the measured limit of the matrix-core kernels at K <= 32
// was the latency of their short per-batch phases and the LDS pipe, not HBM.  Here every distance
// is computed as helpers.rs:18-23 states it -- sum (x_d - w_kd)^2 on the vector ALUs, packed f32 -- which needs no
// screening, no re-measurement, no centring and no guard:
//   * a wave owns S tiles of 32 samples; lane (s, h) holds the 16-byte chunks c with (c & 1) == h of its samples' rows
//     (engine.h's fragment layout) in registers;
//   * the loop runs over CHUNKS outside and prototypes inside: one ds_read_b128 of w_k's chunk (the same address in
//     all 32 lanes of a half: a broadcast) serves S samples per lane, and K * S running sums (pairs, v_pk_fma_f32)
//     stay in registers;
//   * a chunk's registers are dead once its step is done, so the same chunk of the wave's NEXT batch is requested into
//     them right there: one register set, loads in flight all the time, no LDS staging;
//   * argmin in index order with strict '<' on the squared sums (helpers.rs:70), one sqrt for the winner (helpers.rs:23).
// Work per sample: 3 K D flops on the vector ALUs, 4 D + 8 bytes of HBM traffic: K = 8, D = 128 needs 0.6 flop per byte
// and cycle budget to spare, so the kernel streams at what the memory system gives.
#include <cstdlib>

#include "mfma_device.h"

namespace vq {

// Prototype image in LDS: KP rows of R floats holding -w_k (rows past K zero), then KP class ids.
template <int R, int S, int KP, bool MATCHED, bool CUT, int NW = 8>
__global__ void __launch_bounds__(NW * 64, NW / 4) k_direct(const EngineArgs a)
{
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    constexpr int NJ = R / 8;                          // chunks per lane and sample
    constexpr int BATCH = 32 * S;                      // samples per wave and batch
    __shared__ __attribute__((aligned(16))) float w_lds[KP * R];
    __shared__ uint32_t cls_lds[KP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s_col = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < KP * R; i += NW * 64) w_lds[i] = a.img1[i];
    if (MATCHED && threadIdx.x < KP) cls_lds[threadIdx.x] = ((const uint32_t *)a.img1)[KP * R + threadIdx.x];
    __syncthreads();

    const uint32_t wave_id = blockIdx.x * NW + (uint32_t)__builtin_amdgcn_readfirstlane(wave);
    BlockQueue<1> queue(a.queue, lane, wave_id, gridDim.x * NW, a.nbatches);
    const uint32_t row_bytes = a.D * 4u;
    const uint32_t voff = (uint32_t)s_col * row_bytes + (uint32_t)h * 16u;
    auto records = [&](uint64_t first, uint32_t unit) -> uint32_t {       // bytes from sample `first` to the end, clamped
        if (first >= a.N) return 0u;
        const uint64_t left = (a.N - first) * unit;
        return left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)left;
    };
    // rows past N read as zeros (hardware range check); the descriptor is rebased per tile (wave-uniform)
    auto rows_of = [&](uint64_t bb, int t) {
        const uint64_t first = bb * BATCH + (uint64_t)t * 32;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(a.X + first * a.D), 0, (int)records(first, row_bytes), 0x00020000);
    };
    f32x4 x[S][NJ];
    uint32_t label[S];
#pragma unroll
    for (int t = 0; t < S; ++t) label[t] = 0u;
    auto fetch_labels = [&](uint64_t bb) {
        if (MATCHED) {
#pragma unroll
            for (int t = 0; t < S; ++t) {
                const uint64_t first = bb * BATCH + (uint64_t)t * 32;
                const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)(a.y + first), 0, (int)records(first, 4u), 0x00020000);
                label[t] = __builtin_amdgcn_raw_buffer_load_b32(ry, (uint32_t)s_col * 4u, 0, 0);
            }
        }
    };

    uint64_t b = wave_id;
    queue.request();
    if (b < a.nbatches) {
#pragma unroll
        for (int t = 0; t < S; ++t) {
            const __amdgpu_buffer_rsrc_t rx = rows_of(b, t);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                x[t][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, j * 32, 0));
        }
    }

    while (b < a.nbatches) {
        const uint64_t nb = queue.take();              // the batch after this one (>= nbatches: none; its loads read zeros)
        queue.request();
        uint32_t lab[S];
        fetch_labels(b);
#pragma unroll
        for (int t = 0; t < S; ++t) lab[t] = label[t];
        __amdgpu_buffer_rsrc_t rnext[S];
#pragma unroll
        for (int t = 0; t < S; ++t) rnext[t] = rows_of(nb, t);

        f32x2 acc[KP][S];
#pragma unroll
        for (int k = 0; k < KP; ++k)
#pragma unroll
            for (int t = 0; t < S; ++t) acc[k][t] = f32x2{0.0f, 0.0f};
        // The prototype reads do not depend on the batch: left alone, the compiler hoists all KP * R / 8 of them out of
        // the batch loop and keeps the lane's whole share of the prototypes in registers (and spills).  An offset it
        // cannot see through keeps them where they are.
        uint32_t woff = 4u * (uint32_t)h;
        asm volatile("" : "+v"(woff));

        static_for<0, NJ>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            f32x4 xc[S];
#pragma unroll
            for (int t = 0; t < S; ++t) {
                xc[t] = x[t][j];
                if constexpr (CUT) xc[t] = mask_chunk(xc[t], (uint32_t)(2 * j + h), a.D);
            }
            // A cache line of a row (128 bytes) holds the chunks of four consecutive steps.  Once the fourth is in xc the
            // registers of all four are free, and the same four chunks of the next batch go out back to back: requested one
            // step at a time, the four touches of a line are a quarter of a batch apart and each one fetches the line again
            // (measured: 2.2 TB/s against 3.8).
            if constexpr ((j & 3) == 3 || j == NJ - 1) {
                constexpr int j0 = j & ~3;
#pragma unroll
                for (int t = 0; t < S; ++t)
#pragma unroll
                    for (int i = j0; i <= j; ++i)
                        x[t][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rnext[t], voff, i * 32, 0));
            }
            const float *wj = w_lds + woff + 8 * j;
            static_for<0, KP>([&](auto kk) {
                constexpr int k = decltype(kk)::value;
                const f32x4 w = *(const f32x4 *)(wj + k * R);
#pragma unroll
                for (int t = 0; t < S; ++t) {
                    const f32x2 d0 = f32x2{xc[t][0], xc[t][1]} + f32x2{w[0], w[1]};     // w holds -w_k: x - w_k, one v_pk_add_f32
                    const f32x2 d1 = f32x2{xc[t][2], xc[t][3]} + f32x2{w[2], w[3]};
                    acc[k][t] = __builtin_elementwise_fma(d0, d0, acc[k][t]);
                    acc[k][t] = __builtin_elementwise_fma(d1, d1, acc[k][t]);
                }
            });
            // keep every step's prototype reads inside the step: hoisted ahead by the scheduler they take the registers
            // the next batch's rows are landing in
            __builtin_amdgcn_sched_barrier(0);
        });

        // The sums are used only further down, some inside conditional code: left alone, the compiler sinks whole chains of
        // the unrolled loop there, step by step, and parks the prototype chunks they need in scratch.  An opaque use
        // keeps every chain where it was written.
#pragma unroll
        for (int k = 0; k < KP; ++k)
#pragma unroll
            for (int t = 0; t < S; ++t) asm volatile("" : "+v"(acc[k][t]));

        // ---- argmin in index order (helpers.rs:58-59, 70; labelled: :113 / :122), one sqrt for the winner (:23)
#pragma unroll
        for (int t = 0; t < S; ++t) {
            Best1 wp, wn;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                float v = acc[k][t][0] + acc[k][t][1];
                v += xor32(v);
                if ((uint32_t)k < a.K) {
                    if (!MATCHED || cls_lds[k] == lab[t]) wp.offer(v, (uint32_t)k);
                    else wn.offer(v, (uint32_t)k);
                }
            }
            wp.d = sqrtf(wp.d);
            if (MATCHED) wn.d = sqrtf(wn.d);
            const uint64_t first = b * BATCH + (uint64_t)t * 32;
            const int rec = (int)records(first, 4u);
            const uint32_t so = (uint32_t)s_col * 4u;
            auto put = [&](void *base, uint32_t bits) {
                __builtin_amdgcn_raw_buffer_store_b32(bits, __builtin_amdgcn_make_buffer_rsrc(base, 0, rec, 0x00020000), so, 0, 0);
            };
            if (h == 0) {
                if (a.out.idx) put(a.out.idx + first, wp.i);
                if (MATCHED && a.out.jneg) put(a.out.jneg + first, wn.i);
            } else {
                if (a.out.dist) put(a.out.dist + first, __builtin_bit_cast(uint32_t, wp.d));
                if (MATCHED && a.out.dneg) put(a.out.dneg + first, __builtin_bit_cast(uint32_t, wn.d));
            }
        }
        b = nb;
    }
}

namespace {
template <int R, int S, int KP>
hipError_t launch_direct(const EngineArgs &a, bool matched, hipStream_t st)
{
    constexpr int NW = 8;                                             // waves per workgroup = per CU: 2 per SIMD
    const uint64_t wgs = (a.nbatches + NW - 1) / NW;
    const unsigned grid = (unsigned)(wgs < 256 ? wgs : 256);          // one workgroup per CU
    const hipError_t e = zero_queue(a, st);
    if (e != hipSuccess) return e;
    const bool cut = a.D < (uint32_t)R;
    if (matched && cut)
        hipLaunchKernelGGL((k_direct<R, S, KP, true, true, NW>), dim3(grid), dim3(NW * 64), 0, st, a);
    else if (matched)
        hipLaunchKernelGGL((k_direct<R, S, KP, true, false, NW>), dim3(grid), dim3(NW * 64), 0, st, a);
    else if (cut)
        hipLaunchKernelGGL((k_direct<R, S, KP, false, true, NW>), dim3(grid), dim3(NW * 64), 0, st, a);
    else
        hipLaunchKernelGGL((k_direct<R, S, KP, false, false, NW>), dim3(grid), dim3(NW * 64), 0, st, a);
    return hipGetLastError();
}

template <int R, int S>
LaunchFn pick_kp(int KP)
{
    switch (KP) {
    case 4: return &launch_direct<R, S, 4>;
    case 8: return &launch_direct<R, S, 8>;
    default: return nullptr;
    }
}
}  // namespace

// R: padded row length (32 / 64 / 128: two tiles of 32 samples per wave; 256: one), KP: prototypes padded to 4 / 8.
LaunchFn direct_launcher(int R, int KP, int *samples_per_batch)
{
    // two tiles of 32 samples per wave where the rows fit the register file twice (measured at D = 128, K = 8: 5.0 TB/s;
    // one tile per wave and twice the waves: 2.5 TB/s -- every prototype chunk read from LDS then serves one sample per lane)
    *samples_per_batch = R == 256 ? 32 : 64;
    switch (R) {
    case 32: return pick_kp<32, 2>(KP);
    case 64: return pick_kp<64, 2>(KP);
    case 128: return pick_kp<128, 2>(KP);
    case 256: return pick_kp<256, 1>(KP);
    default: return nullptr;
    }
}

}  // namespace vq
