// MFMA tile engine: shared geometry (host image builder + device kernels agree through this file).
//
// Everything is built from one primitive: a 32 x 32 output tile  C = A . B  computed with
// v_mfma_f32_32x32x2_f32 (exact f32: a k-ordered fmaf chain), where
//   A = 32 "weight rows" (prototypes, Omega rows, Omega_j rows) x R reduction values, read from LDS,
//   B = 32 samples x R, held in registers for the whole pass ("sample-stationary"),
//   C: lane (s = lane & 31, h = lane >> 5) owns column s = one sample, rows 8g + 4h + e
//      (g = reg >> 2, e = reg & 3) = 16 weight rows of that sample.
// So the argmin over prototypes runs inside a lane, with one lane <-> lane^32 exchange at the end.
//
// Reduction index layout: a row of R floats is R/4 chunks of 16 B.  Lane half h owns the chunks
// c with (c & 1) == h, in order: its j-th chunk is c = 2j + h.  MFMA step (j, e) multiplies element
// e of chunk 2j (k = 0, lanes 0-31) and of chunk 2j+1 (k = 1, lanes 32-63).  With this choice the
// 16 accumulator registers of a projection tile ARE the next stage's B fragment (row
// 8g + 4h + e of tile t is element e of chunk 2(4t+g) + h): no cross-lane movement between stages.
#pragma once
#include <cstdint>

namespace vq {

constexpr int TILE_ROWS = 32;
constexpr int TAIL_FLOATS = 64;           // per tile: 32 x f32 accumulator init + 32 x u32 class id
constexpr int TAIL_BYTES = TAIL_FLOATS * 4;

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

template <int R>
struct Geom {
    static_assert(R >= 8 && R <= 256 && (R & (R - 1)) == 0, "reduction length: power of two in [8, 256]");
    static constexpr int C = R / 4;                           // 16-byte chunks per row
    // Rows are padded by one chunk: consecutive rows start 4 banks apart (mod 64), so a ds_read_b128
    // by 16 lanes holding 16 consecutive rows (same chunk) touches 16 different 16-byte bank slots,
    // and every chunk of a row sits at a compile-time offset from one per-lane row address.
    static constexpr int ROW_FLOATS = R + 4;
    static constexpr int W_FLOATS = TILE_ROWS * ROW_FLOATS;
    static constexpr int TILE_FLOATS = W_FLOATS + TAIL_FLOATS;
    static constexpr int TILE_BYTES = TILE_FLOATS * 4;
    static constexpr int G = R >= 128 ? 1 : 128 / R;          // tiles per LDS stage (~16 KiB of weights)
    static constexpr int STAGE_FLOATS = G * TILE_FLOATS;
    static constexpr int STAGE_BYTES = STAGE_FLOATS * 4;
    // float offset of (row, chunk c) inside a tile image
    __host__ __device__ static constexpr int chunk_off(int row, int c) { return row * ROW_FLOATS + c * 4; }
};

// Tiles whose weights are stored as three bf16 pieces per value (v = p1 + p2 + p3 exactly: mfma_device.h split3), the
// A operand of v_mfma_f32_32x32x16_bf16: per row, per step q of 16 reduction values and lane half h, 16 bytes of first
// pieces, 16 of second, 16 of third, for the row's values 16 q + 4 h .. + 3 and 16 q + 8 + 4 h .. + 3.  Rows are padded
// by 16 bytes like Geom's (consecutive rows start 4 banks apart); the tail is Geom's (32 init values, 32 class ids).
template <int R>
struct Geom3 {
    static_assert(R >= 16 && R <= 256 && (R & (R - 1)) == 0, "reduction length: power of two in [16, 256]");
    static constexpr int NQ = R / 16;
    static constexpr int ROW_BYTES = NQ * 96 + 16;
    static constexpr int W_BYTES = TILE_ROWS * ROW_BYTES;
    static constexpr int TILE_BYTES = W_BYTES + TAIL_BYTES;
    static constexpr int G = Geom<R>::G;                       // tiles per LDS stage
    static constexpr int STAGE_BYTES = G * TILE_BYTES;
    __host__ __device__ static constexpr int piece_off(int row, int q, int h, int c) { return row * ROW_BYTES + q * 96 + h * 48 + c * 16; }
};

}  // namespace vq
