// Host-side helpers shared by the model builders (engine_host.hip, kernels_project.hip): tile images in the
// geometry of engine.h, uploads, and the batch-counter slots.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "engine.h"
#include "mfma_host.h"

namespace vq {

// Batch counters, one per launch in flight: launches of one model on different streams (the two staging
// streams of the host path) may overlap at their edges, and a launch that zeroed the counter of the one
// still running would hand its batches to it.  Slots are used round-robin.
constexpr uint32_t QUEUE_SLOTS = 16;                                  // sets of counters (mfma_host.h: QUEUE_BYTES each)
constexpr uint32_t QUEUE_STRIDE = QUEUE_GROUPS * QUEUE_GROUP_STRIDE;  // u32 between sets

// Tile images for `nrows` weight rows of length `len` (zero padded to R), scaled by `scale`;
// init[r] / cls[r] fill the tail; pad rows get (0 weights, pad_init, cls 0xFFFFFFFF).
template <int R>
inline std::vector<float> build_images(const float *rows, size_t nrows, size_t len, size_t ld, float scale, const float *init,
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

inline uint16_t to_bf16(float v)                      // round to nearest even, like v_cvt_pk_bf16_f32
{
    uint32_t b;
    std::memcpy(&b, &v, 4);
    if ((b & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((b >> 16) | 0x40u);
    b += 0x7FFFu + ((b >> 16) & 1u);
    return (uint16_t)(b >> 16);
}
inline float from_bf16(uint16_t p)
{
    const uint32_t b = (uint32_t)p << 16;
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}
// v = p[0] + p[1] + p[2] (mfma_device.h split3: the remainders are exact in f32)
inline void split3_host(float v, uint16_t (&p)[3])
{
    p[0] = to_bf16(v);
    volatile float r = v - from_bf16(p[0]);
    p[1] = to_bf16(r);
    volatile float s = r - from_bf16(p[1]);
    p[2] = to_bf16(s);
}

// The same as Geom3 tiles (engine.h): three bf16 pieces per weight (their sum is the weight exactly), Geom's tail.
// Returned as floats for to_device; ngroups stages of Geom3<R>::STAGE_BYTES.
template <int R>
inline std::vector<float> build_images3(const float *rows, size_t nrows, size_t len, size_t ld, const float *init, const uint32_t *cls,
                                        uint32_t &ngroups)
{
    using G3 = Geom3<R>;
    const size_t tiles = (nrows + 31) / 32;
    ngroups = (uint32_t)((tiles + G3::G - 1) / G3::G);
    std::vector<unsigned char> bytes((size_t)ngroups * G3::STAGE_BYTES, 0);
    for (size_t t = 0; t < (size_t)ngroups * G3::G; ++t) {
        unsigned char *tile = bytes.data() + t * G3::TILE_BYTES;
        float *tail = reinterpret_cast<float *>(tile + G3::W_BYTES);
        for (int r = 0; r < 32; ++r) {
            const size_t gr = t * 32 + r;
            uint32_t c = 0xFFFFFFFFu;
            if (gr < nrows) {
                for (size_t d = 0; d < len; ++d) {
                    // value d of the row: step d / 16, lane half (d / 4) & 1, element (d & 3) + 4 ((d / 8) & 1)
                    uint16_t p[3];
                    split3_host(rows[gr * ld + d], p);
                    const int q = (int)(d / 16), hh = (int)((d / 4) & 1), e = (int)((d & 3) + 4 * ((d / 8) & 1));
                    for (int cc = 0; cc < 3; ++cc) std::memcpy(tile + G3::piece_off(r, q, hh, cc) + 2 * e, &p[cc], 2);
                }
                tail[r] = init ? init[gr] : 0.0f;
                if (cls) c = cls[gr];
            }
            std::memcpy(&tail[32 + r], &c, 4);
        }
    }
    std::vector<float> img(bytes.size() / 4);
    std::memcpy(img.data(), bytes.data(), bytes.size());
    return img;
}

template <class T>
inline hipError_t to_device(T **dst, const std::vector<T> &src)
{
    hipError_t e = hipMalloc((void **)dst, src.size() * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
}

}  // namespace vq
