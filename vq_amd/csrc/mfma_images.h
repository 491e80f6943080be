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

template <class T>
inline hipError_t to_device(T **dst, const std::vector<T> &src)
{
    hipError_t e = hipMalloc((void **)dst, src.size() * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
}

}  // namespace vq
