// Counter-based synthetic data.  Host and device produce identical bits.
#include <cmath>

#include "vq_common.h"

namespace vq {

__global__ void __launch_bounds__(256) k_synth_fill(float *__restrict__ out, uint64_t count, uint64_t first,
                                                    uint64_t key, float scale, const float *__restrict__ base,
                                                    uint64_t base_len)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint64_t g = first + i;
        const float b = base ? base[g % base_len] : 0.0f;
        out[i] = fmaf(scale, synth_value(key, g), b);
    }
}

hipError_t launch_synth_fill(float *dout, uint64_t count, uint64_t first, uint64_t seed, uint32_t stream_id,
                             float scale, const float *dbase, uint64_t base_len, hipStream_t s)
{
    if (count == 0) return hipSuccess;
    uint64_t blocks = (count + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_synth_fill, dim3((unsigned)blocks), dim3(256), 0, s, dout, count, first,
                       synth_key(seed, stream_id), scale, dbase, base_len);
    return hipGetLastError();
}

}  // namespace vq

extern "C" int vqhip_synth_fill_host(float *out, uint64_t count, uint64_t first, uint64_t seed,
                                     uint32_t stream_id, float scale, const float *base, uint64_t base_len)
{
    if ((!out && count) || (base && base_len == 0)) return VQHIP_ERR_INVALID;
    const uint64_t key = vq::synth_key(seed, stream_id);
    for (uint64_t i = 0; i < count; ++i) {
        const uint64_t g = first + i;
        const float b = base ? base[g % base_len] : 0.0f;
        out[i] = fmaf(scale, vq::synth_value(key, g), b);
    }
    return VQHIP_OK;
}
