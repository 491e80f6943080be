// MFMA engine -- placeholder until the tile kernels land: reports "not eligible" so that
// VQHIP_KERNEL_AUTO uses the scalar kernels and VQHIP_KERNEL_MFMA fails loudly.
#include "vq_common.h"

namespace vq {

struct EngineModel {};

EngineModel *engine_build(const std::vector<float> &, uint32_t, uint32_t, const std::vector<float> &, int, uint32_t,
                          const std::vector<uint32_t> &, std::string &why, hipError_t &herr)
{
    herr = hipSuccess;
    why = "engine not implemented yet";
    return nullptr;
}
void engine_free(EngineModel *m) { delete m; }
const char *engine_name(const EngineModel *) { return "mfma"; }
hipError_t engine_assign(const EngineModel *, const float *, uint64_t, const uint32_t *, Outputs, hipStream_t)
{
    return hipErrorNotSupported;
}

}  // namespace vq
