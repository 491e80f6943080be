// Host-side interface between the kernel translation units (kernels_*.hip) and the model builder
// (engine_host.hip): one launcher per kernel family, looked up by its compile-time shape.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace vq {

struct EngineArgs;
// Enqueues one pass over a.N samples on `st` (zeroes the batch counter a.queue first where the kernel claims batches).
using LaunchFn = hipError_t (*)(const EngineArgs &a, bool matched, hipStream_t st);

enum { MODE_EUCLID = 0, MODE_OMEGA = 1, MODE_LOCAL = 2 };

// nullptr when the shape is not instantiated.  samples_per_batch: samples one claimed batch covers.
LaunchFn streamed_launcher(int mode, int R1, int R2, int *samples_per_batch);    // kernels_engine.hip: k_engine
LaunchFn omega_finish_launcher(int R);                                            // k_omega_pipe FINISH, or nullptr
LaunchFn streamed_finish_launcher(int mode, int R1, int R2);                      // k_engine FINISH (the finish pass), or nullptr
LaunchFn resident_launcher(int R, int *samples_per_batch);                       // kernels_resident.hip: k_resident
LaunchFn resident_pipe_launcher(int R, bool diag);                               // kernels_resident.hip: k_resident_pipe
LaunchFn resident_cut_launcher(int R);                                           // k_resident_pipe for rows shorter than R
LaunchFn resident_finish_launcher(int R);   // k_resident_pipe FINISH: the finish pass over the marked samples (bf16 ranking only)
// k_resident_pipe with rows staged through LDS (few prototypes); staging_bytes: LDS it needs behind the image
LaunchFn resident_staged_launcher(int R, size_t *staging_bytes);
LaunchFn omega_pipe_launcher(int R);                                             // kernels_omega.hip: k_omega_pipe
LaunchFn lowrank_launcher(int R1);                                               // kernels_omega.hip: k_lowrank_pipe
// kernels_direct.hip: k_direct (at most 8 prototypes, direct form on the vector ALUs); R in {32, 64, 128, 256}, KP in {4, 8}
LaunchFn direct_launcher(int R, int KP, int *samples_per_batch);
LaunchFn long_launcher(int tiles);                                               // kernels_long.hip: k_long (rows above 256 values)
LaunchFn local_long_launcher(int tiles);    // k_local_long<tiles>: LGMLVQ over rows above 256 values
LaunchFn omega_long_launcher(int R2);                                            // kernels_long.hip: k_omega_long
// kernels_local.hip: k_local_pj (LGMLVQ, projections on the bf16 matrix cores); R1 in {16, 32, 64, 128, 256}
LaunchFn local_pj_launcher(int R1, int *samples_per_batch);

// batch counters of one launch (BlockQueue, mfma_device.h): QUEUE_GROUPS counters, 4 KiB apart
constexpr uint32_t QUEUE_GROUPS = 16;
constexpr uint32_t QUEUE_GROUP_STRIDE = 1024;                                    // in u32
constexpr size_t QUEUE_BYTES = (size_t)QUEUE_GROUPS * QUEUE_GROUP_STRIDE * sizeof(uint32_t);
// a launch's batch counters start at zero: by this memset, or by one the caller has issued over a buffer that holds them
// (engine_assign: counters of both passes and the marks in one piece, one memset instead of three)
hipError_t zero_queue(const EngineArgs &a, hipStream_t st);      // (engine_host.hip)

constexpr int RES_BYTES = 156 * 1024;       // LDS one resident workgroup may fill with images
constexpr float PAD_SCORE = 3.0e38f;        // accumulator init of pad rows: finite, above every real score

}  // namespace vq
