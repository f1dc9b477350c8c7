// Shared helpers for the gfx950 kernels of libcnr_hip.so (MI355X only; no CUDA / dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#include "../../include/cnr_hip.h"

#define CNR_LAUNCH_CHECK()                                   \
  do {                                                       \
    hipError_t _e = hipGetLastError();                       \
    if (_e != hipSuccess) return (int)_e;                    \
  } while (0)

namespace cnr {
// A kernel that needs more than 64 KB of dynamic LDS must be told so once per DEVICE (hipFuncSetAttribute is per function and
// device).  One bit per device ordinal and kernel instantiation, set after the first successful call: idempotent, safe from
// several host threads, and a process that drives several GPUs gets the attribute on each of them.
struct DeviceOnce { std::atomic<uint64_t> mask{0}; };
inline int set_max_dynamic_lds(DeviceOnce& once, const void* fn, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  const uint64_t bit = 1ull << (dev & 63);
  if (dev < 64 && (once.mask.load(std::memory_order_acquire) & bit)) return 0;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  if (dev < 64) once.mask.fetch_or(bit, std::memory_order_release);
  return 0;
}
// true exactly once per device (for one-time device-side tables in static device storage)
inline bool first_on_device(DeviceOnce& once) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev >= 64) return true;
  const uint64_t bit = 1ull << dev;
  return (once.mask.fetch_or(bit, std::memory_order_acq_rel) & bit) == 0;
}
}  // namespace cnr

// fp32 trunk blob offsets (floats) -- see CNR_TRUNK_PARAMS in cnr_hip.h
namespace cnr {
constexpr int E1 = CNR_E1, E2 = CNR_E2, E = CNR_E, W = CNR_W;
constexpr int OFF_XYZ_W = 0, OFF_XYZ_B = 2784;
constexpr int OFF_S1_W = 2816, OFF_S1_B = 3840;
constexpr int OFF_S2_W = 3872, OFF_S2_B = 4896;
constexpr int OFF_CAT_W = 4928, OFF_CAT_B = 8736;
constexpr int OFF_ES_W = 8768, OFF_ES_B = 9792;
constexpr int OFF_SG_W = 9824, OFF_SG_B = 9856;
constexpr int OFF_VD_W = 9857, OFF_VD_B = 12225;
constexpr int OFF_T1_W = 12257, OFF_T1_B = 13281;
constexpr int OFF_R0_W = 13313, OFF_R0_B = 13825;
constexpr int OFF_R2_W = 13841, OFF_R2_B = 13889;
constexpr int TRUNK = CNR_TRUNK_PARAMS;
static_assert(OFF_R2_B + 3 == TRUNK, "trunk layout");

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// wave64 sum via DPP-friendly shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
}  // namespace cnr
