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
// The value of lane ^ j (j a constant after unrolling; lane = the lane's index in its wave).  Distances inside a row of 16 lanes
// are DPP moves on the vector unit -- quad permutes for 1 and 2, a row rotate for 8, a left and a right row shift and a select
// for 4 --; only 16 and 32 go through ds_bpermute (the LDS crossbar, ~100 cycles of latency each).
__device__ __forceinline__ float xor_lane(float v, int j, int lane) {
  const int x = __builtin_bit_cast(int, v);
  switch (j) {
    case 1: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    case 2: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    case 4: {
      const int up = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xf, 0xf, false);   // row_shl:4: lane i <- i + 4
      const int dn = __builtin_amdgcn_update_dpp(x, x, 0x114, 0xf, 0xf, false);   // row_shr:4: lane i <- i - 4
      return __builtin_bit_cast(float, (lane & 4) ? dn : up);
    }
    case 8: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false));   // row_ror:8
    default: return __shfl_xor(v, j, 64);
  }
}
// butterfly reductions over the wave, every lane gets the result; the same order of operations as a __shfl_xor butterfly
// (32, 16, .., 1: bit-identical sums), four of the six steps without the LDS crossbar
__device__ __forceinline__ float wave_sum(float v) {
  const int lane = (int)(threadIdx.x & 63);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += xor_lane(v, o, lane);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  const int lane = (int)(threadIdx.x & 63);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, xor_lane(v, o, lane));
  return v;
}
}  // namespace cnr
