// Shared by render_loss.hip and tail.hip: the block shape of cnr_render_loss and the reduction of its per-block loss
// partials into loss values and flags.
#pragma once
#include "cnr_common.h"

namespace cnr_rl {
constexpr int RL_WAVES = 16, RL_THREADS = RL_WAVES * 64, RL_MAX_CHUNKS = 8;
// rays per block = 16 waves x rays per wave: one ray per wave (all rays in flight at once) until the grid is
// several blocks per CU, then more rays per wave so that the per-block mask count (2 bytes x C x R) stays small
__host__ __device__ inline int rl_rays_per_block(int C, int R) {
  int rpw = (int)(((int64_t)C * R + 8191) / 8192);
  if (rpw < 1) rpw = 1;
  if (rpw > 8) rpw = 8;
  return RL_WAVES * rpw;
}

// one wave per class: partials in a fixed order, flags
__device__ __forceinline__ void finish_class(const float* __restrict__ partials, int nb, float* __restrict__ losses,
                                             int32_t* __restrict__ flags, int C, int c, int lane) {
  const float* hdr = partials + (size_t)C * nb * 3 + (size_t)c * 4;
  const float wd = hdr[0], wc = hdr[1], wo = hdr[2];
  float sd = 0.f, sc = 0.f, so = 0.f;
  const float* pp = partials + (size_t)c * nb * 3;
  for (int b = lane; b < nb; b += 64) { sd += pp[b * 3 + 0]; sc += pp[b * 3 + 1]; so += pp[b * 3 + 2]; }
  sd = cnr::wave_sum(sd) * wd; sc = cnr::wave_sum(sc) * wc; so = cnr::wave_sum(so) * wo;
  if (lane == 0) {
    losses[0 * C + c] = sd; losses[1 * C + c] = sc; losses[2 * C + c] = so;
    int32_t fl = (int32_t)hdr[3];
    if (sd > 100000.f || sc > 100000.f || so > 100000.f) fl |= 1;
    flags[c] = fl;
  }
}
}  // namespace cnr_rl
