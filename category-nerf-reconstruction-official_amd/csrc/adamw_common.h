// AdamW element update shared by loss.hip (cnr_adamw_step) and render_loss.hip (cnr_adamw_epilogue).
// torch.optim.AdamW as configured at train.py:40,54-64: decoupled weight decay, bias corrections from the step count.
#pragma once
#include "cnr_common.h"

namespace cnr {
struct AdamArgs {
  float* p; const float* g; float* m; float* v; int64_t n;
  float lr, b1, b2, eps, wd, gunscale;
};
// step size and 1 / sqrt(bias correction 2) for optimiser step t (1-based)
__device__ __forceinline__ void adam_coefficients(const AdamArgs& a, int64_t t, float& step_size, float& inv_bc2_sqrt) {
  const double td = (double)t;
  step_size = (float)((double)a.lr / (1.0 - pow((double)a.b1, td)));
  inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)a.b2, td)));
}
// grid-stride over the flat buffer: block `blk` of `nblk` 256-thread blocks
__device__ __forceinline__ void adam_update(const AdamArgs& a, float step_size, float inv_bc2_sqrt, int blk, int nblk) {
  for (int64_t i = (int64_t)blk * 256 + threadIdx.x; i < a.n; i += (int64_t)nblk * 256) {
    const float gi = a.g[i] * a.gunscale;
    float pi = a.p[i] * (1.0f - a.lr * a.wd);
    const float mi = a.m[i] + (gi - a.m[i]) * (1.0f - a.b1);
    const float vi = a.v[i] * a.b2 + (1.0f - a.b2) * gi * gi;
    const float denom = sqrtf(vi) * inv_bc2_sqrt + a.eps;
    pi -= step_size * (mi / denom);
    a.p[i] = pi; a.m[i] = mi; a.v[i] = vi;
  }
}
}  // namespace cnr
