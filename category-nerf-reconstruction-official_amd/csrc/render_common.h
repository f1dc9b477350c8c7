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

// ---- one ray of S <= 64 samples on one wave (lane = sample): composite, the ray's loss terms and their gradient down to
// d sigma / d colour of this lane's sample.  The expressions of composite.hip + loss.hip, evaluated without fused multiply-add
// contraction wherever this is inlined (cnr_render_loss is tested bitwise against the three-call form; the background
// backward, bg_fused.hip, calls the same function so that its in-kernel composite is cnr_render_loss's bit for bit).
__device__ __forceinline__ float incl_prod(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float p = __shfl_up(v, o, 64);
    if (lane >= o) v *= p;
  }
  return v;
}
__device__ __forceinline__ float incl_suffix_sum(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float p = __shfl_down(v, o, 64);
    if (lane + o < 64) v += p;
  }
  return v;
}
__device__ __forceinline__ float sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

struct RayOut {
  float sd, sv, so, sr, sg, sb;      // depth, variance, opacity, rgb of the ray (all lanes)
  float ld, lc, lo;                  // the ray's un-normalised loss terms (all lanes)
  float dsig, dc0, dc1, dc2;         // this lane's sample: d sigma, d colour (x grad_scale); 0 for lanes >= S
  bool live;
};
__device__ __forceinline__ RayOut ray_small(const float* __restrict__ sigmas, const float* __restrict__ colors,
                                            const float* __restrict__ z, size_t base, int S, int lane, float gtd, float g0,
                                            float g1, float g2, uint8_t lab, uint8_t dmask, float wd, float wc, float wo,
                                            float color_scaling, float opacity_scaling, float grad_scale) {
#pragma clang fp contract(off)
  RayOut r;
  const int s = lane;
  const bool live = s < S;
  const float occ = live ? sigmoid_exact(sigmas[base + s]) : 0.0f;
  const float f = live ? (1.0f - occ + 1e-10f) : 1.0f;
  const float incl = incl_prod(f, lane);
  float T = __shfl_up(incl, 1, 64);
  if (lane == 0) T = 1.0f;
  const float term = occ * (1.0f * T);
  float zz = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
  if (live) { zz = z[base + s]; const float* cp = colors + (base + s) * 3; c0 = cp[0]; c1 = cp[1]; c2 = cp[2]; }
  float sd = 0.f, so = 0.f, sr = 0.f, sg = 0.f, sb = 0.f;
  if (live) { sd += term * zz; so += term; sr += term * c0; sg += term * c1; sb += term * c2; }
  sd = cnr::wave_sum(sd); so = cnr::wave_sum(so);
  sr = cnr::wave_sum(sr); sg = cnr::wave_sum(sg); sb = cnr::wave_sum(sb);
  float sv = 0.f;
  if (live) { const float dz = zz - sd; sv += occ * 1.0f * T * dz * dz; }
  sv = cnr::wave_sum(sv);
  r.sd = sd; r.sv = sv; r.so = so; r.sr = sr; r.sg = sg; r.sb = sb;
  const bool mo = lab != 0, ms = lab != 2, md = (dmask != 0) && mo;
  const float fd = md ? 1.f : 0.f, fo = mo ? 1.f : 0.f, fs = ms ? 1.f : 0.f;
  const float rd = sd - gtd;
  const float info = 1.0f / (sqrtf(sv) + 1e-4f);
  const float rc0 = sr - g0, rc1 = sg - g1, rc2 = sb - g2;
  const float ro = so - fo;
  r.ld = fabsf(rd) * fd * info;
  r.lc = (fabsf(rc0) + fabsf(rc1) + fabsf(rc2)) * fo;
  r.lo = fabsf(ro) * fs;
  const float dD = grad_scale * sgn(rd) * fd * info * wd;
  const float dR = grad_scale * color_scaling * sgn(rc0) * fo * wc;
  const float dG = grad_scale * color_scaling * sgn(rc1) * fo * wc;
  const float dBl = grad_scale * color_scaling * sgn(rc2) * fo * wc;
  const float dO = grad_scale * opacity_scaling * sgn(ro) * fs * wo;
  const float g = live ? dD * zz + dR * c0 + dG * c1 + dBl * c2 + dO : 0.0f;
  const float tg = term * g;
  const float incl_suf = incl_suffix_sum(tg, lane);
  const float suf = (incl_suf - tg) + 0.0f;
  r.live = live;
  r.dsig = 0.0f; r.dc0 = 0.0f; r.dc1 = 0.0f; r.dc2 = 0.0f;
  if (live) {
    const float docc = T * g - suf / f;
    r.dsig = docc * occ * (1.0f - occ);
    r.dc0 = term * dR; r.dc1 = term * dG; r.dc2 = term * dBl;
  }
  return r;
}
}  // namespace cnr_rl
