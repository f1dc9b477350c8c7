// a11-a13: occupancy activation + exclusive-cumprod termination + the four renders, and their
// backward (src/render_rays.py:3-7, 25-33, 46-50; src/loss.py:41-48).  Modular exact-fp32 kernels:
// one wavefront per ray, lane = sample (S <= 64: one pass; larger S: 64-sample chunks with a carried
// transmittance), inclusive product scan by wave shuffles.  The fused path does the same scan inside
// fused_mfma.hip without ever writing alpha/color to HBM.
//
//   occ_i = sigmoid(alpha_i) ; f_i = 1 - occ_i + 1e-10 ; T_i = prod_{j<i} f_j ; term_i = occ_i T_i
//   depth = sum term z ; var = sum term (z - depth)^2 (detached) ; rgb = sum term c ; opacity = sum term
// backward, with g_i = dD z_i + dC.c_i + dO (+ d_term_i) and Suf_i = sum_{k>i} term_k g_k:
//   d occ_i = T_i g_i - Suf_i / f_i          (same division ATen's cumprod backward performs)
//   d alpha_i = d occ_i occ_i (1 - occ_i) ; d c_i = term_i dC
#include "cnr_common.h"

// no fused multiply-add contraction in this file: composite.hip and render_loss.hip evaluate the same expressions and
// must round them the same way whatever the surrounding code looks like (the one-launch form is tested bitwise
// against the three-call form)
#pragma clang fp contract(off)

namespace {

// inclusive product scan across the wave (Hillis-Steele, 6 steps)
__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float p = __shfl_up(v, o, 64);
    if (lane >= o) v *= p;
  }
  return v;
}
// inclusive suffix sum across the wave
__device__ __forceinline__ float wave_incl_suffix_sum(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float p = __shfl_down(v, o, 64);
    if (lane + o < 64) v += p;
  }
  return v;
}

__device__ __forceinline__ float sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void composite_fwd_kernel(
    const float* __restrict__ alpha, const float* __restrict__ color, const float* __restrict__ z,
    float* __restrict__ term_out, float* __restrict__ depth, float* __restrict__ var,
    float* __restrict__ rgb, float* __restrict__ opacity, int64_t NR, int S, int in_is_occ) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (ray >= NR) return;
  const int64_t base = ray * S;
  float carry = 1.0f;  // transmittance entering this chunk
  float sd = 0.f, so = 0.f, s2 = 0.f, sr = 0.f, sg = 0.f, sb = 0.f;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool live = s < S;
    const float a = live ? alpha[base + s] : 0.0f;
    const float occ = live ? (in_is_occ ? a : sigmoid_exact(a)) : 0.0f;
    const float f = live ? (1.0f - occ + 1e-10f) : 1.0f;
    const float incl = wave_incl_prod(f, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.0f;
    const float T = carry * excl;
    const float term = occ * T;
    if (live) {
      const float zz = z ? z[base + s] : 0.0f;
      if (term_out) term_out[base + s] = term;
      sd += term * zz; s2 += term * zz * zz; so += term;
      if (color) {
        const float* cp = color + (base + s) * 3;
        sr += term * cp[0]; sg += term * cp[1]; sb += term * cp[2];
      }
    }
    carry *= __shfl(incl, 63, 64);
  }
  sd = cnr::wave_sum(sd); so = cnr::wave_sum(so);
  sr = cnr::wave_sum(sr); sg = cnr::wave_sum(sg); sb = cnr::wave_sum(sb);
  // var = sum term (z - depth)^2, second pass for the exact two-pass form the reference uses
  float sv = 0.f;
  carry = 1.0f;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool live = s < S;
    const float a = live ? alpha[base + s] : 0.0f;
    const float occ = live ? (in_is_occ ? a : sigmoid_exact(a)) : 0.0f;
    const float f = live ? (1.0f - occ + 1e-10f) : 1.0f;
    const float incl = wave_incl_prod(f, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.0f;
    if (live) { const float dz = (z ? z[base + s] : 0.0f) - sd; sv += occ * carry * excl * dz * dz; }
    carry *= __shfl(incl, 63, 64);
  }
  sv = cnr::wave_sum(sv);
  (void)s2;
  if (lane == 0) {
    if (depth) depth[ray] = sd;
    if (var) var[ray] = sv;
    if (opacity) opacity[ray] = so;
    if (rgb) { rgb[ray * 3 + 0] = sr; rgb[ray * 3 + 1] = sg; rgb[ray * 3 + 2] = sb; }
  }
}

__global__ __launch_bounds__(256) void composite_bwd_kernel(
    const float* __restrict__ alpha, const float* __restrict__ color, const float* __restrict__ z,
    const float* __restrict__ d_depth, const float* __restrict__ d_rgb, const float* __restrict__ d_opacity,
    const float* __restrict__ d_term, float* __restrict__ d_alpha, float* __restrict__ d_color,
    int64_t NR, int S, int in_is_occ) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (ray >= NR) return;
  const int64_t base = ray * S;
  const float dD = d_depth ? d_depth[ray] : 0.f, dO = d_opacity ? d_opacity[ray] : 0.f;
  const float dR = d_rgb ? d_rgb[ray * 3 + 0] : 0.f, dG = d_rgb ? d_rgb[ray * 3 + 1] : 0.f,
              dBl = d_rgb ? d_rgb[ray * 3 + 2] : 0.f;
  const int nchunk = (S + 63) / 64;
  // pass 1 (front to back): transmittance entering each chunk is needed back to front; S <= 64*8
  float carry_in[8];
  float carry = 1.0f;
  for (int ch = 0; ch < nchunk; ++ch) {
    carry_in[ch] = carry;
    const int s = ch * 64 + lane;
    const bool live = s < S;
    const float occ = live ? (in_is_occ ? alpha[base + s] : sigmoid_exact(alpha[base + s])) : 0.0f;
    const float f = live ? (1.0f - occ + 1e-10f) : 1.0f;
    const float incl = wave_incl_prod(f, lane);
    carry *= __shfl(incl, 63, 64);
  }
  // pass 2 (back to front) with the suffix sum carried across chunks
  float suf_carry = 0.0f;  // sum_{k in later chunks} term_k g_k
  for (int ch = nchunk - 1; ch >= 0; --ch) {
    const int s = ch * 64 + lane;
    const bool live = s < S;
    const float occ = live ? (in_is_occ ? alpha[base + s] : sigmoid_exact(alpha[base + s])) : 0.0f;
    const float f = live ? (1.0f - occ + 1e-10f) : 1.0f;
    const float incl = wave_incl_prod(f, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.0f;
    const float T = carry_in[ch] * excl;
    const float term = occ * T;
    float g = 0.0f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (live) {
      if (color) { const float* cp = color + (base + s) * 3; c0 = cp[0]; c1 = cp[1]; c2 = cp[2]; }
      g = dD * (z ? z[base + s] : 0.0f) + dR * c0 + dG * c1 + dBl * c2 + dO;
      if (d_term) g += d_term[base + s];
    }
    const float tg = term * g;
    const float incl_suf = wave_incl_suffix_sum(tg, lane);
    const float suf = (incl_suf - tg) + suf_carry;  // exclusive: k > i
    if (live) {
      const float docc = T * g - suf / f;
      d_alpha[base + s] = in_is_occ ? docc : docc * occ * (1.0f - occ);
      if (d_color) { float* dc = d_color + (base + s) * 3; dc[0] = term * dR; dc[1] = term * dG; dc[2] = term * dBl; }
    }
    suf_carry += __shfl(incl_suf, 0, 64);
  }
}
}  // namespace

extern "C" int cnr_composite_fwd(const float* alpha, const float* color, const float* z, float* term,
                                 float* depth, float* var, float* rgb, float* opacity, int64_t NR, int S,
                                 int in_is_occ, void* stream) {
  if (!alpha || NR <= 0 || S <= 0) return CNR_E_ARG;
  if ((rgb && !color) || ((depth || var) && !z)) return CNR_E_ARG;
  const int64_t blocks = (NR + 3) / 4;
  hipLaunchKernelGGL(composite_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     alpha, color, z, term, depth, var, rgb, opacity, NR, S, in_is_occ);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_composite_bwd(const float* alpha, const float* color, const float* z,
                                 const float* d_depth, const float* d_rgb, const float* d_opacity,
                                 const float* d_term, float* d_alpha, float* d_color, int64_t NR, int S,
                                 int in_is_occ, void* stream) {
  if (!alpha || !d_alpha || NR <= 0 || S <= 0) return CNR_E_ARG;
  if ((d_rgb && !color) || (d_depth && !z) || (d_color && !color)) return CNR_E_ARG;
  if (S > 512) return CNR_E_SHAPE;
  const int64_t blocks = (NR + 3) / 4;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     alpha, color, z, d_depth, d_rgb, d_opacity, d_term, d_alpha, d_color, NR, S, in_is_occ);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}
