// a11-a15 in ONE launch for the fused trainer: occupancy activation, exclusive-cumprod termination, the four
// renders, the three masked-L1 losses with their normalisers, the loss gradients w.r.t. the renders and the
// composite backward down to d sigma / d colour per sample (src/render_rays.py:3-7,25-33,46-95; src/loss.py:18-74).
//
// What makes one launch possible: the gradient of each loss term w.r.t. one ray's renders depends on that ray's
// own values and on the mask COUNTS only (masked means: 1 / count), and the counts are known from the sampled
// labels before any rendering.  So nothing between the forward and the backward composite needs a grid-wide
// reduction; the loss VALUES (logging, the explode flag) are per-block partial sums that a second, one-block-per-
// class kernel (cnr_render_loss_finish) adds up in a fixed order -- bitwise reproducible, no float atomics.  It is a
// separate entry point so that the trainer can put it on a side stream, off the critical path.  (A "last block
// finishes" ticket inside the main kernel was measured 3x slower than three separate launches: the agent-scope
// release fence it needs writes the 8 XCD L2s back once per wave.)
//
// Same arithmetic, expression for expression, as composite.hip + loss.hip (the modular path): d sigma / d colour
// are bit-identical to cnr_composite_fwd -> cnr_loss_fwd_bwd -> cnr_composite_bwd, the loss values agree to
// summation order.  Block = 16 waves, one ray per wave at a time.
#include "adamw_common.h"
#include "render_common.h"

// no fused multiply-add contraction in this file: composite.hip and render_loss.hip evaluate the same expressions and
// must round them the same way whatever the surrounding code looks like (the one-launch form is tested bitwise
// against the three-call form)
#pragma clang fp contract(off)

namespace {
using namespace cnr_rl;
// grid (ceil(R / rays_per_block), C)
__global__ __launch_bounds__(RL_THREADS) void render_loss_kernel(
    const float* __restrict__ sigmas, const float* __restrict__ colors, const float* __restrict__ z,
    const float* __restrict__ gt_depth, const float* __restrict__ gt_rgb, const uint8_t* __restrict__ labels,
    const uint8_t* __restrict__ depth_mask, float color_scaling, float opacity_scaling, float grad_scale,
    float* __restrict__ d_sigmas, float* __restrict__ d_colors, float* __restrict__ depth_out,
    float* __restrict__ var_out, float* __restrict__ rgb_out, float* __restrict__ opacity_out, int C, int R, int S,
    float* __restrict__ partials, int rays_per_block, const float* __restrict__ counts_tab,
    const int64_t* __restrict__ d_state) {
  __shared__ float cnt[3 * RL_WAVES];
  __shared__ float wsum[3 * RL_WAVES];
  const int c = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // ---- mask counts of every class: the empty-mask rule of render_rays.py:67-72 couples the classes --------
  bool empty_d = false, empty_c = false, empty_o = false;
  float nd = 0.f, nc = 0.f, no = 0.f;
  if (counts_tab) {  // per-epoch table (cnr_slice_maskcounts), see cnr_field_fwd_render
    const float* t = counts_tab + (size_t)(d_state ? d_state[0] / R : 0) * (size_t)(C + 1) * 4;
    nd = t[c * 4 + 0]; nc = t[c * 4 + 1]; no = t[c * 4 + 2];
    empty_d = t[C * 4 + 0] != 0.f; empty_c = t[C * 4 + 1] != 0.f; empty_o = t[C * 4 + 2] != 0.f;
  }
  for (int cc = 0; cc < (counts_tab ? 0 : C); ++cc) {
    float a = 0.f, b = 0.f, d = 0.f;
    for (int r = threadIdx.x; r < R; r += RL_THREADS) {
      const uint8_t lab = labels[(size_t)cc * R + r];
      const bool mo = lab != 0, ms = lab != 2, md = depth_mask[(size_t)cc * R + r] != 0;
      a += (md && mo) ? 1.f : 0.f; b += mo ? 1.f : 0.f; d += ms ? 1.f : 0.f;
    }
    a = cnr::wave_sum(a); b = cnr::wave_sum(b); d = cnr::wave_sum(d);
    __syncthreads();
    if (lane == 0) { cnt[wv] = a; cnt[RL_WAVES + wv] = b; cnt[2 * RL_WAVES + wv] = d; }
    __syncthreads();
    a = b = d = 0.f;
#pragma unroll
    for (int w = 0; w < RL_WAVES; ++w) { a += cnt[w]; b += cnt[RL_WAVES + w]; d += cnt[2 * RL_WAVES + w]; }  // counts: exact
    empty_d |= (a == 0.f); empty_c |= (b == 0.f); empty_o |= (d == 0.f);
    if (cc == c) { nd = a; nc = b; no = d; }
  }
  const float wd = empty_d ? 0.f : 1.0f / (nd + 1e-10f);
  const float wc = empty_c ? 0.f : 1.0f / (nc + 1e-10f);
  const float wo = empty_o ? 0.f : 1.0f / (no + 1e-10f);

  const int nchunk = (S + 63) / 64;
  float ld = 0.f, lc = 0.f, lo = 0.f;  // this wave's loss partials (lane 0)
  for (int rr = wv; rr < rays_per_block; rr += RL_WAVES) {
    const int r = blockIdx.x * rays_per_block + rr;
    if (r >= R) break;  // wave-uniform
    const size_t ray = (size_t)c * R + r;
    const size_t base = ray * S;
    if (nchunk == 1) {
      // ---- S <= 64: the whole ray sits in one register per lane; sigmoid, scan and loads happen once (render_common.h) --
      const RayOut q = ray_small(sigmas, colors, z, base, S, lane, gt_depth[ray], gt_rgb[ray * 3 + 0], gt_rgb[ray * 3 + 1],
                                 gt_rgb[ray * 3 + 2], labels[ray], depth_mask[ray], wd, wc, wo, color_scaling, opacity_scaling,
                                 grad_scale);
      if (lane == 0) {
        if (depth_out) depth_out[ray] = q.sd;
        if (var_out) var_out[ray] = q.sv;
        if (opacity_out) opacity_out[ray] = q.so;
        if (rgb_out) { rgb_out[ray * 3 + 0] = q.sr; rgb_out[ray * 3 + 1] = q.sg; rgb_out[ray * 3 + 2] = q.sb; }
      }
      ld += q.ld; lc += q.lc; lo += q.lo;
      if (q.live) {
        d_sigmas[base + lane] = q.dsig;
        float* dc = d_colors + (base + lane) * 3;
        dc[0] = q.dc0; dc[1] = q.dc1; dc[2] = q.dc2;
      }
      continue;
    }
    // ---- forward composite ----------------------------------------------------------------------------------
    float carry_in[RL_MAX_CHUNKS];
    float carry = 1.0f;
    float sd = 0.f, so = 0.f, sr = 0.f, sg = 0.f, sb = 0.f;
    for (int ch = 0; ch < nchunk; ++ch) {
      carry_in[ch] = carry;
      const int s = ch * 64 + lane;
      const bool live = s < S;
      const float occ = live ? sigmoid_exact(sigmas[base + s]) : 0.0f;
      const float f = live ? (1.0f - occ + 1e-10f) : 1.0f;
      const float incl = incl_prod(f, lane);
      float excl = __shfl_up(incl, 1, 64);
      if (lane == 0) excl = 1.0f;
      const float term = occ * (carry * excl);
      if (live) {
        const float zz = z[base + s];
        const float* cp = colors + (base + s) * 3;
        sd += term * zz; so += term;
        sr += term * cp[0]; sg += term * cp[1]; sb += term * cp[2];
      }
      carry *= __shfl(incl, 63, 64);
    }
    sd = cnr::wave_sum(sd); so = cnr::wave_sum(so);
    sr = cnr::wave_sum(sr); sg = cnr::wave_sum(sg); sb = cnr::wave_sum(sb);
    float sv = 0.f;  // var = sum term (z - depth)^2 (two-pass form, as the reference)
    for (int ch = 0; ch < nchunk; ++ch) {
      const int s = ch * 64 + lane;
      const bool live = s < S;
      const float occ = live ? sigmoid_exact(sigmas[base + s]) : 0.0f;
      const float f = live ? (1.0f - occ + 1e-10f) : 1.0f;
      const float incl = incl_prod(f, lane);
      float excl = __shfl_up(incl, 1, 64);
      if (lane == 0) excl = 1.0f;
      if (live) { const float dz = z[base + s] - sd; sv += occ * carry_in[ch] * excl * dz * dz; }
    }
    sv = cnr::wave_sum(sv);
    if (lane == 0) {
      if (depth_out) depth_out[ray] = sd;
      if (var_out) var_out[ray] = sv;
      if (opacity_out) opacity_out[ray] = so;
      if (rgb_out) { rgb_out[ray * 3 + 0] = sr; rgb_out[ray * 3 + 1] = sg; rgb_out[ray * 3 + 2] = sb; }
    }
    // ---- losses of this ray and their gradients w.r.t. its renders (loss.hip, same expressions) -----------------------
    const uint8_t lab = labels[ray];
    const bool mo = lab != 0, ms = lab != 2, md = (depth_mask[ray] != 0) && mo;
    const float fd = md ? 1.f : 0.f, fo = mo ? 1.f : 0.f, fs = ms ? 1.f : 0.f;
    const float rd = sd - gt_depth[ray];
    const float info = 1.0f / (sqrtf(sv) + 1e-4f);
    const float rc0 = sr - gt_rgb[ray * 3 + 0], rc1 = sg - gt_rgb[ray * 3 + 1], rc2 = sb - gt_rgb[ray * 3 + 2];
    const float ro = so - fo;
    ld += fabsf(rd) * fd * info;
    lc += (fabsf(rc0) + fabsf(rc1) + fabsf(rc2)) * fo;
    lo += fabsf(ro) * fs;
    const float dD = grad_scale * sgn(rd) * fd * info * wd;
    const float dR = grad_scale * color_scaling * sgn(rc0) * fo * wc;
    const float dG = grad_scale * color_scaling * sgn(rc1) * fo * wc;
    const float dBl = grad_scale * color_scaling * sgn(rc2) * fo * wc;
    const float dO = grad_scale * opacity_scaling * sgn(ro) * fs * wo;
    // ---- backward composite (back to front, suffix sum carried across chunks) ------------------------------------
    float suf_carry = 0.0f;
    for (int ch = nchunk - 1; ch >= 0; --ch) {
      const int s = ch * 64 + lane;
      const bool live = s < S;
      const float occ = live ? sigmoid_exact(sigmas[base + s]) : 0.0f;
      const float f = live ? (1.0f - occ + 1e-10f) : 1.0f;
      const float incl = incl_prod(f, lane);
      float excl = __shfl_up(incl, 1, 64);
      if (lane == 0) excl = 1.0f;
      const float T = carry_in[ch] * excl;
      const float term = occ * T;
      float g = 0.0f;
      if (live) {
        const float* cp = colors + (base + s) * 3;
        g = dD * z[base + s] + dR * cp[0] + dG * cp[1] + dBl * cp[2] + dO;
      }
      const float tg = term * g;
      const float incl_suf = incl_suffix_sum(tg, lane);
      const float suf = (incl_suf - tg) + suf_carry;
      if (live) {
        const float docc = T * g - suf / f;
        d_sigmas[base + s] = docc * occ * (1.0f - occ);
        float* dc = d_colors + (base + s) * 3;
        dc[0] = term * dR; dc[1] = term * dG; dc[2] = term * dBl;
      }
      suf_carry += __shfl(incl_suf, 0, 64);
    }
  }
  // ---- loss values: block partials, summed in block order by the last block to finish --------------------------------
  if (lane == 0) { wsum[wv] = ld; wsum[RL_WAVES + wv] = lc; wsum[2 * RL_WAVES + wv] = lo; }
  __syncthreads();
  const int nb = gridDim.x;
  if (threadIdx.x < 3) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < RL_WAVES; ++w) t += wsum[threadIdx.x * RL_WAVES + w];
    partials[((size_t)c * nb + blockIdx.x) * 3 + threadIdx.x] = t;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // the normalisers and the empty-mask bits, for the finish kernel
    float* hdr = partials + (size_t)C * nb * 3 + (size_t)c * 4;
    hdr[0] = wd; hdr[1] = wc; hdr[2] = wo;
    hdr[3] = (float)((empty_d ? 2 : 0) | (empty_c ? 4 : 0) | (empty_o ? 8 : 0));
  }
}

__global__ __launch_bounds__(64) void render_loss_finish_kernel(const float* __restrict__ partials, int nb,
                                                                float* __restrict__ losses, int32_t* __restrict__ flags,
                                                                int C) {
  finish_class(partials, nb, losses, flags, C, blockIdx.x, threadIdx.x);
}
// last node of the fused trainer's step, ONE 256-thread block (so that "every reader of the step state has read it"
// is a plain __syncthreads): the loss values, the NEXT step's slice maximum of the depth pool
// (cnr_sample_maxdepth, scene_cateogries.py:486 -- the next step then starts with the sampler straight away), and
// the device-side step state advanced (cnr_step_advance)
__global__ __launch_bounds__(256) void step_epilogue_kernel(const float* __restrict__ partials, int nb,
                                                            float* __restrict__ losses, int32_t* __restrict__ flags,
                                                            int C, int64_t* __restrict__ d_state, int64_t add_rows,
                                                            const float* __restrict__ depth, int64_t pool_rows,
                                                            const int* __restrict__ perm, int R,
                                                            float* __restrict__ max_bound) {
  __shared__ float sm[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int c = wv; c < C; c += 4) finish_class(partials, nb, losses, flags, C, c, lane);
  const int64_t cursor = d_state[0] + add_rows;
  if (max_bound) {
    for (int c = 0; c < C; ++c) {
      float m = -INFINITY;
      const int64_t base = (int64_t)c * pool_rows + cursor;
      for (int r = threadIdx.x; r < R; r += 256)
        m = fmaxf(m, depth[perm ? (int64_t)c * pool_rows + perm[base + r] : base + r]);
      m = cnr::wave_max(m);
      __syncthreads();
      if (lane == 0) sm[wv] = m;
      __syncthreads();
      if (threadIdx.x == 0) max_bound[c] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) { d_state[0] = cursor; d_state[1] += 1; d_state[2] += 1; }
}

// The step's last launch when the step state ping-pongs between two buffers: AdamW on the flat parameter buffer
// (the first blocks, reading the optimiser step from the CURRENT state) beside the epilogue work (one block per class:
// loss values, next slice's max depth; class 0 also writes NEXT state = current + (add_rows, 1, 1) to the other buffer).
// Nobody writes the current state during the launch, so the two jobs need no ordering between them.
__global__ __launch_bounds__(256) void adamw_epilogue_kernel(cnr::AdamArgs a, const int64_t* __restrict__ state_cur,
                                                             int64_t* __restrict__ state_next, int64_t add_rows,
                                                             const float* __restrict__ partials, int nb,
                                                             float* __restrict__ losses, int32_t* __restrict__ flags,
                                                             int C, const float* __restrict__ depth, int64_t pool_rows,
                                                             const int* __restrict__ perm, int R,
                                                             float* __restrict__ max_bound) {
  const int nadam = gridDim.x - C;   // AdamW blocks first, then one epilogue block per class
  if ((int)blockIdx.x < nadam) {
    float step_size, inv_bc2_sqrt;
    cnr::adam_coefficients(a, state_cur[2] + 1, step_size, inv_bc2_sqrt);
    cnr::adam_update(a, step_size, inv_bc2_sqrt, blockIdx.x, nadam);
    return;
  }
  __shared__ float sm[4];
  const int c = blockIdx.x - nadam, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t cursor = state_cur[0] + add_rows;
  // the loss values first: their loads are in flight while the max-depth chain (state -> perm -> depth) runs
  if (wv == 0) finish_class(partials, nb, losses, flags, C, c, lane);
  if (max_bound) {
    float m = -INFINITY;
    const int64_t base = (int64_t)c * pool_rows + cursor;
    for (int r = threadIdx.x; r < R; r += 256)
      m = fmaxf(m, depth[perm ? (int64_t)c * pool_rows + perm[base + r] : base + r]);
    m = cnr::wave_max(m);
    if (lane == 0) sm[wv] = m;
    __syncthreads();
    if (threadIdx.x == 0) max_bound[c] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  }
  if (c == 0 && threadIdx.x == 0) {
    state_next[0] = cursor; state_next[1] = state_cur[1] + 1; state_next[2] = state_cur[2] + 1;
  }
}
}  // namespace

extern "C" int64_t cnr_render_loss_workspace_bytes(int C, int R) {
  if (C <= 0 || R <= 0) return 0;
  const int rpb = rl_rays_per_block(C, R);
  const int64_t nb = (R + rpb - 1) / rpb;
  return ((int64_t)C * nb * 3 + (int64_t)C * 4) * (int64_t)sizeof(float);
}

extern "C" int cnr_render_loss(const float* sigmas, const float* colors, const float* z, const float* gt_depth,
                               const float* gt_rgb, const uint8_t* labels, const uint8_t* depth_mask,
                               float color_scaling, float opacity_scaling, float grad_scale, float* d_sigmas,
                               float* d_colors, float* depth, float* var, float* rgb, float* opacity, int C, int R,
                               int S, void* workspace, int64_t workspace_bytes, const float* counts_tab,
                               const int64_t* d_state, void* stream) {
  if (!sigmas || !colors || !z || !gt_depth || !gt_rgb || !labels || !depth_mask || !d_sigmas || !d_colors ||
      !workspace || C <= 0 || R <= 0 || S <= 0)
    return CNR_E_ARG;
  if (S > 64 * RL_MAX_CHUNKS) return CNR_E_SHAPE;
  if (workspace_bytes < cnr_render_loss_workspace_bytes(C, R)) return CNR_E_ARG;
  const int rpb = rl_rays_per_block(C, R);
  const int nb = (R + rpb - 1) / rpb;
  hipLaunchKernelGGL(render_loss_kernel, dim3(nb, C), dim3(RL_THREADS), 0, (hipStream_t)stream, sigmas, colors, z,
                     gt_depth, gt_rgb, labels, depth_mask, color_scaling, opacity_scaling, grad_scale, d_sigmas,
                     d_colors, depth, var, rgb, opacity, C, R, S, (float*)workspace, rpb, counts_tab, d_state);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_render_loss_finish(const void* workspace, float* losses, int32_t* flags, int C, int R,
                                      int rl_blocks, void* stream) {
  if (!workspace || !losses || !flags || C <= 0 || R <= 0) return CNR_E_ARG;
  const int rpb = rl_rays_per_block(C, R);
  const int nb = rl_blocks > 0 ? rl_blocks : (R + rpb - 1) / rpb;
  hipLaunchKernelGGL(render_loss_finish_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, (const float*)workspace,
                     nb, losses, flags, C);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_step_epilogue(int64_t* d_state, int64_t add_rows, const void* workspace, float* losses,
                                 int32_t* flags, const float* depth, int64_t pool_rows, const int* perm,
                                 float* next_max_bound, int C, int R, void* stream) {
  if (!d_state || !workspace || !losses || !flags || C <= 0 || R <= 0) return CNR_E_ARG;
  if (next_max_bound && (!depth || pool_rows < R)) return CNR_E_ARG;
  const int rpb = rl_rays_per_block(C, R);
  const int nb = (R + rpb - 1) / rpb;
  hipLaunchKernelGGL(step_epilogue_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nb,
                     losses, flags, C, d_state, add_rows, depth, pool_rows, perm, R, next_max_bound);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_adamw_epilogue(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                  float lr, float beta1, float beta2, float eps, float weight_decay,
                                  float grad_unscale, const int64_t* state_cur, int64_t* state_next, int64_t add_rows,
                                  const void* workspace, float* losses, int32_t* flags, const float* depth,
                                  int64_t pool_rows, const int* perm, float* next_max_bound, int C, int R,
                                  void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || !state_cur || !state_next || state_cur == state_next ||
      !workspace || !losses || !flags || C <= 0 || R <= 0)
    return CNR_E_ARG;
  if (next_max_bound && (!depth || pool_rows < R)) return CNR_E_ARG;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const int rpb = rl_rays_per_block(C, R);
  const int nb = (R + rpb - 1) / rpb;
  cnr::AdamArgs a{param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, grad_unscale};
  hipLaunchKernelGGL(adamw_epilogue_kernel, dim3((unsigned)blocks + C), dim3(256), 0, (hipStream_t)stream, a, state_cur,
                     state_next, add_rows, (const float*)workspace, nb, losses, flags, C, depth, pool_rows, perm, R,
                     next_max_bound);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}
