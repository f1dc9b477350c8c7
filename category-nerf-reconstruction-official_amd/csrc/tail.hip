// The tail of the fused trainer's step in ONE launch: latent backward, AdamW on every parameter, and the step
// epilogue -- three jobs that the stock sequence (cnr_latent_bwd -> cnr_adamw_step -> cnr_step_epilogue) runs as
// dependent launches.  What makes them independent here:
//   * parameters ping-pong like the step state: every kernel of step k READS theta_in and this launch WRITES
//     theta_out (AdamW out of place; exp_avg / exp_avg_sq are touched by AdamW only).  The latent backward reads
//     trunk weights, latent weights and codes while other blocks update exactly those -- out of place, no race;
//   * the thread that produces a latent-path gradient element applies AdamW to it straight away (one producer per
//     element); the trunk entries the latent path adds to are finished by the AdamW blocks themselves
//     (latent_trunk_term: n_obj multiply-adds from dbiasrows and zl);
//   * the epilogue blocks only read the render-loss partials, the depth pool and the CURRENT state.
// grid: [ do_latent ? NL latent blocks per class : 0 ] [ NA AdamW blocks over the flat (C, P) buffer ] [ C epilogue ].
// With do_latent = 0 (the gradient was completed by cnr_latent_bwd, e.g. before a multi-GPU all-reduce) the AdamW
// blocks cover every parameter and nothing else changes.
#include "adamw_common.h"
#include "latent_common.h"
#include "render_common.h"

namespace {
using namespace cnr;

struct TailArgs {
  const float* theta_in; float* theta_out; float* grad; float* m; float* v;
  FlatLayout lay; int64_t off_B; int C;
  const float* zl; const float* dbiasrows; float reg_scale;
  float lr, b1, b2, eps, wd;
  const int64_t* state_cur; int64_t* state_next; int64_t add_rows;
  const float* partials; int nb; float* losses; int32_t* flags;
  const float* depth; int64_t pool_rows; const int* perm; int R; float* max_bound;
  int do_latent, NL, NA;
};

__device__ __forceinline__ void adam_one(const TailArgs& a, int64_t e, float g, float step_size, float inv_bc2_sqrt) {
  float pi = a.theta_in[e] * (1.0f - a.lr * a.wd);
  const float mi = a.m[e] + (g - a.m[e]) * (1.0f - a.b1);
  const float vi = a.v[e] * a.b2 + (1.0f - a.b2) * g * g;
  const float denom = sqrtf(vi) * inv_bc2_sqrt + a.eps;
  pi -= step_size * (mi / denom);
  a.theta_out[e] = pi; a.m[e] = mi; a.v[e] = vi;
}

struct AdamSink {  // latent-path gradient element -> gradient buffer (kept for inspection) + AdamW, same thread
  const TailArgs& a; int64_t row0; float step_size, inv_bc2_sqrt;
  __device__ __forceinline__ void trunk_add(int, float) const {}
  __device__ __forceinline__ void latent_set(int64_t idx, float v) const {
    a.grad[row0 + idx] = v;
    adam_one(a, row0 + idx, v, step_size, inv_bc2_sqrt);
  }
};

__global__ __launch_bounds__(256) void tail_kernel(TailArgs a) {
  extern __shared__ float sm[];
  const int C = a.C, P = (int)a.lay.stride;
  const int nlat = a.do_latent ? a.NL * C : 0;
  int b = blockIdx.x;
  float step_size, inv_bc2_sqrt;
  {
    AdamArgs co{nullptr, nullptr, nullptr, nullptr, 0, a.lr, a.b1, a.b2, a.eps, a.wd, 1.0f};
    adam_coefficients(co, a.state_cur[2] + 1, step_size, inv_bc2_sqrt);
  }
  if (b < nlat) {  // ---- latent backward of class c + AdamW on what it produces
    const int c = b / a.NL, blk = b % a.NL;
    AdamSink sink{a, (int64_t)c * P, step_size, inv_bc2_sqrt};
    latent_bwd_block(a.theta_in + (int64_t)c * P, a.lay, a.zl + (int64_t)c * a.lay.n_obj * 128,
                     a.dbiasrows + (int64_t)c * a.lay.n_obj * 128, a.reg_scale, sm, sink, blk, a.NL, false);
    return;
  }
  b -= nlat;
  if (b < a.NA) {  // ---- AdamW over the flat buffer; with do_latent only the trunk and B ranges (+ the latent-path term)
    const int64_t n = (int64_t)C * P;
    for (int64_t e = (int64_t)b * 256 + threadIdx.x; e < n; e += (int64_t)a.NA * 256) {
      const int c = (int)(e / P), q = (int)(e - (int64_t)c * P);
      float g = a.grad[e];
      if (a.do_latent) {
        const bool is_trunk = q < CNR_TRUNK_PARAMS, is_B = q >= a.off_B && q < a.off_B + 63;
        if (!is_trunk && !is_B) continue;  // a latent block owns this element
        if (is_trunk) {
          const float t = latent_trunk_term(q, a.zl + (int64_t)c * a.lay.n_obj * 128,
                                            a.dbiasrows + (int64_t)c * a.lay.n_obj * 128, a.lay.n_obj);
          if (t != 0.0f) { g += t; a.grad[e] = g; }
        }
      }
      adam_one(a, e, g, step_size, inv_bc2_sqrt);
    }
    return;
  }
  b -= a.NA;
  // ---- epilogue of class b: loss values + flags, next slice's max depth, next step state (class 0)
  const int c = b, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* red = sm;
  const int64_t cursor = a.state_cur[0] + a.add_rows;
  if (wv == 0) cnr_rl::finish_class(a.partials, a.nb, a.losses, a.flags, C, c, lane);
  if (a.max_bound) {
    float mx = -INFINITY;
    const int64_t base = (int64_t)c * a.pool_rows + cursor;
    for (int r = threadIdx.x; r < a.R; r += 256)
      mx = fmaxf(mx, a.depth[a.perm ? (int64_t)c * a.pool_rows + a.perm[base + r] : base + r]);
    mx = wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    if (threadIdx.x == 0) a.max_bound[c] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  }
  if (c == 0 && threadIdx.x == 0) {
    a.state_next[0] = cursor; a.state_next[1] = a.state_cur[1] + 1; a.state_next[2] = a.state_cur[2] + 1;
  }
}
}  // namespace

extern "C" int cnr_step_tail(const float* theta_in, float* theta_out, float* grad, float* exp_avg, float* exp_avg_sq,
                             int64_t class_stride, int64_t off_B, int64_t off_latW, int64_t off_latb, int64_t off_shape,
                             int64_t off_tex, int L, int n_obj, int C, const float* zl, const float* dbiasrows,
                             float reg_scale, int do_latent, float lr, float beta1, float beta2, float eps,
                             float weight_decay, const int64_t* state_cur, int64_t* state_next, int64_t add_rows,
                             const void* rl_workspace, float* losses, int32_t* flags, const float* depth,
                             int64_t pool_rows, const int* perm, float* next_max_bound, int R, void* stream) {
  if (!theta_in || !theta_out || theta_in == theta_out || !grad || !exp_avg || !exp_avg_sq || class_stride <= 0 ||
      L <= 0 || n_obj <= 0 || C <= 0 || !state_cur || !state_next || state_cur == state_next || !rl_workspace ||
      !losses || !flags || R <= 0)
    return CNR_E_ARG;
  if (do_latent && (!zl || !dbiasrows)) return CNR_E_ARG;
  if (n_obj > 64) return CNR_E_SHAPE;
  if (next_max_bound && (!depth || pool_rows < R)) return CNR_E_ARG;
  TailArgs a{};
  a.theta_in = theta_in; a.theta_out = theta_out; a.grad = grad; a.m = exp_avg; a.v = exp_avg_sq;
  a.lay = FlatLayout{class_stride, off_latW, off_latb, off_shape, off_tex, L, n_obj};
  a.off_B = off_B; a.C = C; a.zl = zl; a.dbiasrows = dbiasrows; a.reg_scale = n_obj > 1 ? reg_scale : 0.0f;
  a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay;
  a.state_cur = state_cur; a.state_next = state_next; a.add_rows = add_rows;
  const int rpb = cnr_rl::rl_rays_per_block(C, R);
  a.partials = (const float*)rl_workspace; a.nb = (R + rpb - 1) / rpb; a.losses = losses; a.flags = flags;
  a.depth = depth; a.pool_rows = pool_rows; a.perm = perm; a.R = R; a.max_bound = next_max_bound;
  a.do_latent = do_latent ? 1 : 0;
  const int64_t nlat_out = (int64_t)4 * 32 * L + 128 + (int64_t)2 * n_obj * L;
  a.NL = (int)((nlat_out + 255) / 256);
  if (a.NL > 256) a.NL = 256;
  const int64_t n = (int64_t)C * class_stride;
  int64_t na = (n + 255) / 256;
  if (na > 2048) na = 2048;
  a.NA = (int)na;
  const unsigned grid = (unsigned)((a.do_latent ? a.NL * C : 0) + a.NA + C);
  const size_t lds = (size_t)(n_obj * 128 + 2 * n_obj + 8) * sizeof(float);
  hipLaunchKernelGGL(tail_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}
