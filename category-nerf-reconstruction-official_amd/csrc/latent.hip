// a7 + the four latent-conditioning layers of CodeNeRF (src/model.py:38-39,44,50-51 ; train.py:136-137) as two
// small kernels over the FLAT parameter row of the fused trainer (fused.py):
//     [ trunk 13892 | latent W (4,32,L) | latent b (4,32) | B 63 | shape codes (n_obj,L) | texture codes (n_obj,L) ]
// These layers depend on the OBJECT only: forward produces, per object row,
//     z_k = relu(Wl_k code_k + bl_k)            (k = 0 shape_latent_1, 1 cat_latent, 2 shape_latent_2, 3 texture_latent_1)
//     biasrows[row][k] = Wt_k[:, :32] z_k + bt_k  (the effective bias the fused field kernels consume)
// and backward turns dbiasrows into the gradients of Wt_k[:, :32], bt_k (added into the trunk gradient), Wl_k, bl_k
// and both code tables, plus the code-norm regulariser of src/loss.py:5-15 (reg_scale * code / ||code||).
// Work is a few hundred kFLOP: one launch each instead of ~45 tiny PyTorch kernels is the whole point.
#include "latent_common.h"

namespace {
using namespace cnr;

// grid (n_obj * 4, C), 256 threads: one block per (object, slot), see latent_fwd_block
__global__ __launch_bounds__(256) void latent_fwd_kernel(const float* __restrict__ theta, FlatLayout lay,
                                                         float* __restrict__ zl, float* __restrict__ biasrows) {
  const float* th = theta + (int64_t)blockIdx.y * lay.stride;
  latent_fwd_block(th, th, lay, zl, biasrows, blockIdx.x >> 2, blockIdx.x & 3, blockIdx.y);
}

// grid (NB, C), 256 threads, NB = outputs / 256.  Every block recomputes the tiny d pre table (n_obj x 128 values) and the code
// norms into LDS, then takes its grid-stride share of the concatenated output space
//   [ d Wt_k[:, :32] and d bt_k : 4*32*33 | d Wl : 4*32*L | d bl : 128 | d shape codes : n_obj*L | d tex codes : n_obj*L ]
__global__ __launch_bounds__(256) void latent_bwd_kernel(const float* __restrict__ theta, FlatLayout lay,
                                                         const float* __restrict__ zl,
                                                         const float* __restrict__ dbiasrows, float reg_scale,
                                                         float* __restrict__ grad) {
  extern __shared__ float sm[];  // dpre [n_obj][4][32] | inv_norm_shape [n_obj] | inv_norm_tex [n_obj]
  const int c = blockIdx.y, n_obj = lay.n_obj, L = lay.L;
  const float* th = theta + (int64_t)c * lay.stride;
  float* g = grad + (int64_t)c * lay.stride;
  float* dpre = sm;
  float* inv_s = sm + n_obj * 128;
  float* inv_t = inv_s + n_obj;
  const float* dbr = dbiasrows + (int64_t)c * n_obj * 128;
  const float* z = zl + (int64_t)c * n_obj * 128;
  for (int i = threadIdx.x; i < n_obj * 128; i += 256) {   // d z -> d pre
    const int ob = i >> 7, k = (i >> 5) & 3, j = i & 31;
    int w_off, b_off, ld;
    latent_target(k, w_off, b_off, ld);
    float s = 0.0f;
#pragma unroll
    for (int o = 0; o < 32; ++o) s = fmaf(dbr[(ob * 4 + k) * 32 + o], th[w_off + o * ld + j], s);
    dpre[i] = z[i] > 0.0f ? s : 0.0f;
  }
  {  // code norms for the regulariser: one wave per code row
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int t = wv; t < 2 * n_obj; t += 4) {
      const int ob = t % n_obj;
      const float* code = th + (t < n_obj ? lay.shape : lay.tex) + (int64_t)ob * L;
      float s = 0.0f;
      for (int l = lane; l < L; l += 64) s = fmaf(code[l], code[l], s);
      s = wave_sum(s);
      if (lane == 0) (t < n_obj ? inv_s : inv_t)[ob] = reg_scale / sqrtf(s);
    }
  }
  __syncthreads();
  const int n0 = 4 * 32 * 33, n1 = n0 + 4 * 32 * L, n2 = n1 + 128, n3 = n2 + n_obj * L, n4 = n3 + n_obj * L;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < n4; t += gridDim.x * 256) {
    if (t < n0) {  // added: the field backward wrote the a-part of these weights already
      const int k = t / (32 * 33), r = t % (32 * 33), o = r / 33, j = r % 33;
      int w_off, b_off, ld;
      latent_target(k, w_off, b_off, ld);
      float s = 0.0f;
      for (int ob = 0; ob < n_obj; ++ob) {
        const float d = dbr[(ob * 4 + k) * 32 + o];
        s += j < 32 ? d * z[(ob * 4 + k) * 32 + j] : d;
      }
      if (j < 32) g[w_off + o * ld + j] += s; else g[b_off + o] += s;
    } else if (t < n1) {
      const int i = t - n0, k = i / (32 * L), r = i % (32 * L), o = r / L, l = r % L;
      float s = 0.0f;
      for (int ob = 0; ob < n_obj; ++ob)
        s = fmaf(dpre[(ob * 4 + k) * 32 + o], th[(k == 3 ? lay.tex : lay.shape) + (int64_t)ob * L + l], s);
      g[lay.latW + i] = s;
    } else if (t < n2) {
      const int i = t - n1;
      float s = 0.0f;
      for (int ob = 0; ob < n_obj; ++ob) s += dpre[ob * 128 + i];
      g[lay.latb + i] = s;
    } else if (t < n3) {
      const int i = t - n2, ob = i / L, l = i % L;
      float s = 0.0f;
      for (int ko = 0; ko < 96; ++ko) s = fmaf(dpre[ob * 128 + ko], th[lay.latW + (int64_t)ko * L + l], s);
      g[lay.shape + i] = s + inv_s[ob] * th[lay.shape + i];
    } else {
      const int i = t - n3, ob = i / L, l = i % L;
      float s = 0.0f;
      for (int o = 0; o < 32; ++o) s = fmaf(dpre[ob * 128 + 96 + o], th[lay.latW + (int64_t)(96 + o) * L + l], s);
      g[lay.tex + i] = s + inv_t[ob] * th[lay.tex + i];
    }
  }
}
}  // namespace

extern "C" int cnr_latent_fwd(const float* theta, int64_t class_stride, int64_t off_latW, int64_t off_latb,
                              int64_t off_shape, int64_t off_tex, int L, int n_obj, int C, float* zl,
                              float* biasrows, void* stream) {
  if (!theta || !zl || !biasrows || L <= 0 || n_obj <= 0 || C <= 0) return CNR_E_ARG;
  FlatLayout lay{class_stride, off_latW, off_latb, off_shape, off_tex, L, n_obj};
  hipLaunchKernelGGL(latent_fwd_kernel, dim3(n_obj * 4, C), dim3(256), 0, (hipStream_t)stream, theta, lay, zl, biasrows);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_latent_bwd(const float* theta, int64_t class_stride, int64_t off_latW, int64_t off_latb,
                              int64_t off_shape, int64_t off_tex, int L, int n_obj, int C, const float* zl,
                              const float* dbiasrows, float reg_scale, float* grad, void* stream) {
  if (!theta || !zl || !dbiasrows || !grad || L <= 0 || n_obj <= 0 || C <= 0) return CNR_E_ARG;
  if (n_obj > 64) return CNR_E_SHAPE;
  FlatLayout lay{class_stride, off_latW, off_latb, off_shape, off_tex, L, n_obj};
  const size_t lds = (size_t)(n_obj * 128 + 2 * n_obj) * sizeof(float);
  // one output per thread in the second phase (every block repeats the small first phase)
  const int64_t n4 = 4 * 32 * 33 + (int64_t)4 * 32 * L + 128 + (int64_t)2 * n_obj * L;
  int nblk = (int)((n4 + 255) / 256);
  if (nblk > 256) nblk = 256;
  hipLaunchKernelGGL(latent_bwd_kernel, dim3(nblk, C), dim3(256), lds, (hipStream_t)stream, theta, lay, zl, dbiasrows,
                     n_obj > 1 ? reg_scale : 0.0f, grad);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}
