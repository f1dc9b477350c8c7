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

// grid (NB, C), 256 threads, NB = outputs / 256: see latent_bwd_block (latent_common.h).  Gradients are stored
// (latent parameters) or added (trunk entries the field backward wrote already).
struct GradStoreSink {
  float* g;  // this class's gradient row
  __device__ __forceinline__ void trunk_add(int idx, float v) const { g[idx] += v; }
  __device__ __forceinline__ void latent_set(int64_t idx, float v) const { g[idx] = v; }
  __device__ __forceinline__ void prefetch(int64_t) const {}
};
__global__ __launch_bounds__(256) void latent_bwd_kernel(const float* __restrict__ theta, FlatLayout lay,
                                                         const float* __restrict__ zl,
                                                         const float* __restrict__ dbiasrows, float reg_scale,
                                                         float* __restrict__ grad, const int* __restrict__ n_obj_cls) {
  extern __shared__ float sm[];
  const int c = blockIdx.y;
  GradStoreSink sink{grad + (int64_t)c * lay.stride};
  const int n_real = n_obj_cls ? n_obj_cls[c] : lay.n_obj;   // src/loss.py:5-15: the regulariser needs > 1 object
  latent_bwd_block(theta + (int64_t)c * lay.stride, lay, zl + (int64_t)c * lay.n_obj * 128,
                   dbiasrows + (int64_t)c * lay.n_obj * 128, n_real > 1 ? reg_scale : 0.0f, sm, sink, blockIdx.x,
                   gridDim.x, true, n_real);
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
                              const float* dbiasrows, float reg_scale, float* grad, const int* n_obj_cls,
                              void* stream) {
  if (!theta || !zl || !dbiasrows || !grad || L <= 0 || n_obj <= 0 || C <= 0) return CNR_E_ARG;
  if (n_obj > 64) return CNR_E_SHAPE;
  FlatLayout lay{class_stride, off_latW, off_latb, off_shape, off_tex, L, n_obj};
  const size_t lds = (size_t)(n_obj * 128 + 2 * n_obj) * sizeof(float);
  // one output per thread in the second phase (every block repeats the small first phase)
  const int64_t n4 = 4 * 32 * 33 + (int64_t)4 * 32 * L + 128 + (int64_t)2 * n_obj * L;
  int nblk = (int)((n4 + 255) / 256);
  if (nblk > 256) nblk = 256;
  hipLaunchKernelGGL(latent_bwd_kernel, dim3(nblk, C), dim3(256), lds, (hipStream_t)stream, theta, lay, zl, dbiasrows,
                     reg_scale, grad, n_obj_cls);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}
