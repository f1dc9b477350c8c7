// Shared by latent.hip (cnr_latent_fwd / cnr_latent_bwd) and fused_fwd.hip (cnr_param_prep): the flat parameter
// row layout of the fused trainer and the per-(object, latent layer) forward block.
#pragma once
#include "cnr_common.h"

namespace cnr {
struct FlatLayout {
  int64_t stride;  // floats per class row
  int64_t latW, latb, shape, tex;
  int L, n_obj;
};
__device__ __forceinline__ void latent_target(int k, int& w_off, int& b_off, int& ld) {
  if (k == 0) { w_off = OFF_S1_W; b_off = OFF_S1_B; ld = 32; }
  else if (k == 1) { w_off = OFF_CAT_W; b_off = OFF_CAT_B; ld = 32 + E1; }
  else if (k == 2) { w_off = OFF_S2_W; b_off = OFF_S2_B; ld = 32; }
  else { w_off = OFF_T1_W; b_off = OFF_T1_B; ld = 32; }
}

// One 256-thread block per (object, slot k): a wave per output for the L-long dot product (coalesced weight rows,
// the 8 outputs of a wave unrolled so that their loads are in flight together), then the 32x32 product with the
// trunk layer: z_k = relu(Wl_k code + bl_k), biasrows[row][k] = Wt_k[:, :32] z_k + bt_k.
// th = this class's parameter row, trunk = its trunk blob (th itself when the trunk sits at offset 0).
__device__ __forceinline__ void latent_fwd_block(const float* __restrict__ th, const float* __restrict__ trunk,
                                                 const FlatLayout& lay, float* __restrict__ zl,
                                                 float* __restrict__ biasrows, int obj, int k, int c) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* code = th + (k == 3 ? lay.tex : lay.shape) + (int64_t)obj * lay.L;
  __shared__ float zs[32];
  // second-phase operands do not depend on the first phase: fetch them now, ahead of the barrier
  int w_off, b_off, ld;
  latent_target(k, w_off, b_off, ld);
  const int o2 = threadIdx.x >> 3, p2 = threadIdx.x & 7;
  float tw[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) tw[q] = trunk[w_off + o2 * ld + p2 * 4 + q];
  const float tb = trunk[b_off + o2];
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int o = wv + 4 * i;
    const float* w = th + lay.latW + ((int64_t)k * 32 + o) * lay.L;
    float a = 0.0f;
    for (int l = lane; l < lay.L; l += 64) a = fmaf(code[l], w[l], a);
    acc[i] = a;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int o = wv + 4 * i;
    const float a = wave_sum(acc[i]);
    if (lane == 0) zs[o] = fmaxf(a + th[lay.latb + k * 32 + o], 0.0f);
  }
  __syncthreads();
  {  // 32 outputs x 32 inputs on 256 threads: 8 lanes per output, 4 inputs each, then an 8-lane sum
    const int o = o2, p = p2;
    float br = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) br = fmaf(tw[q], zs[p * 4 + q], br);
    br += __shfl_xor(br, 1, 64);
    br += __shfl_xor(br, 2, 64);
    br += __shfl_xor(br, 4, 64);
    if (p == 0) {
      const int64_t row = (int64_t)c * lay.n_obj + obj;
      zl[(row * 4 + k) * 32 + o] = zs[o];
      biasrows[(row * 4 + k) * 32 + o] = br + tb;
    }
  }
}
}  // namespace cnr
