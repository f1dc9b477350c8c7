// a8: UniDirsEmbed forward / backward, exact-fp32 modular kernels (src/embedding.py:82-92).
//   t = x / scale ; p_j = B_j . t ; e[0:3] = t ; e[3 + 21 k + j] = sin(pi * 2^k * p_j), k = 0..5
// HBM-bound by construction (the reference materialises 516 B per sample); the fused MFMA path
// (fused_mfma.hip) never writes e.  Forward: one thread per OUTPUT element so the 129-float rows are
// written fully coalesced; backward: one thread per sample, grid-stride, 63 register partial sums of
// dB reduced wave -> block -> one atomic per block and element.
#include "cnr_common.h"

namespace {
constexpr float PI_F = 3.14159265358979323846f;

__global__ __launch_bounds__(256) void pe_fwd_kernel(const float* __restrict__ x, const float* __restrict__ B,
                                                     float* __restrict__ e, int64_t N, float scale) {
  const int c = blockIdx.y;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * cnr::E) return;
  const int64_t n = idx / cnr::E;
  const int f = (int)(idx - n * cnr::E);
  const float* xp = x + ((int64_t)c * N + n) * 3;
  float out;
  if (f < 3) {
    out = xp[f] / scale;
  } else {
    const int k = (f - 3) / CNR_NDIR, j = (f - 3) - k * CNR_NDIR;
    const float* b = B + ((int64_t)c * CNR_NDIR + j) * 3;
    const float t0 = xp[0] / scale, t1 = xp[1] / scale, t2 = xp[2] / scale;
    // same association as at::linear's dot: ((t0*b0 + t1*b1) + t2*b2)
    const float p = t0 * b[0] + t1 * b[1] + t2 * b[2];
    out = sinf((p * (float)(1 << k)) * PI_F);
  }
  e[(int64_t)c * N * cnr::E + idx] = out;
}

__global__ __launch_bounds__(256) void pe_bwd_kernel(const float* __restrict__ x, const float* __restrict__ B,
                                                     const float* __restrict__ de, float* __restrict__ dB,
                                                     float* __restrict__ dx, int64_t N, float scale) {
  const int c = blockIdx.y;
  float acc[CNR_NDIR * 3];
#pragma unroll
  for (int i = 0; i < CNR_NDIR * 3; ++i) acc[i] = 0.0f;
  const float* Bc = B + (int64_t)c * CNR_NDIR * 3;
  const float inv = 1.0f / scale;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
    const float* xp = x + ((int64_t)c * N + n) * 3;
    const float* dep = de + ((int64_t)c * N + n) * cnr::E;
    const float t0 = xp[0] / scale, t1 = xp[1] / scale, t2 = xp[2] / scale;
    float gt0 = dep[0], gt1 = dep[1], gt2 = dep[2];
#pragma unroll
    for (int j = 0; j < CNR_NDIR; ++j) {
      const float b0 = Bc[j * 3 + 0], b1 = Bc[j * 3 + 1], b2 = Bc[j * 3 + 2];
      const float p = t0 * b0 + t1 * b1 + t2 * b2;
      float gp = 0.0f;  // dL/dp_j
#pragma unroll
      for (int k = 0; k < CNR_NFREQ; ++k) {
        const float fk = (float)(1 << k);
        gp += dep[3 + CNR_NDIR * k + j] * cosf((p * fk) * PI_F) * (fk * PI_F);
      }
      acc[j * 3 + 0] += gp * t0; acc[j * 3 + 1] += gp * t1; acc[j * 3 + 2] += gp * t2;
      gt0 += gp * b0; gt1 += gp * b1; gt2 += gp * b2;
    }
    if (dx) {
      float* dxp = dx + ((int64_t)c * N + n) * 3;
      dxp[0] = gt0 * inv; dxp[1] = gt1 * inv; dxp[2] = gt2 * inv;
    }
  }
  __shared__ float sm[4][CNR_NDIR * 3];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < CNR_NDIR * 3; ++i) {
    const float v = cnr::wave_sum(acc[i]);
    if (lane == 0) sm[wv][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < CNR_NDIR * 3)
    atomicAdd(dB + (int64_t)c * CNR_NDIR * 3 + threadIdx.x,
              sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}
// dB alone (no gradient to x: poses are not optimised): one block per (256 samples, direction j) instead of
// one thread per sample walking all 21 directions x 6 bands of precise cosines -- 21 x the blocks, a 21st of the dependent
// work per thread: 46.6 -> ~6 us for the background's 16 800 samples (the sample-per-thread form left 3/4 of the CUs idle).
__global__ __launch_bounds__(256) void pe_bwd_dir_kernel(const float* __restrict__ x, const float* __restrict__ B,
                                                         const float* __restrict__ de, float* __restrict__ dB, int64_t N,
                                                         float scale) {
  const int c = blockIdx.y, j = blockIdx.z;
  const float* Bc = B + (int64_t)c * CNR_NDIR * 3;
  const float b0 = Bc[j * 3 + 0], b1 = Bc[j * 3 + 1], b2 = Bc[j * 3 + 2];
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
    const float* xp = x + ((int64_t)c * N + n) * 3;
    const float* dep = de + ((int64_t)c * N + n) * cnr::E;
    const float t0 = xp[0] / scale, t1 = xp[1] / scale, t2 = xp[2] / scale;
    const float p = t0 * b0 + t1 * b1 + t2 * b2;
    float gp = 0.0f;
#pragma unroll
    for (int k = 0; k < CNR_NFREQ; ++k) {
      const float fk = (float)(1 << k);
      gp += dep[3 + CNR_NDIR * k + j] * cosf((p * fk) * PI_F) * (fk * PI_F);
    }
    a0 += gp * t0; a1 += gp * t1; a2 += gp * t2;
  }
  __shared__ float sm[4][3];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  a0 = cnr::wave_sum(a0); a1 = cnr::wave_sum(a1); a2 = cnr::wave_sum(a2);
  if (lane == 0) { sm[wv][0] = a0; sm[wv][1] = a1; sm[wv][2] = a2; }
  __syncthreads();
  if (threadIdx.x < 3)
    atomicAdd(dB + (int64_t)c * CNR_NDIR * 3 + j * 3 + threadIdx.x,
              sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}
}  // namespace

extern "C" int cnr_pe_fwd(const float* x, const float* B, float* e, int C, int64_t N, float scale, void* stream) {
  if (!x || !B || !e || C <= 0 || N <= 0 || !(scale > 0.0f)) return CNR_E_ARG;
  const int64_t total = N * cnr::E;
  dim3 grid((unsigned)((total + 255) / 256), (unsigned)C);
  hipLaunchKernelGGL(pe_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, B, e, N, scale);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_pe_bwd(const float* x, const float* B, const float* de, float* dB, float* dx, int C,
                          int64_t N, float scale, void* stream) {
  if (!x || !B || !de || !dB || C <= 0 || N <= 0 || !(scale > 0.0f)) return CNR_E_ARG;
  int64_t blocks = (N + 255) / 256;
  if (!dx) {
    if (blocks > 256) blocks = 256;
    dim3 gridj((unsigned)blocks, (unsigned)C, CNR_NDIR);
    hipLaunchKernelGGL(pe_bwd_dir_kernel, gridj, dim3(256), 0, (hipStream_t)stream, x, B, de, dB, N, scale);
    CNR_LAUNCH_CHECK();
    return CNR_OK;
  }
  if (blocks > 1024) blocks = 1024;
  dim3 grid((unsigned)blocks, (unsigned)C);
  hipLaunchKernelGGL(pe_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, B, de, dB, dx, N, scale);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}
