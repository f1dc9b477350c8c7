// a9: CodeNeRF trunk, EXACT fp32, modular (reads a materialised embedding, writes sigmas / rgbs) --
// src/model.py:56-84 with do_cat=True, noise_std=None.  This is the precision-reference path on the
// GPU (parity ~1e-6 against the oracle) and the backend of the reference's un-fused call surface
// (model.CodeNeRF.forward); the throughput path is fused_mfma.hip.
//
// Mapping: one thread per sample, activations in VGPRs, weights read with wave-uniform addresses so
// hipcc emits s_load + v_fmac with an SGPR operand (no LDS traffic for weights, no per-lane weight
// registers).  Backward recomputes the forward in registers, walks the chain back per thread
// (dX = W^T dPre, again SGPR weights) and forms dW = dPre^T X -- a contraction over SAMPLES, i.e.
// over lanes -- with the exact-f32 matrix instruction v_mfma_f32_32x32x2_f32 on LDS-staged
// [sample][feature] tiles; per-workgroup dW lives in LDS and is flushed once with global atomics.
#include "cnr_common.h"

namespace {
using namespace cnr;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int IN, int OUT, int LD>
__device__ __forceinline__ void gemv_acc(const float* __restrict__ Wm, const float (&x)[IN], float (&y)[OUT]) {
#pragma unroll
  for (int o = 0; o < OUT; ++o) {
    float a = y[o];
#pragma unroll
    for (int i = 0; i < IN; ++i) a = fmaf(Wm[o * LD + i], x[i], a);
    y[o] = a;
  }
}
template <int IN, int OUT, int LD>
__device__ __forceinline__ void gemvT_acc(const float* __restrict__ Wm, const float (&d)[OUT], float (&dx)[IN]) {
#pragma unroll
  for (int o = 0; o < OUT; ++o) {
#pragma unroll
    for (int i = 0; i < IN; ++i) dx[i] = fmaf(Wm[o * LD + i], d[o], dx[i]);
  }
}
template <int N>
__device__ __forceinline__ void load_bias(const float* __restrict__ b, float (&y)[N]) {
#pragma unroll
  for (int o = 0; o < N; ++o) y[o] = b[o];
}
template <int N>
__device__ __forceinline__ void relu_(float (&y)[N]) {
#pragma unroll
  for (int o = 0; o < N; ++o) y[o] = fmaxf(y[o], 0.0f);
}
template <int N>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&y)[N]) {
#pragma unroll
  for (int o = 0; o < N; ++o) y[o] = p[o];
}
template <int N>
__device__ __forceinline__ void add_row(const float* __restrict__ p, const float (&a)[N], float (&y)[N]) {
#pragma unroll
  for (int o = 0; o < N; ++o) y[o] = a[o] + p[o];
}

template <int N>
__device__ __forceinline__ unsigned relu_mask(float (&y)[N]) {  // ReLU in place, bit o = (y[o] > 0)
  unsigned m = 0;
#pragma unroll
  for (int o = 0; o < N; ++o) {
    m |= (y[o] > 0.0f) ? (1u << o) : 0u;
    y[o] = fmaxf(y[o], 0.0f);
  }
  return m;
}
template <int N>
__device__ __forceinline__ void mask_bits(float (&d)[N], unsigned m) {
#pragma unroll
  for (int o = 0; o < N; ++o) d[o] = ((m >> o) & 1u) ? d[o] : 0.0f;
}

// forward for one sample; ep -> its 129-float embedding row, zl -> its ray's (4,32) latent rows
__device__ __forceinline__ void forward_sample(const float* __restrict__ Wt, const float* __restrict__ ep,
                                               const float* __restrict__ zl, float& sigma, float (&rgb)[3],
                                               float (&x1)[W], float (&x2)[W], float (&x3)[W], float (&a3)[W],
                                               float (&y4)[W], float (&x6)[W], float (&a6)[W], float (&a7)[16],
                                               unsigned (&mk)[4]) {
  float a[W];
  {
    float e1[E1];
    load_row<E1>(ep, e1);
    load_bias<W>(Wt + OFF_XYZ_B, a);
    gemv_acc<E1, W, E1>(Wt + OFF_XYZ_W, e1, a);
    mk[0] = relu_mask<W>(a);
    add_row<W>(zl + 0 * W, a, x1);
    load_bias<W>(Wt + OFF_S1_B, a);
    gemv_acc<W, W, W>(Wt + OFF_S1_W, x1, a);
    mk[1] = relu_mask<W>(a);
    add_row<W>(zl + 1 * W, a, x2);
    load_bias<W>(Wt + OFF_CAT_B, a);
    gemv_acc<W, W, W + E1>(Wt + OFF_CAT_W, x2, a);
    gemv_acc<E1, W, W + E1>(Wt + OFF_CAT_W + W, e1, a);
    mk[2] = relu_mask<W>(a);
  }
  add_row<W>(zl + 2 * W, a, x3);
  load_bias<W>(Wt + OFF_S2_B, a3);
  gemv_acc<W, W, W>(Wt + OFF_S2_W, x3, a3);
  relu_<W>(a3);
  load_bias<W>(Wt + OFF_ES_B, y4);
  gemv_acc<W, W, W>(Wt + OFF_ES_W, a3, y4);
  {
    float raw[1] = {Wt[OFF_SG_B]};
    gemv_acc<W, 1, W>(Wt + OFF_SG_W, y4, raw);
    sigma = raw[0] * 10.0f;
  }
  {
    float e2[E2];
    load_row<E2>(ep + E1, e2);
    load_bias<W>(Wt + OFF_VD_B, a);
    gemv_acc<W, W, W + E2>(Wt + OFF_VD_W, y4, a);
    gemv_acc<E2, W, W + E2>(Wt + OFF_VD_W + W, e2, a);
    mk[3] = relu_mask<W>(a);
  }
  add_row<W>(zl + 3 * W, a, x6);
  load_bias<W>(Wt + OFF_T1_B, a6);
  gemv_acc<W, W, W>(Wt + OFF_T1_W, x6, a6);
  relu_<W>(a6);
  load_bias<16>(Wt + OFF_R0_B, a7);
  gemv_acc<W, 16, W>(Wt + OFF_R0_W, a6, a7);
  relu_<16>(a7);
  float o3[3];
  load_bias<3>(Wt + OFF_R2_B, o3);
  gemv_acc<16, 3, 16>(Wt + OFF_R2_W, a7, o3);
#pragma unroll
  for (int k = 0; k < 3; ++k) rgb[k] = 1.0f / (1.0f + expf(-o3[k]));
}

__global__ __launch_bounds__(256) void mlp_fwd_kernel(const float* __restrict__ e, const float* __restrict__ zlat,
                                                      const float* __restrict__ trunk, float* __restrict__ sigmas,
                                                      float* __restrict__ rgbs, int64_t N, int S) {
  const int c = blockIdx.y;
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float* Wt = trunk + (int64_t)c * TRUNK;
  const int64_t gs = (int64_t)c * N + n;
  const float* ep = e + gs * E;
  const float* zl = zlat + ((int64_t)c * (N / S) + n / S) * (CNR_NLAT * W);
  float sigma, rgb[3];
  float x1[W], x2[W], x3[W], a3[W], y4[W], x6[W], a6[W], a7[16];
  unsigned mk[4];
  forward_sample(Wt, ep, zl, sigma, rgb, x1, x2, x3, a3, y4, x6, a6, a7, mk);
  sigmas[gs] = sigma;
  rgbs[gs * 3 + 0] = rgb[0]; rgbs[gs * 3 + 1] = rgb[1]; rgbs[gs * 3 + 2] = rgb[2];
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
constexpr int ST_LD = 33;  // padded row (floats) of the [sample][feature] staging tiles

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int OUT>
__device__ __forceinline__ void stage_rows(float* st, int lane, const float (&d)[OUT]) {
#pragma unroll
  for (int o = 0; o < 32; ++o) st[lane * ST_LD + o] = (o < OUT) ? d[o < OUT ? o : 0] : 0.0f;
}

// dW[o][j] += sum_s A[s][o] * X[s][j]  for o < OUT, j < NJ ; A staged in stA, this thread's NJ x-values in xc
template <int OUT, int NJ>
__device__ __forceinline__ void dw_block(const float* stA, float* stB, int lane, const float* xc /*NJ regs*/,
                                         float* dW, int ld) {
  wave_lds_sync();  // previous readers of stB are done
#pragma unroll
  for (int j = 0; j < 32; ++j) stB[lane * ST_LD + j] = (j < NJ) ? xc[j < NJ ? j : 0] : 0.0f;
  wave_lds_sync();
  f32x16 acc = {0.f};
  const int r0 = lane >> 5, cidx = lane & 31;
#pragma unroll 8
  for (int k = 0; k < 32; ++k) {
    const float a = stA[(2 * k + r0) * ST_LD + cidx];
    const float b = stB[(2 * k + r0) * ST_LD + cidx];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = (r & 3) + 8 * (r >> 2) + 4 * r0;
    if (o < OUT && cidx < NJ) atomicAdd(&dW[o * ld + cidx], acc[r]);
  }
}

template <int OUT>
__device__ __forceinline__ void db_accum(const float (&d)[OUT], int lane, float* db) {
#pragma unroll
  for (int o = 0; o < OUT; ++o) {
    const float v = wave_sum(d[o]);
    if (lane == 0) atomicAdd(&db[o], v);
  }
}

// per-ray segment sums of a [64][32] staged tile into dzlat rows (rays may straddle the wave)
__device__ __forceinline__ void dz_flush(const float* st, int lane, int64_t n0, int64_t N, int S,
                                         float* __restrict__ dz_base /* class base: (R,4,32) */, int slot) {
  wave_lds_sync();
  const int j = lane & 31, half = lane >> 5;
  float acc = 0.0f;
  int64_t cur = -1;
  for (int row = half * 32; row < half * 32 + 32; ++row) {
    const int64_t n = n0 + row;
    if (n >= N) break;
    const int64_t ray = n / S;
    if (ray != cur) {
      if (cur >= 0) atomicAdd(&dz_base[(cur * CNR_NLAT + slot) * W + j], acc);
      cur = ray; acc = 0.0f;
    }
    acc += st[row * ST_LD + j];
  }
  if (cur >= 0) atomicAdd(&dz_base[(cur * CNR_NLAT + slot) * W + j], acc);
  wave_lds_sync();
}

template <int N_>
__device__ __forceinline__ void zero_(float (&y)[N_]) {
#pragma unroll
  for (int o = 0; o < N_; ++o) y[o] = 0.0f;
}
template <int N_>
__device__ __forceinline__ void mask_relu(float (&d)[N_], const float (&a)[N_]) {
#pragma unroll
  for (int o = 0; o < N_; ++o) d[o] = (a[o] > 0.0f) ? d[o] : 0.0f;
}

__global__ __launch_bounds__(256, 1) void mlp_bwd_kernel(
    const float* __restrict__ e, const float* __restrict__ zlat, const float* __restrict__ trunk,
    const float* __restrict__ dsig, const float* __restrict__ drgb, float* __restrict__ de,
    float* __restrict__ dzlat, float* __restrict__ dtrunk, int64_t N, int S, int tiles_per_block) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dWacc = smem;                                        // TRUNK floats
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* stA = smem + ((TRUNK + 3) & ~3) + wv * (2 * 64 * ST_LD);  // [64][33]
  float* stB = stA + 64 * ST_LD;
  const int c = blockIdx.y;
  const float* Wt = trunk + (int64_t)c * TRUNK;
  float* dz_base = dzlat + (int64_t)c * (N / S) * (CNR_NLAT * W);

  for (int i = threadIdx.x; i < TRUNK; i += 256) dWacc[i] = 0.0f;
  __syncthreads();

  for (int t = 0; t < tiles_per_block; ++t) {
    const int64_t tile = (int64_t)blockIdx.x * tiles_per_block + t;
    const int64_t n0w = tile * 256 + wv * 64;  // first sample of this wave
    if (n0w >= N) break;                       // wave-uniform
    const int64_t n = n0w + lane;
    const bool live = n < N;
    const int64_t nc = live ? n : N - 1;
    const int64_t gs = (int64_t)c * N + nc;
    const float* ep = e + gs * E;
    const float* zl = zlat + ((int64_t)c * (N / S) + nc / S) * (CNR_NLAT * W);

    float sigma, rgb[3];
    float x1[W], x2[W], x3[W], a3[W], y4[W], x6[W], a6[W], a7[16];
    unsigned mk[4];  // ReLU masks of encoding_xyz, shape_layer_1, cat_layer, encoding_viewdir
    forward_sample(Wt, ep, zl, sigma, rgb, x1, x2, x3, a3, y4, x6, a6, a7, mk);

    // ---- head: rgb.2 / rgb.0 -----------------------------------------------------------------
    float d3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) d3[k] = live ? drgb[gs * 3 + k] * rgb[k] * (1.0f - rgb[k]) : 0.0f;
    const float draw = live ? dsig[gs] * 10.0f : 0.0f;

    stage_rows<3>(stA, lane, d3);
    dw_block<3, 16>(stA, stB, lane, a7, dWacc + OFF_R2_W, 16);
    db_accum<3>(d3, lane, dWacc + OFF_R2_B);
    float d7[16];
    zero_<16>(d7);
    gemvT_acc<16, 3, 16>(Wt + OFF_R2_W, d3, d7);
    mask_relu<16>(d7, a7);

    wave_lds_sync();
    stage_rows<16>(stA, lane, d7);
    dw_block<16, 32>(stA, stB, lane, a6, dWacc + OFF_R0_W, W);
    db_accum<16>(d7, lane, dWacc + OFF_R0_B);
    float d[W];
    zero_<W>(d);
    gemvT_acc<W, 16, W>(Wt + OFF_R0_W, d7, d);
    mask_relu<W>(d, a6);

    // ---- texture_layer_1 ------------------------------------------------------------------------
    wave_lds_sync();
    stage_rows<W>(stA, lane, d);
    dw_block<W, 32>(stA, stB, lane, x6, dWacc + OFF_T1_W, W);
    db_accum<W>(d, lane, dWacc + OFF_T1_B);
    float dx[W];
    zero_<W>(dx);
    gemvT_acc<W, W, W>(Wt + OFF_T1_W, d, dx);          // d x6 = d a5 + d z3
    wave_lds_sync();
    stage_rows<W>(stA, lane, dx);
    dz_flush(stA, lane, n0w, N, S, dz_base, 3);
    {
      mask_bits<W>(dx, mk[3]);
      float e2[E2];
      load_row<E2>(ep + E1, e2);
      // ---- encoding_viewdir: inputs [y4 | e2] ------------------------------------------------------
      stage_rows<W>(stA, lane, dx);
      dw_block<W, 32>(stA, stB, lane, y4, dWacc + OFF_VD_W, W + E2);
      dw_block<W, 32>(stA, stB, lane, e2, dWacc + OFF_VD_W + W, W + E2);
      dw_block<W, E2 - 32>(stA, stB, lane, e2 + 32, dWacc + OFF_VD_W + W + 32, W + E2);
      db_accum<W>(dx, lane, dWacc + OFF_VD_B);
      float de2[E2];
      zero_<E2>(de2);
      gemvT_acc<E2, W, W + E2>(Wt + OFF_VD_W + W, dx, de2);
      if (live) {
#pragma unroll
        for (int i = 0; i < E2; ++i) de[gs * E + E1 + i] = de2[i];
      }
    }
    float dy4[W];
    zero_<W>(dy4);
    gemvT_acc<W, W, W + E2>(Wt + OFF_VD_W, dx, dy4);
    // ---- sigma head -----------------------------------------------------------------------------
    {
      float dr[1] = {draw};
      wave_lds_sync();
      stage_rows<1>(stA, lane, dr);
      dw_block<1, 32>(stA, stB, lane, y4, dWacc + OFF_SG_W, W);
      db_accum<1>(dr, lane, dWacc + OFF_SG_B);
#pragma unroll
      for (int i = 0; i < W; ++i) dy4[i] = fmaf(Wt[OFF_SG_W + i], draw, dy4[i]);
    }
    // ---- encoding_shape (no activation) ---------------------------------------------------------------
    wave_lds_sync();
    stage_rows<W>(stA, lane, dy4);
    dw_block<W, 32>(stA, stB, lane, a3, dWacc + OFF_ES_W, W);
    db_accum<W>(dy4, lane, dWacc + OFF_ES_B);
    zero_<W>(d);
    gemvT_acc<W, W, W>(Wt + OFF_ES_W, dy4, d);
    mask_relu<W>(d, a3);
    // ---- shape_layer_2 -------------------------------------------------------------------------------
    wave_lds_sync();
    stage_rows<W>(stA, lane, d);
    dw_block<W, 32>(stA, stB, lane, x3, dWacc + OFF_S2_W, W);
    db_accum<W>(d, lane, dWacc + OFF_S2_B);
    zero_<W>(dx);
    gemvT_acc<W, W, W>(Wt + OFF_S2_W, d, dx);           // d x3 = d a2 + d z2
    wave_lds_sync();
    stage_rows<W>(stA, lane, dx);
    dz_flush(stA, lane, n0w, N, S, dz_base, 2);
    {
      // cat_layer; inputs [x2 | e1]
      float e1[E1];
      load_row<E1>(ep, e1);
      mask_bits<W>(dx, mk[2]);
      stage_rows<W>(stA, lane, dx);
      dw_block<W, 32>(stA, stB, lane, x2, dWacc + OFF_CAT_W, W + E1);
      dw_block<W, 32>(stA, stB, lane, e1, dWacc + OFF_CAT_W + W, W + E1);
      dw_block<W, 32>(stA, stB, lane, e1 + 32, dWacc + OFF_CAT_W + W + 32, W + E1);
      dw_block<W, E1 - 64>(stA, stB, lane, e1 + 64, dWacc + OFF_CAT_W + W + 64, W + E1);
      db_accum<W>(dx, lane, dWacc + OFF_CAT_B);
      float de1[E1];
      zero_<E1>(de1);
      gemvT_acc<E1, W, W + E1>(Wt + OFF_CAT_W + W, dx, de1);
      zero_<W>(d);
      gemvT_acc<W, W, W + E1>(Wt + OFF_CAT_W, dx, d);    // d x2 = d a1 + d z1
      wave_lds_sync();
      stage_rows<W>(stA, lane, d);
      dz_flush(stA, lane, n0w, N, S, dz_base, 1);
      mask_bits<W>(d, mk[1]);
      // ---- shape_layer_1 ------------------------------------------------------------------------------
      stage_rows<W>(stA, lane, d);
      dw_block<W, 32>(stA, stB, lane, x1, dWacc + OFF_S1_W, W);
      db_accum<W>(d, lane, dWacc + OFF_S1_B);
      zero_<W>(dx);
      gemvT_acc<W, W, W>(Wt + OFF_S1_W, d, dx);          // d x1 = d a0 + d z0
      wave_lds_sync();
      stage_rows<W>(stA, lane, dx);
      dz_flush(stA, lane, n0w, N, S, dz_base, 0);
      mask_bits<W>(dx, mk[0]);
      // ---- encoding_xyz ---------------------------------------------------------------------------------
      stage_rows<W>(stA, lane, dx);
      dw_block<W, 32>(stA, stB, lane, e1, dWacc + OFF_XYZ_W, E1);
      dw_block<W, 32>(stA, stB, lane, e1 + 32, dWacc + OFF_XYZ_W + 32, E1);
      dw_block<W, E1 - 64>(stA, stB, lane, e1 + 64, dWacc + OFF_XYZ_W + 64, E1);
      db_accum<W>(dx, lane, dWacc + OFF_XYZ_B);
      gemvT_acc<E1, W, E1>(Wt + OFF_XYZ_W, dx, de1);
      if (live) {
#pragma unroll
        for (int i = 0; i < E1; ++i) de[gs * E + i] = de1[i];
      }
    }
    wave_lds_sync();
  }
  __syncthreads();
  float* out = dtrunk + (int64_t)c * TRUNK;
  for (int i = threadIdx.x; i < TRUNK; i += 256) {
    const float v = dWacc[i];
    if (v != 0.0f) atomicAdd(&out[i], v);
  }
}
}  // namespace

extern "C" int cnr_mlp_fwd_f32(const float* e, const float* zlat, const float* trunk, float* sigmas, float* rgbs,
                               int C, int R, int S, void* stream) {
  if (!e || !zlat || !trunk || !sigmas || !rgbs || C <= 0 || R <= 0 || S <= 0) return CNR_E_ARG;
  const int64_t N = (int64_t)R * S;
  dim3 grid((unsigned)((N + 255) / 256), (unsigned)C);
  hipLaunchKernelGGL(mlp_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, e, zlat, trunk, sigmas, rgbs, N, S);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_mlp_bwd_f32(const float* e, const float* zlat, const float* trunk, const float* dsig,
                               const float* drgb, float* de, float* dzlat, float* dtrunk, int C, int R, int S,
                               void* stream) {
  if (!e || !zlat || !trunk || !dsig || !drgb || !de || !dzlat || !dtrunk || C <= 0 || R <= 0 || S <= 0)
    return CNR_E_ARG;
  const int64_t N = (int64_t)R * S;
  const int64_t tiles = (N + 255) / 256;
  int64_t blocks = tiles < 512 ? tiles : 512;
  const int tpb = (int)((tiles + blocks - 1) / blocks);
  blocks = (tiles + tpb - 1) / tpb;
  const size_t lds = (size_t)(((TRUNK + 3) & ~3) + 4 * 2 * 64 * ST_LD) * sizeof(float);
  {
    static cnr::DeviceOnce once;
    const int er = cnr::set_max_dynamic_lds(once, (const void*)mlp_bwd_kernel, (int)lds);
    if (er) return er;
  }
  dim3 grid((unsigned)blocks, (unsigned)C);
  hipLaunchKernelGGL(mlp_bwd_kernel, grid, dim3(256), lds, (hipStream_t)stream, e, zlat, trunk, dsig, drgb, de,
                     dzlat, dtrunk, N, S, tpb);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}
