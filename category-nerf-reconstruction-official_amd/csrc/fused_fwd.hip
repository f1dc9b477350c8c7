// Fused field forward for the train / render path: pts -> UniDirsEmbed (a8) -> CodeNeRF trunk (a9)
// -> sigmas, rgbs, one launch, nothing but 12 B in and 16 B out per sample touches HBM.
// f16 MFMA operands (v_mfma_f32_32x32x16_f16), fp32 accumulate, fp32 positional encoding with the
// hardware sine, fp32 VALU sigma head (the x10 logit is the precision-critical output: it never sees an
// f16 rounding of its inputs' last layer).  See fused_common.h for the operand-layout contract.
//
// Work split: tile = 32 consecutive samples; one wave processes a tile at a time (grid-stride), four
// waves per workgroup share the packed weight image in LDS (31 KB), >= 2 waves per SIMD for VALU rate.
// Latent conditioning arrives as per-row effective biases  bias'_l[row] = W_l z_l[row] + b_l  (row =
// object, or ray for per-ray codes): W_l (a + z) + b = W_l a + bias'_l, so the latent add costs no VALU.
#include <cstdlib>
#include "fused_common.h"
#include "latent_common.h"
#include "render_common.h"
#include "sample_common.h"

namespace {
using namespace fz;

// ---------------------------------------------------------------------------------------------------
// weight packing (device): fp32 trunk blob (C, 13892) -> packed image (C, PK_BYTES)
// ---------------------------------------------------------------------------------------------------
struct LayerRef { int w_off, ld, out; };

__device__ __forceinline__ float bwd_elem(const float* __restrict__ Wt, int kt, int r, int h, int j) {
  // A[r][k] of the transposed product: r = input feature (or E slot row), k = output feature
  int s, w_off, ld, out, in_col = -1;
  const int hh = (r >> 2) & 1, reg = (r & 3) + 4 * (r >> 3);  // inverse of acc_row
  if (kt < KT_R0) { s = 0; w_off = OFF_R2_W; ld = 16; out = 3; in_col = r < 16 ? r : -1; }
  else if (kt < KT_T1) { s = 0; w_off = OFF_R0_W; ld = 32; out = 16; in_col = r; }
  else if (kt < KT_VD_Y) { s = kt - KT_T1; w_off = OFF_T1_W; ld = 32; out = 32; in_col = r; }
  else if (kt < KT_VD_E) { s = kt - KT_VD_Y; w_off = OFF_VD_W; ld = 32 + E2; out = 32; in_col = r; }
  else if (kt < KT_ES) { const int b = (kt - KT_VD_E) >> 1; s = (kt - KT_VD_E) & 1; w_off = OFF_VD_W; ld = 32 + E2; out = 32;
    const int q = 16 * b + reg; const int f = q < 24 ? slot_feature(1, hh, q) : -1; in_col = f < 0 ? -1 : 32 + (f - E1); }
  else if (kt < KT_S2) { s = kt - KT_ES; w_off = OFF_ES_W; ld = 32; out = 32; in_col = r; }
  else if (kt < KT_CAT_Y) { s = kt - KT_S2; w_off = OFF_S2_W; ld = 32; out = 32; in_col = r; }
  else if (kt < KT_CAT_E) { s = kt - KT_CAT_Y; w_off = OFF_CAT_W; ld = 32 + E1; out = 32; in_col = r; }
  else if (kt < KT_S1) { const int b = (kt - KT_CAT_E) >> 1; s = (kt - KT_CAT_E) & 1; w_off = OFF_CAT_W; ld = 32 + E1; out = 32;
    const int f = slot_feature(0, hh, 16 * b + reg); in_col = f < 0 ? -1 : 32 + f; }
  else if (kt < KT_XYZ_E) { s = kt - KT_S1; w_off = OFF_S1_W; ld = 32; out = 32; in_col = r; }
  else { const int b = (kt - KT_XYZ_E) >> 1; s = (kt - KT_XYZ_E) & 1; w_off = OFF_XYZ_W; ld = E1; out = 32;
    const int f = slot_feature(0, hh, 16 * b + reg); in_col = f; }
  const int o = acc_feature(s, h, j);
  if (in_col < 0 || o >= out) return 0.0f;
  return Wt[w_off + o * ld + in_col];
}

// one block = one 1 KB operand fragment (kk < NKK_FWD + NKK_BWD) or the constants block (kk == that)
__device__ __forceinline__ void pack_block(const float* __restrict__ Wt, unsigned char* __restrict__ out, int kk) {
  if (kk < NKK_FWD + NKK_BWD) {
    for (int e = threadIdx.x; e < 64 * 8; e += 256) {
      const int lane = e >> 3, j = e & 7, r = lane & 31, h = lane >> 5;
      const float v = kk < NKK_FWD ? fwd_elem(Wt, kk, r, h, j) : bwd_elem(Wt, kk - NKK_FWD, r, h, j);
      const size_t base = kk < NKK_FWD ? (size_t)PK_OFF_FWD + (size_t)kk * FRAG_BYTES
                                       : (size_t)PK_OFF_BWD + (size_t)(kk - NKK_FWD) * FRAG_BYTES;
      reinterpret_cast<_Float16*>(out + base)[lane * 8 + j] = (_Float16)v;
    }
  } else {
    float* cf = reinterpret_cast<float*>(out + PK_OFF_CONST);
    for (int i = threadIdx.x; i < CF_FLOATS; i += 256) {
      float v = 0.0f;
      const int o = i & 31;
      if (i < 32) v = Wt[OFF_XYZ_B + o];
      else if (i < 64) v = Wt[OFF_ES_B + o];
      else if (i < 96) v = Wt[OFF_VD_B + o];
      else if (i < 128) v = o < 16 ? Wt[OFF_R0_B + o] : 0.0f;
      else if (i < 160) v = o < 3 ? Wt[OFF_R2_B + o] : 0.0f;
      else if (i < 192) v = Wt[OFF_SG_W + o];
      else if (i == CF_B_SG) v = Wt[OFF_SG_B];
      cf[i] = v;
    }
  }
}
// residual fragments f16(W - f16(W)) of the geometry branch's forward k-steps (fused_common.h, NKK_GEO): one block = one
// fragment of the (C, PK_LO_BYTES) image
__device__ __forceinline__ void pack_lo_block(const float* __restrict__ Wt, unsigned char* __restrict__ lo, int kk) {
  _Float16* out = reinterpret_cast<_Float16*>(lo + (size_t)kk * FRAG_BYTES);
  for (int e = threadIdx.x; e < 64 * 8; e += 256) {
    const int lane = e >> 3, j = e & 7, r = lane & 31, h = lane >> 5;
    const float v = fwd_elem(Wt, kk, r, h, j);
    out[lane * 8 + j] = (_Float16)(v - (float)(_Float16)v);
  }
}
__global__ __launch_bounds__(256) void pack_lo_kernel(const float* __restrict__ trunk, unsigned char* __restrict__ lo) {
  const int c = blockIdx.y;
  pack_lo_block(trunk + (size_t)c * TRUNK, lo + (size_t)c * PK_LO_BYTES, blockIdx.x);
}
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ trunk, unsigned char* __restrict__ packed) {
  const int c = blockIdx.y;
  pack_block(trunk + (size_t)c * TRUNK, packed + (size_t)c * PK_BYTES, blockIdx.x);
}

// Parameter-only work of one train step in ONE launch (three independent jobs side by side in the grid instead of
// three launches in a row): operand image of the trunk (pack_block), per-object latent rows (latent_fwd_block),
// and the zero fill of the gradient buffers.  grid (NPACK + 4 n_obj + nzero, C).
// With sample blocks appended (cnr_step_prologue) the sampler -- which depends on the ray pool and the step state
// only -- runs beside them as well: grid (NPACK + 4 n_obj + nzero + ceil(R / 4), C), one wavefront per ray.
#ifdef CNR_PREP_STAMPS  // tools/exp only: earliest start / latest end per job of the prologue launch (100 MHz counter)
__device__ unsigned long long g_prep_t[10];  // [job 0..3][min start, max end], [8] = min start over all
#define PREP_T0() const unsigned long long tt0 = __builtin_amdgcn_s_memrealtime(); \
  if (threadIdx.x == 0) atomicMin(&g_prep_t[8], tt0)
#define PREP_T1(job) do { __syncthreads(); if (threadIdx.x == 0) { atomicMin(&g_prep_t[2 * (job)], tt0); \
  atomicMax(&g_prep_t[2 * (job) + 1], __builtin_amdgcn_s_memrealtime()); } } while (0)
extern "C" int cnr_prep_stamps(unsigned long long* host, int reset) {
  if (reset) {
    unsigned long long init[10] = {~0ull, 0, ~0ull, 0, ~0ull, 0, ~0ull, 0, ~0ull, 0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_prep_t), init, sizeof(init));
  }
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_prep_t), sizeof(unsigned long long) * 10);
}
__device__ unsigned long long g_fr_t[8];  // field_fwd_render, block 0 / wave 0: phase boundaries (shader clock)
#define FR_T(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_fr_t[k] = (unsigned long long)__builtin_readcyclecounter(); } while (0)
extern "C" int cnr_fwd_render_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fr_t), sizeof(unsigned long long) * 8);
}
#else
#define PREP_T0() do {} while (0)
#define PREP_T1(job) do {} while (0)
#define FR_T(k) do {} while (0)
#endif

__global__ __launch_bounds__(256) void param_prep_kernel(const float* __restrict__ theta, cnr::FlatLayout lay,
                                                         int64_t off_trunk, unsigned char* __restrict__ packed,
                                                         float* __restrict__ zl, float* __restrict__ biasrows,
                                                         float* __restrict__ zero_buf, int64_t zero_count, int nzero,
                                                         cnr_sample::SampleArgs sa, int nsample,
                                                         unsigned char* __restrict__ packed_lo) {
  constexpr int NPACK = NKK_FWD + NKK_BWD + 1;
  const int c = blockIdx.y, C = gridDim.y;
  const float* th = theta + (int64_t)c * lay.stride;
  int b = blockIdx.x;
  PREP_T0();
  if (b < NPACK) { pack_block(th + off_trunk, packed + (size_t)c * PK_BYTES, b); PREP_T1(0); return; }
  b -= NPACK;
  if (packed_lo) {   // + the residual image of the geometry branch (grid grown by NKK_GEO blocks)
    if (b < NKK_GEO) { pack_lo_block(th + off_trunk, packed_lo + (size_t)c * PK_LO_BYTES, b); PREP_T1(0); return; }
    b -= NKK_GEO;
  }
  if (b < 4 * lay.n_obj) { cnr::latent_fwd_block(th, th + off_trunk, lay, zl, biasrows, b >> 2, b & 3, c); PREP_T1(1); return; }
  b -= 4 * lay.n_obj;
  if (b >= nzero) {  // a2-a5: four rays of class c per block
    const int r = (b - nzero) * 4 + (threadIdx.x >> 6);
    if (nsample > 0 && r < sa.R) cnr_sample::sample_ray(sa, (int64_t)c * sa.R + r, threadIdx.x & 63);
    PREP_T1(3);
    return;
  }
  // zero fill: block (b, c) of nzero * C takes a contiguous float4 range
  const int64_t nvec = zero_count >> 2, nblk = (int64_t)nzero * C, me = (int64_t)c * nzero + b;
  const int64_t per = (nvec + nblk - 1) / nblk;
  const int64_t lo = me * per, hi = lo + per < nvec ? lo + per : nvec;
  f4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) reinterpret_cast<f4*>(zero_buf)[i] = z4;
  if (me == 0 && threadIdx.x < (zero_count & 3)) zero_buf[(nvec << 2) + threadIdx.x] = 0.0f;
  PREP_T1(2);
}

}  // namespace

namespace {

// One 32-sample tile through PE + the ten layers: this lane's sample is (t0, t1, t2) (already / scale), brow its ray's
// effective bias rows.  Returns the sigma logit BEFORE the x10 (valid in every lane) and leaves the three colour logits in
// acc[0..2] of lane half 0.  SPLIT: the geometry branch (encoding_xyz .. encoding_shape) as three products per fragment,
// Wh xh + Wl xh + Wh xl with the residual weight image smem_lo (fused_common.h, NKK_GEO); the colour branch plain f16.
// (the order of the products below is the order of the one-launch kernel, fused_bwd_pipe8.hip: the two forwards then
//  agree bit for bit, which tests/test_trainer_gpu.py relies on)
template <bool SPLIT>
__device__ __forceinline__ f16v pe_products(const unsigned char* smem, const unsigned char* smem_lo, int kk0, int lane,
                                            const h8 (&E1f)[6], const h8 (&E1l)[6], f16v acc) {
  h8 w[6];
#pragma unroll
  for (int s = 0; s < 6; ++s) w[s] = lds_frag(smem, kk0 + s, lane);
#pragma unroll
  for (int s = 0; s < 6; ++s) {   // (fragment by fragment, as the one-launch kernel pipelines them with the encoding)
    acc = MFMA(w[s], E1f[s], acc);
    if (SPLIT) {
      acc = MFMA(w[s], E1l[s], acc);
      acc = MFMA(lds_frag(smem_lo, kk0 + s, lane), E1f[s], acc);
    }
  }
  return acc;
}
// a 32-wide hidden layer of the geometry branch: input = the previous layer's accumulators (ReLU), start = bias row
template <bool SPLIT>
__device__ __forceinline__ f16v hidden_layer(const unsigned char* smem, const unsigned char* smem_lo, int kk0, int lane,
                                             const f16v& prev, const f16v& init) {
  const h8 w0 = lds_frag(smem, kk0, lane), w1 = lds_frag(smem, kk0 + 1, lane);
  const h8 xa = SPLIT ? pack8_relu32(prev, 0) : pack8(prev, 0, true), xb = SPLIT ? pack8_relu32(prev, 1) : pack8(prev, 1, true);
  f16v o = MFMA(w0, xa, init);
  o = MFMA(w1, xb, o);
  if (SPLIT) {
    o = MFMA(lds_frag(smem_lo, kk0, lane), xa, o);
    o = MFMA(lds_frag(smem_lo, kk0 + 1, lane), xb, o);
    const h8 la = pack8_lo(prev, 0, true, xa), lb = pack8_lo(prev, 1, true, xb);
    o = MFMA(w0, la, o);
    o = MFMA(w1, lb, o);
  }
  return o;
}
template <bool SPLIT>
__device__ __forceinline__ float forward_tile(const unsigned char* smem, const unsigned char* smem_lo, const float* cf, const float (&Bh)[33],
                                              float t0, float t1, float t2, const float* __restrict__ brow, int lane,
                                              int h, f16v& acc_out) {
  h8 E1f[6], E2f[3], E1l[6];
  pe_slots<false, SPLIT>(Bh, t0, t1, t2, h, E1f, E2f, E1l);
  // L0 encoding_xyz, and the e1 part of L2 cat_layer (its own accumulator, started from the cat bias row): the E1 operands die here
  f16v acc = pe_products<SPLIT>(smem, smem_lo, KK_XYZ, lane, E1f, E1l, acc_init(cf + CF_B_XYZ, h));
  const f16v catp = pe_products<SPLIT>(smem, smem_lo, KK_CAT + 2, lane, E1f, E1l, acc_init(brow + 1 * 32, h));
  // L1 shape_layer_1 (latent slot 0 folded into the bias row)
  acc = hidden_layer<SPLIT>(smem, smem_lo, KK_S1, lane, acc, acc_init(brow + 0 * 32, h));
  // L2 cat_layer: [a1 | e1]
  acc = hidden_layer<SPLIT>(smem, smem_lo, KK_CAT, lane, acc, catp);
  // L3 shape_layer_2
  acc = hidden_layer<SPLIT>(smem, smem_lo, KK_S2, lane, acc, acc_init(brow + 2 * 32, h));
  // L4 encoding_shape (no activation on its output)
  acc = hidden_layer<SPLIT>(smem, smem_lo, KK_ES, lane, acc, acc_init(cf + CF_B_ES, h));
  h8 H0, H1;
  // sigma head in fp32 on the VALU: raw = w_sigma . y4 + b
  float raw;
  {
    const f16v ws = acc_init(cf + CF_W_SG, h);
    float part = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) part = fmaf(ws[i], acc[i], part);
    raw = part + __shfl_xor(part, 32, 64) + cf[CF_B_SG];
  }
  H0 = pack8(acc, 0, false); H1 = pack8(acc, 1, false);
  // L6 encoding_viewdir: [y4 | e2]   (colour branch: plain f16 from here on)
  acc = acc_init(cf + CF_B_VD, h);
  acc = MFMA(lds_frag(smem, KK_VD + 0, lane), H0, acc);
  acc = MFMA(lds_frag(smem, KK_VD + 1, lane), H1, acc);
#pragma unroll
  for (int s = 0; s < 3; ++s) acc = MFMA(lds_frag(smem, KK_VD + 2 + s, lane), E2f[s], acc);
  H0 = pack8(acc, 0, true); H1 = pack8(acc, 1, true);
  // L7 texture_layer_1
  acc = acc_init(brow + 3 * 32, h);
  acc = MFMA(lds_frag(smem, KK_T1 + 0, lane), H0, acc);
  acc = MFMA(lds_frag(smem, KK_T1 + 1, lane), H1, acc);
  H0 = pack8(acc, 0, true); H1 = pack8(acc, 1, true);
  // L8 rgb.0 (16 outputs = rows 0..15 = registers 0..7)
  acc = acc_init(cf + CF_B_R0, h);
  acc = MFMA(lds_frag(smem, KK_R0 + 0, lane), H0, acc);
  acc = MFMA(lds_frag(smem, KK_R0 + 1, lane), H1, acc);
  H0 = pack8(acc, 0, true);
  // L9 rgb.2 (3 outputs = rows 0..2 = registers 0..2 of half 0)
  acc = acc_init(cf + CF_B_R2, h);
  acc = MFMA(lds_frag(smem, KK_R2, lane), H0, acc);
  acc_out = acc;
  return raw;
}

constexpr int LDS_LO_OFF = (PK_OFF_BWD + 66 * 4 + 8 + 15) / 16 * 16;  // residual fragments behind the [2][33] B rows
// stage the residual fragments of class c (split-weight mode)
__device__ __forceinline__ void stage_lo(unsigned char* smem, const unsigned char* __restrict__ packed_lo, int c) {
  const unsigned char* src = packed_lo + (size_t)c * PK_LO_BYTES;
  for (int i = threadIdx.x * 16; i < PK_LO_BYTES; i += 256 * 16)
    *reinterpret_cast<f4*>(smem + LDS_LO_OFF + i) = *reinterpret_cast<const f4*>(src + i);
}

template <bool SPLIT>
__global__ __launch_bounds__(256, 4) void field_fwd_kernel(
    const float* __restrict__ pts, const float* __restrict__ Bdir, const unsigned char* __restrict__ packed,
    const float* __restrict__ biasrows, const int* __restrict__ ray_row, float inv_scale,
    float* __restrict__ sigmas, float* __restrict__ rgbs, int64_t N /* samples per class */, int S, int R,
    int64_t B_stride, const unsigned char* __restrict__ packed_lo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int c = blockIdx.y;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, h = lane >> 5, col = lane & 31;
  if (SPLIT) stage_lo(smem, packed_lo, c);
  // stage forward fragments + constants + this class's direction matrix into LDS
  {
    const unsigned char* src = packed + (size_t)c * PK_BYTES;
    for (int i = threadIdx.x * 16; i < PK_OFF_BWD; i += 256 * 16)
      *reinterpret_cast<f4*>(smem + i) = *reinterpret_cast<const f4*>(src + i);
    float* Bl = reinterpret_cast<float*>(smem + PK_OFF_BWD);  // [2][33] per-half rows
    for (int i = threadIdx.x; i < 66; i += 256) {
      const int hh = i / 33, k = i % 33, d = k / 3;
      Bl[i] = (hh == 1 && d == 10) ? 0.0f : Bdir[(size_t)c * B_stride + (11 * hh + d) * 3 + (k % 3)];
    }
  }
  __syncthreads();
  const float* cf = reinterpret_cast<const float*>(smem + PK_OFF_CONST);
  float Bh[33];
  {
    const float* Bl = reinterpret_cast<const float*>(smem + PK_OFF_BWD) + 33 * h;
#pragma unroll
    for (int i = 0; i < 33; ++i) Bh[i] = Bl[i];
  }
  const int64_t ntiles = (N + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    // the weight fragments are loop-invariant LDS loads: stop LICM from hoisting 120 VGPRs of them
    asm volatile("" ::: "memory");
    const int64_t n = tile * 32 + col;
    const bool live = n < N;
    const int64_t nc = live ? n : N - 1;
    const int64_t gs = (int64_t)c * N + nc;
    const float* pp = pts + gs * 3;
    const float t0 = pp[0] * inv_scale, t1 = pp[1] * inv_scale, t2 = pp[2] * inv_scale;
    const int64_t ray = (int64_t)c * R + nc / S;
    const int64_t row = ray_row ? (int64_t)ray_row[ray] : ray;
    const float* brow = biasrows + row * (CNR_NLAT * 32);

    f16v acc;
    const float raw = forward_tile<SPLIT>(smem, smem + LDS_LO_OFF, cf, Bh, t0, t1, t2, brow, lane, h, acc);
    if (live && h == 0) {
      sigmas[gs] = raw * 10.0f;
      float* o = rgbs + gs * 3;
      o[0] = 1.0f / (1.0f + __expf(-acc[0]));
      o[1] = 1.0f / (1.0f + __expf(-acc[1]));
      o[2] = 1.0f / (1.0f + __expf(-acc[2]));
    }
  }
}

// Mask counts of every class, one block per class: out[c][3] = #(depth mask & this-object-or-other), #(label != 0),
// #(label != 2).  cnr_field_fwd_render launches it first when the counts do not fit its in-register path (many classes
// x many rays): every one of its blocks would otherwise scan all classes' masks itself (C^2 R work: 604 us instead of
// 290 us at 8 classes x 4096 rays x 128 samples).
__global__ __launch_bounds__(256) void mask_counts_kernel(const uint8_t* __restrict__ labels,
                                                          const uint8_t* __restrict__ depth_mask, int R,
                                                          float* __restrict__ out) {
  __shared__ float cnt[12];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float a = 0.f, b = 0.f, d = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) {
    const uint8_t lab = labels[(size_t)c * R + r];
    const bool mo = lab != 0, ms = lab != 2, md = depth_mask[(size_t)c * R + r] != 0;
    a += (md && mo) ? 1.f : 0.f; b += mo ? 1.f : 0.f; d += ms ? 1.f : 0.f;
  }
  a = cnr::wave_sum(a); b = cnr::wave_sum(b); d = cnr::wave_sum(d);
  if (lane == 0) { cnt[wv] = a; cnt[4 + wv] = b; cnt[8 + wv] = d; }
  __syncthreads();
  if (threadIdx.x < 3)
    out[c * 3 + threadIdx.x] = (cnt[threadIdx.x * 4] + cnt[threadIdx.x * 4 + 1]) + (cnt[threadIdx.x * 4 + 2] + cnt[threadIdx.x * 4 + 3]);
}

// a8-a15 for the fused trainer in ONE launch: the field forward of every sample of a ray, the alpha composite, the
// loss gradient w.r.t. the ray's renders and the composite backward, with the per-sample sigma / colour never leaving
// registers (cnr_field_fwd + cnr_render_loss write and re-read 16 B per sample between two launches).  A wave owns
// whole rays: K = S / 32 consecutive tiles, lane (half 0) = sample; the exclusive-cumprod runs as a 32-lane scan per
// tile with the transmittance carried from tile to tile, the backward's suffix sums the other way.  Same expressions as
// render_loss.hip; sums over lanes run over 32-lane tiles instead of 64-lane chunks, so the results agree to summation
// order, not bitwise.  grid (blocks, C); block b of class c writes loss partial (c, b) -- nb = gridDim.x for
// finish_class.
template <int K, bool SPLIT>
__global__ __launch_bounds__(256, 2) void field_fwd_render_kernel(
    const float* __restrict__ pts, const float* __restrict__ Bdir, const unsigned char* __restrict__ packed,
    const float* __restrict__ biasrows, const int* __restrict__ ray_row, float inv_scale, const float* __restrict__ z,
    const float* __restrict__ gt_depth, const float* __restrict__ gt_rgb, const uint8_t* __restrict__ labels,
    const uint8_t* __restrict__ depth_mask, float color_scaling, float opacity_scaling, float grad_scale,
    float* __restrict__ d_sigmas, float* __restrict__ d_colors, float* __restrict__ depth_out,
    float* __restrict__ var_out, float* __restrict__ rgb_out, float* __restrict__ opacity_out, int C, int R,
    int64_t B_stride, float* __restrict__ partials, const unsigned char* __restrict__ packed_lo,
    const float* __restrict__ counts_ext, const float* __restrict__ counts_tab, const int64_t* __restrict__ d_state) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float cnt[12];
  __shared__ float cntw[4][3 * 16];  // per wave, per class: the three mask counts (register path below)
  FR_T(0);
  constexpr int S = 32 * K;
  const int c = blockIdx.y;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, h = lane >> 5, col = lane & 31;
  // ---- mask counts of every class (the empty-mask rule of render_rays.py:67-72 couples the classes).  Every block needs
  // them, but only for the loss gradients: their loads go out FIRST, four rays per 32-bit word, and are counted after
  // the weight copy below has landed -- the scan used to be a memory round trip of its own behind that copy (5.7 k of
  // the 28 k cycles a block lives).  Register path: C * ceil(R / 1024) <= 8 words per array and thread, R % 4 == 0,
  // C <= 16; anything else takes the loop further down.
  constexpr int MAXW = 8;
  const int nw = (R + 1023) / 1024;
  const bool counts_in_regs = !counts_tab && !counts_ext && C * nw <= MAXW && (R & 3) == 0 && C <= 16;
  unsigned int lw[MAXW], mw[MAXW];
  if (counts_in_regs) {
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
      const int cc = i / nw, r4 = ((i % nw) * 256 + (int)threadIdx.x) * 4;
      const bool on = i < C * nw && r4 < R;
      lw[i] = on ? *reinterpret_cast<const unsigned int*>(labels + (size_t)cc * R + r4) : 0u;
      mw[i] = on ? *reinterpret_cast<const unsigned int*>(depth_mask + (size_t)cc * R + r4) : 0xffffffffu;
      if (!on) lw[i] = 0xffffffffu;  // marks "no rays here" (labels are 0, 1 or 2)
    }
  }
  if (SPLIT) stage_lo(smem, packed_lo, blockIdx.y);
  {
    const unsigned char* src = packed + (size_t)c * PK_BYTES;
    for (int i = threadIdx.x * 16; i < PK_OFF_BWD; i += 256 * 16)
      *reinterpret_cast<f4*>(smem + i) = *reinterpret_cast<const f4*>(src + i);
    float* Bl = reinterpret_cast<float*>(smem + PK_OFF_BWD);
    for (int i = threadIdx.x; i < 66; i += 256) {
      const int hh = i / 33, k = i % 33, d = k / 3;
      Bl[i] = (hh == 1 && d == 10) ? 0.0f : Bdir[(size_t)c * B_stride + (11 * hh + d) * 3 + (k % 3)];
    }
  }
  FR_T(1);
  bool empty_d = false, empty_c = false, empty_o = false;
  float nd = 0.f, nc = 0.f, no = 0.f;
  if (counts_in_regs) {
    for (int cc = 0; cc < C; ++cc) {
      float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
      for (int i = 0; i < MAXW; ++i) {
        if (i / nw != cc || lw[i] == 0xffffffffu) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned int lab = (lw[i] >> (8 * j)) & 0xffu, dm = (mw[i] >> (8 * j)) & 0xffu;
          const bool mo = lab != 0, ms = lab != 2, md = dm != 0;
          a += (md && mo) ? 1.f : 0.f; b += mo ? 1.f : 0.f; d += ms ? 1.f : 0.f;
        }
      }
      a = cnr::wave_sum(a); b = cnr::wave_sum(b); d = cnr::wave_sum(d);  // counts: exact in fp32 in any order
      if (lane == 0) { cntw[wv][cc * 3 + 0] = a; cntw[wv][cc * 3 + 1] = b; cntw[wv][cc * 3 + 2] = d; }
    }
    __syncthreads();  // also: the weight fragments are in LDS
    for (int cc = 0; cc < C; ++cc) {
      const float a = (cntw[0][cc * 3 + 0] + cntw[1][cc * 3 + 0]) + (cntw[2][cc * 3 + 0] + cntw[3][cc * 3 + 0]);
      const float b = (cntw[0][cc * 3 + 1] + cntw[1][cc * 3 + 1]) + (cntw[2][cc * 3 + 1] + cntw[3][cc * 3 + 1]);
      const float d = (cntw[0][cc * 3 + 2] + cntw[1][cc * 3 + 2]) + (cntw[2][cc * 3 + 2] + cntw[3][cc * 3 + 2]);
      empty_d |= (a == 0.f); empty_c |= (b == 0.f); empty_o |= (d == 0.f);
      if (cc == c) { nd = a; nc = b; no = d; }
    }
  } else if (counts_tab) {  // per-epoch table (cnr_slice_maskcounts): this slice's counts and the any-class-empty flags,
                            // possibly summed / or-ed over GPUs by the host -- the kernel never looks at other classes
    const float* t = counts_tab + (size_t)(d_state ? d_state[0] / R : 0) * (size_t)(C + 1) * 4;
    nd = t[c * 4 + 0]; nc = t[c * 4 + 1]; no = t[c * 4 + 2];
    empty_d = t[C * 4 + 0] != 0.f; empty_c = t[C * 4 + 1] != 0.f; empty_o = t[C * 4 + 2] != 0.f;
    __syncthreads();  // the weight fragments are in LDS
  } else if (counts_ext) {  // counted by mask_counts_kernel just before this launch
    for (int cc = 0; cc < C; ++cc) {
      const float a = counts_ext[cc * 3 + 0], b = counts_ext[cc * 3 + 1], d = counts_ext[cc * 3 + 2];
      empty_d |= (a == 0.f); empty_c |= (b == 0.f); empty_o |= (d == 0.f);
      if (cc == c) { nd = a; nc = b; no = d; }
    }
    __syncthreads();  // the weight fragments are in LDS
  } else {
    for (int cc = 0; cc < C; ++cc) {
      float a = 0.f, b = 0.f, d = 0.f;
      for (int r = threadIdx.x; r < R; r += 256) {
        const uint8_t lab = labels[(size_t)cc * R + r];
        const bool mo = lab != 0, ms = lab != 2, md = depth_mask[(size_t)cc * R + r] != 0;
        a += (md && mo) ? 1.f : 0.f; b += mo ? 1.f : 0.f; d += ms ? 1.f : 0.f;
      }
      a = cnr::wave_sum(a); b = cnr::wave_sum(b); d = cnr::wave_sum(d);
      __syncthreads();
      if (lane == 0) { cnt[wv] = a; cnt[4 + wv] = b; cnt[8 + wv] = d; }
      __syncthreads();
      a = (cnt[0] + cnt[1]) + (cnt[2] + cnt[3]); b = (cnt[4] + cnt[5]) + (cnt[6] + cnt[7]);
      d = (cnt[8] + cnt[9]) + (cnt[10] + cnt[11]);
      empty_d |= (a == 0.f); empty_c |= (b == 0.f); empty_o |= (d == 0.f);
      if (cc == c) { nd = a; nc = b; no = d; }
    }
    __syncthreads();
  }
  FR_T(2);
  const float wd = empty_d ? 0.f : 1.0f / (nd + 1e-10f);
  const float wc = empty_c ? 0.f : 1.0f / (nc + 1e-10f);
  const float wo = empty_o ? 0.f : 1.0f / (no + 1e-10f);
  const float* cf = reinterpret_cast<const float*>(smem + PK_OFF_CONST);
  float Bh[33];
  {
    const float* Bl = reinterpret_cast<const float*>(smem + PK_OFF_BWD) + 33 * h;
#pragma unroll
    for (int i = 0; i < 33; ++i) Bh[i] = Bl[i];
  }
  auto sgn = [](float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); };
  // 32-lane scans on lane half 0 (half 1 holds the neutral element, so the 64-lane shuffles below are 32-lane scans)
  auto incl_prod32 = [&](float v) {
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { const float p = __shfl_up(v, o, 64); if (col >= o && h == 0) v *= p; }
    return v;
  };
  auto incl_suffix32 = [&](float v) {
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { const float p = __shfl_down(v, o, 64); if (col + o < 32 && h == 0) v += p; }
    return v;
  };
  float ld = 0.f, lc = 0.f, lo = 0.f;
  for (int r = blockIdx.x * 4 + wv; r < R; r += gridDim.x * 4) {
    asm volatile("" ::: "memory");
    const int64_t ray = (int64_t)c * R + r;
    const int64_t row = ray_row ? (int64_t)ray_row[ray] : ray;
    const float* brow = biasrows + row * (CNR_NLAT * 32);
    float occ[K], T[K], zz[K], c0[K], c1[K], c2[K];
    float carry = 1.0f, sd = 0.f, so = 0.f, sr = 0.f, sg = 0.f, sb = 0.f;
#pragma unroll
    for (int t = 0; t < K; ++t) {
      // (unrolled: occ[t] ... stay in registers; the clobber keeps the K copies of the forward from sharing -- and
      //  keeping alive -- each other's weight-fragment loads)
      asm volatile("" ::: "memory");
      const int64_t gs = ray * S + t * 32 + col;
      const float* pp = pts + gs * 3;
      f16v acc;
      const float raw = forward_tile<SPLIT>(smem, smem + LDS_LO_OFF, cf, Bh, pp[0] * inv_scale, pp[1] * inv_scale,
                                            pp[2] * inv_scale, brow, lane, h, acc);
      // composite of this tile (lane half 0 = the 32 samples; half 1 neutral)
      occ[t] = h == 0 ? 1.0f / (1.0f + expf(-(raw * 10.0f))) : 0.0f;
      c0[t] = 1.0f / (1.0f + __expf(-acc[0])); c1[t] = 1.0f / (1.0f + __expf(-acc[1]));
      c2[t] = 1.0f / (1.0f + __expf(-acc[2]));
      zz[t] = z[gs];
      const float f = h == 0 ? (1.0f - occ[t] + 1e-10f) : 1.0f;
      const float incl = incl_prod32(f);
      float excl = __shfl_up(incl, 1, 64);
      if (col == 0) excl = 1.0f;
      T[t] = carry * excl;
      const float term = occ[t] * T[t];
      sd += term * zz[t]; so += term; sr += term * c0[t]; sg += term * c1[t]; sb += term * c2[t];
      carry *= __shfl(incl, 31, 64);
      FR_T(3 + t);
    }
    sd = cnr::wave_sum(sd); so = cnr::wave_sum(so);
    sr = cnr::wave_sum(sr); sg = cnr::wave_sum(sg); sb = cnr::wave_sum(sb);
    float sv = 0.f;
#pragma unroll
    for (int t = 0; t < K; ++t) { const float dz = zz[t] - sd; sv += occ[t] * T[t] * dz * dz; }
    sv = cnr::wave_sum(sv);
    if (lane == 0) {
      if (depth_out) depth_out[ray] = sd;
      if (var_out) var_out[ray] = sv;
      if (opacity_out) opacity_out[ray] = so;
      if (rgb_out) { rgb_out[ray * 3 + 0] = sr; rgb_out[ray * 3 + 1] = sg; rgb_out[ray * 3 + 2] = sb; }
    }
    // ---- losses of this ray and their gradients w.r.t. its renders (loss.hip, same expressions) --------------------
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
    // ---- composite backward, last tile first, suffix sum carried across tiles ---------------------------------------
    float suf_carry = 0.0f;
#pragma unroll
    for (int t = K - 1; t >= 0; --t) {
      const float term = occ[t] * T[t];
      const float g = dD * zz[t] + dR * c0[t] + dG * c1[t] + dBl * c2[t] + dO;
      const float tg = h == 0 ? term * g : 0.0f;
      const float incl_suf = incl_suffix32(tg);
      const float suf = (incl_suf - tg) + suf_carry;
      if (h == 0) {
        const int64_t gs = ray * S + t * 32 + col;
        const float f = 1.0f - occ[t] + 1e-10f;
        const float docc = T[t] * g - suf / f;
        d_sigmas[gs] = docc * occ[t] * (1.0f - occ[t]);
        float* dc = d_colors + gs * 3;
        dc[0] = term * dR; dc[1] = term * dG; dc[2] = term * dBl;
      }
      suf_carry += __shfl(incl_suf, 0, 64);
    }
  }
  FR_T(7);
  // ---- loss values: block partials for finish_class -------------------------------------------------------------------
  __syncthreads();
  if (lane == 0) { cnt[wv] = ld; cnt[4 + wv] = lc; cnt[8 + wv] = lo; }
  __syncthreads();
  const int nb = gridDim.x;
  if (threadIdx.x < 3)
    partials[((size_t)c * nb + blockIdx.x) * 3 + threadIdx.x] =
        (cnt[threadIdx.x * 4] + cnt[threadIdx.x * 4 + 1]) + (cnt[threadIdx.x * 4 + 2] + cnt[threadIdx.x * 4 + 3]);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float* hdr = partials + (size_t)C * nb * 3 + (size_t)c * 4;
    hdr[0] = wd; hdr[1] = wc; hdr[2] = wo;
    hdr[3] = (float)((empty_d ? 2 : 0) | (empty_c ? 4 : 0) | (empty_o ? 8 : 0));
  }
}
}  // namespace

extern "C" int64_t cnr_pack_bytes(void) { return (int64_t)fz::PK_BYTES; }

extern "C" int cnr_pack_weights(const float* trunk, void* packed, int C, void* stream) {
  if (!trunk || !packed || C <= 0) return CNR_E_ARG;
  if (((uintptr_t)packed & 15) != 0) return CNR_E_ALIGN;
  dim3 grid(fz::NKK_FWD + fz::NKK_BWD + 1, (unsigned)C);
  hipLaunchKernelGGL(pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, trunk, (unsigned char*)packed);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

// blocks per class of cnr_field_fwd_render (= loss partials per class: what cnr_render_loss_finish / cnr_step_tail
// need to know), 0 when the shape is not supported (S must be 32, 64, 96 or 128)
extern "C" int cnr_field_fwd_render_blocks(int R, int S) {
  if (R <= 0 || (S != 32 && S != 64 && S != 96 && S != 128)) return 0;
  const int64_t b = ((int64_t)R + 3) / 4;
  return (int)(b > 2048 ? 2048 : b);
}
extern "C" int64_t cnr_field_fwd_render_workspace_bytes(int C, int R, int S) {
  const int nb = cnr_field_fwd_render_blocks(R, S);
  return nb ? ((int64_t)C * nb * 3 + (int64_t)C * 4 + (int64_t)C * 3) * (int64_t)sizeof(float) : 0;  // + mask counts
}

extern "C" int cnr_field_fwd_render(const float* pts, const float* B, const void* packed, const float* biasrows,
                                    const int* ray_row, float scale, const float* z, const float* gt_depth,
                                    const float* gt_rgb, const uint8_t* labels, const uint8_t* depth_mask,
                                    float color_scaling, float opacity_scaling, float grad_scale, float* d_sigmas,
                                    float* d_colors, float* depth, float* var, float* rgb, float* opacity, int C, int R,
                                    int S, int64_t B_stride, void* workspace, int64_t workspace_bytes,
                                    const void* packed_lo, const float* counts_tab, const int64_t* d_state,
                                    void* stream) {
  if (!pts || !B || !packed || !biasrows || !z || !gt_depth || !gt_rgb || !labels || !depth_mask || !d_sigmas ||
      !d_colors || !workspace || C <= 0 || R <= 0 || !(scale > 0.f))
    return CNR_E_ARG;
  const int nb = cnr_field_fwd_render_blocks(R, S);
  if (!nb) return CNR_E_SHAPE;
  if (workspace_bytes < cnr_field_fwd_render_workspace_bytes(C, R, S)) return CNR_E_ARG;
  if (((uintptr_t)packed & 15) != 0 || ((uintptr_t)biasrows & 15) != 0) return CNR_E_ALIGN;
  const size_t lds = packed_lo ? (size_t)LDS_LO_OFF + fz::PK_LO_BYTES
                               : (size_t)fz::PK_OFF_BWD + 66 * sizeof(float) + 8;
  dim3 grid((unsigned)nb, (unsigned)C);
  // mask counts: inside the kernel from registers when they fit (few classes / rays), else by a one-block-per-class
  // kernel into the workspace's tail
  const float* counts_ext = nullptr;
  if (!counts_tab && !(C * ((R + 1023) / 1024) <= 8 && (R & 3) == 0 && C <= 16)) {
    float* cx = (float*)workspace + (size_t)C * nb * 3 + (size_t)C * 4;
    hipLaunchKernelGGL(mask_counts_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, labels, depth_mask, R, cx);
    counts_ext = cx;
  }
#define CNR_FFR(KK, SP)                                                                                             \
  do {                                                                                                              \
    if (SP) {                                                                                                       \
      static cnr::DeviceOnce once;                                                                                  \
      const int er_ = cnr::set_max_dynamic_lds(once, (const void*)field_fwd_render_kernel<KK, SP>, (int)lds);       \
      if (er_) return er_;                                                                                          \
    }                                                                                                               \
    hipLaunchKernelGGL((field_fwd_render_kernel<KK, SP>), grid, dim3(256), lds, (hipStream_t)stream, pts, B,         \
                       (const unsigned char*)packed, biasrows, ray_row, 1.0f / scale, z, gt_depth, gt_rgb, labels,   \
                       depth_mask, color_scaling, opacity_scaling, grad_scale, d_sigmas, d_colors, depth, var, rgb,  \
                       opacity, C, R, B_stride > 0 ? B_stride : (int64_t)63, (float*)workspace,                      \
                       (const unsigned char*)packed_lo, counts_ext, counts_tab, d_state);                           \
  } while (0)
  if (packed_lo) {
    switch (S / 32) {
      case 1: CNR_FFR(1, true); break;
      case 2: CNR_FFR(2, true); break;
      case 3: CNR_FFR(3, true); break;
      default: CNR_FFR(4, true); break;
    }
  } else {
    switch (S / 32) {
      case 1: CNR_FFR(1, false); break;
      case 2: CNR_FFR(2, false); break;
      case 3: CNR_FFR(3, false); break;
      default: CNR_FFR(4, false); break;
    }
  }
#undef CNR_FFR
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_param_prep(const float* theta, int64_t class_stride, int64_t off_trunk, int64_t off_latW,
                              int64_t off_latb, int64_t off_shape, int64_t off_tex, int L, int n_obj, int C,
                              void* packed, float* zl, float* biasrows, float* zero_buf, int64_t zero_count,
                              void* stream) {
  if (!theta || !packed || !zl || !biasrows || L <= 0 || n_obj <= 0 || C <= 0 || zero_count < 0 ||
      (zero_count > 0 && !zero_buf))
    return CNR_E_ARG;
  if (((uintptr_t)packed & 15) != 0 || ((uintptr_t)zero_buf & 15) != 0) return CNR_E_ALIGN;
  cnr::FlatLayout lay{class_stride, off_latW, off_latb, off_shape, off_tex, L, n_obj};
  int nzero = zero_count > 0 ? (int)((zero_count / 4 + 256 * 8 - 1) / (256 * 8) / C) : 0;  // ~8 float4 per thread
  if (zero_count > 0 && nzero < 1) nzero = 1;
  if (nzero > 256) nzero = 256;
  dim3 grid(fz::NKK_FWD + fz::NKK_BWD + 1 + 4 * n_obj + nzero, (unsigned)C);
  hipLaunchKernelGGL(param_prep_kernel, grid, dim3(256), 0, (hipStream_t)stream, theta, lay, off_trunk,
                     (unsigned char*)packed, zl, biasrows, zero_buf, zero_count, nzero, cnr_sample::SampleArgs{}, 0,
                     (unsigned char*)nullptr);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_step_prologue(const cnr_step_prologue_args* a, void* stream) {
  if (!a || a->struct_size != sizeof(cnr_step_prologue_args) || a->abi_version != CNR_ABI_VERSION) return CNR_E_ARG;
  const int L = a->L, n_obj = a->n_obj, C = a->C, R = a->R, n1 = a->n1, n2 = a->n2;
  if (a->rng_c0 < 0 || a->rng_cstride < 0 || a->rng_R < 0 || a->rng_r0 < 0 || (a->rng_R > 0 && a->rng_r0 + R > a->rng_R))
    return CNR_E_ARG;
  if (a->max_bound_slices < 0 || (a->max_bound_slices > 1 && (a->pool_rows <= 0 || !a->d_state))) return CNR_E_ARG;
  if (!a->theta || !a->packed || !a->zl || !a->biasrows || L <= 0 || n_obj <= 0 || C <= 0 || a->zero_count < 0 ||
      (a->zero_count > 0 && !a->zero_buf))
    return CNR_E_ARG;
  if (((uintptr_t)a->packed & 15) != 0 || ((uintptr_t)a->zero_buf & 15) != 0 || ((uintptr_t)a->packed_lo & 15) != 0)
    return CNR_E_ALIGN;
  const bool sample = a->rgbs != nullptr;   // rgbs = NULL: the parameter-only jobs, no sampler blocks (the rays come from elsewhere)
  if (sample) {
    if (!a->depth || !a->dirs_c || !a->T || !a->max_bound || !a->z || !a->pts || !a->gt_rgb || !a->depth_mask || !a->labels)
      return CNR_E_ARG;
    if (R <= 0 || n1 < 0 || n2 <= 0) return CNR_E_ARG;
    if (n2 > 128) return CNR_E_SHAPE;
    if ((a->u == nullptr) != (a->g == nullptr)) return CNR_E_ARG;
    if (a->pool_rows < 0 || (a->pool_rows > 0 && (!a->d_state || a->pool_rows < R))) return CNR_E_ARG;
    if (a->ray_row && !a->pool_indices) return CNR_E_ARG;
    if (a->perm && a->pool_rows == 0) return CNR_E_ARG;
  }
  cnr::FlatLayout lay{a->class_stride, a->off_latW, a->off_latb, a->off_shape, a->off_tex, L, n_obj};
  int nzero = a->zero_count > 0 ? (int)((a->zero_count / 4 + 256 * 8 - 1) / (256 * 8) / C) : 0;
  if (a->zero_count > 0 && nzero < 1) nzero = 1;
  if (nzero > 256) nzero = 256;
  const int nsample = sample ? (R + 3) / 4 : 0;
  cnr_sample::SampleArgs sa{a->rgbs, a->depth, a->dirs_c, a->T, a->u, a->g, a->seed, a->offset, a->d_state, a->pool_rows,
                            a->max_bound, a->world_frame, C, R, n1, n2, a->eps, a->stop_eps, a->min_bound, a->z, a->pts,
                            a->origins, a->dirs_o, a->gt_rgb, a->gt_depth, a->depth_mask, a->labels, a->pool_indices, n_obj,
                            a->ray_row, a->perm, a->max_bound_slices, a->rng_c0, a->rng_cstride, a->rng_R, a->rng_r0};
  dim3 grid(fz::NKK_FWD + fz::NKK_BWD + 1 + (a->packed_lo ? fz::NKK_GEO : 0) + 4 * n_obj + nzero + nsample, (unsigned)C);
  hipLaunchKernelGGL(param_prep_kernel, grid, dim3(256), 0, (hipStream_t)stream, a->theta, lay, a->off_trunk,
                     (unsigned char*)a->packed, a->zl, a->biasrows, a->zero_buf, a->zero_count, nzero, sa, nsample,
                     (unsigned char*)a->packed_lo);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int64_t cnr_pack_lo_bytes() { return (int64_t)fz::PK_LO_BYTES; }
extern "C" int cnr_pack_weights_lo(const float* trunk, void* packed_lo, int C, void* stream) {
  if (!trunk || !packed_lo || C <= 0) return CNR_E_ARG;
  if (((uintptr_t)packed_lo & 15) != 0) return CNR_E_ALIGN;
  hipLaunchKernelGGL(pack_lo_kernel, dim3(fz::NKK_GEO, (unsigned)C), dim3(256), 0, (hipStream_t)stream, trunk,
                     (unsigned char*)packed_lo);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_field_fwd(const float* pts, const float* B, const void* packed, const float* biasrows,
                             const int* ray_row, float scale, float* sigmas, float* rgbs, int C, int R, int S,
                             int64_t B_stride, const void* packed_lo, void* stream) {
  if (!pts || !B || !packed || !biasrows || !sigmas || !rgbs || C <= 0 || R <= 0 || S <= 0 || !(scale > 0.f))
    return CNR_E_ARG;
  if (((uintptr_t)packed & 15) != 0 || ((uintptr_t)biasrows & 15) != 0) return CNR_E_ALIGN;
  const int64_t N = (int64_t)R * S;
  const int64_t ntiles = (N + 31) / 32;
  int64_t blocks = (ntiles + 3) / 4;
  const int64_t cap = 2048;     // workgroups per class (grid-stride over tiles beyond)
  if (blocks > cap) blocks = cap;
  const size_t lds = packed_lo ? (size_t)LDS_LO_OFF + fz::PK_LO_BYTES
                               : (size_t)fz::PK_OFF_BWD + 66 * sizeof(float) + 8;
  dim3 grid((unsigned)blocks, (unsigned)C);
  if (packed_lo) {
    static cnr::DeviceOnce once;
    const int er = cnr::set_max_dynamic_lds(once, (const void*)field_fwd_kernel<true>, (int)lds);
    if (er) return er;
    hipLaunchKernelGGL(field_fwd_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, pts, B,
                       (const unsigned char*)packed, biasrows, ray_row, 1.0f / scale, sigmas, rgbs, N, S, R,
                       B_stride > 0 ? B_stride : (int64_t)63, (const unsigned char*)packed_lo);
  } else {
    hipLaunchKernelGGL(field_fwd_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, pts, B,
                       (const unsigned char*)packed, biasrows, ray_row, 1.0f / scale, sigmas, rgbs, N, S, R,
                       B_stride > 0 ? B_stride : (int64_t)63, (const unsigned char*)nullptr);
  }
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}
