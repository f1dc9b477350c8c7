// Fused field backward: one launch turns per-sample dL/dsigma, dL/drgb into the gradients of the ten
// trunk layers (dtrunk), of the direction matrix (dB) and of the per-row effective biases (dbiasrows).
// It RECOMPUTES the forward per 32-sample tile (PE + 10 layers on f16 MFMA), so nothing but
// pts (12 B), d_sigma (4 B) and d_rgb (12 B) per sample is read from HBM -- the reference materialises
// ~2.9 KB of activations per sample for autograd.
//
// Three kinds of matrix work per tile (layout contract: fused_common.h):
//   1. forward chain        D = W X^T           A = packed W fragments,      B = activations (registers)
//   2. data-gradient chain  dX^T = W^T dPre^T   A = packed W^T fragments,    B = dPre (registers): the same
//                           "accumulator is the next B operand" trick, so the chain never touches LDS
//   3. weight gradients     dW = dPre^T X       a contraction over SAMPLES, i.e. over lanes: dPre and X go
//                           once through a wave-private LDS image [sample][feature] (ds_write_b64/b128) and
//                           come back transposed with ds_read_b64_tr_b16 as MFMA operands; the 17 dW blocks
//                           of 32x32 stay in 272 accumulator registers for the whole kernel.
// Bias gradients ride along: free "ones" slots of the PE images give db of encoding_xyz / viewdir / rgb.2;
// the other layers use (one-hot(ray slot))^T x dPre, which also yields the per-row sums for dbiasrows.
// One wave per SIMD (512-register budget), 4 waves per workgroup, no workgroup barrier inside the tile loop.
#include "fused_common.h"

namespace {
using namespace fz;
typedef short s4v __attribute__((ext_vector_type(4)));

constexpr int ST_H = 72;    // [32 samples][32 features] f16, 64 B + 8 B pad per row
constexpr int ST_E1 = 208;  // [32][half0: 48 slots | half1: 48 slots] = 192 B + 16 B pad
constexpr int ST_E2 = 112;  // [32][half0: 24 | half1: 24] = 96 B + 16 B pad
constexpr int XIMG_BYTES = 32 * ST_E1 + 128;
constexpr int DIMG_BYTES = 32 * ST_H + 64;
constexpr int WAVE_SCRATCH = XIMG_BYTES + DIMG_BYTES;
constexpr int LDS_BL = PK_BYTES;                      // 66 floats: per-half direction rows
constexpr int LDS_ROWTAB = LDS_BL + 272;              // [32 rows][4][32] f32 (when rows_per_class <= 32)
constexpr int LDS_PLAIN = LDS_ROWTAB + 32 * 128 * 4;  // db of encoding_shape, rgb.0: [2][32] f32
constexpr int LDS_SCRATCH = LDS_PLAIN + 256;
constexpr int LDS_TOTAL = LDS_SCRATCH + 4 * WAVE_SCRATCH;
static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
static_assert(TRUNK * 4 + 64 * 4 <= PK_BYTES + 272, "flush slab aliases the packed image");

enum BlockKind { BK_R2, BK_R0, BK_T1, BK_VD_Y, BK_VD_E0, BK_VD_E1, BK_ES, BK_S2, BK_CAT_Y, BK_CAT_E0, BK_CAT_E1,
                 BK_CAT_E2, BK_S1, BK_XYZ_E0, BK_XYZ_E1, BK_XYZ_E2, NBLOCKS };

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// [sample][feature] image of an accumulator-layout tile held as two f16 fragments
__device__ __forceinline__ void stage_h(unsigned char* img, const h8& f0, const h8& f1, int col, int h) {
  unsigned char* base = img + col * ST_H + h * 8;
  *reinterpret_cast<h4*>(base + 0) = f0.lo;
  *reinterpret_cast<h4*>(base + 16) = f0.hi;
  *reinterpret_cast<h4*>(base + 32) = f1.lo;
  *reinterpret_cast<h4*>(base + 48) = f1.hi;
}
// transposing read: operand fragment (feature = col0 + (lane & 31), k = sample 16 s + 8 h + j)
__device__ __forceinline__ h8 tr_frag(const unsigned char* img, int stride, int col0, int s, int lane) {
  const int i = lane & 15, g16 = lane >> 4, q = i >> 2, p = i & 3, hh = g16 >> 1;
  const unsigned char* a = img + (16 * s + 8 * hh + q) * stride + (col0 + 16 * (g16 & 1) + 4 * p) * 2;
  typedef __attribute__((address_space(3))) s4v* lds_s4;
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(a));
  const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(a + 4 * stride));
  h8 r;
  r.lo = __builtin_bit_cast(h4, lo);
  r.hi = __builtin_bit_cast(h4, hi);
  return r;
}
__device__ __forceinline__ f16v zero16() {
  f16v z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.0f;
  return z;
}
__device__ __forceinline__ h8 zero8() {
  h8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (_Float16)0;
  return z;
}
// dpre = upstream * (activation > 0) for the 8 accumulator registers of k-step s
__device__ __forceinline__ void mask8(f16v& a, int s, const h8& act) {
#pragma unroll
  for (int j = 0; j < 8; ++j) a[8 * s + j] = (act[j] > (_Float16)0) ? a[8 * s + j] : 0.0f;
}

// trunk-blob index of element (out row o, image column c) of a dW block ; -1 = not a parameter
__device__ __forceinline__ int block_index(int kind, int o, int c) {
  int f;
  switch (kind) {
    case BK_R2: if (o >= 3) return -1; return c < 16 ? OFF_R2_W + o * 16 + c : (c == 16 ? OFF_R2_B + o : -1);
    case BK_R0: return o < 16 ? OFF_R0_W + o * 32 + c : -1;
    case BK_T1: return OFF_T1_W + o * 32 + c;
    case BK_VD_Y: return OFF_VD_W + o * (32 + E2) + c;
    case BK_VD_E0: case BK_VD_E1: {
      const int cc = c + (kind == BK_VD_E1 ? 32 : 0);
      if (cc >= 48) return -1;
      const int hh = cc / 24, q = cc % 24;
      if (hh == 0 && q == 22) return OFF_VD_B + o;
      f = slot_feature(1, hh, q);
      return f < 0 ? -1 : OFF_VD_W + o * (32 + E2) + 32 + (f - E1); }
    case BK_ES: return OFF_ES_W + o * 32 + c;
    case BK_S2: return OFF_S2_W + o * 32 + c;
    case BK_CAT_Y: return OFF_CAT_W + o * (32 + E1) + c;
    case BK_CAT_E0: case BK_CAT_E1: case BK_CAT_E2: {
      const int cc = c + 32 * (kind - BK_CAT_E0);
      f = slot_feature(0, cc / 48, cc % 48);
      return f < 0 ? -1 : OFF_CAT_W + o * (32 + E1) + 32 + f; }
    case BK_S1: return OFF_S1_W + o * 32 + c;
    default: {
      const int cc = c + 32 * (kind - BK_XYZ_E0);
      const int hh = cc / 48, q = cc % 48;
      if (hh == 0 && q == 47) return OFF_XYZ_B + o;
      f = slot_feature(0, hh, q);
      return f < 0 ? -1 : OFF_XYZ_W + o * E1 + f; }
  }
}

__global__ __launch_bounds__(256, 1) void field_bwd_kernel(
    const float* __restrict__ pts, const float* __restrict__ Bdir, const unsigned char* __restrict__ packed,
    const float* __restrict__ biasrows, const int* __restrict__ ray_row, float inv_scale,
    const float* __restrict__ d_sigma, const float* __restrict__ d_rgb, float gscale,
    float* __restrict__ dtrunk, float* __restrict__ dB, float* __restrict__ dbiasrows,
    int64_t N, int S, int R, int rows_per_class) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int c = blockIdx.y;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, h = lane >> 5, col = lane & 31;
  const bool lds_rows = rows_per_class > 0 && rows_per_class <= 32;
  {
    const unsigned char* src = packed + (size_t)c * PK_BYTES;
    for (int i = threadIdx.x * 16; i < PK_BYTES; i += 256 * 16)
      *reinterpret_cast<f4*>(smem + i) = *reinterpret_cast<const f4*>(src + i);
    float* Bl = reinterpret_cast<float*>(smem + LDS_BL);
    for (int i = threadIdx.x; i < 66; i += 256) {
      const int hh = i / 33, k = i % 33, d = k / 3;
      Bl[i] = (hh == 1 && d == 10) ? 0.0f : Bdir[(size_t)c * 63 + (11 * hh + d) * 3 + (k % 3)];
    }
    float* rt = reinterpret_cast<float*>(smem + LDS_ROWTAB);
    for (int i = threadIdx.x; i < 32 * 128 + 64; i += 256) rt[i] = 0.0f;
  }
  __syncthreads();
  const float* cf = reinterpret_cast<const float*>(smem + PK_OFF_CONST);
  const unsigned char* bwf = smem + PK_OFF_BWD;
  float* rowtab = reinterpret_cast<float*>(smem + LDS_ROWTAB);
  float* plain = reinterpret_cast<float*>(smem + LDS_PLAIN);
  unsigned char* Ximg = smem + LDS_SCRATCH + wv * WAVE_SCRATCH;
  unsigned char* Dimg = Ximg + XIMG_BYTES;
  float Bh[33];
  {
    const float* Bl = reinterpret_cast<const float*>(smem + LDS_BL) + 33 * h;
#pragma unroll
    for (int i = 0; i < 33; ++i) Bh[i] = Bl[i];
  }
  // ---- persistent accumulators ---------------------------------------------------------------------
  f16v Wacc[NBLOCKS];
#pragma unroll
  for (int b = 0; b < NBLOCKS; ++b) Wacc[b] = zero16();
  float dBacc[33];
#pragma unroll
  for (int i = 0; i < 33; ++i) dBacc[i] = 0.0f;
  f16v dws = zero16();
  float dbs = 0.0f;
  const f16v wsg = acc_init(cf + CF_W_SG, h);
  const float inv_gs = 1.0f / gscale;
  const int slot_inv = (65536 + S - 1) / S;  // (x * slot_inv) >> 16 == x / S for x < 64 + S, S <= 240

  const int64_t ntiles = (N + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    asm volatile("" ::: "memory");  // keep the weight-fragment LDS loads inside the loop
    const int64_t n0 = tile * 32;
    const int64_t n = n0 + col;
    const bool live = n < N;
    const int64_t nc = live ? n : N - 1;
    const int64_t gs = (int64_t)c * N + nc;
    const float* pp = pts + gs * 3;
    const float t0 = pp[0] * inv_scale, t1 = pp[1] * inv_scale, t2 = pp[2] * inv_scale;
    const int64_t ray0 = n0 / S;                 // first ray (within class) touched by this tile
    const int off = (int)(n0 - ray0 * S);        // position of sample 0 inside that ray
    const int64_t ray = (int64_t)c * R + nc / S;
    const int64_t row = ray_row ? (int64_t)ray_row[ray] : ray;
    const float* brow = biasrows + row * (CNR_NLAT * 32);
    // upstream gradients of this sample (both lane halves hold the same sample)
    float draw = live ? d_sigma[gs] * gscale : 0.0f;
    draw = fminf(fmaxf(draw, -8192.0f), 8192.0f) * 10.0f;  // sigmas = raw * 10 (src/model.py:75)
    float dr0 = 0.f, dr1 = 0.f, dr2 = 0.f;
    if (live) { dr0 = d_rgb[gs * 3 + 0] * gscale; dr1 = d_rgb[gs * 3 + 1] * gscale; dr2 = d_rgb[gs * 3 + 2] * gscale; }

    // one-hot(ray slot)^T fragments: row r = slot, k = sample
    h8 OH[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 16 * s + 8 * h + j;
        const int sl = S >= 32 ? ((off + k) >= S ? 1 : 0) : (((off + k) * slot_inv) >> 16);
        OH[s][j] = (sl == col) ? (_Float16)1 : (_Float16)0;
      }

    // =================================== forward recompute =========================================
    h8 E1f[6], E2f[3];
    pe_slots<true>(Bh, t0, t1, t2, h, E1f, E2f);
    f16v acc = acc_init(cf + CF_B_XYZ, h);
#pragma unroll
    for (int s = 0; s < 6; ++s) acc = MFMA(lds_frag(smem, KK_XYZ + s, lane), E1f[s], acc);
    const h8 A0a = pack8(acc, 0, true), A0b = pack8(acc, 1, true);
    acc = acc_init(brow + 0 * 32, h);
    acc = MFMA(lds_frag(smem, KK_S1 + 0, lane), A0a, acc);
    acc = MFMA(lds_frag(smem, KK_S1 + 1, lane), A0b, acc);
    const h8 A1a = pack8(acc, 0, true), A1b = pack8(acc, 1, true);
    acc = acc_init(brow + 1 * 32, h);
    acc = MFMA(lds_frag(smem, KK_CAT + 0, lane), A1a, acc);
    acc = MFMA(lds_frag(smem, KK_CAT + 1, lane), A1b, acc);
#pragma unroll
    for (int s = 0; s < 6; ++s) acc = MFMA(lds_frag(smem, KK_CAT + 2 + s, lane), E1f[s], acc);
    const h8 A2a = pack8(acc, 0, true), A2b = pack8(acc, 1, true);
    acc = acc_init(brow + 2 * 32, h);
    acc = MFMA(lds_frag(smem, KK_S2 + 0, lane), A2a, acc);
    acc = MFMA(lds_frag(smem, KK_S2 + 1, lane), A2b, acc);
    const h8 A3a = pack8(acc, 0, true), A3b = pack8(acc, 1, true);
    acc = acc_init(cf + CF_B_ES, h);
    acc = MFMA(lds_frag(smem, KK_ES + 0, lane), A3a, acc);
    acc = MFMA(lds_frag(smem, KK_ES + 1, lane), A3b, acc);
    // sigma head gradient (fp32 VALU): d w_sigma += draw * y4 ; d b_sigma += draw
#pragma unroll
    for (int i = 0; i < 16; ++i) dws[i] = fmaf(draw, acc[i], dws[i]);
    dbs += (h == 0) ? draw : 0.0f;
    const h8 Y4a = pack8(acc, 0, false), Y4b = pack8(acc, 1, false);
    acc = acc_init(cf + CF_B_VD, h);
    acc = MFMA(lds_frag(smem, KK_VD + 0, lane), Y4a, acc);
    acc = MFMA(lds_frag(smem, KK_VD + 1, lane), Y4b, acc);
#pragma unroll
    for (int s = 0; s < 3; ++s) acc = MFMA(lds_frag(smem, KK_VD + 2 + s, lane), E2f[s], acc);
    const h8 A5a = pack8(acc, 0, true), A5b = pack8(acc, 1, true);
    acc = acc_init(brow + 3 * 32, h);
    acc = MFMA(lds_frag(smem, KK_T1 + 0, lane), A5a, acc);
    acc = MFMA(lds_frag(smem, KK_T1 + 1, lane), A5b, acc);
    const h8 A6a = pack8(acc, 0, true), A6b = pack8(acc, 1, true);
    acc = acc_init(cf + CF_B_R0, h);
    acc = MFMA(lds_frag(smem, KK_R0 + 0, lane), A6a, acc);
    acc = MFMA(lds_frag(smem, KK_R0 + 1, lane), A6b, acc);
    const h8 A7a = pack8(acc, 0, true);
    acc = acc_init(cf + CF_B_R2, h);
    acc = MFMA(lds_frag(smem, KK_R2, lane), A7a, acc);

    // ======================================= backward ================================================
    // helper lambdas -----------------------------------------------------------------------------------
    auto dW_h = [&](f16v& W, const h8& trD0, const h8& trD1) {  // X image = 32-feature tile at Ximg
      W = MFMA(trD0, tr_frag(Ximg, ST_H, 0, 0, lane), W);
      W = MFMA(trD1, tr_frag(Ximg, ST_H, 0, 1, lane), W);
    };
    // per-row sums of the staged dPre tile: (one-hot^T) x dPre ; rows = tile slot, cols = out feature
    auto row_sums = [&](const h8& trD0, const h8& trD1, int latent_slot /* -1: plain bias */, int plain_idx) {
      f16v rs = zero16();
      rs = MFMA(OH[0], trD0, rs);
      rs = MFMA(OH[1], trD1, rs);
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int sl = acc_row(reg, h);
        const int64_t rr = ray0 + sl;
        const bool ok = rr < R && (sl == 0 || (int64_t)sl * S < off + 32);
        if (!ok) continue;
        if (latent_slot < 0) {
          atomicAdd(&plain[plain_idx * 32 + col], rs[reg]);
        } else {
          const int64_t grow = ray_row ? (int64_t)ray_row[(int64_t)c * R + rr] : (int64_t)c * R + rr;
          if (lds_rows) atomicAdd(&rowtab[(grow - (int64_t)c * rows_per_class) * 128 + latent_slot * 32 + col], rs[reg]);
          else atomicAdd(&dbiasrows[grow * 128 + latent_slot * 32 + col], rs[reg] * inv_gs);
        }
      }
    };

    // ---- rgb.2 : dPre9 = drgb * rgb (1 - rgb) lives in rows 0..2 (registers 0..2 of half 0) -----------
    h8 D0 = zero8(), D1 = zero8();
    {
      const float r0 = 1.0f / (1.0f + __expf(-acc[0])), r1 = 1.0f / (1.0f + __expf(-acc[1])),
                  r2 = 1.0f / (1.0f + __expf(-acc[2]));
      if (h == 0) {
        D0[0] = (_Float16)(dr0 * r0 * (1.0f - r0));
        D0[1] = (_Float16)(dr1 * r1 * (1.0f - r1));
        D0[2] = (_Float16)(dr2 * r2 * (1.0f - r2));
      }
    }
    wave_lds_sync();
    stage_h(Dimg, D0, D1, col, h);
    {
      h8 one = zero8();
      if (h == 0) one[0] = (_Float16)1;  // feature 16 of the a7 image := 1 -> d b(rgb.2)
      stage_h(Ximg, A7a, one, col, h);
    }
    wave_lds_sync();
    {
      const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
      dW_h(Wacc[BK_R2], tD0, tD1);
    }
    acc = MFMA(lds_frag(bwf, KT_R2, lane), D0, zero16());   // d a7 (rows 0..15)
    mask8(acc, 0, A7a);
    D0 = pack8(acc, 0, false); D1 = zero8();
    // ---- rgb.0 ------------------------------------------------------------------------------------------
    wave_lds_sync();
    stage_h(Dimg, D0, D1, col, h);
    stage_h(Ximg, A6a, A6b, col, h);
    wave_lds_sync();
    {
      const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
      dW_h(Wacc[BK_R0], tD0, tD1);
      row_sums(tD0, tD1, -1, 1);
    }
    acc = MFMA(lds_frag(bwf, KT_R0, lane), D0, zero16());   // d a6
    mask8(acc, 0, A6a); mask8(acc, 1, A6b);
    D0 = pack8(acc, 0, false); D1 = pack8(acc, 1, false);
    // ---- texture_layer_1 ----------------------------------------------------------------------------------
    wave_lds_sync();
    stage_h(Dimg, D0, D1, col, h);
    stage_h(Ximg, A5a, A5b, col, h);
    wave_lds_sync();
    {
      const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
      dW_h(Wacc[BK_T1], tD0, tD1);
      row_sums(tD0, tD1, 3, 0);
    }
    acc = MFMA(lds_frag(bwf, KT_T1 + 0, lane), D0, zero16());
    acc = MFMA(lds_frag(bwf, KT_T1 + 1, lane), D1, acc);    // d a5
    mask8(acc, 0, A5a); mask8(acc, 1, A5b);
    D0 = pack8(acc, 0, false); D1 = pack8(acc, 1, false);
    // ---- encoding_viewdir : inputs [y4 | e2] -----------------------------------------------------------------
    f16v DE2[2];
    wave_lds_sync();
    stage_h(Dimg, D0, D1, col, h);
    stage_h(Ximg, Y4a, Y4b, col, h);
    wave_lds_sync();
    {
      const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
      dW_h(Wacc[BK_VD_Y], tD0, tD1);
      wave_lds_sync();
      {
        unsigned char* base = Ximg + col * ST_E2 + h * 48;
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<h8*>(base + 16 * s) = E2f[s];
      }
      wave_lds_sync();
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        Wacc[BK_VD_E0 + b] = MFMA(tD0, tr_frag(Ximg, ST_E2, 32 * b, 0, lane), Wacc[BK_VD_E0 + b]);
        Wacc[BK_VD_E0 + b] = MFMA(tD1, tr_frag(Ximg, ST_E2, 32 * b, 1, lane), Wacc[BK_VD_E0 + b]);
      }
    }
    acc = MFMA(lds_frag(bwf, KT_VD_Y + 0, lane), D0, zero16());
    acc = MFMA(lds_frag(bwf, KT_VD_Y + 1, lane), D1, acc);  // d y4 from the colour branch
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      DE2[b] = MFMA(lds_frag(bwf, KT_VD_E + 2 * b + 0, lane), D0, zero16());
      DE2[b] = MFMA(lds_frag(bwf, KT_VD_E + 2 * b + 1, lane), D1, DE2[b]);
    }
    // + sigma head: d y4 += w_sigma * draw
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = fmaf(wsg[i], draw, acc[i]);
    D0 = pack8(acc, 0, false); D1 = pack8(acc, 1, false);
    // ---- encoding_shape (no activation) --------------------------------------------------------------------------
    wave_lds_sync();
    stage_h(Dimg, D0, D1, col, h);
    stage_h(Ximg, A3a, A3b, col, h);
    wave_lds_sync();
    {
      const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
      dW_h(Wacc[BK_ES], tD0, tD1);
      row_sums(tD0, tD1, -1, 0);
    }
    acc = MFMA(lds_frag(bwf, KT_ES + 0, lane), D0, zero16());
    acc = MFMA(lds_frag(bwf, KT_ES + 1, lane), D1, acc);    // d a3
    mask8(acc, 0, A3a); mask8(acc, 1, A3b);
    D0 = pack8(acc, 0, false); D1 = pack8(acc, 1, false);
    // ---- shape_layer_2 ------------------------------------------------------------------------------------------------
    wave_lds_sync();
    stage_h(Dimg, D0, D1, col, h);
    stage_h(Ximg, A2a, A2b, col, h);
    wave_lds_sync();
    {
      const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
      dW_h(Wacc[BK_S2], tD0, tD1);
      row_sums(tD0, tD1, 2, 0);
    }
    acc = MFMA(lds_frag(bwf, KT_S2 + 0, lane), D0, zero16());
    acc = MFMA(lds_frag(bwf, KT_S2 + 1, lane), D1, acc);    // d a2
    mask8(acc, 0, A2a); mask8(acc, 1, A2b);
    D0 = pack8(acc, 0, false); D1 = pack8(acc, 1, false);
    // ---- cat_layer : inputs [a1 | e1] ------------------------------------------------------------------------------------
    f16v DE1[3];
    wave_lds_sync();
    stage_h(Dimg, D0, D1, col, h);
    stage_h(Ximg, A1a, A1b, col, h);
    wave_lds_sync();
    {
      const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
      dW_h(Wacc[BK_CAT_Y], tD0, tD1);
      row_sums(tD0, tD1, 1, 0);
      wave_lds_sync();
      {
        unsigned char* base = Ximg + col * ST_E1 + h * 96;
#pragma unroll
        for (int s = 0; s < 6; ++s) *reinterpret_cast<h8*>(base + 16 * s) = E1f[s];
      }
      wave_lds_sync();
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        Wacc[BK_CAT_E0 + b] = MFMA(tD0, tr_frag(Ximg, ST_E1, 32 * b, 0, lane), Wacc[BK_CAT_E0 + b]);
        Wacc[BK_CAT_E0 + b] = MFMA(tD1, tr_frag(Ximg, ST_E1, 32 * b, 1, lane), Wacc[BK_CAT_E0 + b]);
      }
    }
    acc = MFMA(lds_frag(bwf, KT_CAT_Y + 0, lane), D0, zero16());
    acc = MFMA(lds_frag(bwf, KT_CAT_Y + 1, lane), D1, acc);  // d a1
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      DE1[b] = MFMA(lds_frag(bwf, KT_CAT_E + 2 * b + 0, lane), D0, zero16());
      DE1[b] = MFMA(lds_frag(bwf, KT_CAT_E + 2 * b + 1, lane), D1, DE1[b]);
    }
    mask8(acc, 0, A1a); mask8(acc, 1, A1b);
    D0 = pack8(acc, 0, false); D1 = pack8(acc, 1, false);
    // ---- shape_layer_1 --------------------------------------------------------------------------------------------------------
    wave_lds_sync();
    stage_h(Dimg, D0, D1, col, h);
    stage_h(Ximg, A0a, A0b, col, h);  // overwrites the E1 image; it is re-staged for encoding_xyz below
    wave_lds_sync();
    {
      const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
      dW_h(Wacc[BK_S1], tD0, tD1);
      row_sums(tD0, tD1, 0, 0);
    }
    acc = MFMA(lds_frag(bwf, KT_S1 + 0, lane), D0, zero16());
    acc = MFMA(lds_frag(bwf, KT_S1 + 1, lane), D1, acc);    // d a0
    mask8(acc, 0, A0a); mask8(acc, 1, A0b);
    D0 = pack8(acc, 0, false); D1 = pack8(acc, 1, false);
    // ---- encoding_xyz : input e1 ---------------------------------------------------------------------------------------------------
    wave_lds_sync();
    stage_h(Dimg, D0, D1, col, h);
    {
      unsigned char* base = Ximg + col * ST_E1 + h * 96;
#pragma unroll
      for (int s = 0; s < 6; ++s) *reinterpret_cast<h8*>(base + 16 * s) = E1f[s];
    }
    wave_lds_sync();
    {
      const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        Wacc[BK_XYZ_E0 + b] = MFMA(tD0, tr_frag(Ximg, ST_E1, 32 * b, 0, lane), Wacc[BK_XYZ_E0 + b]);
        Wacc[BK_XYZ_E0 + b] = MFMA(tD1, tr_frag(Ximg, ST_E1, 32 * b, 1, lane), Wacc[BK_XYZ_E0 + b]);
      }
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      DE1[b] = MFMA(lds_frag(bwf, KT_XYZ_E + 2 * b + 0, lane), D0, DE1[b]);
      DE1[b] = MFMA(lds_frag(bwf, KT_XYZ_E + 2 * b + 1, lane), D1, DE1[b]);
    }
    // ---- positional-encoding backward: dB_d += (sum_b de[b,d] cos(pi 2^b p_d) pi 2^b) * t ----------------------------------------------
#pragma unroll
    for (int d = 0; d < 11; ++d) {
      const float p = Bh[3 * d] * t0 + Bh[3 * d + 1] * t1 + Bh[3 * d + 2] * t2;
      float gp = 0.0f;
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const float arg = p * (0.5f * (float)(1 << b));
        const float cs = __builtin_amdgcn_cosf(arg);
        const int q = b < 4 ? 11 * b + d : 11 * (b - 4) + d;
        const float de = b < 4 ? DE1[q >> 4][q & 15] : DE2[q >> 4][q & 15];
        gp = fmaf(de * cs, 3.14159265358979f * (float)(1 << b), gp);
      }
      dBacc[3 * d + 0] = fmaf(gp, t0, dBacc[3 * d + 0]);
      dBacc[3 * d + 1] = fmaf(gp, t1, dBacc[3 * d + 1]);
      dBacc[3 * d + 2] = fmaf(gp, t2, dBacc[3 * d + 2]);
    }
  }

  // ========================================= flush ====================================================
  __syncthreads();  // every wave is done with the packed image and its scratch
  float* slab = reinterpret_cast<float*>(smem);  // TRUNK floats, then 63 floats of dB
  for (int i = threadIdx.x; i < TRUNK + 64; i += 256) slab[i] = 0.0f;
  __syncthreads();
#pragma unroll
  for (int b = 0; b < NBLOCKS; ++b) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int idx = block_index(b, acc_row(reg, h), col);
      if (idx >= 0) atomicAdd(&slab[idx], Wacc[b][reg]);
    }
  }
  // sigma head, direction matrix: reduce over the 32 samples of each half, then one lane adds
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float v = dws[i];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (col == 0) atomicAdd(&slab[OFF_SG_W + acc_row(i, h)], v);
  }
  {
    float v = dbs;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) atomicAdd(&slab[OFF_SG_B], v);
  }
#pragma unroll
  for (int i = 0; i < 33; ++i) {
    float v = dBacc[i];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int d = i / 3;
    if (col == 0 && !(h == 1 && d == 10)) atomicAdd(&slab[TRUNK + (11 * h + d) * 3 + (i % 3)], v);
  }
  __syncthreads();
  // plain biases collected by the one-hot products
  for (int i = threadIdx.x; i < 64; i += 256) {
    const int which = i >> 5, o = i & 31;
    const float v = plain[i];
    if (which == 0) slab[OFF_ES_B + o] += v;
    else if (o < 16) slab[OFF_R0_B + o] += v;
  }
  __syncthreads();
  float* out = dtrunk + (size_t)c * TRUNK;
  for (int i = threadIdx.x; i < TRUNK; i += 256) {
    const float v = slab[i];
    if (v != 0.0f) atomicAdd(&out[i], v * inv_gs);
  }
  for (int i = threadIdx.x; i < 63; i += 256) atomicAdd(&dB[(size_t)c * 63 + i], slab[TRUNK + i] * inv_gs);
  if (lds_rows) {
    for (int i = threadIdx.x; i < rows_per_class * 128; i += 256) {
      const float v = rowtab[i];
      if (v != 0.0f) atomicAdd(&dbiasrows[(size_t)c * rows_per_class * 128 + i], v * inv_gs);
    }
  }
}
}  // namespace

extern "C" int cnr_field_bwd(const float* pts, const float* B, const void* packed, const float* biasrows,
                             const int* ray_row, float scale, const float* d_sigma, const float* d_rgb,
                             float grad_scale, float* dtrunk, float* dB, float* dbiasrows, int C, int R, int S,
                             int rows_per_class, int max_blocks, void* stream) {
  if (!pts || !B || !packed || !biasrows || !d_sigma || !d_rgb || !dtrunk || !dB || !dbiasrows) return CNR_E_ARG;
  if (C <= 0 || R <= 0 || S <= 0 || !(scale > 0.f) || !(grad_scale > 0.f)) return CNR_E_ARG;
  if (S > 240) return CNR_E_SHAPE;
  if (((uintptr_t)packed & 15) != 0 || ((uintptr_t)biasrows & 15) != 0) return CNR_E_ALIGN;
  if (ray_row == nullptr && rows_per_class > 0 && rows_per_class != R) return CNR_E_ARG;
  const int64_t N = (int64_t)R * S;
  const int64_t ntiles = (N + 31) / 32;
  int64_t blocks = (ntiles + 3) / 4;
  const int64_t cap = max_blocks > 0 ? max_blocks : 256;
  if (blocks > cap) blocks = cap;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t er = hipFuncSetAttribute((const void*)field_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        LDS_TOTAL);
    if (er != hipSuccess) return (int)er;
    attr_set = true;
  }
  dim3 grid((unsigned)blocks, (unsigned)C);
  hipLaunchKernelGGL(field_bwd_kernel, grid, dim3(256), LDS_TOTAL, (hipStream_t)stream, pts, B,
                     (const unsigned char*)packed, biasrows, ray_row, 1.0f / scale, d_sigma, d_rgb, grad_scale,
                     dtrunk, dB, dbiasrows, N, S, R, rows_per_class);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}
