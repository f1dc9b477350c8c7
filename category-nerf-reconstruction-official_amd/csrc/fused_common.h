// Layout contract of the fused f16-MFMA path (fused_fwd.hip / fused_bwd.hip / pack kernel).
//
// Orientation: every layer is computed TRANSPOSED, D = W * X^T with v_mfma_f32_32x32x16_f16:
//   A = W        (rows = output features, K = input features)      -- pre-packed fragments in LDS
//   B = X^T      (K = input features, columns = 32 samples = lanes) -- built in registers
//   D = (W X^T)  32 output features x 32 samples: column (sample) = lane & 31,
//                row(reg, h) = (reg & 3) + 8 * (reg >> 2) + 4 * h,  h = lane >> 5, reg in [0,16)
// so that a layer's accumulators, converted pairwise to f16, ARE the next layer's B operand with no
// lane movement (MI355X guide, "An accumulator tile as the next MFMA's operand"): B fragment of
// k-step s, element j of lane half h  <->  feature 16 s + 8 (j >> 2) + 4 h + (j & 3).  The packed A
// fragments carry the matching K permutation.
//
// Positional-encoding features are produced directly in B-operand layout.  Each lane half owns a set
// of directions (half 0: dirs 0..10, half 1: dirs 11..20 + one zero dummy) so a lane evaluates only 33
// dot-product terms and 66 sines per sample:
//   E1 (6 k-steps, 48 slots per half): slot q = 11 * band + d, band 0..3, d 0..10 ; q 44..46 = t (half 0)
//   E2 (3 k-steps, 24 slots per half): slot q = 11 * (band - 4) + d, band 4..5
//   slot q lives in k-step q / 8, element q % 8.
#pragma once
#include "cnr_common.h"

namespace fz {
using namespace cnr;

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// ---- forward A-fragment k-step table ---------------------------------------------------------------
constexpr int KK_XYZ = 0, KK_S1 = 6, KK_CAT = 8, KK_S2 = 16, KK_ES = 18, KK_VD = 20, KK_T1 = 25,
              KK_R0 = 27, KK_R2 = 29, NKK_FWD = 30;
// ---- backward (transposed) A-fragment table: rows = INPUT features of the layer (one 32-row block
// each), K = output features (k-steps of 16).  E1 / E2 row blocks are in slot order (16 slots per half
// per block: block row (reg, h) <-> slot 16 * blk + reg of half h).
constexpr int KT_R2 = 0;     // rgb.2^T     : rows a7 (16 used)      K = 3   -> 1 k-step
constexpr int KT_R0 = 1;     // rgb.0^T     : rows a6 (32)           K = 16  -> 1
constexpr int KT_T1 = 2;     // texture_1^T : rows a5 (32)           K = 32  -> 2
constexpr int KT_VD_Y = 4;   // viewdir^T   : rows y4 (32)                    -> 2
constexpr int KT_VD_E = 6;   //               rows E2 slots, 2 blocks         -> 2 x 2
constexpr int KT_ES = 10;    // enc_shape^T : rows a3                         -> 2
constexpr int KT_S2 = 12;    // shape_2^T   : rows a2                         -> 2
constexpr int KT_CAT_Y = 14; // cat^T       : rows a1                         -> 2
constexpr int KT_CAT_E = 16; //               rows E1 slots, 3 blocks         -> 3 x 2
constexpr int KT_S1 = 22;    // shape_1^T   : rows a0                         -> 2
constexpr int KT_XYZ_E = 24; // xyz^T       : rows E1 slots, 3 blocks         -> 3 x 2
constexpr int NKK_BWD = 30;

constexpr int FRAG_BYTES = 1024;  // 64 lanes x 16 B
// f32 constants section (floats): acc-init biases of the non-latent layers, sigma head
constexpr int CF_B_XYZ = 0, CF_B_ES = 32, CF_B_VD = 64, CF_B_R0 = 96, CF_B_R2 = 128, CF_W_SG = 160,
              CF_B_SG = 192, CF_FLOATS = 256;
constexpr int PK_OFF_FWD = 0;
constexpr int PK_OFF_CONST = NKK_FWD * FRAG_BYTES;                 // 30720
constexpr int PK_OFF_BWD = PK_OFF_CONST + CF_FLOATS * 4;           // 31744
constexpr int PK_BYTES = PK_OFF_BWD + NKK_BWD * FRAG_BYTES;        // 62464
// Residual image ("lo"): f16(W - f16(W)) of the forward fragments of the GEOMETRY branch -- k-steps [0, NKK_GEO) =
// encoding_xyz, shape_layer_1, cat_layer, shape_layer_2, encoding_shape, the layers between the ray sample and the x10
// occupancy logit (src/model.py:56-75).  A kernel given this image computes those layers as three products
//   W x  ~=  Wh xh + Wl xh + Wh xl      (Wh = f16(W), Wl = f16(W - Wh), xh = f16(x), xl = f16(x - xh); fp32 accumulate)
// which carries ~22 mantissa bits through the branch: north_star's 1e-3 on the occupancy holds for TRAINED weights (plain
// f16 operands give 1.8e-3 .. 2.0e-3 after 400 .. 5000 steps).  The colour branch stays plain f16.
constexpr int NKK_GEO = KK_VD;                                      // 20 fragments
constexpr int PK_LO_BYTES = NKK_GEO * FRAG_BYTES;                   // 20480

// e-feature (0..128) held by slot q of lane-half h ; -1 = empty slot.  part 0 = E1, 1 = E2.
__host__ __device__ inline int slot_feature(int part, int h, int q) {
  if (part == 0) {
    if (q < 44) {
      const int band = q / 11, d = q % 11;
      if (h == 0) return 3 + 21 * band + d;
      return d < 10 ? 3 + 21 * band + 11 + d : -1;
    }
    if (q < 47 && h == 0) return q - 44;
    return -1;
  }
  if (q < 22) {
    const int band = 4 + q / 11, d = q % 11;
    if (h == 0) return 3 + 21 * band + d;
    return d < 10 ? 3 + 21 * band + 11 + d : -1;
  }
  return -1;
}
// feature (0..31) of an accumulator-derived B operand: k-step s, half h, element j
__host__ __device__ inline int acc_feature(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }
// accumulator row of (reg, h)
__host__ __device__ inline int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
// element A[r][k] of forward weight fragment kk (k-step of 16, K permuted as above) for lane half h, element j ; 0 = padding
__device__ __forceinline__ float fwd_elem(const float* __restrict__ Wt, int kk, int r, int h, int j) {
    int s, w_off, ld, out, col = -1;
  if (kk < KK_S1) { s = kk - KK_XYZ; w_off = OFF_XYZ_W; ld = E1; out = 32;
    const int f = slot_feature(0, h, 8 * s + j); col = f; }
  else if (kk < KK_CAT) { s = kk - KK_S1; w_off = OFF_S1_W; ld = 32; out = 32; col = acc_feature(s, h, j); }
  else if (kk < KK_S2) { s = kk - KK_CAT; w_off = OFF_CAT_W; ld = 32 + E1; out = 32;
    if (s < 2) col = acc_feature(s, h, j);
    else { const int f = slot_feature(0, h, 8 * (s - 2) + j); col = f < 0 ? -1 : 32 + f; } }
  else if (kk < KK_ES) { s = kk - KK_S2; w_off = OFF_S2_W; ld = 32; out = 32; col = acc_feature(s, h, j); }
  else if (kk < KK_VD) { s = kk - KK_ES; w_off = OFF_ES_W; ld = 32; out = 32; col = acc_feature(s, h, j); }
  else if (kk < KK_T1) { s = kk - KK_VD; w_off = OFF_VD_W; ld = 32 + E2; out = 32;
    if (s < 2) col = acc_feature(s, h, j);
    else { const int f = slot_feature(1, h, 8 * (s - 2) + j); col = f < 0 ? -1 : 32 + (f - E1); } }
  else if (kk < KK_R0) { s = kk - KK_T1; w_off = OFF_T1_W; ld = 32; out = 32; col = acc_feature(s, h, j); }
  else if (kk < KK_R2) { s = kk - KK_R0; w_off = OFF_R0_W; ld = 32; out = 16; col = acc_feature(s, h, j); }
  else { s = 0; w_off = OFF_R2_W; ld = 16; out = 3; col = acc_feature(0, h, j); }
  if (col < 0 || r >= out) return 0.0f;
  return Wt[w_off + r * ld + col];
}

// ---------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ h8 pack8(const f16v& a, int s, bool relu) {
  h8 r;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    f2 v = {a[8 * s + j], a[8 * s + j + 1]};
    h2 p = __builtin_convertvector(v, h2);
    if (relu) { h2 z = {(_Float16)0, (_Float16)0}; p = __builtin_elementwise_max(p, z); }
    r[j] = p[0]; r[j + 1] = p[1];
  }
  return r;
}
// x - (float)h as ONE v_fma_mix_f32 (f16 source read in place): the -1 is opaque to the optimiser, which otherwise rewrites
// fma(h, -1, x) into v_cvt_f32_f16 + v_sub_f32
__device__ __forceinline__ float minus_one() {
  float m = -1.0f;
  asm volatile("" : "+s"(m));   // no instruction
  return m;
}
// max(x, 0) as ONE integer maximum on the bit pattern (a non-negative float is a non-negative integer, anything with the sign bit
// set -- -0 included -- is below 0): fmaxf(x, 0.0f) is TWO instructions, the compiler first canonicalises x (v_max_f32 x, x, x: a
// matrix-core result is not known to be quiet) -- 64 of the chain wave's instructions per tile, seen in the ISA
__device__ __forceinline__ float relu_f32(float x) {
  const int b = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}
// pack8(a, s, true) from the fp32 ReLU: f16(max(a, 0)) == max(f16(a), 0) (rounding is monotonic), and where the residual is
// formed too (pack8_lo) the maxima are shared with it -- no packed-f16 maximum on top: four instructions less per fragment
__device__ __forceinline__ h8 pack8_relu32(const f16v& a, int s) {
  h8 r;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    f2 v = {relu_f32(a[8 * s + j]), relu_f32(a[8 * s + j + 1])};
    h2 p = __builtin_convertvector(v, h2);
    r[j] = p[0]; r[j + 1] = p[1];
  }
  return r;
}
// the residual of pack8: f16(x - f16(x)) for the same 8 accumulator registers, x = max(a, 0) when relu
__device__ __forceinline__ h8 pack8_lo(const f16v& a, int s, bool relu, const h8& hi) {
  h8 r;
  const float m1 = minus_one();
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const float x0 = relu ? relu_f32(a[8 * s + j]) : a[8 * s + j];
    const float x1 = relu ? relu_f32(a[8 * s + j + 1]) : a[8 * s + j + 1];
    f2 v = {__builtin_fmaf((float)hi[j], m1, x0), __builtin_fmaf((float)hi[j + 1], m1, x1)};   // v_fma_mix_f32
    h2 p = __builtin_convertvector(v, h2);
    r[j] = p[0]; r[j + 1] = p[1];
  }
  return r;
}
__device__ __forceinline__ h8 pack8f_lo(const float* v, const h8& hi) {
  h8 r;
  const float m1 = minus_one();
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    f2 x = {__builtin_fmaf((float)hi[j], m1, v[j]), __builtin_fmaf((float)hi[j + 1], m1, v[j + 1])};
    h2 p = __builtin_convertvector(x, h2);
    r[j] = p[0]; r[j + 1] = p[1];
  }
  return r;
}
__device__ __forceinline__ h8 pack8f(const float* v) {
  h8 r;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    f2 x = {v[j], v[j + 1]};
    h2 p = __builtin_convertvector(x, h2);
    r[j] = p[0]; r[j + 1] = p[1];
  }
  return r;
}
// accumulator init from 32 floats in natural feature order at p (LDS or global): lane needs rows
// 8 g + 4 h + (0..3), g = 0..3
__device__ __forceinline__ f16v acc_init(const float* p, int h) {
  f16v a;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f4 v = *reinterpret_cast<const f4*>(p + 8 * g + 4 * h);
    a[4 * g + 0] = v[0]; a[4 * g + 1] = v[1]; a[4 * g + 2] = v[2]; a[4 * g + 3] = v[3];
  }
  return a;
}
__device__ __forceinline__ h8 lds_frag(const unsigned char* lds_w, int kk, int lane) {
  return *reinterpret_cast<const h8*>(lds_w + kk * FRAG_BYTES + lane * 16);
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// Positional encoding of one sample in B-operand layout.  Bh: this lane-half's 11 direction rows
// (row 10 of half 1 is zero).  ONES puts 1.0 into two empty half-0 slots (E1 slot 47, E2 slot 22): their
// forward weights are zero, and in the backward's dW products they collect the bias gradients.
// (the pieces of pe_slots, for callers that interleave the encoding with the matrix products fragment by fragment)
// the direction products p_d = B_d . t of this lane half's 11 directions
__device__ __forceinline__ void pe_dirs(const float (&Bh)[33], float t0, float t1, float t2, float (&pd)[11]) {
#pragma unroll
  for (int d = 0; d < 11; ++d) pd[d] = Bh[3 * d] * t0 + Bh[3 * d + 1] * t1 + Bh[3 * d + 2] * t2;
}
// E1 feature slots 8 s .. 8 s + 7 (s = 0..5): slot q < 44 = sin(pi 2^b p_d) with b = q / 11, d = q % 11 (hardware sine takes
// revolutions: v_sin(2^(b-1) p)); 44..46 = t (lane half 0), 47 = the ones slot
template <bool ONES>
__device__ __forceinline__ void pe_e1_slots(const float (&pd)[11], float t0, float t1, float t2, int h, int s, float (&v)[8]) {
  const float one = (ONES && h == 0) ? 1.0f : 0.0f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int q = 8 * s + j;
    if (q < 44) v[j] = __builtin_amdgcn_sinf(pd[q % 11] * (0.5f * (float)(1 << (q / 11))));
    else v[j] = q == 47 ? one : (h == 0 ? (q == 44 ? t0 : q == 45 ? t1 : t2) : 0.0f);
  }
}
// the 24 E2 slots: bands 4 and 5 of the 11 directions, the ones slot (22), an empty one
template <bool ONES>
__device__ __forceinline__ void pe_e2_slots(const float (&pd)[11], int h, float (&v)[24]) {
#pragma unroll
  for (int j = 0; j < 22; ++j) v[j] = __builtin_amdgcn_sinf(pd[j % 11] * (0.5f * (float)(1 << (4 + j / 11))));
  v[22] = (ONES && h == 0) ? 1.0f : 0.0f;
  v[23] = 0.0f;
}
template <bool ONES, bool LO = false>
__device__ __forceinline__ void pe_slots(const float (&Bh)[33], float t0, float t1, float t2, int h,
                                         h8 (&E1f)[6], h8 (&E2f)[3], h8* E1lo = nullptr) {
  float pd[11];
  pe_dirs(Bh, t0, t1, t2, pd);
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    float v[8];
    pe_e1_slots<ONES>(pd, t0, t1, t2, h, s, v);
    E1f[s] = pack8f(v);
    if constexpr (LO) E1lo[s] = pack8f_lo(v, E1f[s]);   // residual of the E1 features (the geometry branch's inputs)
  }
  float v2[24];
  pe_e2_slots<ONES>(pd, h, v2);
#pragma unroll
  for (int s = 0; s < 3; ++s) E2f[s] = pack8f(&v2[8 * s]);
}

}  // namespace fz
