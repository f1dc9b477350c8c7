// Fused f16-MFMA train step of the BACKGROUND field: vMAP-style OccupancyMap(hidden 128), src/model.py:86-155, trained every
// iteration on 1200 world-frame rays x 14 samples (train.py:113-121,172-180).  Round 2 ran its seven Linear layers one launch
// each under torch autograd (14 x 20.8 us + 7 x 26.6 us + ~40 glue kernels = 0.69 ms); here the step is
//   cnr_bg_pack      fp32 parameters -> f16 MFMA fragment images (forward hi / geometry-branch residual / transposed); once, and
//                    after outside changes of the parameters: inside the loop the tail launch keeps the images current
//   cnr_bg_forward   per 32-sample tile: PE -> in_layer -> mid1 -> cat_layer -> mid2 -> {out_alpha | color_linear -> out_color}
//                    with every activation in LDS; writes sigma, rgb and the f16 activations the backward needs
//   (cnr_render_loss: composite, losses, d sigma, d colour -- the category path's kernel)
//   cnr_bg_backward  per tile: the data-gradient chain, masks from the stored activations, PE backward; writes the f16
//                    pre-activation gradients, and the small gradients (out_color, out_alpha, B) as per-workgroup records
//   cnr_bg_dw        weight / bias gradients of the five 128-wide layers: dW = dPre^T X over all samples, split over sample
//                    chunks (partials, fixed-order sum)
//   cnr_bg_tail      partial + record reduction, AdamW in place, fragment refresh, loss values, step state advance
// Layout: a workgroup = 4 waves owns a tile of TS = 32 samples; every layer is computed transposed, D = W X^T
// (v_mfma_f32_32x32x16_f16): wave w owns output features [32 w, 32 w + 32) of the tile; the A operand (weights) comes
// pre-packed from global / L2 straight into registers, the B operand (activations) from a [sample][feature] f16 image in LDS
// (16-byte reads, padded rows), the accumulators go back to such an image.
// Precision: the geometry branch (in_layer, mid1, cat_layer, mid2 -> out_alpha) as three f16 products per fragment,
// Wh xh + Wl xh + Wh xl (residual weights and residual activation images), the x10 occupancy head as an fp32 dot product
// of the fp32 accumulators -- the same reasoning as the category kernel (include/cnr_hip.h, cnr_pack_weights_lo); the
// colour branch and the whole backward use plain f16 operands with a power-of-two loss scale.  Deterministic: no float
// atomics anywhere (records and partials summed in a fixed order).
#include "adamw_common.h"
#include "fused_common.h"
#include "render_common.h"
#include "sample_common.h"

namespace {
using namespace fz;

constexpr int BH = 128, BE1 = 87, BE2 = 42, BE1P = 96, BE2P = 48, BEP = BE1P + BE2P;   // hidden, PE widths, padded
// flat parameter buffer (the modules' registration order: src/model.py:100-113, then B_layer of the PE)
constexpr int O_IN_W = 0, O_IN_B = 11136, O_M1_W = 11264, O_M1_B = 27648, O_CAT_W = 27776, O_CAT_B = 55296,
              O_M2_W = 55424, O_M2_B = 71808, O_OA_W = 71936, O_OA_B = 72064, O_CL_W = 72065, O_CL_B = 93825,
              O_OC_W = 93953, O_OC_B = 94337, O_PE_B = 94340, BG_NPARAM = 94403;
enum BgLayer { L_IN, L_M1, L_CAT, L_M2, L_CL, L_OC, NLAYER };
// k-steps (16 input features each) per layer: first source (hidden activations / E1 for in_layer), second source (PE part)
__host__ __device__ constexpr int ks_a(int l) { return l == L_IN ? 6 : 8; }
__host__ __device__ constexpr int ks_b(int l) { return l == L_CAT ? 6 : l == L_CL ? 3 : 0; }
__host__ __device__ constexpr int ks(int l) { return ks_a(l) + ks_b(l); }
// forward fragment index (hi image) of (layer, out block w, k-step s); OC has one out block
__host__ __device__ constexpr int fwd_base(int l) { return l == L_IN ? 0 : l == L_M1 ? 24 : l == L_CAT ? 56 : l == L_M2 ? 112 : l == L_CL ? 144 : 188; }
constexpr int NF_FWD = 196, NF_LO = 144;     // residual image: the four geometry layers, same indexing (fwd_base < 144)
// transposed fragments (data-gradient chain): rows = 32 input features of block ib, K = output features
__host__ __device__ constexpr int bwd_base(int l) { return l == L_OC ? 0 : l == L_CL ? 4 : l == L_M2 ? 52 : l == L_CAT ? 84 : l == L_M1 ? 140 : 172; }
__host__ __device__ constexpr int bwd_blocks(int l) { return l == L_OC ? 4 : l == L_CL ? 6 : l == L_CAT ? 7 : l == L_IN ? 3 : 4; }
constexpr int NF_BWD = 196;
constexpr int PK_FWD_OFF = 0, PK_LO_OFF = NF_FWD * FRAG_BYTES, PK_BWD_OFF = PK_LO_OFF + NF_LO * FRAG_BYTES,
              BG_PACK_BYTES = PK_BWD_OFF + NF_BWD * FRAG_BYTES;
__host__ __device__ constexpr int w_off(int l) { return l == L_IN ? O_IN_W : l == L_M1 ? O_M1_W : l == L_CAT ? O_CAT_W : l == L_M2 ? O_M2_W : l == L_CL ? O_CL_W : O_OC_W; }
__host__ __device__ constexpr int b_off(int l) { return l == L_IN ? O_IN_B : l == L_M1 ? O_M1_B : l == L_CAT ? O_CAT_B : l == L_M2 ? O_M2_B : l == L_CL ? O_CL_B : O_OC_B; }
__host__ __device__ constexpr int ld_of(int l) { return l == L_IN ? BE1 : l == L_CAT ? BH + BE1 : l == L_CL ? BH + BE2 : BH; }
__host__ __device__ constexpr int n_out(int l) { return l == L_OC ? 3 : BH; }
// column of the layer's weight matrix that extended input index k (16 s + 8 h + j) reads, -1 = padding
__device__ __forceinline__ int in_col(int l, int k) {
  if (l == L_IN) return k < BE1 ? k : -1;
  if (k < BH) return k;
  const int q = k - BH;
  if (l == L_CAT) return q < BE1 ? BH + q : -1;
  if (l == L_CL) return q < BE2 ? BH + q : -1;
  return -1;
}

// LDS image strides (bytes): [sample][feature] f16 rows padded so that the 16-byte operand reads of 16 consecutive samples
// fall on 16 different bank quads
constexpr int ST_X = 272, ST_E1B = 208, ST_E2B = 112, ST_OC = 48;
// Samples per workgroup tile of the forward / backward kernels: 32 = one MFMA column block.  M = 16 800 gives 525 tiles at
// 51 KB (forward) / 45 KB (backward) of LDS: all resident at once, two or three workgroups per CU to cover each other's
// dependent latency (weight fragments from L2 -> products -> epilogue -> barrier, per layer).  With 64-sample tiles (103 KB, one
// workgroup per CU) 263 tiles took two rounds on 256 CUs.
constexpr int TS = 32, NH = TS / 32;
constexpr int XIMG = TS * ST_X, E1IMG = TS * ST_E1B, E2IMG = TS * ST_E2B;

// ------------------------------------------------------------------------------------------------------------------------
// pack: one block per fragment
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bg_pack_kernel(const float* __restrict__ theta, unsigned char* __restrict__ packed) {
  const int f = blockIdx.x;
  _Float16* out = reinterpret_cast<_Float16*>(packed + (size_t)f * FRAG_BYTES);
  int kind, idx;   // 0 hi, 1 lo, 2 transposed
  if (f < NF_FWD) { kind = 0; idx = f; } else if (f < NF_FWD + NF_LO) { kind = 1; idx = f - NF_FWD; } else { kind = 2; idx = f - NF_FWD - NF_LO; }
  for (int e = threadIdx.x; e < 512; e += 256) {
    const int lane = e >> 3, j = e & 7, r = lane & 31, h = lane >> 5;
    float v = 0.0f;
    if (kind < 2) {
      int l = idx < 24 ? L_IN : idx < 56 ? L_M1 : idx < 112 ? L_CAT : idx < 144 ? L_M2 : idx < 188 ? L_CL : L_OC;
      const int rel = idx - fwd_base(l), w = l == L_OC ? 0 : rel / ks(l), s = l == L_OC ? rel : rel % ks(l);
      const int o = 32 * w + r, c = in_col(l, 16 * s + 8 * h + j);
      if (o < n_out(l) && c >= 0) v = theta[w_off(l) + o * ld_of(l) + c];
      if (kind == 1) v = v - (float)(_Float16)v;
    } else {
      int l = idx < 4 ? L_OC : idx < 52 ? L_CL : idx < 84 ? L_M2 : idx < 140 ? L_CAT : idx < 172 ? L_M1 : L_IN;
      const int rel = idx - bwd_base(l), ib = l == L_OC ? rel : rel / 8, s = l == L_OC ? 0 : rel % 8;
      // row = input feature 32 ib + r of the layer's extended input; for in_layer the blocks are its E1 inputs
      const int kin = l == L_IN ? 32 * ib + r : 32 * ib + r;
      const int c = l == L_IN ? (kin < BE1 ? kin : -1) : in_col(l, kin);
      const int o = 16 * s + 8 * h + j;
      if (c >= 0 && o < n_out(l)) v = theta[w_off(l) + o * ld_of(l) + c];
    }
    out[lane * 8 + j] = (_Float16)v;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// shared device helpers
// ------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ h8 gfrag(const unsigned char* __restrict__ img, int f, int lane) {
  return *reinterpret_cast<const h8*>(img + (size_t)f * FRAG_BYTES + lane * 16);
}
// B operand of k-step s of a [sample][feature] image: lane (sample, h) reads features 16 s + 8 h .. + 7
__device__ __forceinline__ h8 xfrag(const unsigned char* img, int stride, int sample, int s, int h) {
  return *reinterpret_cast<const h8*>(img + sample * stride + (16 * s + 8 * h) * 2);
}
// accumulator rows of wave block w as a function of (reg, h): feature = 32 w + acc_row(reg, h)
// store 16 accumulator values (f16) of one sample into an image: four 8-byte pieces
__device__ __forceinline__ void store_acc_h(unsigned char* img, int stride, int sample, int w, int h, const f16v& a, bool relu,
                                            unsigned char* img_lo) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float x[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = relu ? fmaxf(a[4 * g + q], 0.0f) : a[4 * g + q];
    h4 hi = {(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3]};
    unsigned char* p = img + sample * stride + (32 * w + 8 * g + 4 * h) * 2;
    *reinterpret_cast<h4*>(p) = hi;
    if (img_lo) {
      h4 lo = {(_Float16)(x[0] - (float)hi[0]), (_Float16)(x[1] - (float)hi[1]), (_Float16)(x[2] - (float)hi[2]),
               (_Float16)(x[3] - (float)hi[3])};
      *reinterpret_cast<h4*>(img_lo + sample * stride + (32 * w + 8 * g + 4 * h) * 2) = lo;
    }
  }
}
__device__ __forceinline__ f16v bias_acc(const float* __restrict__ b, int w, int h, int nvalid) {
  f16v a;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int o = 32 * w + acc_row(reg, h);
    a[reg] = o < nvalid ? b[o] : 0.0f;
  }
  return a;
}
// copy a [TS][128] f16 image (stride ST_X) to global rows [m0, m0 + TS) of an (M, 128) f16 array, 16 bytes per thread and trip
__device__ __forceinline__ void image_to_global(const unsigned char* img, _Float16* __restrict__ dst, int m0, int M) {
#pragma unroll
  for (int i = threadIdx.x; i < TS * 16; i += 256) {
    const int row = i >> 4, ch = i & 15;
    if (m0 + row < M)
      *reinterpret_cast<f4*>(reinterpret_cast<unsigned char*>(dst + (size_t)(m0 + row) * BH) + ch * 16) =
          *reinterpret_cast<const f4*>(img + row * ST_X + ch * 16);
  }
}
// The opposite copy in two halves, so that the rows' round trip runs under the layer step in front of it: image_fetch issues this
// thread's two 16-byte loads (TS x 16 pieces over 256 threads), image_commit stores them.  (As one loop the compiler kept the two
// trips serial -- load, wait, store, load, wait, store: two dependent memory round trips per layer step of the backward.)
struct ImgRows { f4 v[TS * 16 / 256]; };
__device__ __forceinline__ ImgRows image_fetch(const _Float16* __restrict__ src, int m0, int M) {
  ImgRows r;
#pragma unroll
  for (int t = 0; t < TS * 16 / 256; ++t) {
    const int i = threadIdx.x + 256 * t, row = i >> 4, ch = i & 15;
    const int rc = m0 + row < M ? m0 + row : M - 1;
    r.v[t] = *reinterpret_cast<const f4*>(reinterpret_cast<const unsigned char*>(src + (size_t)rc * BH) + ch * 16);
  }
  return r;
}
__device__ __forceinline__ void image_commit(unsigned char* img, const ImgRows& r, int m0, int M) {
#pragma unroll
  for (int t = 0; t < TS * 16 / 256; ++t) {
    const int i = threadIdx.x + 256 * t, row = i >> 4, ch = i & 15;
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f4*>(img + row * ST_X + ch * 16) = m0 + row < M ? r.v[t] : z;
  }
}

// positional encoding of the tile: thread = (sample c, direction group g); natural feature order e[0..2] = t,
// e[3 + 21 b + d] = sin(pi 2^b p_d) (src/embedding.py:82-92) into the E1 (features 0..86, hi and residual) and E2 (87..128)
// images; pads zeroed.  Also the hi features to global rows (M, 144) for the weight-gradient kernel.
__device__ __forceinline__ void pe_images(const float* __restrict__ pts, const float* __restrict__ Bm, float inv_scale, int m0,
                                          int M, unsigned char* E1h, unsigned char* E1l, unsigned char* E2h) {
  constexpr int NG = 256 / TS;     // direction groups
  const int c = threadIdx.x % TS, g = threadIdx.x / TS;
  const int m = m0 + c < M ? m0 + c : M - 1;
  const float t0 = pts[(size_t)m * 3 + 0] * inv_scale, t1 = pts[(size_t)m * 3 + 1] * inv_scale, t2 = pts[(size_t)m * 3 + 2] * inv_scale;
  _Float16* e1h = reinterpret_cast<_Float16*>(E1h + c * ST_E1B);
  _Float16* e1l = E1l ? reinterpret_cast<_Float16*>(E1l + c * ST_E1B) : nullptr;
  _Float16* e2h = reinterpret_cast<_Float16*>(E2h + c * ST_E2B);
  auto put1 = [&](int f, float v) {
    const _Float16 hi = (_Float16)v;
    e1h[f] = hi;
    if (e1l) e1l[f] = (_Float16)(v - (float)hi);
  };
  if (g == 0) {
    put1(0, t0); put1(1, t1); put1(2, t2);
    for (int f = BE1; f < BE1P; ++f) put1(f, 0.0f);
    for (int f = BE2; f < BE2P; ++f) e2h[f] = (_Float16)0;
  }
  for (int d = g; d < 21; d += NG) {
    const float p = Bm[3 * d] * t0 + Bm[3 * d + 1] * t1 + Bm[3 * d + 2] * t2;
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const float v = __builtin_amdgcn_sinf(p * (0.5f * (float)(1 << b)));   // sin(pi 2^b p): the hardware sine takes revolutions
      const int f = 3 + 21 * b + d;
      if (b < 4) put1(f, v); else e2h[f - BE1] = (_Float16)v;
    }
  }
}

// The weight fragments of a k-range (hi, and for the geometry branch the residual) as a register set, loaded from global / L2 right
// in front of their use.  (Loading them a layer ahead was measured: 256 instead of 82 registers, two instead of three workgroups
// per CU, forward 25.5 instead of 20.5 us -- the neighbours on the CU hide the round trip better than a prefetch does.  Requesting
// the next layer's set behind this layer's products, in front of its epilogue and barrier -- one set at a time, 138 registers --
// changed nothing here either: 28.7 against 28.3 us eager; the backward, whose sets are half the size, gains from it.)
template <int NS, bool GEO>
struct Frags {
  h8 wh[NS], wl[GEO ? NS : 1];
  __device__ __forceinline__ void load(const unsigned char* __restrict__ pk, int f0, int lane) {
#pragma unroll
    for (int s = 0; s < NS; ++s) wh[s] = gfrag(pk + PK_FWD_OFF, f0 + s, lane);
    if constexpr (GEO) {
#pragma unroll
      for (int s = 0; s < NS; ++s) wl[s] = gfrag(pk + PK_LO_OFF, f0 + s, lane);
    }
  }
  // acc[half] += W x image columns; GEO: three products with the residual weights / image
  __device__ __forceinline__ void mma(const unsigned char* Xh, const unsigned char* Xl, int stride, int lane, f16v (&acc)[NH]) const {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int half = 0; half < NH; ++half) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const h8 xh = xfrag(Xh, stride, c + 32 * half, s, h);
        acc[half] = MFMA(wh[s], xh, acc[half]);
        if constexpr (GEO) {
          acc[half] = MFMA(wl[s], xh, acc[half]);
          acc[half] = MFMA(wh[s], xfrag(Xl, stride, c + 32 * half, s, h), acc[half]);
        }
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------------
struct BgFwdArgs {
  const float* pts; const float* theta; const unsigned char* packed; float inv_scale; int M;
  float* sigma; float* rgb;            // (M,), (M,3)
  _Float16* act;                       // 5 x (M,128): a1 .. a5 (post-ReLU, f16)
  _Float16* eimg;                      // (M,144): PE features (hi), E1 in columns 0..95, E2 in 96..143
};
__device__ __forceinline__ void store_tile(unsigned char* img, unsigned char* img_lo, int c, int w, int h, const f16v (&acc)[NH], bool relu) {
#pragma unroll
  for (int half = 0; half < NH; ++half) store_acc_h(img, ST_X, c + 32 * half, w, h, acc[half], relu, img_lo);
}
__device__ __forceinline__ void init_tile(f16v (&acc)[NH], const float* __restrict__ b, int w, int h, int nvalid) {
  const f16v v = bias_acc(b, w, h, nvalid);
#pragma unroll
  for (int half = 0; half < NH; ++half) acc[half] = v;
}

__global__ __launch_bounds__(256, 2) void bg_fwd_kernel(BgFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* E1h = smem;
  unsigned char* E1l = E1h + E1IMG;
  unsigned char* E2h = E1l + E1IMG;
  unsigned char* Xh0 = E2h + E2IMG;
  unsigned char* Xl0 = Xh0 + XIMG;
  unsigned char* Xh1 = Xl0 + XIMG;
  unsigned char* Xl1 = Xh1 + XIMG;
  float* sigp = reinterpret_cast<float*>(Xl1 + XIMG);   // [4][TS]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * TS, M = a.M;
  const float* th = a.theta;
  pe_images(a.pts, th + O_PE_B, a.inv_scale, m0, M, E1h, E1l, E2h);
  __syncthreads();
  // PE features (hi) -> global for the weight-gradient kernel: rows of 288 bytes = 18 x 16
  // (a.act / a.eimg == NULL: the forward alone -- Trainer.eval_points' meshing queries: nothing is kept for a backward)
  const bool keep = a.act != nullptr;
  for (int i = threadIdx.x; keep && i < TS * 18; i += 256) {
    const int row = i / 18, ch = i % 18;
    if (m0 + row < M) {
      const f4 v = ch < 12 ? *reinterpret_cast<const f4*>(E1h + row * ST_E1B + ch * 16)
                           : *reinterpret_cast<const f4*>(E2h + row * ST_E2B + (ch - 12) * 16);
      *reinterpret_cast<f4*>(reinterpret_cast<unsigned char*>(a.eimg + (size_t)(m0 + row) * BEP) + ch * 16) = v;
    }
  }
  f16v acc[NH];
  const size_t MH = (size_t)M * BH;
  // ---- in_layer: E1 -> a1 -----------------------------------------------------------------------------------------
  init_tile(acc, th + O_IN_B, w, h, BH);
  Frags<6, true> f_in;
  f_in.load(a.packed, fwd_base(L_IN) + 6 * w, lane);            // in flight under the positional encoding
  f_in.mma(E1h, E1l, ST_E1B, lane, acc);
  store_tile(Xh0, Xl0, c, w, h, acc, true);
  __syncthreads();
  if (keep) image_to_global(Xh0, a.act + 0 * MH, m0, M);
  // ---- mid1: a1 -> a2 -------------------------------------------------------------------------------------------------
  init_tile(acc, th + O_M1_B, w, h, BH);
  Frags<8, true> f_m1;
  f_m1.load(a.packed, fwd_base(L_M1) + 8 * w, lane);
  f_m1.mma(Xh0, Xl0, ST_X, lane, acc);
  store_tile(Xh1, Xl1, c, w, h, acc, true);
  __syncthreads();
  if (keep) image_to_global(Xh1, a.act + 1 * MH, m0, M);
  // ---- cat_layer: [a2 | e1] -> a3 (image 0 = a1 was last read before the barrier above) ------------------------------
  init_tile(acc, th + O_CAT_B, w, h, BH);
  Frags<8, true> f_ca;
  Frags<6, true> f_ce;
  f_ca.load(a.packed, fwd_base(L_CAT) + 14 * w, lane);
  f_ce.load(a.packed, fwd_base(L_CAT) + 14 * w + 8, lane);
  f_ca.mma(Xh1, Xl1, ST_X, lane, acc);
  f_ce.mma(E1h, E1l, ST_E1B, lane, acc);
  store_tile(Xh0, Xl0, c, w, h, acc, true);
  __syncthreads();
  if (keep) image_to_global(Xh0, a.act + 2 * MH, m0, M);
  // ---- mid2: a3 -> a4, and the x10 occupancy head as an fp32 dot product of the fp32 activations -----------------------
  init_tile(acc, th + O_M2_B, w, h, BH);
  Frags<8, true> f_m2;
  f_m2.load(a.packed, fwd_base(L_M2) + 8 * w, lane);
  f_m2.mma(Xh0, Xl0, ST_X, lane, acc);
  {
    float wa[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) wa[reg] = th[O_OA_W + 32 * w + acc_row(reg, h)];
#pragma unroll
    for (int half = 0; half < NH; ++half) {
      float part = 0.0f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) part = fmaf(wa[reg], fmaxf(acc[half][reg], 0.0f), part);
      part += __shfl_xor(part, 32, 64);
      if (h == 0) sigp[w * TS + c + 32 * half] = part;
    }
  }
  store_tile(Xh1, nullptr, c, w, h, acc, true);   // image 1 = a2: last read by cat_layer, a barrier ago
  __syncthreads();
  if (keep) image_to_global(Xh1, a.act + 3 * MH, m0, M);
  if (threadIdx.x < TS && m0 + threadIdx.x < M) {
    const int s = threadIdx.x;   // alpha = raw * 10 (src/model.py:142)
    a.sigma[m0 + s] = 10.0f * (((sigp[s] + sigp[TS + s]) + (sigp[2 * TS + s] + sigp[3 * TS + s])) + th[O_OA_B]);
  }
  // ---- color_linear: [a4 | e2] -> a5 (colour branch: plain f16) --------------------------------------------------------
  init_tile(acc, th + O_CL_B, w, h, BH);
  Frags<8, false> f_cl;
  Frags<3, false> f_c2;
  f_cl.load(a.packed, fwd_base(L_CL) + 11 * w, lane);
  f_c2.load(a.packed, fwd_base(L_CL) + 11 * w + 8, lane);
  f_cl.mma(Xh1, nullptr, ST_X, lane, acc);
  f_c2.mma(E2h, nullptr, ST_E2B, lane, acc);
  store_tile(Xh0, nullptr, c, w, h, acc, true);   // image 0 = a3: last read by mid2
  __syncthreads();
  if (keep) image_to_global(Xh0, a.act + 4 * MH, m0, M);
  // ---- out_color (3 outputs): wave 0, rows 0..2 = registers 0..2 of lane half 0 ------------------------------------------
  if (w == 0) {
    init_tile(acc, th + O_OC_B, 0, h, 3);
    Frags<8, false> f_oc;
    f_oc.load(a.packed, fwd_base(L_OC), lane);
    f_oc.mma(Xh0, nullptr, ST_X, lane, acc);
    if (h == 0) {
#pragma unroll
      for (int half = 0; half < NH; ++half) {
        const int m = m0 + c + 32 * half;
        if (m < M) {
          float* o = a.rgb + (size_t)m * 3;
          o[0] = 1.0f / (1.0f + __expf(-acc[half][0])); o[1] = 1.0f / (1.0f + __expf(-acc[half][1]));
          o[2] = 1.0f / (1.0f + __expf(-acc[half][2]));
        }
      }
    }
  }
}
constexpr int BG_FWD_LDS = 2 * E1IMG + E2IMG + 4 * XIMG + 4 * TS * 4;

// ------------------------------------------------------------------------------------------------------------------------
// backward (data-gradient chain)
// ------------------------------------------------------------------------------------------------------------------------
constexpr int BG_REC = 640;   // per-workgroup record (floats): dW out_color 384 | db out_color 3 (+1) | dW out_alpha 128 | db out_alpha 1 (+3) | dB 63 (+..)
constexpr int R_OCW = 0, R_OCB = 384, R_OAW = 388, R_OAB = 516, R_PEB = 520;
struct BgBwdArgs {
  const float* pts; const float* theta; const unsigned char* packed; float inv_scale; int M;
  const float* d_sigma; const float* d_rgb; const float* rgb;       // upstream gradients (already x grad_scale), forward colours
  const _Float16* act;                 // 5 x (M,128)
  _Float16* dpre;                      // 5 x (M,128): scaled pre-activation gradients of in_layer, mid1, cat, mid2, color_linear
  float* records;                      // (blocks, BG_REC)
  int64_t* d_state; int64_t add_rows;  // d_state != NULL: block 0 advances the step state (see cnr_bg_backward)
  // RENDER (cnr_bg_backward_render): the composite, the losses and their gradient in front of the chain -- d_sigma / d_rgb unused
  const float* sigma; const float* z; const float* gt_depth; const float* gt_rgb; const uint8_t* labels; const uint8_t* depth_mask;
  const float* counts_tab; const int64_t* cursor; int R, S; float color_scaling, opacity_scaling, grad_scale;
  float* depth_out; float* var_out; float* rgb_out; float* opacity_out; float* rl_partials;
  float* d_sigma_out; float* d_rgb_out;   // optional: the loss gradient per sample, as cnr_render_loss writes it
};

// the 8 k-steps (128 output features) of W^T block `fb`; acc[half] += W^T x dPre image
struct BFrags {
  h8 wt[8];
  __device__ __forceinline__ void load(const unsigned char* __restrict__ pk, int fb, int lane) {
#pragma unroll
    for (int s = 0; s < 8; ++s) wt[s] = gfrag(pk + PK_BWD_OFF, fb + s, lane);
  }
  __device__ __forceinline__ void mma(const unsigned char* D, int lane, f16v (&acc)[NH]) const {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int half = 0; half < NH; ++half)
#pragma unroll
      for (int s = 0; s < 8; ++s) acc[half] = MFMA(wt[s], xfrag(D, ST_X, c + 32 * half, s, h), acc[half]);
  }
};
// dPre = d act * (act > 0): the activation image gives this lane's 16 features in the accumulator layout
__device__ __forceinline__ void mask_store(unsigned char* Dimg, const unsigned char* Aimg, int sample, int w, int h, const f16v& a) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const h4 act = *reinterpret_cast<const h4*>(Aimg + sample * ST_X + (32 * w + 8 * g + 4 * h) * 2);
    h4 d;
#pragma unroll
    for (int q = 0; q < 4; ++q) d[q] = act[q] > (_Float16)0 ? (_Float16)a[4 * g + q] : (_Float16)0;
    *reinterpret_cast<h4*>(Dimg + sample * ST_X + (32 * w + 8 * g + 4 * h) * 2) = d;
  }
}
__device__ __forceinline__ void mask_tile(unsigned char* Dimg, const unsigned char* Aimg, int c, int w, int h, const f16v (&acc)[NH]) {
#pragma unroll
  for (int half = 0; half < NH; ++half) mask_store(Dimg, Aimg, c + 32 * half, w, h, acc[half]);
}
__device__ __forceinline__ void zero_tile(f16v (&acc)[NH]) {
#pragma unroll
  for (int half = 0; half < NH; ++half)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[half][i] = 0.0f;
}

// dPre of out_color for one colour channel, d rgb * rgb (1 - rgb), rounded to fp32 and THEN to f16 in both forms of the kernel
// (gradient from memory / from the in-kernel composite): left alone, the compiler turned the last product and the conversion
// into one v_fma_mixlo_f16 -- a single rounding -- in one form and not in the other, three samples of 3360 one f16 ulp apart.
__device__ __forceinline__ _Float16 oc_dpre(float d, float r) {
  float v = d * r * (1.0f - r);
  asm volatile("" : "+v"(v));   // no instruction: the product exists as an fp32 value
  return (_Float16)v;
}
// (The mirror image at the other end of the step -- the sampler inside the FORWARD launch, every tile sampling the rays its samples
//  belong to with cnr_sample_rays' per-ray function -- was built the same way, bit-equal to the two calls, and changed nothing:
//  85.4 against 85.1 us per step.  The sampler's three dependent round trips (cursor -> permutation -> pool row) then stand in
//  front of every forward workgroup at once; as a launch of its own they cost the same.  Not kept.)
template <bool RENDER>
__global__ __launch_bounds__(256, 2) void bg_bwd_kernel(BgBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* D0 = smem;
  unsigned char* D1 = D0 + XIMG;
  unsigned char* Aim = D1 + XIMG;
  float* DE1 = reinterpret_cast<float*>(Aim + XIMG);        // [TS][96] d e1 (fp32, scaled)
  float* DE2 = DE1 + TS * BE1P;                              // [TS][48]
  unsigned char* OCimg = reinterpret_cast<unsigned char*>(DE2 + TS * BE2P);   // [TS][16] f16 (+ pad): dPre of out_color in columns 0..2
  float* dsg = reinterpret_cast<float*>(OCimg + TS * ST_OC);                   // [TS] 10 x scaled d sigma
  float* tpos = dsg + TS;                                                      // [TS][3] t = x / scale
  float* lsum = tpos + 3 * TS;                                                 // [4][3] RENDER: the waves' loss terms
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * TS, M = a.M;
  const float* th = a.theta;
  const size_t MH = (size_t)M * BH;
  float* rec = a.records + (size_t)blockIdx.x * BG_REC;
  // the step state moves here: the sampler (which reads the cursor) ran earlier in the step, the optimiser launch (which reads
  // the step count) comes later -- no launch of its own for three integer adds.  (RENDER: every block reads the cursor for its
  // mask-count entry, so the state moves one launch later, in cnr_bg_dw.)
  if (!RENDER && a.d_state && blockIdx.x == 0 && threadIdx.x == 0) { a.d_state[0] += a.add_rows; a.d_state[1] += 1; a.d_state[2] += 1; }
  // the f16 chain needs |10 d sigma| within range: clipped like the category kernel's (include/cnr_hip.h, clamp_flags)
  auto clip10 = [](float v) { return fminf(fmaxf(v, -8192.0f), 8192.0f) * 10.0f; };
  if constexpr (RENDER) {
    // ---- a11-a15 for the rays this tile's samples belong to (src/render_rays.py:3-7,25-33,46-95; src/loss.py:18-74): one wave
    // per ray, lane = sample (S <= 64), cnr_render_loss's arithmetic (render_common.h: ray_small).  A ray that straddles two tiles
    // is composited by both; the tile that holds its FIRST sample accounts for its renders and loss terms.
    const int S = a.S;
    const float* t = a.counts_tab + (size_t)(a.cursor ? a.cursor[0] / a.R : 0) * 8;      // (slices, C + 1 = 2, 4), C = 1
    const float wd = t[4] != 0.f ? 0.f : 1.0f / (t[0] + 1e-10f), wc = t[5] != 0.f ? 0.f : 1.0f / (t[1] + 1e-10f),
                wo = t[6] != 0.f ? 0.f : 1.0f / (t[2] + 1e-10f);
    if (threadIdx.x < TS) {   // columns 3..15 of the out_color image, rows beyond M, sample positions
      const int s = threadIdx.x, m = m0 + s;
      _Float16* o = reinterpret_cast<_Float16*>(OCimg + s * ST_OC);
#pragma unroll
      for (int j = 3; j < 16; ++j) o[j] = (_Float16)0.0f;
      if (m >= M) { o[0] = o[1] = o[2] = (_Float16)0.0f; dsg[s] = 0.0f; }
      const int mc = m < M ? m : M - 1;
      tpos[3 * s + 0] = a.pts[(size_t)mc * 3 + 0] * a.inv_scale; tpos[3 * s + 1] = a.pts[(size_t)mc * 3 + 1] * a.inv_scale;
      tpos[3 * s + 2] = a.pts[(size_t)mc * 3 + 2] * a.inv_scale;
    }
    const int mlast = (m0 + TS < M ? m0 + TS : M) - 1;
    float ld = 0.f, lc = 0.f, lo = 0.f;
    for (int ray = m0 / S + w; ray <= mlast / S; ray += 4) {   // wave-uniform
      const size_t base = (size_t)ray * S;
      const cnr_rl::RayOut q = cnr_rl::ray_small(a.sigma, a.rgb, a.z, base, S, lane, a.gt_depth[ray], a.gt_rgb[ray * 3 + 0],
                                                 a.gt_rgb[ray * 3 + 1], a.gt_rgb[ray * 3 + 2], a.labels[ray], a.depth_mask[ray],
                                                 wd, wc, wo, a.color_scaling, a.opacity_scaling, a.grad_scale);
      const bool owner = (int64_t)base >= m0;                  // the ray's first sample lies in this tile
      if (owner) {
        ld += q.ld; lc += q.lc; lo += q.lo;
        if (lane == 0) {
          if (a.depth_out) a.depth_out[ray] = q.sd;
          if (a.var_out) a.var_out[ray] = q.sv;
          if (a.opacity_out) a.opacity_out[ray] = q.so;
          if (a.rgb_out) { a.rgb_out[ray * 3 + 0] = q.sr; a.rgb_out[ray * 3 + 1] = q.sg; a.rgb_out[ray * 3 + 2] = q.sb; }
        }
      }
      const int64_t m = (int64_t)base + lane;
      if (q.live && m >= m0 && m <= mlast) {
        const int s = (int)(m - m0);
        const float* cp = a.rgb + (size_t)m * 3;
        _Float16* o = reinterpret_cast<_Float16*>(OCimg + s * ST_OC);
        o[0] = oc_dpre(q.dc0, cp[0]); o[1] = oc_dpre(q.dc1, cp[1]); o[2] = oc_dpre(q.dc2, cp[2]);
        dsg[s] = clip10(q.dsig);
        if (a.d_sigma_out) a.d_sigma_out[m] = q.dsig;
        if (a.d_rgb_out) { float* dc = a.d_rgb_out + (size_t)m * 3; dc[0] = q.dc0; dc[1] = q.dc1; dc[2] = q.dc2; }
      }
    }
    if (lane == 0) { lsum[3 * w + 0] = ld; lsum[3 * w + 1] = lc; lsum[3 * w + 2] = lo; }
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // the normalisers and the empty-mask bits (cnr_rl::finish_class reads them)
      float* hdr = a.rl_partials + (size_t)gridDim.x * 3;
      hdr[0] = wd; hdr[1] = wc; hdr[2] = wo;
      hdr[3] = (float)((t[4] != 0.f ? 2 : 0) | (t[5] != 0.f ? 4 : 0) | (t[6] != 0.f ? 8 : 0));
    }
  } else if (threadIdx.x < TS) {
    // ---- upstream: dPre of out_color = d rgb * rgb (1 - rgb); 10 x d sigma; sample positions ----------------------------
    const int s = threadIdx.x, m = m0 + s;
    const bool ok = m < M;
    _Float16* o = reinterpret_cast<_Float16*>(OCimg + s * ST_OC);
#pragma unroll
    for (int j = 0; j < 16; ++j)
      o[j] = (ok && j < 3) ? oc_dpre(a.d_rgb[(size_t)m * 3 + j], a.rgb[(size_t)m * 3 + j]) : (_Float16)0.0f;
    dsg[s] = ok ? clip10(a.d_sigma[m]) : 0.0f;
    const int mc = ok ? m : M - 1;
    tpos[3 * s + 0] = a.pts[(size_t)mc * 3 + 0] * a.inv_scale; tpos[3 * s + 1] = a.pts[(size_t)mc * 3 + 1] * a.inv_scale;
    tpos[3 * s + 2] = a.pts[(size_t)mc * 3 + 2] * a.inv_scale;
  }
  BFrags bf;
  // activation images: the rows of layer step k + 1 are requested when step k's have been stored, a whole step ahead of their use
  ImgRows nx = image_fetch(a.act + 4 * MH, m0, M);
  image_commit(Aim, nx, m0, M);                   // a5
  nx = image_fetch(a.act + 3 * MH, m0, M);
  __syncthreads();
  if constexpr (RENDER) {
    if (threadIdx.x < 3)
      a.rl_partials[(size_t)blockIdx.x * 3 + threadIdx.x] =
          ((lsum[threadIdx.x] + lsum[3 + threadIdx.x]) + lsum[6 + threadIdx.x]) + lsum[9 + threadIdx.x];
  }
  // ---- out_color: weight / bias gradient on the VALU (3 x 128 + 3), d a5 on the matrix core -----------------------------
  {
    const int f = threadIdx.x & 127, jj = threadIdx.x >> 7;   // outputs j = jj and jj + 2
    float s0 = 0.0f, s1 = 0.0f;
    for (int s = 0; s < TS; ++s) {
      const float x = (float)reinterpret_cast<const _Float16*>(Aim + s * ST_X)[f];
      const _Float16* o = reinterpret_cast<const _Float16*>(OCimg + s * ST_OC);
      s0 = fmaf((float)o[jj], x, s0);
      if (jj == 0) s1 = fmaf((float)o[2], x, s1);
    }
    rec[R_OCW + jj * BH + f] = s0;
    if (jj == 0) rec[R_OCW + 2 * BH + f] = s1;
    if (threadIdx.x < 3) {
      float sb = 0.0f;
      for (int s = 0; s < TS; ++s) sb += (float)reinterpret_cast<const _Float16*>(OCimg + s * ST_OC)[threadIdx.x];
      rec[R_OCB + threadIdx.x] = sb;
    }
  }
  f16v acc[NH], ae[NH];
  zero_tile(acc);
  {
    const h8 wt = gfrag(a.packed + PK_BWD_OFF, bwd_base(L_OC) + w, lane);
#pragma unroll
    for (int half = 0; half < NH; ++half) {
      const h8 x = *reinterpret_cast<const h8*>(OCimg + (c + 32 * half) * ST_OC + 8 * h * 2);
      acc[half] = MFMA(wt, x, acc[half]);
    }
  }
  // (every layer step's transposed fragments are requested HERE, behind the previous step's products and in front of its mask /
  //  barrier / store phase: the register set is free at that point and the L2 round trip runs under that phase)
  bf.load(a.packed, bwd_base(L_CL) + 8 * w, lane);
  mask_tile(D0, Aim, c, w, h, acc);      // dPre5
  __syncthreads();
  image_to_global(D0, a.dpre + 4 * MH, m0, M);
  // ---- color_linear^T: d a4 (colour part) + d e2 ; + out_alpha ------------------------------------------------------------
  zero_tile(acc);
  bf.mma(D0, lane, acc);
  if (w < 2) {
    zero_tile(ae);
    bf.load(a.packed, bwd_base(L_CL) + 8 * (4 + w), lane);
    bf.mma(D0, lane, ae);
#pragma unroll
    for (int half = 0; half < NH; ++half)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int q = 32 * w + acc_row(reg, h);
        if (q < BE2P) DE2[(c + 32 * half) * BE2P + q] = ae[half][reg];
      }
  }
  bf.load(a.packed, bwd_base(L_M2) + 8 * w, lane);
  image_commit(Aim, nx, m0, M);                   // a4 (the a5 image was last read before the barrier above)
  nx = image_fetch(a.act + 2 * MH, m0, M);
  __syncthreads();
  {
    // out_alpha: dW = sum_s (10 d sigma_s) a4[s][f], db = sum_s 10 d sigma_s ; d a4 += w_alpha 10 d sigma
    if (threadIdx.x < 128) {
      float sw = 0.0f;
      for (int s = 0; s < TS; ++s) sw = fmaf(dsg[s], (float)reinterpret_cast<const _Float16*>(Aim + s * ST_X)[threadIdx.x], sw);
      rec[R_OAW + threadIdx.x] = sw;
    } else if (threadIdx.x == 128) {
      float sb = 0.0f;
      for (int s = 0; s < TS; ++s) sb += dsg[s];
      rec[R_OAB] = sb;
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const float wa = th[O_OA_W + 32 * w + acc_row(reg, h)];
#pragma unroll
      for (int half = 0; half < NH; ++half) acc[half][reg] = fmaf(wa, dsg[c + 32 * half], acc[half][reg]);
    }
  }
  mask_tile(D1, Aim, c, w, h, acc);      // dPre4
  __syncthreads();
  image_to_global(D1, a.dpre + 3 * MH, m0, M);
  // ---- mid2^T -> dPre3 ---------------------------------------------------------------------------------------------------
  zero_tile(acc);
  bf.mma(D1, lane, acc);
  bf.load(a.packed, bwd_base(L_CAT) + 8 * w, lane);
  image_commit(Aim, nx, m0, M);                   // a3 (the a4 image was last read before the barrier above)
  nx = image_fetch(a.act + 1 * MH, m0, M);
  __syncthreads();
  mask_tile(D0, Aim, c, w, h, acc);
  __syncthreads();
  image_to_global(D0, a.dpre + 2 * MH, m0, M);
  // ---- cat_layer^T -> d a2, d e1 -------------------------------------------------------------------------------------------
  zero_tile(acc);
  bf.mma(D0, lane, acc);
  if (w < 3) {
    zero_tile(ae);
    bf.load(a.packed, bwd_base(L_CAT) + 8 * (4 + w), lane);
    bf.mma(D0, lane, ae);
  }
  bf.load(a.packed, bwd_base(L_M1) + 8 * w, lane);
  image_commit(Aim, nx, m0, M);                   // a2
  nx = image_fetch(a.act + 0 * MH, m0, M);
  __syncthreads();
  mask_tile(D1, Aim, c, w, h, acc);      // dPre2
  __syncthreads();
  image_to_global(D1, a.dpre + 1 * MH, m0, M);
  // ---- mid1^T -> dPre1 -----------------------------------------------------------------------------------------------------
  zero_tile(acc);
  bf.mma(D1, lane, acc);
  if (w < 3) bf.load(a.packed, bwd_base(L_IN) + 8 * w, lane);
  image_commit(Aim, nx, m0, M);                   // a1
  __syncthreads();
  mask_tile(D0, Aim, c, w, h, acc);
  __syncthreads();
  image_to_global(D0, a.dpre + 0 * MH, m0, M);
  // ---- in_layer^T -> d e1 (on top of cat_layer's) --------------------------------------------------------------------------
  if (w < 3) {
    bf.mma(D0, lane, ae);
#pragma unroll
    for (int half = 0; half < NH; ++half)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        DE1[(c + 32 * half) * BE1P + 32 * w + acc_row(reg, h)] = ae[half][reg];
  }
  __syncthreads();
  // ---- PE backward -> d B_layer.weight: a wave takes 64 / TS directions at a time; lane = (direction slot, sample) ---------
  {
    constexpr int DPW = 64 / TS;                       // directions per wave and trip
    const int s = lane % TS, dsel = lane / TS;
    const bool ok = m0 + s < M;
    const float t0 = tpos[3 * s], t1 = tpos[3 * s + 1], t2 = tpos[3 * s + 2];
    for (int d0 = w * DPW; d0 < 21; d0 += 4 * DPW) {
      const int d = d0 + dsel, dc = d < 21 ? d : 20;
      const float p = th[O_PE_B + 3 * dc] * t0 + th[O_PE_B + 3 * dc + 1] * t1 + th[O_PE_B + 3 * dc + 2] * t2;
      float gp = 0.0f;
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const int f = 3 + 21 * b + dc;
        const float de = b < 4 ? DE1[s * BE1P + f] : DE2[s * BE2P + (f - BE1)];
        gp = fmaf(de * __builtin_amdgcn_cosf(p * (0.5f * (float)(1 << b))), 3.14159265358979f * (float)(1 << b), gp);
      }
      if (!ok || d >= 21) gp = 0.0f;
      float g0 = gp * t0, g1 = gp * t1, g2 = gp * t2;
#pragma unroll
      for (int o = TS / 2; o > 0; o >>= 1) { g0 += __shfl_xor(g0, o, 64); g1 += __shfl_xor(g1, o, 64); g2 += __shfl_xor(g2, o, 64); }
      if (s == 0 && d < 21) { rec[R_PEB + 3 * d] = g0; rec[R_PEB + 3 * d + 1] = g1; rec[R_PEB + 3 * d + 2] = g2; }
    }
  }
}
constexpr int BG_BWD_LDS = 3 * XIMG + TS * (BE1P + BE2P) * 4 + TS * ST_OC + TS * 4 + TS * 12 + 64;   // (+ 12 floats of loss terms inside the 64)

// ------------------------------------------------------------------------------------------------------------------------
// weight gradients of the five 128-wide layers: dW[n][k] = sum_m dPre[m][n] X[m][k], db[n] = sum_m dPre[m][n]
// grid (5, chunks): block = layer x sample chunk.  The workgroup stages 64-sample tiles of dPre (all 128 outputs) and of the
// layer's input rows in LDS (the next tile's global loads are in flight under the products of the current one); a
// [sample][feature] tile gives both MFMA operands through transposing reads (ds_read_b64_tr_b16).  Wave w owns output block
// [32 w, 32 w + 32) with ALL of the layer's 32-wide input blocks (+ one "ones" block whose column 0 is the bias gradient):
// up to eight accumulators, kept for the whole chunk and written out by the wave itself -- every input row is read from
// memory once per chunk (a block per (layer, output block) read it four times: the kernel was bound by those reads).
// (Two tiles' loads in flight instead of one -- a second register set, 407 registers -- measured 21.9 against 19.5 us eager.  The
//  five layers of a chunk dealt to ONE XCD, so that the three readers of the chunk's PE rows share an L2 copy: no change, 79.2
//  against 79.2 us per step -- what this kernel re-reads comes out of the memory-side cache either way.)
// ------------------------------------------------------------------------------------------------------------------------
typedef short s4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ h8 tr_frag_b(const unsigned char* img, int stride, int col0, int s, int lane) {
  const int i = lane & 15, g16 = lane >> 4, q = i >> 2, p = i & 3, hh = g16 >> 1;
  const unsigned char* ad = img + (16 * s + 8 * hh + q) * stride + (col0 + 16 * (g16 & 1) + 4 * p) * 2;
  typedef __attribute__((address_space(3))) s4v* lds_s4;
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(ad));
  const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(ad + 4 * stride));
  h8 r;
  r.lo = __builtin_bit_cast(h4, lo);
  r.hi = __builtin_bit_cast(h4, hi);
  return r;
}
struct BgDwArgs {
  const _Float16* act; const _Float16* dpre; const _Float16* eimg; int M; int chunk;   // chunk: samples per block (multiple of 64)
  float* partials;                                                                       // (chunks, BG_NPARAM)
  int64_t* d_state; int64_t add_rows;      // d_state != NULL: block (0, 0) advances the step state (see cnr_bg_dw)
};
// tile row strides (bytes), = 48 and 16 mod 64 dwords: the four rows a transposing read's 32 lanes touch (64 bytes each) then
// sit in four different quarters of the 64 banks (272 / 464 bytes put them 4 / 52 dwords apart: 2- to 4-way conflicts)
constexpr int DW_XST = 448;          // X tile: up to 224 features
constexpr int DW_DST = 320;          // dPre tile: 128 features + pad
constexpr int DW_BUF = 64 * DW_XST + 64 * DW_DST;
template <int LAYER>
__device__ __forceinline__ void dw_layer(const BgDwArgs& a, unsigned char* smem) {
  constexpr int KB = LAYER == L_IN ? 3 : LAYER == L_CAT ? 7 : LAYER == L_CL ? 6 : 4;    // 32-wide input blocks
  constexpr int NA = LAYER == L_IN ? 0 : 4;                                                // of which from the hidden activation
  constexpr int AIDX = LAYER == L_M1 ? 0 : LAYER == L_CAT ? 1 : LAYER == L_M2 ? 2 : 3;   // which activation array feeds it
  constexpr int DIDX = LAYER == L_IN ? 0 : LAYER == L_M1 ? 1 : LAYER == L_CAT ? 2 : LAYER == L_M2 ? 3 : 4;
  constexpr int ECOL = LAYER == L_CL ? BE1P : 0, EW = LAYER == L_CL ? BE2P : BE1P;          // PE columns of the (M,144) image
  constexpr int XCH = NA * 4 + (KB > NA ? EW / 8 : 0);                                    // 16-byte pieces per X row (a 32-feature block = 4)
  constexpr int XPT = (64 * XCH + 255) / 256;                                            // X pieces per thread and tile
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t MH = (size_t)a.M * BH;
  f16v acc[KB + 1];
#pragma unroll
  for (int k = 0; k <= KB; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.0f;
  h8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (lane & 31) == 0 ? (_Float16)1 : (_Float16)0;   // column 0 of the bias block = sum over samples
  const int mc0 = blockIdx.y * a.chunk, mc1 = mc0 + a.chunk < a.M ? mc0 + a.chunk : a.M;
  f4 dreg[4], xreg[XPT];
  auto load_tile = [&](int m0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = e * 256 + threadIdx.x, row = i >> 4, ch = i & 15;
      dreg[e] = f4{0.f, 0.f, 0.f, 0.f};
      if (m0 + row < mc1)
        dreg[e] = *reinterpret_cast<const f4*>(reinterpret_cast<const unsigned char*>(a.dpre + DIDX * MH + (size_t)(m0 + row) * BH) + ch * 16);
    }
#pragma unroll
    for (int e = 0; e < XPT; ++e) {
      const int i = e * 256 + threadIdx.x, row = i / XCH, ch = i % XCH;
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < 64 * XCH && m0 + row < mc1) {
        if (ch < NA * 4)
          v = *reinterpret_cast<const f4*>(reinterpret_cast<const unsigned char*>(a.act + AIDX * MH + (size_t)(m0 + row) * BH) + ch * 16);
        else
          v = *reinterpret_cast<const f4*>(reinterpret_cast<const unsigned char*>(a.eimg + (size_t)(m0 + row) * BEP + ECOL) + (ch - NA * 4) * 16);
      }
      xreg[e] = v;
    }
  };
  auto store_tile_lds = [&](unsigned char* buf) {
    unsigned char* Xt = buf;
    unsigned char* Dt = buf + 64 * DW_XST;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = e * 256 + threadIdx.x;
      *reinterpret_cast<f4*>(Dt + (i >> 4) * DW_DST + (i & 15) * 16) = dreg[e];
    }
#pragma unroll
    for (int e = 0; e < XPT; ++e) {
      const int i = e * 256 + threadIdx.x, row = i / XCH, ch = i % XCH;
      if (i < 64 * XCH) *reinterpret_cast<f4*>(Xt + row * DW_XST + ch * 16) = xreg[e];
    }
    if constexpr (LAYER == L_CL) {   // the second PE block is half empty (48 columns): columns 48..63 of it are zero
      if (threadIdx.x < 128) *reinterpret_cast<f4*>(Xt + (threadIdx.x >> 1) * DW_XST + NA * 64 + 96 + (threadIdx.x & 1) * 16) = f4{0.f, 0.f, 0.f, 0.f};
    }
  };
  int buf = 0;
  if (mc0 < mc1) { load_tile(mc0); store_tile_lds(smem); }
  __syncthreads();
  for (int m0 = mc0; m0 < mc1; m0 += 64) {
    const bool more = m0 + 64 < mc1;
    if (more) load_tile(m0 + 64);                      // global loads of the next tile under this tile's products
    const unsigned char* Xt = smem + buf * DW_BUF;
    const unsigned char* Dt = Xt + 64 * DW_XST;
#pragma unroll
    for (int s = 0; s < 4; ++s) {      // 16 samples per k-step
      const h8 df = tr_frag_b(Dt, DW_DST, 32 * wv, s, lane);
#pragma unroll
      for (int k = 0; k < KB; ++k) acc[k] = MFMA(df, tr_frag_b(Xt, DW_XST, 32 * k, s, lane), acc[k]);
      acc[KB] = MFMA(df, ones, acc[KB]);
    }
    if (more) store_tile_lds(smem + (buf ^ 1) * DW_BUF);
    __syncthreads();
    buf ^= 1;
  }
  // ---- out: accumulator rows = outputs n of block wv, lanes = input columns -----------------------------------------------
  float* part = a.partials + (size_t)blockIdx.y * BG_NPARAM;
  const int kcol = lane & 31, hh = lane >> 5;
  constexpr int WOFF = w_off(LAYER), BOFF = b_off(LAYER), LD = ld_of(LAYER);
#pragma unroll
  for (int k = 0; k < KB; ++k) {
    int wcol;
    if (k < NA) wcol = 32 * k + kcol;
    else {
      const int q = 32 * (k - NA) + kcol;
      const int ew = LAYER == L_CL ? BE2 : BE1;
      wcol = q < ew ? (LAYER == L_IN ? q : BH + q) : -1;
    }
    if (wcol >= 0) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) part[WOFF + (32 * wv + acc_row(reg, hh)) * LD + wcol] = acc[k][reg];
    }
  }
  if (kcol == 0) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) part[BOFF + 32 * wv + acc_row(reg, hh)] = acc[KB][reg];
  }
}
__global__ __launch_bounds__(256, 1) void bg_dw_kernel(BgDwArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (a.d_state && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { a.d_state[0] += a.add_rows; a.d_state[1] += 1; a.d_state[2] += 1; }
  switch (blockIdx.x) {
    case 0: dw_layer<L_CAT>(a, smem); break;       // heaviest layer first in the dispatch order
    case 1: dw_layer<L_CL>(a, smem); break;
    case 2: dw_layer<L_M1>(a, smem); break;
    case 3: dw_layer<L_M2>(a, smem); break;
    default: dw_layer<L_IN>(a, smem); break;
  }
}
constexpr int BG_DW_LDS = 2 * DW_BUF;

// ------------------------------------------------------------------------------------------------------------------------
// tail: fixed-order sum of the partials (big layers) and records (small ones), 1 / grad_scale, AdamW in place, step state
// ------------------------------------------------------------------------------------------------------------------------
struct BgTailArgs {
  float* theta; float* grad; float* m; float* v; const float* partials; int chunks; const float* records; int nrec;
  float inv_gscale, lr, b1, b2, eps, wd; int64_t* d_state; int64_t add_rows;
  unsigned char* packed;               // != NULL: the thread that updates a weight also refreshes its f16 fragment slots
  const float* rl_partials; int rl_nb; float* losses; int32_t* flags;   // != NULL: one more block forms the step's loss values
  cnr_sample::SampleArgs sa; int sample_blocks;   // cnr_bg_tail_sample: the NEXT step's rays, four per block behind the others
};
// Where weight (layer l, output o, input column c) sits in the packed images: the forward fragment (and, geometry layers, its
// residual twin at the same index) and the transposed fragment.  The inverse of bg_pack_kernel's maps: with `packed` given to
// the tail, a parameter's fragments are refreshed by the thread that applies its AdamW update and the next step needs no pack
// launch (padding slots keep the zeros of the first cnr_bg_pack).
__device__ __forceinline__ void refresh_slots(unsigned char* packed, int l, int o, int c, float v) {
  const _Float16 hi = (_Float16)v;
  {
    const int wb = l == L_OC ? 0 : o >> 5, r = o & 31, s = c >> 4, h = (c >> 3) & 1, j = c & 7;
    const int f = fwd_base(l) + wb * ks(l) + s, e = (r + 32 * h) * 8 + j;
    reinterpret_cast<_Float16*>(packed + PK_FWD_OFF + (size_t)f * FRAG_BYTES)[e] = hi;
    if (l <= L_M2) reinterpret_cast<_Float16*>(packed + PK_LO_OFF + (size_t)f * FRAG_BYTES)[e] = (_Float16)(v - (float)hi);
  }
  {
    const int ib = c >> 5, r = c & 31, s = l == L_OC ? 0 : o >> 4, h = (o >> 3) & 1, j = o & 7;
    const int f = bwd_base(l) + (l == L_OC ? ib : ib * 8 + s), e = (r + 32 * h) * 8 + j;
    reinterpret_cast<_Float16*>(packed + PK_BWD_OFF + (size_t)f * FRAG_BYTES)[e] = hi;
  }
}
__device__ __forceinline__ void refresh_param(unsigned char* packed, int i, float v) {
  int l = -1;
  if (i < O_IN_B) l = L_IN;
  else if (i >= O_M1_W && i < O_M1_B) l = L_M1;
  else if (i >= O_CAT_W && i < O_CAT_B) l = L_CAT;
  else if (i >= O_M2_W && i < O_M2_B) l = L_M2;
  else if (i >= O_CL_W && i < O_CL_B) l = L_CL;
  else if (i >= O_OC_W && i < O_OC_B) l = L_OC;
  if (l < 0) return;                    // biases, out_alpha and B_layer are read from theta in fp32
  const int rel = i - w_off(l), ld = ld_of(l);
  refresh_slots(packed, l, rel / ld, rel % ld, v);
}
__device__ __forceinline__ int rec_slot(int i) {   // record entry of small parameter i, -1 = a big-layer parameter
  if (i >= O_OA_W && i < O_OA_W + BH) return R_OAW + (i - O_OA_W);
  if (i == O_OA_B) return R_OAB;
  if (i >= O_OC_W && i < O_OC_W + 3 * BH) return R_OCW + (i - O_OC_W);
  if (i >= O_OC_B && i < O_OC_B + 3) return R_OCB + (i - O_OC_B);
  if (i >= O_PE_B) return R_PEB + (i - O_PE_B);
  return -1;
}
// blocks [0, NBIG): one parameter per thread, its partials summed over the chunks; blocks [NBIG, NBIG + BG_REC / 16): 16 record
// entries each, the workgroups' records split over 16 thread groups (eight loads in flight each), combined in group order
__device__ __forceinline__ int rec_param(int e) {   // parameter of record entry e, -1 = padding
  if (e < R_OCW + 3 * BH) return O_OC_W + (e - R_OCW);
  if (e >= R_OCB && e < R_OCB + 3) return O_OC_B + (e - R_OCB);
  if (e >= R_OAW && e < R_OAW + BH) return O_OA_W + (e - R_OAW);
  if (e == R_OAB) return O_OA_B;
  if (e >= R_PEB && e < R_PEB + 63) return O_PE_B + (e - R_PEB);
  return -1;
}
constexpr int TAIL_NBIG = (BG_NPARAM + 255) / 256;
constexpr int TAIL_NREC = BG_REC / 16;
__global__ __launch_bounds__(256) void bg_tail_kernel(BgTailArgs a) {
  if ((int)blockIdx.x >= TAIL_NBIG + TAIL_NREC) {
    // ---- a2-a5 for the NEXT step (cnr_bg_tail_sample): nothing in this launch reads the sampler's outputs, the step state was
    // advanced earlier in the step (cnr_bg_dw), so cursor and RNG step are already the next step's.  At the epoch's last step
    // there is no next slice yet (the host reshuffles, then samples with cnr_sample_rays): nothing is written.
    const int64_t ray = (int64_t)((int)blockIdx.x - TAIL_NBIG - TAIL_NREC) * 4 + (threadIdx.x >> 6);
    if (ray < a.sa.R && a.sa.d_state[0] + a.sa.R <= a.sa.pool_rows) cnr_sample::sample_ray(a.sa, ray, threadIdx.x & 63);
    return;
  }
  __shared__ float part[16][16];
  const int64_t t = a.d_state[2] + (a.add_rows >= 0 ? 1 : 0);   // add_rows < 0: the state was advanced earlier in the step
  cnr::AdamArgs ad{a.theta, a.grad, a.m, a.v, BG_NPARAM, a.lr, a.b1, a.b2, a.eps, a.wd, 1.0f};
  float step_size, inv_bc2;
  cnr::adam_coefficients(ad, t, step_size, inv_bc2);
  int i = -1;
  float g = 0.0f;
  if ((int)blockIdx.x < TAIL_NBIG) {
    i = blockIdx.x * 256 + threadIdx.x;
    if (i >= BG_NPARAM || rec_slot(i) >= 0) i = -1;
    if (i >= 0) {
      float sv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      int c = 0;
      for (; c + 8 <= a.chunks; c += 8) {       // eight loads in flight (the sum is a memory round trip per trip)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = a.partials[(size_t)(c + u) * BG_NPARAM + i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) sv[u] += v[u];
      }
      for (; c < a.chunks; ++c) sv[0] += a.partials[(size_t)c * BG_NPARAM + i];
      g = ((sv[0] + sv[1]) + (sv[2] + sv[3])) + ((sv[4] + sv[5]) + (sv[6] + sv[7]));
    }
  } else {
    // 16 record entries x 16 groups of records per block: ~nrec / 16 loads per thread, eight in flight
    const int el = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int e = ((int)blockIdx.x - TAIL_NBIG) * 16 + el;
    const int per = (a.nrec + 15) / 16, r0 = q * per, r1 = r0 + per < a.nrec ? r0 + per : a.nrec;
    float sv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int r = r0;
    for (; r + 8 <= r1; r += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = a.records[(size_t)(r + u) * BG_REC + e];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) sv[u] += v[u];
    }
    for (; r < r1; ++r) sv[0] += a.records[(size_t)r * BG_REC + e];
    part[q][el] = ((sv[0] + sv[1]) + (sv[2] + sv[3])) + ((sv[4] + sv[5]) + (sv[6] + sv[7]));
    __syncthreads();
    if (q == 0) {
      i = rec_param(e);
      float t = 0.0f;
#pragma unroll
      for (int u = 0; u < 16; ++u) t += part[u][el];
      g = t;
    }
  }
  if (i >= 0) {
    g *= a.inv_gscale;
    a.grad[i] = g;
    float p = a.theta[i] * (1.0f - a.lr * a.wd);
    const float mi = a.m[i] + (g - a.m[i]) * (1.0f - a.b1);
    const float vi = a.v[i] * a.b2 + (1.0f - a.b2) * g * g;
    p -= step_size * (mi / (sqrtf(vi) * inv_bc2 + a.eps));
    a.theta[i] = p; a.m[i] = mi; a.v[i] = vi;
    if (a.packed) refresh_param(a.packed, i, p);
  }
  // the step's loss values out of the render kernel's per-block partials (cnr_render_loss_finish's job, one block of this launch)
  if (a.rl_partials && blockIdx.x == TAIL_NBIG + TAIL_NREC - 1 && threadIdx.x < 64)
    cnr_rl::finish_class(a.rl_partials, a.rl_nb, a.losses, a.flags, 1, 0, threadIdx.x);
}
__global__ void bg_advance_kernel(int64_t* d_state, int64_t add_rows) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { d_state[0] += add_rows; d_state[1] += 1; d_state[2] += 1; }
}
}  // namespace

// ========================================================================================================================
// C-ABI
// ========================================================================================================================
extern "C" int64_t cnr_bg_pack_bytes(void) { return (int64_t)BG_PACK_BYTES; }
extern "C" int cnr_bg_param_count(void) { return BG_NPARAM; }
extern "C" int cnr_bg_blocks(int M) { return M > 0 ? (M + TS - 1) / TS : 0; }
extern "C" int cnr_bg_dw_chunks(int M, int chunk) { return (M > 0 && chunk > 0) ? (M + chunk - 1) / chunk : 0; }
extern "C" int cnr_bg_record_floats(void) { return BG_REC; }
extern "C" int64_t cnr_bg_backward_render_workspace_bytes(int M) {
  return M > 0 ? ((int64_t)cnr_bg_blocks(M) * 3 + 4) * (int64_t)sizeof(float) : 0;
}

extern "C" int cnr_bg_pack(const float* theta, void* packed, void* stream) {
  if (!theta || !packed) return CNR_E_ARG;
  if (((uintptr_t)packed & 15) != 0) return CNR_E_ALIGN;
  hipLaunchKernelGGL(bg_pack_kernel, dim3(NF_FWD + NF_LO + NF_BWD), dim3(256), 0, (hipStream_t)stream, theta, (unsigned char*)packed);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_bg_forward(const float* pts, const float* theta, const void* packed, float scale, int M, float* sigma,
                              float* rgb, void* act, void* eimg, void* stream) {
  if (!pts || !theta || !packed || !sigma || !rgb || M <= 0 || !(scale > 0.f)) return CNR_E_ARG;
  if ((act == nullptr) != (eimg == nullptr)) return CNR_E_ARG;    // both (a training step's forward) or neither (forward only)
  if (((uintptr_t)packed & 15) != 0 || ((uintptr_t)act & 15) != 0 || ((uintptr_t)eimg & 15) != 0) return CNR_E_ALIGN;
  static cnr::DeviceOnce once;
  const int er = cnr::set_max_dynamic_lds(once, (const void*)bg_fwd_kernel, BG_FWD_LDS);
  if (er) return er;
  BgFwdArgs a{pts, theta, (const unsigned char*)packed, 1.0f / scale, M, sigma, rgb, (_Float16*)act, (_Float16*)eimg};
  hipLaunchKernelGGL(bg_fwd_kernel, dim3((unsigned)cnr_bg_blocks(M)), dim3(256), BG_FWD_LDS, (hipStream_t)stream, a);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_bg_backward(const float* pts, const float* theta, const void* packed, float scale, int M,
                               const float* d_sigma, const float* d_rgb, const float* rgb, const void* act, void* dpre,
                               float* records, int64_t* d_state, int64_t add_rows, void* stream) {
  if (!pts || !theta || !packed || !d_sigma || !d_rgb || !rgb || !act || !dpre || !records || M <= 0 || !(scale > 0.f))
    return CNR_E_ARG;
  if (((uintptr_t)packed & 15) != 0 || ((uintptr_t)act & 15) != 0 || ((uintptr_t)dpre & 15) != 0) return CNR_E_ALIGN;
  static cnr::DeviceOnce once;
  const int er = cnr::set_max_dynamic_lds(once, (const void*)bg_bwd_kernel<false>, BG_BWD_LDS);
  if (er) return er;
  BgBwdArgs a{};
  a.pts = pts; a.theta = theta; a.packed = (const unsigned char*)packed; a.inv_scale = 1.0f / scale; a.M = M;
  a.d_sigma = d_sigma; a.d_rgb = d_rgb; a.rgb = rgb; a.act = (const _Float16*)act; a.dpre = (_Float16*)dpre;
  a.records = records; a.d_state = d_state; a.add_rows = add_rows;
  hipLaunchKernelGGL(bg_bwd_kernel<false>, dim3((unsigned)cnr_bg_blocks(M)), dim3(256), BG_BWD_LDS, (hipStream_t)stream, a);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_bg_backward_render(const cnr_bg_backward_render_args* p, void* stream) {
  if (!p || p->struct_size != sizeof(cnr_bg_backward_render_args) || p->abi_version != CNR_ABI_VERSION) return CNR_E_ARG;
  if (!p->pts || !p->theta || !p->packed || !p->sigma || !p->rgb || !p->z || !p->gt_depth || !p->gt_rgb || !p->labels ||
      !p->depth_mask || !p->counts_tab || !p->act || !p->dpre || !p->records || !p->loss_workspace || p->R <= 0 ||
      p->S <= 0 || !(p->scale > 0.f))
    return CNR_E_ARG;
  if (p->S > 64 || (int64_t)p->R * p->S > 0x7fffffff) return CNR_E_SHAPE;
  const int M = p->R * p->S;
  if (p->loss_workspace_bytes < cnr_bg_backward_render_workspace_bytes(M)) return CNR_E_ARG;
  if (((uintptr_t)p->packed & 15) != 0 || ((uintptr_t)p->act & 15) != 0 || ((uintptr_t)p->dpre & 15) != 0) return CNR_E_ALIGN;
  static cnr::DeviceOnce once;
  const int er = cnr::set_max_dynamic_lds(once, (const void*)bg_bwd_kernel<true>, BG_BWD_LDS);
  if (er) return er;
  BgBwdArgs a{};
  a.pts = p->pts; a.theta = p->theta; a.packed = (const unsigned char*)p->packed; a.inv_scale = 1.0f / p->scale; a.M = M;
  a.rgb = p->rgb; a.act = (const _Float16*)p->act; a.dpre = (_Float16*)p->dpre; a.records = p->records;
  a.sigma = p->sigma; a.z = p->z; a.gt_depth = p->gt_depth; a.gt_rgb = p->gt_rgb; a.labels = p->labels;
  a.depth_mask = p->depth_mask; a.counts_tab = p->counts_tab; a.cursor = p->d_state; a.R = p->R; a.S = p->S;
  a.color_scaling = p->color_scaling; a.opacity_scaling = p->opacity_scaling; a.grad_scale = p->grad_scale;
  a.depth_out = p->depth; a.var_out = p->var; a.rgb_out = p->rgb_render; a.opacity_out = p->opacity;
  a.rl_partials = (float*)p->loss_workspace; a.d_sigma_out = p->d_sigma; a.d_rgb_out = p->d_rgb;
  hipLaunchKernelGGL(bg_bwd_kernel<true>, dim3((unsigned)cnr_bg_blocks(M)), dim3(256), BG_BWD_LDS, (hipStream_t)stream, a);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_bg_dw(const void* act, const void* dpre, const void* eimg, int M, int chunk, float* partials,
                         int64_t* d_state, int64_t add_rows, void* stream) {
  if (!act || !dpre || !eimg || !partials || M <= 0 || chunk <= 0 || (chunk & 63) != 0) return CNR_E_ARG;
  if (((uintptr_t)act & 15) != 0 || ((uintptr_t)dpre & 15) != 0 || ((uintptr_t)eimg & 15) != 0) return CNR_E_ALIGN;
  static cnr::DeviceOnce once;
  const int er = cnr::set_max_dynamic_lds(once, (const void*)bg_dw_kernel, BG_DW_LDS);
  if (er) return er;
  BgDwArgs a{(const _Float16*)act, (const _Float16*)dpre, (const _Float16*)eimg, M, chunk, partials, d_state, add_rows};
  hipLaunchKernelGGL(bg_dw_kernel, dim3(5, (unsigned)cnr_bg_dw_chunks(M, chunk)), dim3(256), BG_DW_LDS, (hipStream_t)stream, a);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

static int bg_tail_launch(float* theta, float* grad, float* exp_avg, float* exp_avg_sq, const float* partials, int chunks,
                          const float* records, int nrec, float grad_scale, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int64_t* d_state, int64_t add_rows, void* packed, const void* rl_workspace, int R,
                          float* losses, int32_t* flags, const cnr_sample::SampleArgs* sa, void* stream) {
  if (rl_workspace && (!losses || !flags || R < 0)) return CNR_E_ARG;
  if (((uintptr_t)packed & 15) != 0) return CNR_E_ALIGN;
  if (!theta || !grad || !exp_avg || !exp_avg_sq || !partials || !records || chunks <= 0 || nrec <= 0 || !d_state ||
      !(grad_scale > 0.f) || !(lr > 0.f) || BG_REC % 64 != 0)
    return CNR_E_ARG;
  BgTailArgs a{};
  a.theta = theta; a.grad = grad; a.m = exp_avg; a.v = exp_avg_sq; a.partials = partials; a.chunks = chunks; a.records = records;
  a.nrec = nrec; a.inv_gscale = 1.0f / grad_scale; a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay;
  a.d_state = d_state; a.add_rows = add_rows; a.packed = (unsigned char*)packed; a.rl_partials = (const float*)rl_workspace;
  a.losses = losses; a.flags = flags;
  // R > 0: cnr_render_loss's workspace for R rays; R = 0: cnr_bg_backward_render's (one partial per backward block = nrec)
  if (rl_workspace) { const int rpb = cnr_rl::rl_rays_per_block(1, R > 0 ? R : 1); a.rl_nb = R > 0 ? (R + rpb - 1) / rpb : nrec; }
  if (sa) { a.sa = *sa; a.sample_blocks = (sa->R + 3) / 4; }
  // (the step count is read by every block: the state moves in a second, one-thread launch behind them -- or, add_rows < 0,
  //  it was moved by cnr_bg_backward / cnr_bg_dw already, the launch between the sampler, which reads the cursor, and this one)
  hipLaunchKernelGGL(bg_tail_kernel, dim3(TAIL_NBIG + TAIL_NREC + a.sample_blocks), dim3(256), 0, (hipStream_t)stream, a);
  if (add_rows >= 0) hipLaunchKernelGGL(bg_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_state, add_rows);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_bg_tail(float* theta, float* grad, float* exp_avg, float* exp_avg_sq, const float* partials, int chunks,
                           const float* records, int nrec, float grad_scale, float lr, float beta1, float beta2, float eps,
                           float weight_decay, int64_t* d_state, int64_t add_rows, void* packed, const void* rl_workspace,
                           int R, float* losses, int32_t* flags, void* stream) {
  return bg_tail_launch(theta, grad, exp_avg, exp_avg_sq, partials, chunks, records, nrec, grad_scale, lr, beta1, beta2, eps,
                        weight_decay, d_state, add_rows, packed, rl_workspace, R, losses, flags, nullptr, stream);
}

extern "C" int cnr_bg_tail_sample(const cnr_bg_tail_sample_args* p, void* stream) {
  if (!p || p->struct_size != sizeof(cnr_bg_tail_sample_args) || p->abi_version != CNR_ABI_VERSION) return CNR_E_ARG;
  if (!p->rgbs || !p->depth || !p->dirs_c || !p->T || !p->d_state || !p->z || !p->pts || !p->gt_rgb || !p->depth_mask ||
      !p->labels || p->R <= 0 || p->n1 < 0 || p->n2 <= 0 || p->pool_rows < p->R)
    return CNR_E_ARG;
  if (p->n2 > 128) return CNR_E_SHAPE;
  if (p->max_bound_slices < 0 || (p->max_bound_slices > 1 && !p->max_bound)) return CNR_E_ARG;
  const cnr_sample::SampleArgs sa{p->rgbs, p->depth, p->dirs_c, p->T, nullptr, nullptr, p->seed, p->offset, p->d_state,
                                  p->pool_rows, p->max_bound, p->world_frame, 1, p->R, p->n1, p->n2, p->eps, p->stop_eps,
                                  p->min_bound, p->z, p->pts, p->origins, p->dirs_o, p->gt_rgb, p->gt_depth, p->depth_mask,
                                  p->labels, nullptr, 0, nullptr, p->perm, p->max_bound_slices, 0, 0, 0, 0};
  // the sampler blocks read the NEXT step's cursor: the state must have moved earlier in the step (add_rows = -1 semantics)
  return bg_tail_launch(p->theta, p->grad, p->exp_avg, p->exp_avg_sq, p->partials, p->chunks, p->records, p->nrec, p->grad_scale,
                        p->lr, p->beta1, p->beta2, p->adam_eps, p->weight_decay, (int64_t*)p->d_state, -1, p->packed,
                        p->rl_workspace, p->rl_R, p->losses, p->flags, &sa, stream);
}
