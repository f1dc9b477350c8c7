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
//                           come back transposed with ds_read_b64_tr_b16 as MFMA operands.
// The dW blocks of 32x32 stay in accumulator registers for the whole kernel.  All 16 of them (256 AGPRs) plus
// the chain's working set do not fit one wave's 512 registers (hipcc spills ~200 VGPRs to scratch, and a
// scratch-using dispatch is slow), so the backward runs as TWO launches of this kernel that split the blocks:
//   PART 0 "geometry": encoding_shape, shape_layer_2, cat_layer, shape_layer_1, encoding_xyz (10 blocks) + sigma head
//   PART 1 "texture" : rgb.2, rgb.0, texture_layer_1, encoding_viewdir (6 blocks)
// Both recompute the forward; PART 0 also walks the (cheap) texture data-gradient chain to get d y4.  Bias gradients
// ride along: free "ones" slots of the PE images give db of encoding_xyz / viewdir / rgb.2; the other layers
// use (one-hot(ray slot))^T x dPre, which also yields the per-row sums for dbiasrows.
//
// Nothing here uses an LDS float atomic: on gfx950 ds_add_f32 costs ~190 cycles per wave-instruction and
// serialises across the CU (~770 cycles with four waves issuing, tools/exp/ldsatomic.hip) -- 20x a plain
// read-add-write.  Per-row bias sums therefore go to WAVE-PRIVATE LDS tables with plain read-modify-write and
// the final reduction over the four waves is a two-round plain-store tree.  To leave the VGPR file to the chain,
// the longest-lived activations (a0, a1, a2 and the PE features) are parked in per-wave LDS images already
// during the forward sweep; those images double as the X operands of the weight-gradient products.
// One wave per SIMD (512-register budget), 4 waves per workgroup, no workgroup barrier inside the tile loop.
#include "fused_bwd_common.h"
namespace {
constexpr int WS_E1 = 0, WS_E2 = WS_E1 + E1IMG_BYTES, WS_X = WS_E2 + E2IMG_BYTES, WS_D = WS_X + HIMG_BYTES,
              WS_A0 = WS_D + HIMG_BYTES, WS_A1 = WS_A0 + HIMG_BYTES, WS_A2 = WS_A1 + HIMG_BYTES,
              WS_ROWTAB = WS_A2 + HIMG_BYTES,             // [ROWS_LDS][4][32] f32
              WS_PLAIN = WS_ROWTAB + ROWS_LDS * 128 * 4,  // db of encoding_shape, rgb.0: [2][32] f32
              WAVE_SCRATCH = WS_PLAIN + 256;
constexpr int LDS_BL = PK_BYTES;  // 66 floats: per-half direction rows
constexpr int LDS_SCRATCH = LDS_BL + 272;
constexpr int LDS_TOTAL = LDS_SCRATCH + 4 * WAVE_SCRATCH;
static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
static_assert(2 * NBLOCKS * 4096 <= LDS_TOTAL, "the two flush regions alias the whole allocation");


template <bool BIG_S, int PART>
__global__ __launch_bounds__(256, 1) void field_bwd_kernel(
    const float* __restrict__ pts, const float* __restrict__ Bdir, const unsigned char* __restrict__ packed,
    const float* __restrict__ biasrows, const int* __restrict__ ray_row, float inv_scale,
    const float* __restrict__ d_sigma, const float* __restrict__ d_rgb, float gscale,
    float* __restrict__ records, long long* __restrict__ rows_tab,
    int64_t N, int S, int R, int rows_per_class, int64_t B_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int c = blockIdx.y;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, h = lane >> 5, col = lane & 31;
  const bool lds_rows = rows_per_class > 0 && rows_per_class <= ROWS_LDS;
  unsigned char* ws = smem + LDS_SCRATCH + wv * WAVE_SCRATCH;
  {
    const unsigned char* src = packed + (size_t)c * PK_BYTES;
    for (int i = threadIdx.x * 16; i < PK_BYTES; i += 256 * 16)
      *reinterpret_cast<f4*>(smem + i) = *reinterpret_cast<const f4*>(src + i);
    float* Bl = reinterpret_cast<float*>(smem + LDS_BL);
    for (int i = threadIdx.x; i < 66; i += 256) {
      const int hh = i / 33, k = i % 33, d = k / 3;
      Bl[i] = (hh == 1 && d == 10) ? 0.0f : Bdir[(size_t)c * B_stride + (11 * hh + d) * 3 + (k % 3)];
    }
    float* rt = reinterpret_cast<float*>(ws + WS_ROWTAB);
    for (int i = lane; i < ROWS_LDS * 128 + 64; i += 64) rt[i] = 0.0f;
  }
  __syncthreads();
  const float* cf = reinterpret_cast<const float*>(smem + PK_OFF_CONST);
  const unsigned char* bwf = smem + PK_OFF_BWD;
  unsigned char* E1img = ws + WS_E1;
  unsigned char* E2img = ws + WS_E2;
  unsigned char* Ximg = ws + WS_X;
  unsigned char* Dimg = ws + WS_D;
  unsigned char* A0img = ws + WS_A0;
  unsigned char* A1img = ws + WS_A1;
  unsigned char* A2img = ws + WS_A2;
  float* rowtab = reinterpret_cast<float*>(ws + WS_ROWTAB);
  float* plain = reinterpret_cast<float*>(ws + WS_PLAIN);
  const float* Bl_h = reinterpret_cast<const float*>(smem + LDS_BL) + 33 * h;

  // ---- persistent accumulators ---------------------------------------------------------------------
  f16v Wacc[NBLOCKS];
  Wacc[0] = zero16(); Wacc[1] = zero16(); Wacc[2] = zero16(); Wacc[3] = zero16();
  Wacc[4] = zero16(); Wacc[5] = zero16(); Wacc[6] = zero16(); Wacc[7] = zero16();
  Wacc[8] = zero16(); Wacc[9] = zero16(); Wacc[10] = zero16(); Wacc[11] = zero16();
  Wacc[12] = zero16(); Wacc[13] = zero16(); Wacc[14] = zero16(); Wacc[15] = zero16();
  static_assert(NBLOCKS == 16, "persistent blocks");
  float dBacc[33];
#pragma unroll
  for (int i = 0; i < 33; ++i) dBacc[i] = 0.0f;
  f16v dws = zero16();
  float dbs = 0.0f;
  constexpr bool GEO = PART == 0, TEX = PART == 1;
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
    const int64_t ray0 = n0 / S;           // first ray (within class) touched by this tile
    const int off = (int)(n0 - ray0 * S);  // position of sample 0 inside that ray
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
        const int sl = BIG_S ? ((off + k) >= S ? 1 : 0) : (((off + k) * slot_inv) >> 16);
        OH[s][j] = (sl == col) ? (_Float16)1 : (_Float16)0;
      }
    // ray slots present in this tile and the biasrows row of each (loaded once, up front).
    // slot s <-> accumulator row s: slots 0..3 = registers 0..3 of lane-half 0, 4..7 = registers 0..3 of half 1, ...
    const int cnt = (int)((N - n0) < 32 ? (N - n0) : 32);
    const int nslots = (off + cnt - 1) / S + 1;  // wave-uniform
    constexpr int NSLOTREG = BIG_S ? 2 : 16;     // S >= 32: slots 0, 1 only
    int slot_row[NSLOTREG];
#pragma unroll
    for (int reg = 0; reg < NSLOTREG; ++reg) {
      slot_row[reg] = -1;
      if (acc_row(reg, 0) < nslots) {
        const int sl = acc_row(reg, h);
        const int64_t rr = ray0 + sl;
        if (sl < nslots && rr < R) slot_row[reg] = ray_row ? ray_row[(int64_t)c * R + rr] : (int)((int64_t)c * R + rr);
      }
    }

    // =================================== forward recompute =========================================
    h8 E1f[6], E2f[3];
    {
      float Bh[33];
#pragma unroll
      for (int i = 0; i < 33; ++i) Bh[i] = Bl_h[i];
      pe_slots<true>(Bh, t0, t1, t2, h, E1f, E2f);
    }
    wave_lds_sync();
    {
      if (GEO) {
        unsigned char* b1 = E1img + col * ST_E1 + h * 96;
#pragma unroll
        for (int s = 0; s < 6; ++s) *reinterpret_cast<h8*>(b1 + 16 * s) = E1f[s];
      } else {
        unsigned char* b2 = E2img + col * ST_E2 + h * 48;
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<h8*>(b2 + 16 * s) = E2f[s];
      }
    }
    f16v acc = acc_init(cf + CF_B_XYZ, h);
#pragma unroll
    for (int s = 0; s < 6; ++s) acc = MFMA(lds_frag(smem, KK_XYZ + s, lane), E1f[s], acc);
    {
      const h8 a = pack8(acc, 0, true), b = pack8(acc, 1, true);  // a0
      if (GEO) stage_h(A0img, a, b, col, h);
      acc = acc_init(brow + 0 * 32, h);
      acc = MFMA(lds_frag(smem, KK_S1 + 0, lane), a, acc);
      acc = MFMA(lds_frag(smem, KK_S1 + 1, lane), b, acc);
    }
    {
      const h8 a = pack8(acc, 0, true), b = pack8(acc, 1, true);  // a1
      if (GEO) stage_h(A1img, a, b, col, h);
      acc = acc_init(brow + 1 * 32, h);
      acc = MFMA(lds_frag(smem, KK_CAT + 0, lane), a, acc);
      acc = MFMA(lds_frag(smem, KK_CAT + 1, lane), b, acc);
#pragma unroll
      for (int s = 0; s < 6; ++s) acc = MFMA(lds_frag(smem, KK_CAT + 2 + s, lane), E1f[s], acc);
    }
    {
      const h8 a = pack8(acc, 0, true), b = pack8(acc, 1, true);  // a2
      if (GEO) stage_h(A2img, a, b, col, h);
      acc = acc_init(brow + 2 * 32, h);
      acc = MFMA(lds_frag(smem, KK_S2 + 0, lane), a, acc);
      acc = MFMA(lds_frag(smem, KK_S2 + 1, lane), b, acc);
    }
    const h8 A3a = pack8(acc, 0, true), A3b = pack8(acc, 1, true);
    acc = acc_init(cf + CF_B_ES, h);
    acc = MFMA(lds_frag(smem, KK_ES + 0, lane), A3a, acc);
    acc = MFMA(lds_frag(smem, KK_ES + 1, lane), A3b, acc);
    // sigma head gradient (fp32 VALU): d w_sigma += draw * y4 ; d b_sigma += draw
    if (GEO) {
#pragma unroll
      for (int i = 0; i < 16; ++i) dws[i] = fmaf(draw, acc[i], dws[i]);
      dbs += (h == 0) ? draw : 0.0f;
    }
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
    auto dW_img = [&](f16v& W, const h8& trD0, const h8& trD1, const unsigned char* ximg, int stride, int col0) {
      W = MFMA(trD0, tr_frag(ximg, stride, col0, 0, lane), W);
      W = MFMA(trD1, tr_frag(ximg, stride, col0, 1, lane), W);
    };
    // per-row sums of the staged dPre tile: (one-hot^T) x dPre ; rows = tile slot, cols = out feature;
    // plain read-modify-write on the wave-private tables (or global atomics when rows are per ray)
    auto row_sums = [&](const h8& trD0, const h8& trD1, int latent_slot /* -1: plain bias */, int plain_idx) {
      f16v rs = zero16();
      rs = MFMA(OH[0], trD0, rs);
      rs = MFMA(OH[1], trD1, rs);
#pragma unroll
      for (int reg = 0; reg < NSLOTREG; ++reg) {
        if (acc_row(reg, 0) >= nslots) continue;  // wave-uniform
        const int sl = acc_row(reg, h);
        if (sl >= nslots) continue;
        const int grow = slot_row[reg];
        if (grow < 0) continue;
        if (latent_slot < 0) plain[plain_idx * 32 + col] += rs[reg];
        else if (lds_rows) rowtab[(grow - c * rows_per_class) * 128 + latent_slot * 32 + col] += rs[reg];
        else   // rows that do not fit the LDS tables (one per ray, or more than ROWS_LDS per class): 2^-40 fixed point, integer
               // atomics -- order-free, hence bitwise reproducible (a float atomicAdd here made this path repeatable to rounding only)
          atomicAdd(reinterpret_cast<unsigned long long*>(rows_tab + (int64_t)grow * 128 + latent_slot * 32 + col),
                    (unsigned long long)__double2ll_rn((double)(rs[reg] * inv_gs) * cnr_rec::ROWS_FIX_SCALE));
      }
    };
    // PE backward: p_d recomputed from LDS (cheaper than 11 registers held across the chain), dL/dp_d from
    // 16-slot blocks of d e, dB_d += (sum_b de cos(pi 2^b p_d) pi 2^b) t
    auto pe_backward = [&](const f16v (&de)[3], int nblk, int band0, int nq) {
      float pd[11], gpa[11];
#pragma unroll
      for (int d = 0; d < 11; ++d) {
        pd[d] = Bl_h[3 * d] * t0 + Bl_h[3 * d + 1] * t1 + Bl_h[3 * d + 2] * t2;
        gpa[d] = 0.0f;
      }
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        if (b >= nblk) continue;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int q = 16 * b + reg;
          if (q < nq) {
            const int band = band0 + q / 11, d = q % 11;
            const float cs = __builtin_amdgcn_cosf(pd[d] * (0.5f * (float)(1 << band)));
            gpa[d] = fmaf(de[b][reg] * cs, 3.14159265358979f * (float)(1 << band), gpa[d]);
          }
        }
      }
#pragma unroll
      for (int d = 0; d < 11; ++d) {
        dBacc[3 * d + 0] = fmaf(gpa[d], t0, dBacc[3 * d + 0]);
        dBacc[3 * d + 1] = fmaf(gpa[d], t1, dBacc[3 * d + 1]);
        dBacc[3 * d + 2] = fmaf(gpa[d], t2, dBacc[3 * d + 2]);
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
    if (TEX) {
      wave_lds_sync();
      stage_h(Dimg, D0, D1, col, h);
      h8 one = zero8();
      if (h == 0) one[0] = (_Float16)1;  // feature 16 of the a7 image := 1 -> d b(rgb.2)
      stage_h(Ximg, A7a, one, col, h);
      wave_lds_sync();
      const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
      dW_img(Wacc[BK_R2], tD0, tD1, Ximg, ST_H, 0);
    }
    acc = MFMA(lds_frag(bwf, KT_R2, lane), D0, zero16());  // d a7 (rows 0..15)
    mask8(acc, 0, A7a);
    D0 = pack8(acc, 0, false); D1 = zero8();
    // ---- rgb.0 ------------------------------------------------------------------------------------------
    if (TEX) {
      wave_lds_sync();
      stage_h(Dimg, D0, D1, col, h);
      stage_h(Ximg, A6a, A6b, col, h);
      wave_lds_sync();
      const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
      dW_img(Wacc[BK_R0], tD0, tD1, Ximg, ST_H, 0);
      row_sums(tD0, tD1, -1, 1);
    }
    acc = MFMA(lds_frag(bwf, KT_R0, lane), D0, zero16());  // d a6
    mask8(acc, 0, A6a); mask8(acc, 1, A6b);
    D0 = pack8(acc, 0, false); D1 = pack8(acc, 1, false);
    // ---- texture_layer_1 ----------------------------------------------------------------------------------
    if (TEX) {
      wave_lds_sync();
      stage_h(Dimg, D0, D1, col, h);
      stage_h(Ximg, A5a, A5b, col, h);
      wave_lds_sync();
      const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
      dW_img(Wacc[BK_T1], tD0, tD1, Ximg, ST_H, 0);
      row_sums(tD0, tD1, 3, 0);
    }
    acc = MFMA(lds_frag(bwf, KT_T1 + 0, lane), D0, zero16());
    acc = MFMA(lds_frag(bwf, KT_T1 + 1, lane), D1, acc);  // d a5
    mask8(acc, 0, A5a); mask8(acc, 1, A5b);
    D0 = pack8(acc, 0, false); D1 = pack8(acc, 1, false);
    // ---- encoding_viewdir : inputs [y4 | e2] -----------------------------------------------------------------
    if (TEX) {
      wave_lds_sync();
      stage_h(Dimg, D0, D1, col, h);
      stage_h(Ximg, Y4a, Y4b, col, h);
      wave_lds_sync();
      {
        const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
        dW_img(Wacc[BK_VD_Y], tD0, tD1, Ximg, ST_H, 0);
        dW_img(Wacc[BK_VD_E0], tD0, tD1, E2img, ST_E2, 0);
        dW_img(Wacc[BK_VD_E1], tD0, tD1, E2img, ST_E2, 32);
      }
      // d e2 (two 16-slot blocks) -> dB, bands 4 and 5
      f16v de[3];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        de[b] = MFMA(lds_frag(bwf, KT_VD_E + 2 * b + 0, lane), D0, zero16());
        de[b] = MFMA(lds_frag(bwf, KT_VD_E + 2 * b + 1, lane), D1, de[b]);
      }
      de[2] = de[1];
      pe_backward(de, 2, 4, 22);
      continue;  // the geometry branch belongs to PART 0
    }
    acc = MFMA(lds_frag(bwf, KT_VD_Y + 0, lane), D0, zero16());
    acc = MFMA(lds_frag(bwf, KT_VD_Y + 1, lane), D1, acc);  // d y4 from the colour branch
    {  // + sigma head: d y4 += w_sigma * draw
      const f16v wsg = acc_init(cf + CF_W_SG, h);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fmaf(wsg[i], draw, acc[i]);
    }
    D0 = pack8(acc, 0, false); D1 = pack8(acc, 1, false);
    // ---- encoding_shape (no activation) --------------------------------------------------------------------------
    wave_lds_sync();
    stage_h(Dimg, D0, D1, col, h);
    stage_h(Ximg, A3a, A3b, col, h);
    wave_lds_sync();
    {
      const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
      dW_img(Wacc[BK_ES], tD0, tD1, Ximg, ST_H, 0);
      row_sums(tD0, tD1, -1, 0);
    }
    acc = MFMA(lds_frag(bwf, KT_ES + 0, lane), D0, zero16());
    acc = MFMA(lds_frag(bwf, KT_ES + 1, lane), D1, acc);  // d a3
    mask8(acc, 0, A3a); mask8(acc, 1, A3b);
    D0 = pack8(acc, 0, false); D1 = pack8(acc, 1, false);
    // ---- shape_layer_2 : input a2 (image parked in the forward sweep) ---------------------------------------------------
    wave_lds_sync();
    stage_h(Dimg, D0, D1, col, h);
    wave_lds_sync();
    {
      const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
      dW_img(Wacc[BK_S2], tD0, tD1, A2img, ST_H, 0);
      row_sums(tD0, tD1, 2, 0);
    }
    acc = MFMA(lds_frag(bwf, KT_S2 + 0, lane), D0, zero16());
    acc = MFMA(lds_frag(bwf, KT_S2 + 1, lane), D1, acc);  // d a2
    {
      h8 a, b;
      load_h(A2img, a, b, col, h);
      mask8(acc, 0, a); mask8(acc, 1, b);
    }
    D0 = pack8(acc, 0, false); D1 = pack8(acc, 1, false);
    // ---- cat_layer : inputs [a1 | e1] ------------------------------------------------------------------------------------
    wave_lds_sync();
    stage_h(Dimg, D0, D1, col, h);
    wave_lds_sync();
    {
      const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
      dW_img(Wacc[BK_CAT_Y], tD0, tD1, A1img, ST_H, 0);
      row_sums(tD0, tD1, 1, 0);
      dW_img(Wacc[BK_CAT_E0], tD0, tD1, E1img, ST_E1, 0);
      dW_img(Wacc[BK_CAT_E1], tD0, tD1, E1img, ST_E1, 32);
      dW_img(Wacc[BK_CAT_E2], tD0, tD1, E1img, ST_E1, 64);
    }
    const h8 Dc0 = D0, Dc1 = D1;  // dPre of cat_layer: its d e1 part is formed together with encoding_xyz's below
    acc = MFMA(lds_frag(bwf, KT_CAT_Y + 0, lane), D0, zero16());
    acc = MFMA(lds_frag(bwf, KT_CAT_Y + 1, lane), D1, acc);  // d a1
    {
      h8 a, b;
      load_h(A1img, a, b, col, h);
      mask8(acc, 0, a); mask8(acc, 1, b);
    }
    D0 = pack8(acc, 0, false); D1 = pack8(acc, 1, false);
    // ---- shape_layer_1 : input a0 ---------------------------------------------------------------------------------------------
    wave_lds_sync();
    stage_h(Dimg, D0, D1, col, h);
    wave_lds_sync();
    {
      const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
      dW_img(Wacc[BK_S1], tD0, tD1, A0img, ST_H, 0);
      row_sums(tD0, tD1, 0, 0);
    }
    acc = MFMA(lds_frag(bwf, KT_S1 + 0, lane), D0, zero16());
    acc = MFMA(lds_frag(bwf, KT_S1 + 1, lane), D1, acc);  // d a0
    {
      h8 a, b;
      load_h(A0img, a, b, col, h);
      mask8(acc, 0, a); mask8(acc, 1, b);
    }
    D0 = pack8(acc, 0, false); D1 = pack8(acc, 1, false);
    // ---- encoding_xyz : input e1 ---------------------------------------------------------------------------------------------------
    wave_lds_sync();
    stage_h(Dimg, D0, D1, col, h);
    wave_lds_sync();
    {
      const h8 tD0 = tr_frag(Dimg, ST_H, 0, 0, lane), tD1 = tr_frag(Dimg, ST_H, 0, 1, lane);
      dW_img(Wacc[BK_XYZ_E0], tD0, tD1, E1img, ST_E1, 0);
      dW_img(Wacc[BK_XYZ_E1], tD0, tD1, E1img, ST_E1, 32);
      dW_img(Wacc[BK_XYZ_E2], tD0, tD1, E1img, ST_E1, 64);
    }
    {  // d e1 = Wc_e^T dPre(cat) + Wx^T dPre(xyz) (three 16-slot blocks) -> dB, bands 0..3
      f16v de[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        de[b] = MFMA(lds_frag(bwf, KT_CAT_E + 2 * b + 0, lane), Dc0, zero16());
        de[b] = MFMA(lds_frag(bwf, KT_CAT_E + 2 * b + 1, lane), Dc1, de[b]);
        de[b] = MFMA(lds_frag(bwf, KT_XYZ_E + 2 * b + 0, lane), D0, de[b]);
        de[b] = MFMA(lds_frag(bwf, KT_XYZ_E + 2 * b + 1, lane), D1, de[b]);
      }
      pe_backward(de, 3, 0, 44);
    }
  }

  // ========================================= flush ====================================================
  // No global float atomics on shared addresses either: hundreds of workgroups adding into the same 55 KB
  // serialise at the memory side (measured: +2.6 us per extra workgroup).  Every workgroup writes ONE record
  //   rec = [ trunk 13892 | dB geometry 63 | dB texture 63 | row sums ROWS_LDS x 128 ]  (floats, REC_FLOATS)
  // with plain stores (the two PART launches fill disjoint entries of the same record, every entry the reduction
  // reads is rewritten by every call) and reduce_records_kernel sums the records in a fixed order: the gradient is
  // bitwise reproducible run to run.
  float* rec = records + ((size_t)c * gridDim.x + blockIdx.x) * REC_FLOATS;
  // (1) per-wave small sums -> wave slots in LDS (plain), then one pass over the four waves
  wave_lds_sync();
  float* small = reinterpret_cast<float*>(ws + WS_X);  // [0..31] d w_sigma, [32] d b_sigma, [64..126] dB
  if (GEO) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float v = half_sum_dpp(dws[i]);
      if (col == 31) small[acc_row(i, h)] = v;
    }
    const float v = half_sum_dpp(dbs);   // dbs is zero in lane half 1
    if (lane == 31) small[32] = v;
  }
#pragma unroll
  for (int i = 0; i < 33; ++i) {
    const float v = half_sum_dpp(dBacc[i]);
    const int d = i / 3;
    if (col == 31 && !(h == 1 && d == 10)) small[64 + (11 * h + d) * 3 + (i % 3)] = v;
  }
  __syncthreads();
  {
    const unsigned char* w0 = smem + LDS_SCRATCH;
    auto sum4 = [&](int byte_off, int i) {
      float v = 0.0f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += reinterpret_cast<const float*>(w0 + w * WAVE_SCRATCH + byte_off)[i];
      return v;
    };
    for (int i = threadIdx.x; i < 63; i += 256) rec[TRUNK + (GEO ? 0 : 63) + i] = sum4(WS_X, 64 + i) * inv_gs;
    if (GEO) {
      for (int i = threadIdx.x; i < 32; i += 256) rec[OFF_SG_W + i] = sum4(WS_X, i) * inv_gs;
      if (threadIdx.x == 0) rec[OFF_SG_B] = sum4(WS_X, 32) * inv_gs;
      for (int i = threadIdx.x; i < 32; i += 256) rec[OFF_ES_B + i] = sum4(WS_PLAIN, i) * inv_gs;
    } else {
      for (int i = threadIdx.x; i < 16; i += 256) rec[OFF_R0_B + i] = sum4(WS_PLAIN, 32 + i) * inv_gs;
    }
    if (lds_rows) {
      for (int i = threadIdx.x; i < rows_per_class * 128; i += 256) {
        const int slot = (i >> 5) & 3;
        if ((slot == 3) == TEX) rec[TRUNK + 126 + i] = sum4(WS_ROWTAB, i) * inv_gs;
      }
    }
  }
  // (2) the accumulator blocks: waves 0,1 store into two LDS regions, waves 2,3 add theirs (plain RMW), then
  //     every thread sums the two regions and stores the valid entries of the record
  __syncthreads();  // every wave is done with the packed image and its scratch
  {
    float* region = reinterpret_cast<float*>(smem) + (wv & 1) * (NBLOCKS * 1024);
#define CNR_STORE(KIND, OP)                            \
  _Pragma("unroll") for (int reg = 0; reg < 16; ++reg) \
      region[(KIND) * 1024 + acc_row(reg, h) * 32 + col] OP Wacc[KIND][reg];
#define CNR_ALL(OP)                                                                                 \
  if (TEX) {                                                                                        \
    CNR_STORE(BK_R2, OP) CNR_STORE(BK_R0, OP) CNR_STORE(BK_T1, OP) CNR_STORE(BK_VD_Y, OP)           \
    CNR_STORE(BK_VD_E0, OP) CNR_STORE(BK_VD_E1, OP)                                                 \
  } else {                                                                                          \
    CNR_STORE(BK_ES, OP) CNR_STORE(BK_S2, OP) CNR_STORE(BK_CAT_Y, OP) CNR_STORE(BK_CAT_E0, OP)      \
    CNR_STORE(BK_CAT_E1, OP) CNR_STORE(BK_CAT_E2, OP) CNR_STORE(BK_S1, OP) CNR_STORE(BK_XYZ_E0, OP) \
    CNR_STORE(BK_XYZ_E1, OP) CNR_STORE(BK_XYZ_E2, OP)                                               \
  }
    if (wv < 2) { CNR_ALL(=) }
    __syncthreads();
    if (wv >= 2) { CNR_ALL(+=) }
    __syncthreads();
#undef CNR_ALL
#undef CNR_STORE
  }
  {
    const float* ra = reinterpret_cast<const float*>(smem);
    const float* rb = ra + NBLOCKS * 1024;
    for (int j = threadIdx.x; j < TRUNK; j += 256) {   // coalesced record stores, LDS gather
      const int src = g_param_src[j];
      if (src >= 0 && ((src >> 10) <= BK_VD_E1) == TEX) rec[j] = (ra[src] + rb[src]) * inv_gs;
    }
  }
}
}  // namespace

extern "C" int64_t cnr_field_bwd_workspace_bytes(int C, int max_blocks) {
  const int64_t cap = max_blocks > 0 ? max_blocks : 256;
  return (int64_t)C * cap * REC_FLOATS * (int64_t)sizeof(float);
}
// the fixed-point table of the per-row bias sums, behind the records, when the rows do not fit the kernels' LDS tables
extern "C" int64_t cnr_field_bwd_rows_table_bytes(int64_t total_rows) {
  return total_rows > 0 ? total_rows * 128 * (int64_t)sizeof(long long) : 0;
}
namespace {
__global__ __launch_bounds__(256) void rows_table_to_float_kernel(const long long* __restrict__ tab, float* __restrict__ dbiasrows,
                                                                  int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    dbiasrows[i] += (float)((double)tab[i] * (1.0 / cnr_rec::ROWS_FIX_SCALE));
}
}  // namespace

extern "C" int cnr_field_bwd(const float* pts, const float* B, const void* packed, const float* biasrows,
                             const int* ray_row, float scale, const float* d_sigma, const float* d_rgb,
                             float grad_scale, float* dtrunk, float* dB, float* dbiasrows, int C, int R, int S,
                             int rows_per_class, int max_blocks, void* workspace, int64_t workspace_bytes,
                             int64_t B_stride, int64_t dtrunk_stride, int64_t dB_stride, void* stream) {
  if (!pts || !B || !packed || !biasrows || !d_sigma || !d_rgb || !dtrunk || !dB || !dbiasrows || !workspace)
    return CNR_E_ARG;
  if (C <= 0 || R <= 0 || S <= 0 || !(scale > 0.f) || !(grad_scale > 0.f)) return CNR_E_ARG;
  if (S > 240) return CNR_E_SHAPE;
  if (((uintptr_t)packed & 15) != 0 || ((uintptr_t)biasrows & 15) != 0 || ((uintptr_t)workspace & 15) != 0)
    return CNR_E_ALIGN;
  if (ray_row == nullptr && rows_per_class > 0 && rows_per_class != R) return CNR_E_ARG;
  const int64_t N = (int64_t)R * S;
  const int64_t ntiles = (N + 31) / 32;
  int64_t blocks = (ntiles + 3) / 4;
  const int64_t cap = max_blocks > 0 ? max_blocks : 256;
  if (blocks > cap) blocks = cap;
  const int64_t need = (int64_t)C * blocks * REC_FLOATS * (int64_t)sizeof(float);
  // rows beyond the LDS tables (one per ray, or more than ROWS_LDS per class) are summed in a 2^-40 fixed-point table behind
  // the records (cnr_field_bwd_rows_table_bytes): cleared, filled with integer atomics, added to dbiasrows -- all on `stream`
  const bool rows_lds = rows_per_class > 0 && rows_per_class <= ROWS_LDS;
  const int64_t total_rows = rows_lds ? 0 : (ray_row ? (int64_t)C * rows_per_class : (int64_t)C * R);
  const int64_t tab_bytes = cnr_field_bwd_rows_table_bytes(total_rows);
  if (!rows_lds && ray_row && rows_per_class <= 0) return CNR_E_ARG;
  if (workspace_bytes < need + tab_bytes) return CNR_E_ARG;
  long long* rows_tab = tab_bytes ? reinterpret_cast<long long*>(reinterpret_cast<unsigned char*>(workspace) + need) : nullptr;
  if (rows_tab) {
    const hipError_t e = hipMemsetAsync(rows_tab, 0, (size_t)tab_bytes, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  {  // once per device: the LDS attribute of the four instantiations and the constant table (static device storage)
    static cnr::DeviceOnce once[4], table;
    const void* fns[4] = {(const void*)field_bwd_kernel<true, 0>, (const void*)field_bwd_kernel<true, 1>,
                          (const void*)field_bwd_kernel<false, 0>, (const void*)field_bwd_kernel<false, 1>};
    for (int i = 0; i < 4; ++i) {
      const int er = cnr::set_max_dynamic_lds(once[i], fns[i], LDS_TOTAL);
      if (er) return er;
    }
    if (cnr::first_on_device(table)) {   // stream-ordered before the first use
      hipLaunchKernelGGL(build_param_src_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream);
      hipLaunchKernelGGL(fill_param_src_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream);
    }
  }
  dim3 grid((unsigned)blocks, (unsigned)C);
#define CNR_LAUNCH_BWD(BIG, PART)                                                                               \
  hipLaunchKernelGGL((field_bwd_kernel<BIG, PART>), grid, dim3(256), LDS_TOTAL, (hipStream_t)stream, pts, B,    \
                     (const unsigned char*)packed, biasrows, ray_row, 1.0f / scale, d_sigma, d_rgb, grad_scale, \
                     (float*)workspace, rows_tab, N, S, R, rows_per_class, B_stride > 0 ? B_stride : (int64_t)63)
  if (S >= 32) { CNR_LAUNCH_BWD(true, 1); CNR_LAUNCH_BWD(true, 0); }
  else { CNR_LAUNCH_BWD(false, 1); CNR_LAUNCH_BWD(false, 0); }
#undef CNR_LAUNCH_BWD
  CNR_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_records_kernel, dim3(REC_FLOATS / 64, (unsigned)C), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, (int)blocks, dtrunk, dB, dbiasrows,
                     // rows travel in the records only when these kernels keep them in LDS (<= ROWS_LDS per class);
                     // otherwise they went to dbiasrows by atomics and the records hold none
                     (rows_per_class >= 1 && rows_per_class <= ROWS_LDS) ? rows_per_class : 0,
                     dtrunk_stride > 0 ? dtrunk_stride : (int64_t)TRUNK, dB_stride > 0 ? dB_stride : (int64_t)63);
  CNR_LAUNCH_CHECK();
  if (rows_tab) {
    const int64_t n = total_rows * 128;
    hipLaunchKernelGGL(rows_table_to_float_kernel, dim3((unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const long long*)rows_tab, dbiasrows, n);
    CNR_LAUNCH_CHECK();
  }
  return CNR_OK;
}
