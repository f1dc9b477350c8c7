// Fused field backward, pipelined variant: ONE launch, workgroup = NCH "chain" waves + NDW "dW" waves (4 total,
// one per SIMD).
//   chain wave : owns one 32-sample tile per iteration: forward recompute, data-gradient chain, sigma head, PE backward;
//                after each layer it drops that layer's dPre image (and, when it is not already parked in LDS, the
//                layer-input image) into one of two LDS slots and hits the workgroup barrier.
//   dW wave    : owns the dW accumulator blocks (all 16 for NDW = 1, 8 each for NDW = 2) for the whole kernel; one
//                barrier behind the chain waves it reads each chain wave's slot back transposed
//                (ds_read_b64_tr_b16) and runs the weight-gradient and row-sum MFMAs.  It has no VALU work to speak
//                of, so its MFMAs run on their own SIMD concurrently with the chain waves' VALU-heavy code.
// Compared with fused_bwd.hip's two-launch block split this recomputes the forward and the texture chain once
// instead of twice and takes the 44 dW / row-sum MFMAs per tile off the chain waves' critical path.
// Synchronisation is s_barrier only (10 per iteration, both roles run the same trip count): no flags, no polling.
// The two roles are separate loops (the register allocator then sees max(roles), not their sum).
#include "fused_bwd_common.h"

namespace {

constexpr int SLOT_BYTES = 2 * HIMG_BYTES;  // dPre image, then the layer-input image
constexpr int CW_E1 = 0, CW_E2 = CW_E1 + E1IMG_BYTES, CW_A0 = CW_E2 + E2IMG_BYTES, CW_A1 = CW_A0 + HIMG_BYTES,
              CW_A2 = CW_A1 + HIMG_BYTES, CW_SLOT = CW_A2 + HIMG_BYTES, CW_SMALL = CW_SLOT + 2 * SLOT_BYTES,
              CW_BYTES = CW_SMALL + 512;
constexpr int P_LDS_BL = PK_BYTES;
constexpr int P_LDS_BR = P_LDS_BL + 272;                       // biasrows of this class (ROWS_LDS x 128 floats)
constexpr int P_LDS_CHAIN = P_LDS_BR + ROWS_LDS * 128 * 4;
constexpr int RS_REGION = NBLOCKS;  // flush image: 16 dW blocks + the row-sum block
template <int NCH> constexpr int p_lds_total() {
  return P_LDS_CHAIN + NCH * CW_BYTES > (RS_REGION + 1) * 4096 ? P_LDS_CHAIN + NCH * CW_BYTES : (RS_REGION + 1) * 4096;
}
static_assert(p_lds_total<3>() <= 160 * 1024 && p_lds_total<2>() <= 160 * 1024, "LDS budget");

// backward layer steps, in order
#ifdef CNR_PIPE_STAMPS  // tools/exp only: cycle stamps of one chain wave and one dW wave of workgroup 0
__device__ long long g_pipe_stamps[160];
#define PSTAMP() do { if (blockIdx.x == 0 && lane == 0) \
    g_pipe_stamps[wv * 32 + (pstamp_i++)] = (long long)__builtin_readcyclecounter(); } while (0)
#define PSTAMP_RESET() int pstamp_i = 0
#define FSTAMP(k) do { if (blockIdx.x == 0 && lane == 0 && wv == 0) \
    g_pipe_stamps[128 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define FSTAMP(k) do {} while (0)
#define PSTAMP() do {} while (0)
#define PSTAMP_RESET() do {} while (0)
#endif
#define PSYNC() do { PSTAMP(); role_barrier(); PSTAMP(); } while (0)

// (Tried: the chain role's MFMAs as inline asm in VGPR form -- a 512-register kernel gets AGPR-form MFMAs, which costs
//  the chain role 16 v_accvgpr_read per layer.  With each layer's chain as one asm statement and hand-written wait
//  states it passed the parity tests and gained 2 % (288.8 -> 282.8 us at 8192 x 128): not worth carrying hazard
//  rules the compiler cannot check.  The chain role is bound by per-step dependent latency -- MFMA result -> mask /
//  pack -> image store -> barrier, ~700 cycles for ~40 instructions -- not by instruction count.)

template <int NCH>
__global__ __launch_bounds__(256, 1) void field_bwd_pipe_kernel(
    const float* __restrict__ pts, const float* __restrict__ Bdir, const unsigned char* __restrict__ packed,
    const float* __restrict__ biasrows, const int* __restrict__ ray_row, float inv_scale,
    const float* __restrict__ d_sigma, const float* __restrict__ d_rgb, float gscale,
    float* __restrict__ records, float* __restrict__ dbiasrows, int N, int S, int R, int rows_per_class,
    int64_t B_stride, long long* __restrict__ rows_fix) {
  constexpr int NDW = 4 - NCH;
  constexpr int NW = NBLOCKS / NDW;  // accumulator blocks per dW wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int c = blockIdx.y;
  const int lane = threadIdx.x & 63, h = lane >> 5, col = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool is_chain = wv < NCH;
  const int dwid = wv - NCH;
  {
    const unsigned char* src = packed + (size_t)c * PK_BYTES;
    for (int i = threadIdx.x * 16; i < PK_BYTES; i += 256 * 16)
      *reinterpret_cast<f4*>(smem + i) = *reinterpret_cast<const f4*>(src + i);
    float* Bl = reinterpret_cast<float*>(smem + P_LDS_BL);
    for (int i = threadIdx.x; i < 66; i += 256) {
      const int hh = i / 33, k = i % 33, d = k / 3;
      Bl[i] = (hh == 1 && d == 10) ? 0.0f : Bdir[(size_t)c * B_stride + (11 * hh + d) * 3 + (k % 3)];
    }
    float* br = reinterpret_cast<float*>(smem + P_LDS_BR);  // host guarantees 1 <= rows_per_class <= ROWS_LDS
    for (int i = threadIdx.x; i < rows_per_class * 128; i += 256) br[i] = biasrows[(size_t)c * rows_per_class * 128 + i];
  }
  __syncthreads();
  const float* cf = reinterpret_cast<const float*>(smem + PK_OFF_CONST);
  const unsigned char* bwf = smem + PK_OFF_BWD;
  unsigned char* chain_base = smem + P_LDS_CHAIN;
  const float inv_gs = 1.0f / gscale;
  const int slot_inv = (65536 + S - 1) / S;
  const int ntiles = (N + 31) / 32;
  const int tile_step = gridDim.x * NCH;

  // A chain wave never touches accumulator blocks, so its persistent per-lane partial sums live in the same
  // registers: dB (33 floats) in Wacc[0..2], d b_sigma in Wacc[2][15], d w_sigma in Wacc[3].
  f16v Wacc[NW];
#pragma unroll
  for (int b = 0; b < NW; ++b) Wacc[b] = zero16();
#define DBACC(i) Wacc[(i) >> 4][(i) & 15]
#define DWS(i) Wacc[3][i]
#define DBS Wacc[2][15]

  if (is_chain) {
    // ===================================================================================================
    // chain role
    // ===================================================================================================
    unsigned char* cw = chain_base + wv * CW_BYTES;
    unsigned char* E1img = cw + CW_E1;
    unsigned char* E2img = cw + CW_E2;
    unsigned char* A0img = cw + CW_A0;
    unsigned char* A1img = cw + CW_A1;
    unsigned char* A2img = cw + CW_A2;
    // [5][32 samples] f16: rows 0..3 = (object row of sample k == r), row 4 = ones: the dW wave's row-sum operand
    _Float16* rowoh = reinterpret_cast<_Float16*>(cw + CW_SMALL);
    const float* Bl_h = reinterpret_cast<const float*>(smem + P_LDS_BL) + 33 * h;
    auto slot_D = [&](int step) { return cw + CW_SLOT + (step & 1) * SLOT_BYTES; };

    // this lane's sample of a tile: loaded one iteration ahead (during the drain barrier of the previous tile)
    float in_p0, in_p1, in_p2, in_ds, in_r0, in_r1, in_r2;
    int in_row;
    auto load_inputs = [&](int tile) {
      const int n0 = tile * 32;
      const int ray0 = (int)((unsigned)n0 / (unsigned)S);  // wave-uniform
      const int off = n0 - ray0 * S;
      const int k = off + col;
      const int sl = (k * slot_inv) >> 16;  // k / S for k < S + 32, S <= 240
      const bool live = n0 + col < N;
      const int nc = live ? n0 + col : N - 1;
      const int rayc = live ? ray0 + sl : R - 1;
      const int64_t gs = (int64_t)c * N + nc;
      const float* pp = pts + gs * 3;
      in_p0 = pp[0]; in_p1 = pp[1]; in_p2 = pp[2];
      const int64_t ray = (int64_t)c * R + rayc;
      in_row = ray_row ? ray_row[ray] : (int)ray;
      in_ds = live ? d_sigma[gs] : 0.0f;
      in_r0 = live ? d_rgb[gs * 3 + 0] : 0.0f;
      in_r1 = live ? d_rgb[gs * 3 + 1] : 0.0f;
      in_r2 = live ? d_rgb[gs * 3 + 2] : 0.0f;
    };
    auto clamp_tile = [&](int t) { return t < ntiles ? t : ntiles - 1; };  // past the end: a dead tile (all zero)
    int tile = blockIdx.x * NCH + wv;
    load_inputs(clamp_tile(tile));
    // PE features of the tile, computed one tile ahead (under the dW wave's last products of the previous tile)
    h8 E1f[6], E2f[3];
    auto pe_features = [&]() {
      float Bh[33];
#pragma unroll
      for (int i = 0; i < 33; ++i) Bh[i] = Bl_h[i];
      pe_slots<true>(Bh, in_p0 * inv_scale, in_p1 * inv_scale, in_p2 * inv_scale, h, E1f, E2f);
    };
    pe_features();

    for (int t0 = blockIdx.x * NCH; t0 < ntiles; t0 += tile_step, tile += tile_step) {
      asm volatile("" ::: "memory");
      PSTAMP_RESET();
      PSTAMP();
      const bool tile_ok = tile < ntiles;
      const float t0x = in_p0 * inv_scale, t1x = in_p1 * inv_scale, t2x = in_p2 * inv_scale;
      float draw = tile_ok ? in_ds * gscale : 0.0f;
      draw = fminf(fmaxf(draw, -8192.0f), 8192.0f) * 10.0f;  // sigmas = raw * 10 (src/model.py:75)
      const float dr0 = tile_ok ? in_r0 * gscale : 0.0f, dr1 = tile_ok ? in_r1 * gscale : 0.0f,
                  dr2 = tile_ok ? in_r2 * gscale : 0.0f;
      const int row = in_row;
      const float* brow_l = reinterpret_cast<const float*>(smem + P_LDS_BR) + (row - c * rows_per_class) * 128;

      auto pe_backward = [&](const f16v (&de)[3], int nblk, int band0, int nq) {
        float pd[11], gpa[11];
#pragma unroll
        for (int d = 0; d < 11; ++d) {
          pd[d] = Bl_h[3 * d] * t0x + Bl_h[3 * d + 1] * t1x + Bl_h[3 * d + 2] * t2x;
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
          DBACC(3 * d + 0) = fmaf(gpa[d], t0x, DBACC(3 * d + 0));
          DBACC(3 * d + 1) = fmaf(gpa[d], t1x, DBACC(3 * d + 1));
          DBACC(3 * d + 2) = fmaf(gpa[d], t2x, DBACC(3 * d + 2));
        }
      };

      // ------------------------------- forward recompute --------------------------------------------
      // Every layer's weight fragments and bias are fetched while the previous layer's MFMAs are in flight, in
      // source order BEFORE that layer's image stores (LDS reads cannot be hoisted over the stores by the compiler).
      FSTAMP(0);
      h8 wq[8];
      f16v acc, bq;
#pragma unroll
      for (int s = 0; s < 6; ++s) wq[s] = lds_frag(smem, KK_XYZ + s, lane);
      acc = acc_init(cf + CF_B_XYZ, h);
      {
        const int rl = row - c * rows_per_class;
        rowoh[(2 * h + 0) * 32 + col] = rl == 2 * h + 0 ? (_Float16)1 : (_Float16)0;
        rowoh[(2 * h + 1) * 32 + col] = rl == 2 * h + 1 ? (_Float16)1 : (_Float16)0;
        if (h == 0) rowoh[4 * 32 + col] = (_Float16)1;
      }
      {
        unsigned char* b1 = E1img + col * ST_E1 + h * 96;
#pragma unroll
        for (int s = 0; s < 6; ++s) *reinterpret_cast<h8*>(b1 + 16 * s) = E1f[s];
        unsigned char* b2 = E2img + col * ST_E2 + h * 48;
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<h8*>(b2 + 16 * s) = E2f[s];
      }
      FSTAMP(1);
#pragma unroll
      for (int s = 0; s < 6; ++s) acc = MFMA(wq[s], E1f[s], acc);
      wq[0] = lds_frag(smem, KK_S1 + 0, lane); wq[1] = lds_frag(smem, KK_S1 + 1, lane);
      bq = acc_init(brow_l + 0 * 32, h);
      FSTAMP(2);
      {
        const h8 a = pack8(acc, 0, true), b = pack8(acc, 1, true);  // a0
        acc = MFMA(wq[0], a, bq);
        acc = MFMA(wq[1], b, acc);
#pragma unroll
        for (int s = 0; s < 8; ++s) wq[s] = lds_frag(smem, KK_CAT + s, lane);
        bq = acc_init(brow_l + 1 * 32, h);
        stage_h(A0img, a, b, col, h);
      }
      {
        const h8 a = pack8(acc, 0, true), b = pack8(acc, 1, true);  // a1
        acc = MFMA(wq[0], a, bq);
        acc = MFMA(wq[1], b, acc);
#pragma unroll
        for (int s = 0; s < 6; ++s) acc = MFMA(wq[2 + s], E1f[s], acc);
        wq[0] = lds_frag(smem, KK_S2 + 0, lane); wq[1] = lds_frag(smem, KK_S2 + 1, lane);
        bq = acc_init(brow_l + 2 * 32, h);
        stage_h(A1img, a, b, col, h);
      }
      {
        const h8 a = pack8(acc, 0, true), b = pack8(acc, 1, true);  // a2
        acc = MFMA(wq[0], a, bq);
        acc = MFMA(wq[1], b, acc);
        wq[0] = lds_frag(smem, KK_ES + 0, lane); wq[1] = lds_frag(smem, KK_ES + 1, lane);
        bq = acc_init(cf + CF_B_ES, h);
        stage_h(A2img, a, b, col, h);
      }
      FSTAMP(3);
      const h8 A3a = pack8(acc, 0, true), A3b = pack8(acc, 1, true);
      acc = MFMA(wq[0], A3a, bq);
      acc = MFMA(wq[1], A3b, acc);
#pragma unroll
      for (int s = 0; s < 5; ++s) wq[s] = lds_frag(smem, KK_VD + s, lane);
      bq = acc_init(cf + CF_B_VD, h);
#pragma unroll
      for (int i = 0; i < 16; ++i) DWS(i) = fmaf(draw, acc[i], DWS(i));  // d w_sigma += draw * y4
      DBS += (h == 0) ? draw : 0.0f;
      FSTAMP(4);
      const h8 Y4a = pack8(acc, 0, false), Y4b = pack8(acc, 1, false);
      acc = MFMA(wq[0], Y4a, bq);
      acc = MFMA(wq[1], Y4b, acc);
#pragma unroll
      for (int s = 0; s < 3; ++s) acc = MFMA(wq[2 + s], E2f[s], acc);
      wq[0] = lds_frag(smem, KK_T1 + 0, lane); wq[1] = lds_frag(smem, KK_T1 + 1, lane);
      bq = acc_init(brow_l + 3 * 32, h);
      const h8 A5a = pack8(acc, 0, true), A5b = pack8(acc, 1, true);
      acc = MFMA(wq[0], A5a, bq);
      acc = MFMA(wq[1], A5b, acc);
      wq[0] = lds_frag(smem, KK_R0 + 0, lane); wq[1] = lds_frag(smem, KK_R0 + 1, lane);
      bq = acc_init(cf + CF_B_R0, h);
      const h8 A6a = pack8(acc, 0, true), A6b = pack8(acc, 1, true);
      acc = MFMA(wq[0], A6a, bq);
      acc = MFMA(wq[1], A6b, acc);
      wq[0] = lds_frag(smem, KK_R2, lane);
      wq[1] = lds_frag(bwf, KT_R2, lane);
      bq = acc_init(cf + CF_B_R2, h);
      const h8 A7a = pack8(acc, 0, true);
      acc = MFMA(wq[0], A7a, bq);
      h8 Wn0 = lds_frag(bwf, KT_R0, lane), Wn1;
      FSTAMP(5);

      // ---- step ST_R2: dPre9 = drgb * rgb (1 - rgb) in rows 0..2 (registers 0..2 of half 0)
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
      acc = MFMA(wq[1], D0, zero16());  // d a7 (rows 0..15)
      // staged into feature columns 16..18: the dW wave accumulates rgb.2 into rows 16..18 of rgb.0's block
      // (rgb.0 has 16 outputs), which frees one accumulator block for the row sums
      stage_h(slot_D(ST_R2), D1, D0, col, h);
      {
        h8 one = zero8();
        if (h == 0) one[0] = (_Float16)1;  // feature 16 of the a7 image := 1 -> d b(rgb.2)
        stage_h(slot_D(ST_R2) + HIMG_BYTES, A7a, one, col, h);
      }
      D0 = pack8_masked(acc, 0, A7a); D1 = zero8();
      u4v Mn0 = relu_mask(A6a), Mn1 = relu_mask(A6b);   // masks of the next step, ahead of the barrier
      // (the weight fragments of the next step are fetched before each barrier: its fence would otherwise pin
      //  their LDS reads behind it, in front of the MFMA that needs them)
      PSYNC();
      // ---- step ST_R0 ------------------------------------------------------------------------------
      acc = MFMA(Wn0, D0, zero16());  // d a6
      Wn0 = lds_frag(bwf, KT_T1 + 0, lane); Wn1 = lds_frag(bwf, KT_T1 + 1, lane);
      stage_h(slot_D(ST_R0), D0, D1, col, h);
      stage_h(slot_D(ST_R0) + HIMG_BYTES, A6a, A6b, col, h);
      D0 = pack8_and(acc, 0, Mn0); D1 = pack8_and(acc, 1, Mn1);
      Mn0 = relu_mask(A5a); Mn1 = relu_mask(A5b);
      PSYNC();
      // ---- step ST_T1 ------------------------------------------------------------------------------
      acc = MFMA(Wn0, D0, zero16());
      acc = MFMA(Wn1, D1, acc);  // d a5
      Wn0 = lds_frag(bwf, KT_VD_Y + 0, lane); Wn1 = lds_frag(bwf, KT_VD_Y + 1, lane);
      stage_h(slot_D(ST_T1), D0, D1, col, h);
      stage_h(slot_D(ST_T1) + HIMG_BYTES, A5a, A5b, col, h);
      D0 = pack8_and(acc, 0, Mn0); D1 = pack8_and(acc, 1, Mn1);
      PSYNC();
      // ---- step ST_VD : inputs [y4 | e2] -----------------------------------------------------------
      acc = MFMA(Wn0, D0, zero16());
      acc = MFMA(Wn1, D1, acc);  // d y4 from the colour branch
      f16v de2[3];  // d e2 (two 16-slot blocks); its PE backward runs in the next step, under the dW wave's 3 blocks
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        de2[b] = MFMA(lds_frag(bwf, KT_VD_E + 2 * b + 0, lane), D0, zero16());
        de2[b] = MFMA(lds_frag(bwf, KT_VD_E + 2 * b + 1, lane), D1, de2[b]);
      }
      de2[2] = de2[1];
      Wn0 = lds_frag(bwf, KT_ES + 0, lane); Wn1 = lds_frag(bwf, KT_ES + 1, lane);
      stage_h(slot_D(ST_VD), D0, D1, col, h);
      stage_h(slot_D(ST_VD) + HIMG_BYTES, Y4a, Y4b, col, h);
      {  // + sigma head: d y4 += w_sigma * draw
        const f16v wsg = acc_init(cf + CF_W_SG, h);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fmaf(wsg[i], draw, acc[i]);
      }
      D0 = pack8(acc, 0, false); D1 = pack8(acc, 1, false);
      Mn0 = relu_mask(A3a); Mn1 = relu_mask(A3b);
      PSYNC();
      // ---- step ST_ES (no activation) --------------------------------------------------------------
      acc = MFMA(Wn0, D0, zero16());
      acc = MFMA(Wn1, D1, acc);  // d a3
      Wn0 = lds_frag(bwf, KT_S2 + 0, lane); Wn1 = lds_frag(bwf, KT_S2 + 1, lane);
      stage_h(slot_D(ST_ES), D0, D1, col, h);
      stage_h(slot_D(ST_ES) + HIMG_BYTES, A3a, A3b, col, h);
      D0 = pack8_and(acc, 0, Mn0); D1 = pack8_and(acc, 1, Mn1);
      pe_backward(de2, 2, 4, 22);  // bands 4 and 5 -> dB
      {
        h8 a, b;
        load_h(A2img, a, b, col, h);
        Mn0 = relu_mask(a); Mn1 = relu_mask(b);
      }
      PSYNC();
      // ---- step ST_S2 : input a2 (parked) ------------------------------------------------------------
      acc = MFMA(Wn0, D0, zero16());
      acc = MFMA(Wn1, D1, acc);  // d a2
      Wn0 = lds_frag(bwf, KT_CAT_Y + 0, lane); Wn1 = lds_frag(bwf, KT_CAT_Y + 1, lane);
      stage_h(slot_D(ST_S2), D0, D1, col, h);
      D0 = pack8_and(acc, 0, Mn0); D1 = pack8_and(acc, 1, Mn1);
      {
        h8 a, b;
        load_h(A1img, a, b, col, h);
        Mn0 = relu_mask(a); Mn1 = relu_mask(b);
      }
      PSYNC();
      // ---- step ST_CAT : inputs [a1 | e1] (both parked) ----------------------------------------------
      acc = MFMA(Wn0, D0, zero16());
      acc = MFMA(Wn1, D1, acc);  // d a1
      Wn0 = lds_frag(bwf, KT_S1 + 0, lane); Wn1 = lds_frag(bwf, KT_S1 + 1, lane);
      const h8 Dc0 = D0, Dc1 = D1;  // its d e1 part is formed together with encoding_xyz's
      stage_h(slot_D(ST_CAT), D0, D1, col, h);
      D0 = pack8_and(acc, 0, Mn0); D1 = pack8_and(acc, 1, Mn1);
      {
        h8 a, b;
        load_h(A0img, a, b, col, h);
        Mn0 = relu_mask(a); Mn1 = relu_mask(b);
      }
      PSYNC();
      // ---- step ST_S1 : input a0 (parked) --------------------------------------------------------------
      acc = MFMA(Wn0, D0, zero16());
      acc = MFMA(Wn1, D1, acc);  // d a0
      f16v de[3];  // the cat part of d e1 does not depend on this step's result: under the dW wave's 4 cat blocks
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        de[b] = MFMA(lds_frag(bwf, KT_CAT_E + 2 * b + 0, lane), Dc0, zero16());
        de[b] = MFMA(lds_frag(bwf, KT_CAT_E + 2 * b + 1, lane), Dc1, de[b]);
      }
      stage_h(slot_D(ST_S1), D0, D1, col, h);
      D0 = pack8_and(acc, 0, Mn0); D1 = pack8_and(acc, 1, Mn1);
      load_inputs(clamp_tile(tile + tile_step));  // next tile's sample: in flight across the next two steps
      PSYNC();
      // ---- step ST_XYZ : input e1 (parked) ------------------------------------------------------------------
      stage_h(slot_D(ST_XYZ), D0, D1, col, h);
      // d e1 = Wc_e^T dPre(cat) + Wx^T dPre(xyz) (three 16-slot blocks) -> dB, bands 0..3
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        de[b] = MFMA(lds_frag(bwf, KT_XYZ_E + 2 * b + 0, lane), D0, de[b]);
        de[b] = MFMA(lds_frag(bwf, KT_XYZ_E + 2 * b + 1, lane), D1, de[b]);
      }
      pe_backward(de, 3, 0, 44);
      PSYNC();
      // ---- the dW wave finishes encoding_xyz; this wave computes the next tile's PE features meanwhile ---------
      pe_features();
#pragma unroll
      for (int s = 0; s < 6; ++s) asm volatile("" : "+v"(E1f[s]));  // materialise here, not below the barrier
#pragma unroll
      for (int s = 0; s < 3; ++s) asm volatile("" : "+v"(E2f[s]));
      PSYNC();  // the dW wave is done with this tile's images
    }
  } else {
    // ===================================================================================================
    // dW role
    // ===================================================================================================
    // which dW wave owns a step's blocks (NDW = 2: texture + enc_shape + shape_2 | cat, shape_1, xyz); the row-sum
    // block replaces rgb.2's, so with two dW waves the first one takes the row sums of every layer
    auto owns = [&](int step) { return NDW == 1 || (dwid == 0) == (step <= ST_S2); };
    const bool owns_rs = NDW == 1 || dwid == 0;
    constexpr int LI_RS = 0;  // BK_R2's local index
    // row-sum block RS[m][feature]: m = 4 * latent slot + object row (m < 16), m = 16: enc_shape bias, m = 17: rgb.0 bias
    const int m_row = col < 16 ? (col & 3) : 4, m_grp = col < 16 ? (col >> 2) : (col - 12);  // groups 0..3 latent, 4 / 5 the biases

    for (int t0 = blockIdx.x * NCH; t0 < ntiles; t0 += tile_step) {
      asm volatile("" ::: "memory");
      PSTAMP_RESET();
      PSTAMP();
      h8 RF[NCH][2];  // A operand of the row sums before the layer mask: [m][sample k] = (object row of k == m_row), or 1

      // One layer step for every chain wave's tile.  The transposing LDS reads are convergent operations, which the
      // scheduler keeps in source order: all operand fragments of the step are therefore read first (they are in
      // flight together), then the MFMAs run back to back.
      auto consume = [&](auto step_c) {
        constexpr int step = decltype(step_c)::value;
        constexpr int NX = step == ST_VD ? 3 : step == ST_CAT ? 4 : step == ST_XYZ ? 3 : 1;  // dW blocks of the layer
        constexpr int KIND0 = step == ST_R2 ? BK_R0     // rows 16..18 of rgb.0's block, see the chain role
                            : step == ST_R0 ? BK_R0 : step == ST_T1 ? BK_T1 : step == ST_VD ? BK_VD_Y
                            : step == ST_ES ? BK_ES : step == ST_S2 ? BK_S2 : step == ST_CAT ? BK_CAT_Y
                            : step == ST_S1 ? BK_S1 : BK_XYZ_E0;
        constexpr int RS_GRP = step == ST_R0 ? 5 : step == ST_T1 ? 3 : step == ST_ES ? 4 : step == ST_S2 ? 2
                             : step == ST_CAT ? 1 : step == ST_S1 ? 0 : -1;
        const bool mine = owns(step);
        if (!mine && !(owns_rs && RS_GRP >= 0)) return;
        h8 fD[2][2], fX[2][NX][2];  // two tiles in flight
        auto load_tile = [&](int w) {
          const unsigned char* cb = chain_base + w * CW_BYTES;
          const unsigned char* Dimg = cb + CW_SLOT + (step & 1) * SLOT_BYTES;
          fD[w & 1][0] = tr_frag(Dimg, ST_H, 0, 0, lane);
          fD[w & 1][1] = tr_frag(Dimg, ST_H, 0, 1, lane);
          if (!mine) return;
#pragma unroll
          for (int b = 0; b < NX; ++b) {
            // input image of block b: the slot's X image, or an image the chain wave parked at forward time
            const unsigned char* ximg;
            int stride = ST_H, col0 = 0;
            if (step == ST_R2 || step == ST_R0 || step == ST_T1 || step == ST_ES || (step == ST_VD && b == 0))
              ximg = Dimg + HIMG_BYTES;
            else if (step == ST_VD) { ximg = cb + CW_E2; stride = ST_E2; col0 = 32 * (b - 1); }
            else if (step == ST_S2) ximg = cb + CW_A2;
            else if (step == ST_S1) ximg = cb + CW_A0;
            else if (step == ST_CAT && b == 0) ximg = cb + CW_A1;
            else { ximg = cb + CW_E1; stride = ST_E1; col0 = 32 * (step == ST_CAT ? b - 1 : b); }
            fX[w & 1][b][0] = tr_frag(ximg, stride, col0, 0, lane);
            fX[w & 1][b][1] = tr_frag(ximg, stride, col0, 1, lane);
          }
        };
        load_tile(0);
#pragma unroll
        for (int w = 0; w < NCH; ++w) {
          if (w + 1 < NCH) load_tile(w + 1);
          if (mine) {
#pragma unroll
            for (int b = 0; b < NX; ++b) {
              const int kind = KIND0 + b;  // compile-time after unrolling
              const int li = NDW == 1 ? kind : (kind < BK_CAT_Y ? kind : kind - BK_CAT_Y);
              Wacc[li] = MFMA(fD[w & 1][0], fX[w & 1][b][0], Wacc[li]);
              Wacc[li] = MFMA(fD[w & 1][1], fX[w & 1][b][1], Wacc[li]);
            }
          }
          if (owns_rs && RS_GRP >= 0) {
            // RS[m][:] += sum over the tile's samples in m's group of dPre: no LDS, no result read-back
            typedef unsigned int u4 __attribute__((ext_vector_type(4)));
            const unsigned int lm = (m_grp == RS_GRP) ? 0xffffffffu : 0u;
            const h8 a0 = __builtin_bit_cast(h8, (u4)(__builtin_bit_cast(u4, RF[w][0]) & lm));
            const h8 a1 = __builtin_bit_cast(h8, (u4)(__builtin_bit_cast(u4, RF[w][1]) & lm));
            Wacc[LI_RS] = MFMA(a0, fD[w & 1][0], Wacc[LI_RS]);
            Wacc[LI_RS] = MFMA(a1, fD[w & 1][1], Wacc[LI_RS]);
          }
        }
      };
      PSYNC();
      if (owns_rs) {
#pragma unroll
        for (int w = 0; w < NCH; ++w) {
          const _Float16* rowoh = reinterpret_cast<const _Float16*>(chain_base + w * CW_BYTES + CW_SMALL);
#pragma unroll
          for (int s = 0; s < 2; ++s) RF[w][s] = *reinterpret_cast<const h8*>(rowoh + m_row * 32 + 16 * s + 8 * h);
        }
      }
      consume(IC<ST_R2>{});
      PSYNC();
      consume(IC<ST_R0>{});
      PSYNC();
      consume(IC<ST_T1>{});
      PSYNC();
      consume(IC<ST_VD>{});
      PSYNC();
      consume(IC<ST_ES>{});
      PSYNC();
      consume(IC<ST_S2>{});
      PSYNC();
      consume(IC<ST_CAT>{});
      PSYNC();
      consume(IC<ST_S1>{});
      PSYNC();
      consume(IC<ST_XYZ>{});
      PSYNC();
    }
  }

  // ========================================= flush ====================================================
  __syncthreads();
  float* rec = records + ((size_t)c * gridDim.x + blockIdx.x) * REC_FLOATS;
  if (is_chain) {
    float* small = reinterpret_cast<float*>(chain_base + wv * CW_BYTES + CW_SMALL);  // [0..31] d w_sigma, [32] d b_sigma, [64..126] dB
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float v = half_sum_dpp(DWS(i));
      if (col == 31) small[acc_row(i, h)] = v;
    }
    {
      const float v = half_sum_dpp(DBS);  // zero in lane half 1
      if (lane == 31) small[32] = v;
    }
#pragma unroll
    for (int i = 0; i < 33; ++i) {
      const float v = half_sum_dpp(DBACC(i));
      const int d = i / 3;
      if (col == 31 && !(h == 1 && d == 10)) small[64 + (11 * h + d) * 3 + (i % 3)] = v;
    }
  }
  __syncthreads();
  {
    auto sum_chain = [&](int i) {
      float v = 0.0f;
#pragma unroll
      for (int w = 0; w < NCH; ++w) v += reinterpret_cast<const float*>(chain_base + w * CW_BYTES + CW_SMALL)[i];
      return v;
    };
    for (int i = threadIdx.x; i < 63; i += 256) { rec[TRUNK + i] = sum_chain(64 + i) * inv_gs; rec[TRUNK + 63 + i] = 0.0f; }
    for (int i = threadIdx.x; i < 32; i += 256) rec[OFF_SG_W + i] = sum_chain(i) * inv_gs;
    if (threadIdx.x == 0) rec[OFF_SG_B] = sum_chain(32) * inv_gs;
  }
  __syncthreads();  // everything above has been read: the accumulator image may alias it
  if (!is_chain) {
    float* region = reinterpret_cast<float*>(smem);
    const bool owns_rs_flush = NDW == 1 || dwid == 0;
#define CNR_PSTORE(KIND)                                                                      \
  if (NDW == 1 || (dwid == 0) == ((KIND) < BK_CAT_Y)) {                                       \
    constexpr int li = NDW == 1 ? (KIND) : ((KIND) < BK_CAT_Y ? (KIND) : (KIND)-BK_CAT_Y);    \
    _Pragma("unroll") for (int reg = 0; reg < 16; ++reg)                                      \
        region[(KIND) * 1024 + acc_row(reg, h) * 32 + col] = Wacc[li][reg];                   \
  }
    if (owns_rs_flush) {  // the row-sum block, and rgb.2 out of rows 16..31 of rgb.0's block
      constexpr int li_r0 = BK_R0;  // same local index for NDW = 1 and for dW wave 0 of NDW = 2
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        region[RS_REGION * 1024 + acc_row(reg, h) * 32 + col] = Wacc[0][reg];
        if (reg >= 8) region[BK_R2 * 1024 + (acc_row(reg, h) - 16) * 32 + col] = Wacc[li_r0][reg];
      }
    }
    CNR_PSTORE(BK_R0) CNR_PSTORE(BK_T1) CNR_PSTORE(BK_VD_Y) CNR_PSTORE(BK_VD_E0)
    CNR_PSTORE(BK_VD_E1) CNR_PSTORE(BK_ES) CNR_PSTORE(BK_S2) CNR_PSTORE(BK_CAT_Y) CNR_PSTORE(BK_CAT_E0)
    CNR_PSTORE(BK_CAT_E1) CNR_PSTORE(BK_CAT_E2) CNR_PSTORE(BK_S1) CNR_PSTORE(BK_XYZ_E0) CNR_PSTORE(BK_XYZ_E1)
    CNR_PSTORE(BK_XYZ_E2)
#undef CNR_PSTORE
  }
  __syncthreads();
  {
    const float* region = reinterpret_cast<const float*>(smem);
    for (int j = threadIdx.x; j < TRUNK; j += 256) {  // coalesced record stores, LDS gather
      const int src = g_param_src[j];
      if (src >= 0) rec[j] = region[src] * inv_gs;
    }
    const float* rs = region + RS_REGION * 1024;  // [m][feature]
    for (int i = threadIdx.x; i < 32; i += 256) rec[OFF_ES_B + i] = rs[16 * 32 + i] * inv_gs;
    for (int i = threadIdx.x; i < 16; i += 256) rec[OFF_R0_B + i] = rs[17 * 32 + i] * inv_gs;
    for (int i = threadIdx.x; i < rows_per_class * 128; i += 256) {  // dbiasrows [row][latent slot][feature]
      const float v = rs[(((i >> 5) & 3) * 4 + (i >> 7)) * 32 + (i & 31)] * inv_gs;
      rec[TRUNK + 126 + i] = v;
      // the same sums as 2^-40 fixed point: integer atomics add in any order to the same result, so the consumer
      // (cnr_step_tail) has them without waiting for the record reduction
      if (rows_fix)
        atomicAdd(reinterpret_cast<unsigned long long*>(
                      rows_fix + ((size_t)(blockIdx.x % cnr_rec::ROWS_FIX_COPIES) * gridDim.y + c) * rows_per_class * 128 + i),
                  (unsigned long long)__double2ll_rn((double)v * cnr_rec::ROWS_FIX_SCALE));
    }
  }
}
}  // namespace

#ifdef CNR_PIPE_STAMPS
extern "C" int cnr_pipe_read_stamps(long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pipe_stamps), sizeof(long long) * 160);
}
#endif

// workgroups (= records) per class of a cnr_field_bwd_pipe launch: what a caller that reduces the records itself needs
// the 8-wave kernel (4 chain + 4 dW waves) lives in fused_bwd_pipe8.hip
extern "C" int cnr_field_bwd_pipe8_launch(const float* pts, const float* B, const void* packed, const float* biasrows,
                                          const int* ray_row, float scale, const float* d_sigma, const float* d_rgb,
                                          float grad_scale, int C, int R, int S, int rows_per_class, int blocks,
                                          void* workspace, int64_t B_stride, long long* rows_fix, int* clamp_flags,
                                          void* stream);

static bool pipe_waves_ok(int chain_waves) { return chain_waves == 2 || chain_waves == 3 || chain_waves == 4; }

extern "C" int cnr_field_bwd_pipe_blocks(int R, int S, int chain_waves, int max_blocks) {
  if (R <= 0 || S <= 0 || !pipe_waves_ok(chain_waves)) return 0;
  const int64_t ntiles = ((int64_t)R * S + 31) / 32;
  int64_t blocks = (ntiles + chain_waves - 1) / chain_waves;
  const int64_t cap = max_blocks > 0 ? max_blocks : 256;
  return (int)(blocks > cap ? cap : blocks);
}

extern "C" int cnr_field_bwd_pipe(const float* pts, const float* B, const void* packed, const float* biasrows,
                                  const int* ray_row, float scale, const float* d_sigma, const float* d_rgb,
                                  float grad_scale, float* dtrunk, float* dB, float* dbiasrows, int C, int R, int S,
                                  int rows_per_class, int max_blocks, int chain_waves, void* workspace,
                                  int64_t workspace_bytes, int64_t B_stride, int64_t dtrunk_stride, int64_t dB_stride,
                                  long long* rows_fix, int skip_reduce, int* clamp_flags, void* stream) {
  if (!pts || !B || !packed || !biasrows || !d_sigma || !d_rgb || !dtrunk || !dB || !dbiasrows || !workspace)
    return CNR_E_ARG;
  if (C <= 0 || R <= 0 || S <= 0 || !(scale > 0.f) || !(grad_scale > 0.f)) return CNR_E_ARG;
  if (!pipe_waves_ok(chain_waves)) return CNR_E_ARG;
  // the pipeline keeps per-object row sums in one accumulator block: class-major rows, at most ROWS_LDS per class
  // (ROWS_MAX = 15 in the 8-wave kernel).
  // Everything else (one row per ray, many objects) takes the block-split kernels.
  if (ray_row == nullptr || rows_per_class < 1 || rows_per_class > (chain_waves == 4 ? cnr_rec::ROWS_MAX : ROWS_LDS)) {
    if (rows_fix || skip_reduce) return CNR_E_ARG;   // those two need the pipelined kernel's per-object rows
    return cnr_field_bwd(pts, B, packed, biasrows, ray_row, scale, d_sigma, d_rgb, grad_scale, dtrunk, dB, dbiasrows, C,
                         R, S, rows_per_class, max_blocks, workspace, workspace_bytes, B_stride, dtrunk_stride, dB_stride,
                         stream);
  }
  if (S > 240) return CNR_E_SHAPE;
  if (((uintptr_t)packed & 15) != 0 || ((uintptr_t)biasrows & 15) != 0 || ((uintptr_t)workspace & 15) != 0)
    return CNR_E_ALIGN;
  if (ray_row == nullptr && rows_per_class > 0 && rows_per_class != R) return CNR_E_ARG;
  const int64_t N = (int64_t)R * S;
  if (N > (int64_t)0x7fffff00) return CNR_E_SHAPE;  // tile and sample indices are 32-bit inside the kernel
  const int64_t ntiles = (N + 31) / 32;
  int64_t blocks = (ntiles + chain_waves - 1) / chain_waves;
  const int64_t cap = max_blocks > 0 ? max_blocks : 256;
  if (blocks > cap) blocks = cap;
  const int64_t need = (int64_t)C * blocks * REC_FLOATS * (int64_t)sizeof(float);
  if (workspace_bytes < need) return CNR_E_ARG;
  if (chain_waves == 4) {  // the 8-wave kernel
    const int rc = cnr_field_bwd_pipe8_launch(pts, B, packed, biasrows, ray_row, scale, d_sigma, d_rgb, grad_scale, C, R, S,
                                              rows_per_class, (int)blocks, workspace, B_stride, rows_fix, clamp_flags,
                                              stream);
    if (rc != CNR_OK) return rc;
    if (skip_reduce) return CNR_OK;
    hipLaunchKernelGGL(reduce_records_kernel, dim3(REC_FLOATS / 64, (unsigned)C), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, (int)blocks, dtrunk, dB, dbiasrows, rows_per_class,
                       dtrunk_stride > 0 ? dtrunk_stride : (int64_t)TRUNK, dB_stride > 0 ? dB_stride : (int64_t)63);
    CNR_LAUNCH_CHECK();
    return CNR_OK;
  }
  {  // once per device: the LDS attribute of both instantiations and the constant table (static device storage)
    static cnr::DeviceOnce once3, once2, table;
    int er = cnr::set_max_dynamic_lds(once3, (const void*)field_bwd_pipe_kernel<3>, p_lds_total<3>());
    if (er) return er;
    er = cnr::set_max_dynamic_lds(once2, (const void*)field_bwd_pipe_kernel<2>, p_lds_total<2>());
    if (er) return er;
    if (cnr::first_on_device(table)) {
      hipLaunchKernelGGL(build_param_src_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream);
      hipLaunchKernelGGL(fill_param_src_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream);
    }
  }
  dim3 grid((unsigned)blocks, (unsigned)C);
#define CNR_LAUNCH_PIPE(NCH)                                                                                  \
  hipLaunchKernelGGL((field_bwd_pipe_kernel<NCH>), grid, dim3(256), p_lds_total<NCH>(), (hipStream_t)stream,  \
                     pts, B, (const unsigned char*)packed, biasrows, ray_row, 1.0f / scale, d_sigma, d_rgb,   \
                     grad_scale, (float*)workspace, dbiasrows, (int)N, S, R, rows_per_class,                    \
                     B_stride > 0 ? B_stride : (int64_t)63, rows_fix)
  if (chain_waves == 3) CNR_LAUNCH_PIPE(3); else CNR_LAUNCH_PIPE(2);
#undef CNR_LAUNCH_PIPE
  CNR_LAUNCH_CHECK();
  if (skip_reduce) return CNR_OK;   // the caller reduces the records itself (cnr_step_tail)
  hipLaunchKernelGGL(reduce_records_kernel, dim3(REC_FLOATS / 64, (unsigned)C), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, (int)blocks, dtrunk, dB, dbiasrows, rows_per_class,
                     dtrunk_stride > 0 ? dtrunk_stride : (int64_t)TRUNK, dB_stride > 0 ? dB_stride : (int64_t)63);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}
