// Fused field backward, 8-wave pipeline: workgroup = 4 chain waves + 4 dW waves, one of each per SIMD (TWO waves per SIMD:
// a chain wave's dependent latency -- MFMA result -> mask / pack -> image store -> barrier -- is covered by its SIMD
// partner instead of idling the SIMD as in round 1's 3 + 1 form, one wave per SIMD).
// What it takes to fit eight waves into one CU:
//   * 256 registers per wave: the 16 accumulator blocks are spread over the four dW waves (<= 5 each); a chain wave
//     keeps a0 .. a7 in registers until their step stages them as the layer input (no parked images);
//   * LDS: operand fragments 61 KB + 4 x 15.1 KB per chain wave (PE images 10.5 KB, ONE dPre / input slot of
//     2 x 2 KB, row one-hot table) -- the slot is single-buffered behind TWO barriers per layer step:
//        chain: store step j's images | A | MFMA + mask + pack of step j+1 (registers only) | B | store step j+1 ...
//        dW   :                         A | read step j's images, MFMAs                    | B
//   * the slot images are unpadded 64-byte rows with an XOR chunk swizzle (stage_hs / tr_frag_hs): conflict-free for
//     the transposing reads and the b64 stores;
//   * the next tile's inputs (pts, upstream gradients, object row) are fetched three steps before the iteration ends.
// Every dW wave has one unit of work per layer step (block ownership table below), reads all four tiles' operands up
// front (it has the registers for that) and runs the MFMAs as they land.
// Measured (tools/exp/run_pipe_timed.py, 8192 x 128): iteration = 4 tiles = ~17.5 k cycles against 15.5 k for 3 tiles
// in the 3 + 1 form; forward recompute ~5.3 k (the dW waves wait), nine layer steps at ~1.2 k each with chain and
// dW about level.  6 chain + 2 dW waves was tried first: 2 dW waves with 8 blocks each are the critical path
// (12.2 us per 6 tiles against 7.3 us per 4 here).
// Same records, same results contract as cnr_field_bwd_pipe (chain_waves = 4 selects this kernel).
// (This header holds the kernel and its launcher template; fused_bwd_pipe8.hip instantiates the stand-alone backward and holds the
//  C-ABI entry points, fused_bwd_pipe8_w0 .. w3.hip instantiate the one-launch forms of one WIDE each, so that the ~45
//  instantiations compile in parallel: the single unit took three minutes.)
#pragma once
#include "fused_bwd_common.h"

namespace {
#ifdef CNR_PIPE_STAMPS  // tools/exp only: cycle stamps of every wave of workgroup 0 (its last iteration)
__device__ long long g_pipe8_stamps[8 * 64];
#ifndef CNR_STAMP_ITER
#define CNR_STAMP_ITER 0   // which iteration (1-based) of workgroup 0 the barrier stamps and marks record; 0 = every one (the last stays)
#endif
#define P8STAMP() do { if (blockIdx.x == 0 && lane == 0 && (CNR_STAMP_ITER == 0 || p8_iter == CNR_STAMP_ITER)) \
    g_pipe8_stamps[wv * 64 + (pstamp_i++)] = (long long)__builtin_readcyclecounter(); } while (0)
#define P8STAMP_RESET() int pstamp_i = 0; do { if (blockIdx.x == 0 && lane == 0 && p8_iter < 12) \
    g_pipe8_stamps[wv * 64 + 30 + (p8_iter++)] = (long long)__builtin_readcyclecounter(); } while (0)
#define P8ITER_DECL() int p8_iter = 0
#define P8PHASE(k) do { if (blockIdx.x == 0 && lane == 0) \
    g_pipe8_stamps[wv * 64 + 56 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#define P8MARK(k) do { if (blockIdx.x == 0 && lane == 0 && (CNR_STAMP_ITER == 0 || p8_iter == CNR_STAMP_ITER)) \
    g_pipe8_stamps[wv * 64 + 44 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define P8MARK(k) do {} while (0)
#define P8PHASE(k) do {} while (0)
#define P8STAMP() do {} while (0)
#define P8STAMP_RESET() do {} while (0)
#define P8ITER_DECL() do {} while (0)
#endif
#define P8SYNC() do { P8STAMP(); role_barrier(); P8STAMP(); } while (0)
#ifdef CNR_ISA_MARKS  // tools/isa_mix.py --phases: named comments in the assembly (hipcc --save-temps -DCNR_ISA_MARKS), no instruction
#define P8ISA(name) asm volatile("; CNRMARK " name)
#else
#define P8ISA(name) do {} while (0)
#endif
// LDS layout of the 8-wave kernel, by instantiation (WIDE = which row-sum form, GEO = precise geometry branch):
//   [ packed image PK_BYTES | GEO: residual fragments PK_LO_BYTES | B rows 272 | bias rows of the class | 4 chain waves | exchange ]
// A class's bias rows (512 B per object row) sit in LDS except in the one instantiation where the budget is spent
// (GEO with up to 15 rows: they are read from global there, ~1 k cycles per layer of the forward).
// (WIDE = 3, any number of object rows: no per-object rows in LDS at all -- the row-sum block then holds one row per (chain wave,
//  latent slot), see the kernel's comment; rs_rows8 is the one-hot table's row count, unused there)
__host__ __device__ constexpr int rs_rows8(int wide) { return wide == 2 ? cnr_rec::ROWS_MAX : wide == 1 ? 7 : 4; }
__host__ __device__ constexpr bool brows_in_lds8(int wide, bool geo) { return wide == 3 ? false : !(geo && wide == 2); }
// row one-hot table per chain wave: rows + the ones row, 32 halfs each; the flush reuses it for 127 floats of partial sums
__host__ __device__ constexpr int k8_small(int wide) { return wide != 3 && (rs_rows8(wide) + 1) * 64 > 512 ? (rs_rows8(wide) + 1) * 64 : 512; }
// PE images without the trailing pads of the 4-wave kernels: an E2 read of the (discarded) slot columns 48..63 of row 31
// runs 16 B past the image -- into this wave's dPre slot, which follows it
constexpr int E1IMG8 = 32 * ST_E1, E2IMG8 = 32 * ST_E2;
__host__ __device__ constexpr int c8_bytes(int wide) { return E1IMG8 + E2IMG8 + 4 * HSIMG_BYTES + k8_small(wide); }
__host__ __device__ constexpr int l8_lo() { return PK_BYTES; }
__host__ __device__ constexpr int l8_bl(bool geo) { return PK_BYTES + (geo ? PK_LO_BYTES : 0); }
__host__ __device__ constexpr int l8_br(bool geo) { return l8_bl(geo) + 272; }
__host__ __device__ constexpr int l8_chain(int wide, bool geo) {
  return l8_br(geo) + (brows_in_lds8(wide, geo) ? rs_rows8(wide) * 128 * 4 : 0);
}
constexpr int BK_RS = 100, BK_RS2 = 101;  // pseudo kinds of the row-sum blocks in the ownership tables
constexpr int XCH_BYTES = 2 * 4 * 8 * 4;  // one-launch step: four chain waves x eight floats, two copies (iteration parity)
__host__ __device__ constexpr int l8_xch(int nch, int wide, bool geo) { return l8_chain(wide, geo) + nch * c8_bytes(wide); }
__host__ __device__ constexpr int l8_total(int nch, int wide, bool geo) { return l8_xch(nch, wide, geo) + XCH_BYTES; }
// KR > 0 (cnr_field_train): the render / loss side of the step, so that ONE launch runs field forward -> composite ->
// losses -> their gradient -> composite backward -> field backward with no second forward and no d sigma / d colour
// round trip through HBM.  A ray occupies SP = 32 KR padded sample slots (S <= SP; slots >= S are dead lanes): the KR chain
// waves wv, wv ^ 1, .. of a workgroup iteration hold one ray.  TWO (KR = 1 only): SP = 16, two short rays side by side in a
// tile, one per DPP row -- the reference's real batch shape is 10 samples per ray (config_replica_room0.json:25-28).
struct TrainArgs {
  const float* z; const float* gt_depth; const float* gt_rgb; const uint8_t* labels; const uint8_t* depth_mask;
  const float* counts_tab; const int64_t* d_state; float color_scaling, opacity_scaling, loss_scale;
  float* depth_out; float* var_out; float* rgb_out; float* opacity_out; float* partials;
};
static_assert(l8_total(4, 0, true) <= 160 * 1024 && l8_total(4, 1, true) <= 160 * 1024 && l8_total(4, 2, true) <= 160 * 1024 &&
              l8_total(4, 2, false) <= 160 * 1024 && l8_total(4, 3, true) <= 160 * 1024, "LDS budget");

// Which dW wave owns a block kind, and the block's index among that wave's accumulators: at most 5 accumulators per
// wave, one unit of work per wave and layer step (cat_layer: dW3 two)
//     dW0: row sums (one block; two with more than 7 object rows), viewdir[y], xyz[e1 0..31]          dW1: rgb.0 (+ rgb.2 rows 16..18), texture_1, cat[y], viewdir[e2 0..31], xyz[e1 32..63]
//     dW2: enc_shape, shape_2, cat[e1 0..31], viewdir[e2 32..], xyz[e1 64..]      dW3: shape_1, cat[e1 32..63], cat[e1 64..]
template <int NDW> __host__ __device__ constexpr int owner8(int kind);
template <int NDW> __host__ __device__ constexpr int local8(int kind);
template <> __host__ __device__ constexpr int owner8<4>(int kind) {
  return (kind == BK_RS || kind == BK_RS2 || kind == BK_VD_Y || kind == BK_XYZ_E0) ? 0
       : (kind == BK_R0 || kind == BK_T1 || kind == BK_CAT_Y || kind == BK_VD_E0 || kind == BK_XYZ_E1) ? 1
       : (kind == BK_ES || kind == BK_S2 || kind == BK_CAT_E0 || kind == BK_VD_E1 || kind == BK_XYZ_E2) ? 2 : 3;
}
template <> __host__ __device__ constexpr int local8<4>(int kind) {
  return kind == BK_RS ? 0 : kind == BK_VD_Y ? 1 : kind == BK_XYZ_E0 ? 2 : kind == BK_RS2 ? 3
       : kind == BK_R0 ? 0 : kind == BK_T1 ? 1 : kind == BK_CAT_Y ? 2 : kind == BK_VD_E0 ? 3 : kind == BK_XYZ_E1 ? 4
       : kind == BK_ES ? 0 : kind == BK_S2 ? 1 : kind == BK_CAT_E0 ? 2 : kind == BK_VD_E1 ? 3 : kind == BK_XYZ_E2 ? 4
       : kind == BK_S1 ? 0 : kind == BK_CAT_E1 ? 1 : 2 /* BK_CAT_E2 */;
}

// WIDE: more than four object rows per class.  The row stride of the row-sum block is a constant of the instantiation -- 4, 7
// or 15, the most rows it takes; a class with fewer leaves rows unused -- so that the index arithmetic folds (as a run-time
// value it cost 2.4 us of the kernel at 2048 x 64).  WIDE = 1: up to 7 rows,
// one row-sum block [4 latent slots x rows | 2 plain biases] (4 * 7 + 2 <= 32 block rows).  WIDE = 2: up to ROWS_MAX =
// 15 rows, two blocks: A = [slots 0, 1 x rows | 2 plain biases], B = [slots 2, 3 x rows]; a layer step feeds one of them.
// WIDE = 3 (one-launch form with a whole ray per tile group, i.e. KR >= 1 and not TWO): ANY number of object rows per class.  A
// 32-sample tile lies inside one ray, hence belongs to ONE object: its per-object row sum is the plain column sum of dPre.  The
// row-sum block holds row 4 w + s for chain wave w's tile and latent slot s (A operand: all ones in that row) and rows 16, 17
// for the two plain biases; after every iteration its owner adds rows 0..15 to the fixed-point table at the four tiles' object
// rows (integer atomics: order-free, bitwise reproducible) and clears them.  No one-hot table, bias rows from global memory.
// GEO: the forward's geometry branch as three products per fragment with the residual weight image packed_lo (fused_common.h,
// NKK_GEO) -- what keeps the occupancy within 1e-3 of fp32 for trained weights.  The stand-alone backward (KR = 0) takes it too:
// it must recompute the activations and ReLU masks of the forward that was rendered (cnr_field_fwd / cnr_field_fwd_render with
// the same image), or it is the gradient of a different function.
template <int NCH, int NDW, int WIDE, int KR, bool TWO, bool PAD, bool GEO>
__global__ __launch_bounds__((NCH + NDW) * 64, 1) void field_bwd_pipe8_kernel(
    const float* __restrict__ pts, const float* __restrict__ Bdir, const unsigned char* __restrict__ packed,
    const unsigned char* __restrict__ packed_lo, const float* __restrict__ biasrows, const int* __restrict__ ray_row, float inv_scale,
    const float* __restrict__ d_sigma, const float* __restrict__ d_rgb, float gscale, rec_t* __restrict__ records,
    int N, int S, int R, int rows_per_class, int64_t B_stride, long long* __restrict__ rows_fix,
    int* __restrict__ clamp_flags, TrainArgs ta) {
  static_assert(KR == 0 || KR == 1 || KR == 2 || KR == 4, "tiles per ray");
  static_assert(!TWO || KR == 1, "two rays per tile only with one tile per ray");
  static_assert(!TWO || PAD, "16-slot rays are the padded form");   // PAD = false: S == SP exactly (the plain index arithmetic)
  static_assert(WIDE != 3 || (KR > 0 && !TWO), "per-tile object rows need a whole ray per tile group");
  constexpr int L8_BL = l8_bl(GEO), L8_BR = l8_br(GEO), L8_CHAIN = l8_chain(WIDE, GEO), C8_BYTES = c8_bytes(WIDE),
                K8_SMALL_BYTES = k8_small(WIDE);
  constexpr bool BROWS_LDS = brows_in_lds8(WIDE, GEO);
  constexpr int SUMS_SCRATCH = 25 * HALF_SUMS_STRIDE * 4;   // per wave, behind the loop (half_sums_lds), in the dead operand fragments
  static_assert((NCH + NDW) * SUMS_SCRATCH <= PK_BYTES, "scratch of the closing sums");
  // PEDW: the PE backward (66 cosines and ~300 multiply-adds per sample, pure VALU work on the critical chain wave) runs on the
  // chain wave's dW partner -- the wave that shares its SIMD and idles through the forward phase.  d e2 / d e1 travel lane to
  // lane as f16 through LDS that is free at that point (the E2 image after step VD, three slot images after step XYZ); the
  // partner needs a barrier between its reads and the chain wave's next writes there, which the composite exchange of rays
  // that span tiles (KR > 1) provides.  Measured bound (PE backward deleted): 40.4 -> 36.9 us at 2048 x 64, 259 -> 228 at 8192 x 128.
  constexpr bool PEDW = KR > 1;
  constexpr int SP = TWO ? 16 : (KR > 0 ? 32 * KR : 32);   // padded sample slots per ray (one-launch form)
  constexpr int NCHW = NCH, NTHR = (NCH + NDW) * 64, NACC = 5, LI_RS = local8<NDW>(BK_RS), LI_RS2 = local8<NDW>(BK_RS2);
  constexpr int RS_ROWS = WIDE == 2 ? cnr_rec::ROWS_MAX : WIDE == 1 ? 7 : 4;  // most object rows this instance takes (WIDE = 3: unused)
  constexpr bool ROWTILE = WIDE == 3;
  // per chain wave: E1 image, E2 image, dPre / input slot, row one-hot table (the flush reuses it for the wave's
  // partial sums)
  // The dPre / input images are double-buffered (K_PAR apart): layer step k uses copy k & 1, so the chain wave stages step
  // k + 1 while the dW waves still read step k -- ONE workgroup barrier per layer step ("images of step k are complete";
  // copy (k + 1) & 1 was last read in step k - 1, which the dW waves finished before they arrived at that barrier).
  constexpr int K_E1 = 0, K_E2 = E1IMG8, K_D = K_E2 + E2IMG8, K_X = K_D + HSIMG_BYTES, K_PAR = 2 * HSIMG_BYTES,
                K_SMALL = K_X + HSIMG_BYTES + K_PAR, K_BYTES = C8_BYTES;
  static_assert(K_BYTES == K_SMALL + K8_SMALL_BYTES && K8_SMALL_BYTES >= 512 && K8_SMALL_BYTES >= (RS_ROWS + 1) * 64, "layout");
  static_assert(NACC >= 4, "the chain role parks its partial sums in accumulators 0..3");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int c = blockIdx.y;
  const int lane = threadIdx.x & 63, h = lane >> 5, col = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  P8ITER_DECL();
  P8PHASE(0);
  const bool is_chain = wv < NCHW;
  const int dwid = wv - NCHW;
  // operands of the class -> LDS (weight fragments + constants, direction matrix, bias rows), by every thread.  Called inside the
  // role branches: the chain waves first put their own first global requests in flight (first tile's inputs, step state),
  // whose round trips then run under this copy instead of after it (~2 us of every workgroup at 2048 x 64)
  // SPLIT (rays that span tiles, KR > 1): the transposed (backward) fragments -- 30 of the 81 KB, first needed ~9 k cycles into
  // the first iteration -- are brought in by the dW waves alone, AFTER the workgroup barrier that releases the chain waves into
  // their first forward (the dW waves idle through it); the iteration's composite-exchange barrier, which both roles pass before
  // a chain wave's first read of a transposed fragment, orders the two.  Stamped at 2048 x 64: the copy was 5.4 k of a
  // workgroup's 100 k cycles.
  constexpr bool SPLIT_COPY = KR > 1;
  auto copy_operands = [&](auto&& under_the_loads) {
    // 61 KB (+ 20 KB of residual fragments) per workgroup: EVERY 16-byte load of a thread is issued before its first LDS store
    // (one memory round trip; written as a load -> store loop the copy was eight dependent round trips: 7.5 k cycles of every
    // workgroup, stamped).  under_the_loads(): the caller's own dependent requests, issued while these are in flight.
    constexpr int FIRST = SPLIT_COPY ? PK_OFF_BWD : PK_BYTES;     // bytes of the packed image every thread helps copy
    constexpr int NV = (FIRST / 16 + NTHR - 1) / NTHR, NVL = GEO ? (PK_LO_BYTES / 16 + NTHR - 1) / NTHR : 0;
    const f4* src = reinterpret_cast<const f4*>(packed + (size_t)c * PK_BYTES);
    f4 v[NV], vl[NVL > 0 ? NVL : 1];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = threadIdx.x + k * NTHR;
      v[k] = i < FIRST / 16 ? src[i] : f4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (GEO) {   // residual fragments of the geometry branch behind the packed image
      const f4* lsrc = reinterpret_cast<const f4*>(packed_lo + (size_t)c * PK_LO_BYTES);
#pragma unroll
      for (int k = 0; k < NVL; ++k) {
        const int i = threadIdx.x + k * NTHR;
        vl[k] = i < PK_LO_BYTES / 16 ? lsrc[i] : f4{0.f, 0.f, 0.f, 0.f};
      }
    }
    float* Bl = reinterpret_cast<float*>(smem + L8_BL);
    for (int i = threadIdx.x; i < 66; i += NTHR) {
      const int hh = i / 33, k = i % 33, d = k / 3;
      Bl[i] = (hh == 1 && d == 10) ? 0.0f : Bdir[(size_t)c * B_stride + (11 * hh + d) * 3 + (k % 3)];
    }
    if constexpr (BROWS_LDS) {
      float* br = reinterpret_cast<float*>(smem + L8_BR);  // host guarantees 1 <= rows_per_class <= RS_ROWS
      for (int i = threadIdx.x; i < rows_per_class * 128; i += NTHR) br[i] = biasrows[(size_t)c * rows_per_class * 128 + i];
    }
    under_the_loads();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = threadIdx.x + k * NTHR;
      if (i < FIRST / 16) reinterpret_cast<f4*>(smem)[i] = v[k];
    }
    if constexpr (GEO) {
#pragma unroll
      for (int k = 0; k < NVL; ++k) {
        const int i = threadIdx.x + k * NTHR;
        if (i < PK_LO_BYTES / 16) reinterpret_cast<f4*>(smem + l8_lo())[i] = vl[k];
      }
    }
    __syncthreads();
    P8PHASE(1);
  };
  // (dW waves, SPLIT_COPY: the transposed fragments, behind the barrier above)
  auto copy_transposed = [&]() {
    constexpr int NDT = NDW * 64, NB = ((PK_BYTES - PK_OFF_BWD) / 16 + NDT - 1) / NDT;
    const f4* src = reinterpret_cast<const f4*>(packed + (size_t)c * PK_BYTES + PK_OFF_BWD);
    const int t = threadIdx.x - NCH * 64;
    f4 v[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int i = t + k * NDT;
      v[k] = i < (PK_BYTES - PK_OFF_BWD) / 16 ? src[i] : f4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int i = t + k * NDT;
      if (i < (PK_BYTES - PK_OFF_BWD) / 16) reinterpret_cast<f4*>(smem + PK_OFF_BWD)[i] = v[k];
    }
  };
  const float* cf = reinterpret_cast<const float*>(smem + PK_OFF_CONST);
  const unsigned char* bwf = smem + PK_OFF_BWD;
  unsigned char* chain_base = smem + L8_CHAIN;
  const float inv_gs = 1.0f / gscale;
  const int slot_inv = (65536 + S - 1) / S;
  const int ntiles = KR > 0 ? (int)(((int64_t)R * SP + 31) / 32) : (N + 31) / 32;
  const int tile_step = gridDim.x * NCHW;

  // dW waves: up to 5 accumulator blocks each.  Chain waves never touch them; their own persistent per-lane partial
  // sums (dB 33 floats, d w_sigma 16, d b_sigma) are locals of the chain branch, so that branch pays for 50 registers,
  // not for the dW role's 80.
  f16v Wacc[NACC];
#pragma unroll
  for (int b = 0; b < NACC; ++b) Wacc[b] = zero16();
  // PEDW: the PE backward from the f16 copies of d e1 / d e2 a chain wave left in LDS (dq[2 b + reg / 8][reg % 8] = de[b][reg],
  // dq[6 ..] likewise for d e2), for the lane's own sample at (t0x, t1x, t2x); both roles run it (the dW partner for every tile
  // but a workgroup's last, the chain wave itself for that one, while the dW waves write the record)
  auto pe_backward_h = [&](const h8 (&dq)[9], float t0x, float t1x, float t2x, float (&dbacc)[33]) {
    const float* Blh = reinterpret_cast<const float*>(smem + L8_BL) + 33 * h;
    float pd[11], gpa[11];
#pragma unroll
    for (int d = 0; d < 11; ++d) {
      pd[d] = Blh[3 * d] * t0x + Blh[3 * d + 1] * t1x + Blh[3 * d + 2] * t2x;
      gpa[d] = 0.0f;
    }
    // cos(pi 2^b p) for the six bands from ONE hardware cosine per direction and the double-angle recurrence c' = 2 c^2 - 1: two
    // full-rate operations per band instead of an argument product and a quarter-rate v_cos (the recurrence doubles the error per
    // band: 32 x the hardware cosine's 1e-6 at band 5, three orders under the f16 gradient it multiplies).  The band's factor
    // pi 2^b rides on the cosine, the f16 gradient enters the multiply-add directly (v_fma_mix_f32): four instructions per
    // (band, direction) where the plain form had eight issue slots.
#pragma unroll
    for (int d = 0; d < 11; ++d) {
      float cs = __builtin_amdgcn_cosf(pd[d] * 0.5f);
#pragma unroll
      for (int band = 0; band < 6; ++band) {
        const int q = 11 * band + d, k = q < 44 ? q : 48 + (q - 44);   // q < 44: d e1 (bands 0..3), then d e2 (bands 4, 5)
        gpa[d] = fmaf((float)dq[k >> 3][k & 7], cs * (3.14159265358979f * (float)(1 << band)), gpa[d]);
        if (band < 5) cs = fmaf(cs + cs, cs, -1.0f);
      }
    }
#pragma unroll
    for (int d = 0; d < 11; ++d) {
      dbacc[3 * d + 0] = fmaf(gpa[d], t0x, dbacc[3 * d + 0]);
      dbacc[3 * d + 1] = fmaf(gpa[d], t1x, dbacc[3 * d + 1]);
      dbacc[3 * d + 2] = fmaf(gpa[d], t2x, dbacc[3 * d + 2]);
    }
  };
#define DBACC(i) c_dbacc[i]
#define DWS(i) c_dws[i]
#define DBS c_dbs

  if (is_chain) {
    // ===================================================================================================
    // chain role: one tile per iteration
    // ===================================================================================================
    float c_dbacc[33], c_dws[16], c_dbs = 0.0f;
    if constexpr (PEDW) __builtin_amdgcn_s_setprio(1);   // above its dW partner's PE backward (priority 0), below the partner's layer steps (3)
#pragma unroll
    for (int i = 0; i < 33; ++i) c_dbacc[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) c_dws[i] = 0.0f;
    unsigned char* cw = chain_base + wv * K_BYTES;
    unsigned char* E1img = cw + K_E1;
    unsigned char* E2img = cw + K_E2;
    unsigned char* Dimg0 = cw + K_D;            // layer steps 0, 2, .. (rgb.2, texture_1, enc_shape, cat, xyz)
    unsigned char* Ximg0 = cw + K_X;
    unsigned char* Dimg1 = cw + K_D + K_PAR;    // layer steps 1, 3, .. (rgb.0, viewdir, shape_2, shape_1)
    unsigned char* Ximg1 = cw + K_X + K_PAR;
    // [rows_per_class + 1][32] (<= 8 rows): (object row of sample k == r), last row ones
    _Float16* rowoh = reinterpret_cast<_Float16*>(cw + K_SMALL);
    const float* Bl_h = reinterpret_cast<const float*>(smem + L8_BL) + 33 * h;

    // one lane's inputs of a tile; past the end: a dead tile (all-zero gradients, any valid row)
    struct TileIn { float px, py, pz, dsg, dr0, dr1, dr2; int row; float z, gtd, g0, g1, g2; int lab, dm, live, rlive; };
    auto fetch = [&](int tile) {
      const bool tile_ok = tile < ntiles;
      const int tl = tile_ok ? tile : ntiles - 1;
      const int n0 = tl * 32;
      TileIn t;
      if constexpr (KR > 0) {
        int64_t ray, gs;
        bool ray_ok = tile_ok, lane_ok = tile_ok;
        if constexpr (PAD) {
          // padded slots: lane col of tile tl is slot 32 tl + col = sample slot % SP of ray slot / SP; samples >= S and rays
          // >= R are dead lanes (they run the forward on a clamped address and contribute nothing)
          const int slot = n0 + col, ray_l = slot / SP, sidx = slot % SP;
          ray_ok = tile_ok && ray_l < R; lane_ok = ray_ok && sidx < S;
          ray = (int64_t)c * R + (ray_l < R ? ray_l : R - 1);
          gs = ray * S + (sidx < S ? sidx : S - 1);
        } else {  // S = 32 KR: tile tl is samples [32 (tl % KR), + 32) of ray tl / KR, never partial
          ray = (int64_t)c * R + tl / KR;
          gs = (int64_t)c * N + n0 + col;
        }
        const float* pp = pts + gs * 3;
        t.px = pp[0]; t.py = pp[1]; t.pz = pp[2];
        t.row = ray_row ? ray_row[ray] : (int)ray;
        t.live = lane_ok ? 1 : 0; t.rlive = PAD ? (ray_ok ? 1 : 0) : 0;
        // the render-side inputs (z, targets, label) are first used ~4 k cycles into the iteration, by the composite: they
        // are requested at the START of their own iteration (render_inputs), not with the prefetch three steps before the
        // previous one ends -- seven registers less while the d e1 accumulators (48) are live, which is where the kernel spilled
        t.z = t.gtd = t.g0 = t.g1 = t.g2 = 0.0f; t.lab = t.dm = 0;
        t.dsg = t.dr0 = t.dr1 = t.dr2 = 0.0f;
        return t;
      }
      const int ray0 = (int)((unsigned)n0 / (unsigned)S);
      const int kk = n0 - ray0 * S + col;
      const int sl = (kk * slot_inv) >> 16;  // kk / S for kk < S + 32, S <= 240
      const bool live = tile_ok && n0 + col < N;
      const int nc = n0 + col < N ? n0 + col : N - 1;
      const int rayc = n0 + col < N ? ray0 + sl : R - 1;
      const int64_t gs = (int64_t)c * N + nc;
      const float* pp = pts + gs * 3;
      const int64_t ray = (int64_t)c * R + rayc;
      t.px = pp[0]; t.py = pp[1]; t.pz = pp[2];
      t.row = ray_row ? ray_row[ray] : (int)ray;
      t.dsg = live ? d_sigma[gs] : 0.0f;
      t.dr0 = live ? d_rgb[gs * 3 + 0] : 0.0f;
      t.dr1 = live ? d_rgb[gs * 3 + 1] : 0.0f;
      t.dr2 = live ? d_rgb[gs * 3 + 2] : 0.0f;
      t.z = t.gtd = t.g0 = t.g1 = t.g2 = 0.0f; t.lab = t.dm = 0; t.live = live ? 1 : 0; t.rlive = t.live;
      return t;
    };
    auto render_inputs = [&](TileIn& t, int tile) {   // KR > 0: same index arithmetic as fetch
      const int tl = tile < ntiles ? tile : ntiles - 1;
      const int n0 = tl * 32;
      int64_t ray, gs;
      if constexpr (PAD) {
        const int slot = n0 + col, ray_l = slot / SP, sidx = slot % SP;
        ray = (int64_t)c * R + (ray_l < R ? ray_l : R - 1);
        gs = ray * S + (sidx < S ? sidx : S - 1);
      } else {
        ray = (int64_t)c * R + tl / (KR > 0 ? KR : 1);
        gs = (int64_t)c * N + n0 + col;
      }
      t.z = ta.z[gs]; t.gtd = ta.gt_depth[ray];
      t.g0 = ta.gt_rgb[ray * 3 + 0]; t.g1 = ta.gt_rgb[ray * 3 + 1]; t.g2 = ta.gt_rgb[ray * 3 + 2];
      t.lab = ta.labels[ray]; t.dm = ta.depth_mask[ray];
    };
    TileIn cur = fetch(blockIdx.x * NCHW + wv), nxt = cur;     // in flight under the operand copy
    int64_t cursor0 = 0;
    if constexpr (KR > 0) cursor0 = ta.d_state ? ta.d_state[0] : 0;
    // ---- one-launch step: this class's loss weights from the epoch's mask-count table.  The entry's address depends on the
    // cursor (requested above): its six loads go out while the operand copy's are in flight, not behind the copy's barrier (a
    // dependent round trip of ~2.5 k cycles in front of every workgroup's first tile, stamped)
    float tbv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    copy_operands([&]() {
      if constexpr (KR > 0) {
        const int Cn = gridDim.y;
        const float* tb = ta.counts_tab + (size_t)(cursor0 / R) * (size_t)(Cn + 1) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) { tbv[k] = tb[Cn * 4 + k]; tbv[3 + k] = tb[c * 4 + k]; }
      }
    });
    float wd_c = 0.f, wc_c = 0.f, wo_c = 0.f, ld_acc = 0.f, lc_acc = 0.f, lo_acc = 0.f;
    int tab_flags = 0;
    if constexpr (KR > 0) {
      const bool e_d = tbv[0] != 0.f, e_c = tbv[1] != 0.f, e_o = tbv[2] != 0.f;
      wd_c = e_d ? 0.f : 1.0f / (tbv[3] + 1e-10f);
      wc_c = e_c ? 0.f : 1.0f / (tbv[4] + 1e-10f);
      wo_c = e_o ? 0.f : 1.0f / (tbv[5] + 1e-10f);
      tab_flags = (e_d ? 2 : 0) | (e_c ? 4 : 0) | (e_o ? 8 : 0);
    }
    float* xch = reinterpret_cast<float*>(smem + l8_xch(NCH, WIDE, GEO));   // [4][8]
    bool any_iter = false;
    float l0x = 0.f, l1x = 0.f, l2x = 0.f;   // PEDW: the last tile's sample position (its PE backward runs here, behind the loop)
    for (int tile = blockIdx.x * NCHW + wv, t0 = blockIdx.x * NCHW; t0 < ntiles; t0 += tile_step, tile += tile_step) {
      asm volatile("" ::: "memory");
      P8STAMP_RESET();
      P8STAMP();
      // ---- this lane's sample (fetched during the previous iteration's shape_layer_2 step) -------------------
      if constexpr (KR > 0) render_inputs(cur, tile);
      const float t0x = cur.px * inv_scale, t1x = cur.py * inv_scale, t2x = cur.pz * inv_scale;
      if constexpr (PEDW) { l0x = t0x; l1x = t1x; l2x = t2x; }
      const int row = cur.row;
      // upstream gradients: given (KR == 0) or formed after this tile's forward by the composite / loss block below
      float draw = 0.0f, dr0 = 0.0f, dr1 = 0.0f, dr2 = 0.0f;
      auto scale_dsigma = [&](float dsg) {
        const float dsg_s = dsg * gscale;
        // the f16 data-gradient chain needs |d sigma| * gscale <= 8192; a hit (a ray whose termination is one sample:
        // var -> 0, info -> 1e4) is clipped and REPORTED: bit 4 of the step's flags (cnr_step_tail or-s the word in)
        if (clamp_flags && fabsf(dsg_s) > 8192.0f) atomicOr(clamp_flags + c, 16);
        return fminf(fmaxf(dsg_s, -8192.0f), 8192.0f) * 10.0f;  // sigmas = raw * 10 (src/model.py:75)
      };
      if constexpr (KR == 0) {
        draw = scale_dsigma(cur.dsg);
        dr0 = cur.dr0 * gscale; dr1 = cur.dr1 * gscale; dr2 = cur.dr2 * gscale;
      }
      const float* brow_l = BROWS_LDS ? reinterpret_cast<const float*>(smem + L8_BR) + (row - c * rows_per_class) * 128
                                      : biasrows + (size_t)row * 128;

      auto pe_backward = [&](const f16v (&de)[3], int nblk, int band0, int nq) {
        float pd[11], gpa[11];
#pragma unroll
        for (int d = 0; d < 11; ++d) {
          pd[d] = Bl_h[3 * d] * t0x + Bl_h[3 * d + 1] * t1x + Bl_h[3 * d + 2] * t2x;
          gpa[d] = 0.0f;
        }
        // (the cosines as in pe_backward_h: one hardware cosine per direction at the first band of the call, double-angle upward)
#pragma unroll
        for (int d = 0; d < 11; ++d) {
          float cs = __builtin_amdgcn_cosf(pd[d] * (0.5f * (float)(1 << band0)));
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) {
            const int q = 11 * bb + d;
            if (q < nq && (q >> 4) < nblk) {
              gpa[d] = fmaf(de[q >> 4][q & 15], cs * (3.14159265358979f * (float)(1 << (band0 + bb))), gpa[d]);
              cs = fmaf(cs + cs, cs, -1.0f);
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
      // (the previous iteration's last barrier has passed: every image of this wave is free)
      // ---- positional encoding, encoding_xyz and the e1 part of cat_layer (its own accumulator, started from the cat bias row) as a
      // SOFTWARE PIPELINE over the six E1 fragments: stage s requests the weight fragments of s + 1, encodes fragment s + 1 (eight
      // sines, conversions, residuals) and issues the (GEO: six) products on fragment s -- whose operands were requested and
      // encoded one stage earlier.  The vector unit then encodes while the matrix core multiplies, and no product waits for an
      // LDS read.  As "all 66 sines, then 36 products" the two units took turns: stamped ~3.3 k cycles of every tile.
      // GEO: three products per fragment and accumulator, Wh xh + Wh xl + Wl xh (fused_common.h; fused_fwd.hip runs the same order:
      // the two forwards agree bit for bit); wl* = the residual fragments, el = the residual of the features.
      const unsigned char* lo_w = smem + l8_lo();
      h8 wq[8], wl[GEO ? 6 : 1];
      h8 E2f[3];
      f16v acc, bq, catp;
      float pdv[11];
      {
        float Bh[33];
#pragma unroll
        for (int i = 0; i < 33; ++i) Bh[i] = Bl_h[i];
        pe_dirs(Bh, t0x, t1x, t2x, pdv);
      }
      struct E1Stage { h8 e, el, wx, wc, wlx, wlc; };
      auto e1_request = [&](int s, E1Stage& st) {
        st.wx = lds_frag(smem, KK_XYZ + s, lane); st.wc = lds_frag(smem, KK_CAT + 2 + s, lane);
        if constexpr (GEO) { st.wlx = lds_frag(lo_w, KK_XYZ + s, lane); st.wlc = lds_frag(lo_w, KK_CAT + 2 + s, lane); }
      };
      auto e1_encode = [&](int s, E1Stage& st) {
        float v[8];
        pe_e1_slots<true>(pdv, t0x, t1x, t2x, h, s, v);
        st.e = pack8f(v);
        if constexpr (GEO) st.el = pack8f_lo(v, st.e);
      };
      E1Stage stg[2];
      e1_request(0, stg[0]);
      e1_encode(0, stg[0]);
      acc = acc_init(cf + CF_B_XYZ, h);
      catp = acc_init(brow_l + 1 * 32, h);
      if (any_iter) P8SYNC();   // the previous iteration's last barrier: the dW waves are done with this wave's images
      if constexpr (!ROWTILE) {
        const int rl = row - c * rows_per_class;
        constexpr int rs = RS_ROWS;  // rows per latent slot in the row-sum block (4 / 7 / 15 by WIDE); the ones row follows
#pragma unroll
        for (int r = 0; r < RS_ROWS; r += 2)  // lane half h writes rows h, h + 2, ...
          if (r + h < rs) rowoh[(r + h) * 32 + col] = rl == r + h ? (_Float16)1 : (_Float16)0;
        if (h == 0) rowoh[rs * 32 + col] = (_Float16)1;
      }
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        E1Stage& cu = stg[s & 1];
        E1Stage& nx = stg[(s + 1) & 1];
        if (s + 1 < 6) { e1_request(s + 1, nx); e1_encode(s + 1, nx); }
        *reinterpret_cast<h8*>(E1img + col * ST_E1 + h * 96 + 16 * s) = cu.e;
        acc = MFMA(cu.wx, cu.e, acc);
        catp = MFMA(cu.wc, cu.e, catp);
        if constexpr (GEO) {
          acc = MFMA(cu.wx, cu.el, acc);
          catp = MFMA(cu.wc, cu.el, catp);
          acc = MFMA(cu.wlx, cu.e, acc);
          catp = MFMA(cu.wlc, cu.e, catp);
        }
      }
      {   // the view-direction features (bands 4, 5): used by encoding_viewdir, behind the geometry branch
        float v2[24];
        pe_e2_slots<true>(pdv, h, v2);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          E2f[s] = pack8f(&v2[8 * s]);
          *reinterpret_cast<h8*>(E2img + col * ST_E2 + h * 48 + 16 * s) = E2f[s];
        }
      }
      P8MARK(8);
      P8ISA("pe_forward_done");
      P8MARK(9);
      // a 32-wide hidden layer on the input pair (xa, xb) = f16 of the previous accumulators; GEO: + the two residual products
      auto hidden = [&](const h8& xa, const h8& xb, const f16v& prev, bool relu, const f16v& init) {
        f16v o = MFMA(wq[0], xa, init);
        o = MFMA(wq[1], xb, o);
        if constexpr (GEO) {
          o = MFMA(wl[0], xa, o);
          o = MFMA(wl[1], xb, o);
          const h8 la = pack8_lo(prev, 0, relu, xa), lb = pack8_lo(prev, 1, relu, xb);
          o = MFMA(wq[0], la, o);
          o = MFMA(wq[1], lb, o);
        }
        return o;
      };
      P8ISA("xyz_and_cat_e1_done");
      wq[0] = lds_frag(smem, KK_S1 + 0, lane); wq[1] = lds_frag(smem, KK_S1 + 1, lane);
      if constexpr (GEO) { wl[0] = lds_frag(lo_w, KK_S1 + 0, lane); wl[1] = lds_frag(lo_w, KK_S1 + 1, lane); }
      bq = acc_init(brow_l + 0 * 32, h);
      const h8 A0a = GEO ? pack8_relu32(acc, 0) : pack8(acc, 0, true), A0b = GEO ? pack8_relu32(acc, 1) : pack8(acc, 1, true);
      acc = hidden(A0a, A0b, acc, true, bq);
      wq[0] = lds_frag(smem, KK_CAT + 0, lane); wq[1] = lds_frag(smem, KK_CAT + 1, lane);
      if constexpr (GEO) { wl[0] = lds_frag(lo_w, KK_CAT + 0, lane); wl[1] = lds_frag(lo_w, KK_CAT + 1, lane); }
      const h8 A1a = GEO ? pack8_relu32(acc, 0) : pack8(acc, 0, true), A1b = GEO ? pack8_relu32(acc, 1) : pack8(acc, 1, true);
      acc = hidden(A1a, A1b, acc, true, catp);
      wq[0] = lds_frag(smem, KK_S2 + 0, lane); wq[1] = lds_frag(smem, KK_S2 + 1, lane);
      if constexpr (GEO) { wl[0] = lds_frag(lo_w, KK_S2 + 0, lane); wl[1] = lds_frag(lo_w, KK_S2 + 1, lane); }
      bq = acc_init(brow_l + 2 * 32, h);
      const h8 A2a = GEO ? pack8_relu32(acc, 0) : pack8(acc, 0, true), A2b = GEO ? pack8_relu32(acc, 1) : pack8(acc, 1, true);
      acc = hidden(A2a, A2b, acc, true, bq);
      wq[0] = lds_frag(smem, KK_ES + 0, lane); wq[1] = lds_frag(smem, KK_ES + 1, lane);
      if constexpr (GEO) { wl[0] = lds_frag(lo_w, KK_ES + 0, lane); wl[1] = lds_frag(lo_w, KK_ES + 1, lane); }
      bq = acc_init(cf + CF_B_ES, h);
      const h8 A3a = GEO ? pack8_relu32(acc, 0) : pack8(acc, 0, true), A3b = GEO ? pack8_relu32(acc, 1) : pack8(acc, 1, true);
      acc = hidden(A3a, A3b, acc, true, bq);
      P8MARK(10);
      P8ISA("geometry_hidden_layers_done");
#pragma unroll
      for (int s = 0; s < 5; ++s) wq[s] = lds_frag(smem, KK_VD + s, lane);
      bq = acc_init(cf + CF_B_VD, h);
      f16v y4keep;        // KR > 0: y4 (fp32) until the loss gradient exists
      float raw = 0.0f;   // KR > 0: the sigma logit before the x10, fp32 VALU dot product as in fused_fwd.hip
      if constexpr (KR == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) DWS(i) = fmaf(draw, acc[i], DWS(i));  // d w_sigma += draw * y4
        DBS += (h == 0) ? draw : 0.0f;
      } else {
        y4keep = acc;
        const f16v ws = acc_init(cf + CF_W_SG, h);
        float part = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) part = fmaf(ws[i], acc[i], part);
        raw = part + __shfl_xor(part, 32, 64) + cf[CF_B_SG];
      }
      // the sigma-only half of the composite starts HERE, ahead of the colour branch: its scans are a dependent VALU chain
      // that the scheduler can run under the ten MFMAs (and their result latencies) of the colour layers
      float c_occ = 0.f, c_f = 1.f, c_excl = 1.f, c_Pt = 1.f, c_tl = 0.f, c_wl = 0.f, c_dl = 0.f, c_ml = 0.f, c_M2 = 0.f;
      if constexpr (KR > 0) {
        c_occ = (h == 0 && cur.live != 0) ? __builtin_amdgcn_rcpf(1.0f + expf(-(raw * 10.0f))) : 0.0f;
        c_f = h == 0 ? (1.0f - c_occ + 1e-10f) : 1.0f;
        const float incl = seg_scan_mul<TWO>(c_f);                   // inclusive product of the free probabilities
        c_excl = lane_below(incl, 1.0f);                             // exclusive (first lane of a ray: 1)
        if (TWO && col == 16) c_excl = 1.0f;
        c_Pt = lane_value(incl, 31);                                 // the tile's product (rays spanning tiles)
        c_tl = c_occ * c_excl;                                       // termination with the carry still to come
        c_wl = seg_total<TWO>(seg_scan_add<TWO>(c_tl), lane);
        c_dl = seg_total<TWO>(seg_scan_add<TWO>(c_tl * cur.z), lane);
        c_ml = c_wl > 0.0f ? c_dl * __builtin_amdgcn_rcpf(c_wl) : 0.0f;
        const float dzl = cur.z - c_ml;
        c_M2 = seg_total<TWO>(seg_scan_add<TWO>(c_tl * dzl * dzl), lane);
      }
      P8MARK(11);
      P8ISA("sigma_head_and_sigma_half_of_composite_done");
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
      if constexpr (!SPLIT_COPY) wq[1] = lds_frag(bwf, KT_R2, lane);
      bq = acc_init(cf + CF_B_R2, h);
      const h8 A7a = pack8(acc, 0, true);
      acc = MFMA(wq[0], A7a, bq);
      h8 Wn0, Wn1;
      if constexpr (!SPLIT_COPY) Wn0 = lds_frag(bwf, KT_R0, lane);   // (SPLIT_COPY: behind the exchange barrier, see copy_operands)

      P8MARK(7);
      P8ISA("colour_branch_done");
      // rgb = sigmoid(logits in rows 0..2 = registers 0..2 of lane half 0)
      // (v_rcp_f32, 1 ulp, instead of an IEEE division: ten dependent instructions less per quotient, and this stretch of
      //  the iteration runs as one dependent chain with the SIMD's other wave idle)
      const float r0 = __builtin_amdgcn_rcpf(1.0f + __expf(-acc[0])), r1 = __builtin_amdgcn_rcpf(1.0f + __expf(-acc[1])),
                  r2 = __builtin_amdgcn_rcpf(1.0f + __expf(-acc[2]));
      if constexpr (KR > 0) {
        // ================= composite + losses + their gradient + composite backward (a11-a15) ==========================
        // lane half 0 = the tile's 32 samples (half 1 neutral); the ray's KR tiles sit in chain waves wv0 .. wv0 + KR - 1
        // and exchange tile products / sums through LDS behind workgroup barriers (the dW waves take part in them).
        // Same expressions as field_fwd_render_kernel (fused_fwd.hip), which holds the whole ray in one wave.
        constexpr int wv_mask = KR - 1;
        const int wv0 = wv & ~wv_mask, tin = wv & wv_mask;
        const bool live = cur.live != 0;
        const float occ = c_occ, zz = cur.z, f = c_f, excl = c_excl, Pt = c_Pt, tl_ = c_tl;
        // the tile's own sums (lane 31 of the DPP scans holds the half-0 totals; half 1 is zero); the sigma-only ones
        // (w_l, d_l and the pieces of the variance) were formed ahead of the colour branch
        const float w_l = c_wl, d_l = c_dl, m_l = c_ml, M2_l = c_M2;
        const float r_l = seg_total<TWO>(seg_scan_add<TWO>(tl_ * r0), lane), g_l = seg_total<TWO>(seg_scan_add<TWO>(tl_ * r1), lane);
        const float b_l = seg_total<TWO>(seg_scan_add<TWO>(tl_ * r2), lane);
        // var = sum term (z - depth)^2 needs the ray's depth first.  Tiles exchange ONCE: each publishes its weighted mean
        // m_l and M2_l = sum tl (z - m_l)^2; with mean = depth / opacity over the ray,
        //   var = sum_t c_t (M2_t + w_t (m_t - mean)^2) + opacity (mean - depth)^2      (exact; c_t = carried transmittance)
        float carry = 1.0f, sd = d_l, so = w_l, sr = r_l, sg = g_l, sb = b_l, sv;
        // the ray's tiles: {P, w, d, r} {g, b, M2, m} per tile, kept in registers for the three passes below (re-reading the
        // four tiles of a 128-slot ray from LDS in every pass instead was measured 2 % slower at 8192 x 128)
        constexpr bool XREG = KR >= 2;
        f4 xa[XREG ? KR : 1], xb[XREG ? KR : 1];
        float cpre[KR > 1 ? KR : 1];                    // carried transmittance in front of tile t
        auto XA = [&](int t) { return XREG ? xa[XREG ? t : 0] : reinterpret_cast<const f4*>(xch + (wv0 + t) * 8)[0]; };
        auto XB = [&](int t) { return XREG ? xb[XREG ? t : 0] : reinterpret_cast<const f4*>(xch + (wv0 + t) * 8)[1]; };
        if constexpr (KR > 1) {
          if (lane == 0) {
            f4* x = reinterpret_cast<f4*>(xch + wv * 8);
            x[0] = f4{Pt, w_l, d_l, r_l}; x[1] = f4{g_l, b_l, M2_l, m_l};
          }
          P8SYNC();
          if constexpr (SPLIT_COPY) {   // the first transposed fragments of the backward: in LDS since this barrier at the latest
            wq[1] = lds_frag(bwf, KT_R2, lane);
            Wn0 = lds_frag(bwf, KT_R0, lane);
          }
          if constexpr (XREG) {
#pragma unroll
            for (int t = 0; t < KR; ++t) {
              const f4* x = reinterpret_cast<const f4*>(xch + (wv0 + t) * 8);
              xa[t] = x[0]; xb[t] = x[1];
            }
          }
          sd = so = sr = sg = sb = 0.0f;
          float run = 1.0f;
#pragma unroll
          for (int t = 0; t < KR; ++t) {
            const f4 a4 = XA(t), b4 = XB(t);
            cpre[t] = run;
            so += run * a4[1]; sd += run * a4[2]; sr += run * a4[3]; sg += run * b4[0]; sb += run * b4[1];
            run *= a4[0];
          }
#pragma unroll
          for (int t = 0; t < KR; ++t) carry = t == tin ? cpre[t] : carry;
        }
        P8MARK(0);
        P8ISA("composite_exchange_done");
        const float mean = so > 0.0f ? sd * __builtin_amdgcn_rcpf(so) : 0.0f;
        if constexpr (KR > 1) {
          float M2 = 0.0f;
#pragma unroll
          for (int t = 0; t < KR; ++t) {
            const f4 a4 = XA(t), b4 = XB(t);
            const float dm = b4[3] - mean;
            M2 += cpre[t] * (b4[2] + a4[1] * dm * dm);
          }
          sv = M2 + so * (mean - sd) * (mean - sd);
        } else {
          const float dm = m_l - mean;   // (= 0 up to rounding: one tile is the whole ray)
          sv = (M2_l + w_l * dm * dm) + so * (mean - sd) * (mean - sd);
        }
        P8MARK(1);
        const float T = carry * excl;
        const float term = occ * T;
        auto sgn = [](float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); };
        const bool mo = cur.lab != 0, ms = cur.lab != 2, md = (cur.dm != 0) && mo;
        const float fd = md ? 1.f : 0.f, fo = mo ? 1.f : 0.f, fs = ms ? 1.f : 0.f;
        const float rd = sd - cur.gtd;
        const float info = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(sv) + 1e-4f);
        const float rc0 = sr - cur.g0, rc1 = sg - cur.g1, rc2 = sb - cur.g2;
        const float ro = so - fo;
        if (tin == 0 && (PAD ? cur.rlive : cur.live)) {   // the ray's first tile accounts for it (uniform over its lanes)
          ld_acc += fabsf(rd) * fd * info;
          lc_acc += (fabsf(rc0) + fabsf(rc1) + fabsf(rc2)) * fo;
          lo_acc += fabsf(ro) * fs;
          if (lane == 0 || (TWO && lane == 16)) {
            const int ray = c * R + (PAD ? (tile * 32 + col) / SP : tile / KR);     // (C R < 2^31: checked by the host)
            if (ta.depth_out) ta.depth_out[ray] = sd;
            if (ta.var_out) ta.var_out[ray] = sv;
            if (ta.opacity_out) ta.opacity_out[ray] = so;
            if (ta.rgb_out) { ta.rgb_out[ray * 3 + 0] = sr; ta.rgb_out[ray * 3 + 1] = sg; ta.rgb_out[ray * 3 + 2] = sb; }
          }
        }
        const float dD = ta.loss_scale * sgn(rd) * fd * info * wd_c;
        const float dR = ta.loss_scale * ta.color_scaling * sgn(rc0) * fo * wc_c;
        const float dG = ta.loss_scale * ta.color_scaling * sgn(rc1) * fo * wc_c;
        const float dBl = ta.loss_scale * ta.color_scaling * sgn(rc2) * fo * wc_c;
        const float dO = ta.loss_scale * ta.opacity_scaling * sgn(ro) * fs * wo_c;
        P8MARK(2);
        P8ISA("loss_gradient_scalars_done");
        // composite backward: d occ_i = T_i g_i - (sum_{j > i} term_j g_j) / f_i.  The sum over LATER tiles needs no second
        // exchange: a tile's total of term g is linear in the five ray-level factors, c_t (dD d_t + dR r_t + .. + dO w_t)
        float suf_carry = 0.0f;
        if constexpr (KR > 1) {
#pragma unroll
          for (int t = 0; t < KR; ++t) {
            const f4 a4 = XA(t), b4 = XB(t);
            const float tot = cpre[t] * (dD * a4[2] + dR * a4[3] + dG * b4[0] + dBl * b4[1] + dO * a4[1]);
            suf_carry += t > tin ? tot : 0.0f;
          }
        }
        P8MARK(3);
        const float g = dD * zz + dR * r0 + dG * r1 + dBl * r2 + dO;
        const float tg = h == 0 ? term * g : 0.0f;
        const float isuf = seg_suffix_add<TWO>(tg, lane);
        const float suf = (isuf - tg) + suf_carry;
        const float docc = T * g - suf * __builtin_amdgcn_rcpf(f);
        float dsg = (h == 0 && live) ? docc * occ * (1.0f - occ) : 0.0f;
        P8MARK(4);
        dsg = low_half_to_both(dsg);                // both lane halves of a sample column need it (d y4 rows 4..7 mod 8)
        draw = scale_dsigma(dsg);
        dr0 = term * dR * gscale; dr1 = term * dG * gscale; dr2 = term * dBl * gscale;   // used in lane half 0 only
#pragma unroll
        for (int i = 0; i < 16; ++i) DWS(i) = fmaf(draw, y4keep[i], DWS(i));  // d w_sigma += draw * y4
        DBS += (h == 0) ? draw : 0.0f;
      }
      P8MARK(5);
      P8ISA("composite_backward_done");
      // ---- step R2: dPre9 = drgb * rgb (1 - rgb) in rows 0..2 (registers 0..2 of half 0)
      h8 D0 = zero8(), D1 = zero8();
      {
        if (h == 0) {
          D0[0] = (_Float16)(dr0 * r0 * (1.0f - r0));
          D0[1] = (_Float16)(dr1 * r1 * (1.0f - r1));
          D0[2] = (_Float16)(dr2 * r2 * (1.0f - r2));
        }
      }
      // staged into feature columns 16..18: dW0 accumulates rgb.2 into rows 16..18 of rgb.0's block
      stage_hs(Dimg0, D1, D0, col, h);
      {
        h8 one = zero8();
        if (h == 0) one[0] = (_Float16)1;  // feature 16 of the a7 image := 1 -> d b(rgb.2)
        stage_hs(Ximg0, A7a, one, col, h);
      }
      acc = MFMA(wq[1], D0, zero16());  // d a7 (rows 0..15)
      P8MARK(6);
      P8SYNC();                   // A(R2)
      D0 = pack8_masked(acc, 0, A7a); D1 = zero8();
      u4v Mn0 = relu_mask(A6a), Mn1 = relu_mask(A6b);
      acc = MFMA(Wn0, D0, zero16());  // d a6
      Wn0 = lds_frag(bwf, KT_T1 + 0, lane); Wn1 = lds_frag(bwf, KT_T1 + 1, lane);
      // ---- step R0
      stage_hs(Dimg1, D0, D1, col, h);
      stage_hs(Ximg1, A6a, A6b, col, h);
      P8SYNC();                   // A(R0)
      D0 = pack8_and(acc, 0, Mn0); D1 = pack8_and(acc, 1, Mn1);
      Mn0 = relu_mask(A5a); Mn1 = relu_mask(A5b);
      acc = MFMA(Wn0, D0, zero16());
      acc = MFMA(Wn1, D1, acc);  // d a5
      Wn0 = lds_frag(bwf, KT_VD_Y + 0, lane); Wn1 = lds_frag(bwf, KT_VD_Y + 1, lane);
      // ---- step T1
      stage_hs(Dimg0, D0, D1, col, h);
      stage_hs(Ximg0, A5a, A5b, col, h);
      P8SYNC();                   // A(T1)
      // (the extra fragments of a step -- d e2 here, d e1 in steps CAT and S1 -- are requested at the TOP of the step, under the
      //  packing of dPre: read in front of their products, each product waited out part of an LDS round trip)
      h8 wE[6];
#pragma unroll
      for (int k = 0; k < 4; ++k) wE[k] = lds_frag(bwf, KT_VD_E + k, lane);
      __builtin_amdgcn_sched_barrier(0);
      D0 = pack8_and(acc, 0, Mn0); D1 = pack8_and(acc, 1, Mn1);
      acc = MFMA(Wn0, D0, zero16());
      acc = MFMA(Wn1, D1, acc);  // d y4 from the colour branch
      f16v de2[3];  // d e2 (two 16-slot blocks)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        de2[b] = MFMA(wE[2 * b + 0], D0, zero16());
        de2[b] = MFMA(wE[2 * b + 1], D1, de2[b]);
      }
      Wn0 = lds_frag(bwf, KT_ES + 0, lane); Wn1 = lds_frag(bwf, KT_ES + 1, lane);
      // ---- step VD : inputs [y4 | e2]
      stage_hs(Dimg1, D0, D1, col, h);
      stage_hs(Ximg1, Y4a, Y4b, col, h);
      P8SYNC();                   // A(VD)
      {  // + sigma head: d y4 += w_sigma * draw
        const f16v wsg = acc_init(cf + CF_W_SG, h);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fmaf(wsg[i], draw, acc[i]);
      }
      D0 = pack8(acc, 0, false); D1 = pack8(acc, 1, false);
      Mn0 = relu_mask(A3a); Mn1 = relu_mask(A3b);
      acc = MFMA(Wn0, D0, zero16());
      acc = MFMA(Wn1, D1, acc);  // d a3
      Wn0 = lds_frag(bwf, KT_S2 + 0, lane); Wn1 = lds_frag(bwf, KT_S2 + 1, lane);
      de2[2] = de2[1];
      h8 de2h[3];   // PEDW: d e2 as f16, parked until the E2 image is free (after A(ES))
      if constexpr (PEDW) {
        de2h[0] = pack8(de2[0], 0, false); de2h[1] = pack8(de2[0], 1, false); de2h[2] = pack8(de2[1], 0, false);
      } else {
        pe_backward(de2, 2, 4, 22);  // bands 4 and 5 -> dB
      }
      // ---- step ES (no activation)
      stage_hs(Dimg0, D0, D1, col, h);
      stage_hs(Ximg0, A3a, A3b, col, h);
      P8SYNC();                   // A(ES)
      if constexpr (PEDW) {   // the dW waves have read the E2 image (step VD): it now carries d e2 to the partner, lane to lane
        unsigned char* eq = E2img + lane * 16;
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<h8*>(eq + k * 1024) = de2h[k];
      }
      D0 = pack8_and(acc, 0, Mn0); D1 = pack8_and(acc, 1, Mn1);
      Mn0 = relu_mask(A2a); Mn1 = relu_mask(A2b);
      acc = MFMA(Wn0, D0, zero16());
      acc = MFMA(Wn1, D1, acc);  // d a2
      Wn0 = lds_frag(bwf, KT_CAT_Y + 0, lane); Wn1 = lds_frag(bwf, KT_CAT_Y + 1, lane);
      // ---- step S2 : input a2
      stage_hs(Dimg1, D0, D1, col, h);
      stage_hs(Ximg1, A2a, A2b, col, h);
      P8SYNC();                   // A(S2)
      D0 = pack8_and(acc, 0, Mn0); D1 = pack8_and(acc, 1, Mn1);
      Mn0 = relu_mask(A1a); Mn1 = relu_mask(A1b);
      acc = MFMA(Wn0, D0, zero16());
      acc = MFMA(Wn1, D1, acc);  // d a1
      Wn0 = lds_frag(bwf, KT_S1 + 0, lane); Wn1 = lds_frag(bwf, KT_S1 + 1, lane);
      const h8 Dc0 = D0, Dc1 = D1;  // dPre(cat): its d e1 part is formed together with encoding_xyz's
      nxt = fetch(tile + tile_step);  // next iteration's inputs: their latency hides under the last three steps
      // ---- step CAT : inputs [a1 | e1]
      stage_hs(Dimg0, D0, D1, col, h);
      stage_hs(Ximg0, A1a, A1b, col, h);
      P8SYNC();                   // A(CAT)
#pragma unroll
      for (int k = 0; k < 6; ++k) wE[k] = lds_frag(bwf, KT_CAT_E + k, lane);
      __builtin_amdgcn_sched_barrier(0);
      D0 = pack8_and(acc, 0, Mn0); D1 = pack8_and(acc, 1, Mn1);
      Mn0 = relu_mask(A0a); Mn1 = relu_mask(A0b);
      acc = MFMA(Wn0, D0, zero16());
      acc = MFMA(Wn1, D1, acc);  // d a0
      f16v de[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        de[b] = MFMA(wE[2 * b + 0], Dc0, zero16());
        de[b] = MFMA(wE[2 * b + 1], Dc1, de[b]);
      }
      // ---- step S1 : input a0
      stage_hs(Dimg1, D0, D1, col, h);
      stage_hs(Ximg1, A0a, A0b, col, h);
      P8SYNC();                   // A(S1)
#pragma unroll
      for (int k = 0; k < 6; ++k) wE[k] = lds_frag(bwf, KT_XYZ_E + k, lane);
      __builtin_amdgcn_sched_barrier(0);
      D0 = pack8_and(acc, 0, Mn0); D1 = pack8_and(acc, 1, Mn1);
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        de[b] = MFMA(wE[2 * b + 0], D0, de[b]);
        de[b] = MFMA(wE[2 * b + 1], D1, de[b]);
      }
      // ---- step XYZ : input e1 (its image)
      stage_hs(Dimg0, D0, D1, col, h);
      P8SYNC();                   // A(XYZ)
      if constexpr (PEDW) {
        // d e1 -> this wave's dW partner, lane to lane, through the three slot images no step uses any more (Ximg0, Dimg1, Ximg1:
        // 6 KB = six 16-byte pieces per lane; the dW waves read step XYZ from Dimg0 and the E1 image).  The partner picks them
        // up behind the iteration's last barrier and has finished with them before the composite-exchange barrier of the next
        // iteration, in front of which this wave writes none of the three.
        unsigned char* xq = cw + K_X + lane * 16;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          *reinterpret_cast<h8*>(xq + (2 * b) * 1024) = pack8(de[b], 0, false);
          *reinterpret_cast<h8*>(xq + (2 * b + 1) * 1024) = pack8(de[b], 1, false);
        }
      } else {
        pe_backward(de, 3, 0, 44);        // d e1 -> dB, bands 0..3
      }
      cur = nxt;
      any_iter = true;   // the barrier "dW waves are done with this tile's images" follows at the next iteration's image
    }                    // writes (below the PE arithmetic, which so runs beside the dW waves' last step), or here:
    P8PHASE(3);
    if (any_iter) role_barrier();
    P8PHASE(4);
    if constexpr (PEDW) {
      if (any_iter) {   // the workgroup's last tile: its d e1 / d e2 are still where this wave put them
        h8 dq[9];
#pragma unroll
        for (int k = 0; k < 6; ++k) dq[k] = *reinterpret_cast<const h8*>(cw + K_X + lane * 16 + k * 1024);
#pragma unroll
        for (int k = 0; k < 3; ++k) dq[6 + k] = *reinterpret_cast<const h8*>(E2img + lane * 16 + k * 1024);
        pe_backward_h(dq, l0x, l1x, l2x, c_dbacc);
      }
    }
    P8PHASE(6);
    {  // publish this wave's partial sums (the last barrier has passed: the dW waves no longer read the row table
       // this aliases): [0..31] d w_sigma, [32] d b_sigma, [64..126] dB.  The 50 half-wave sums go through LDS (half_sums_lds; the
       // operand fragments at the front of the LDS are dead behind the loop's last barrier: a wave's scratch lies there), in two
       // batches of 25: stamped, 50 DPP reductions were 4.8 k cycles of every workgroup here.
      float* small = reinterpret_cast<float*>(cw + K_SMALL);
      float* scr = reinterpret_cast<float*>(smem + wv * SUMS_SCRATCH);
      auto put_db = [&](int i, int hh, float s) {   // dB entry i = 3 d + axis of lane half hh (direction 11 hh + d; 21 exist)
        const int d = i / 3;
        if (!(hh == 1 && d == 10)) small[64 + (11 * hh + d) * 3 + (i - 3 * d)] = s;
      };
      float va[25], vb[25];
#pragma unroll
      for (int i = 0; i < 16; ++i) va[i] = DWS(i);
      va[16] = DBS;   // zero in lane half 1
#pragma unroll
      for (int i = 0; i < 8; ++i) va[17 + i] = DBACC(i);
#pragma unroll
      for (int i = 0; i < 25; ++i) vb[i] = DBACC(8 + i);
      half_sums_lds<25>(va, scr, lane, [&](int i, int hh, float s) {
        if (i < 16) small[acc_row(i, hh)] = s;
        else if (i == 16) { if (hh == 0) small[32] = s; }
        else put_db(i - 17, hh, s);
      });
      half_sums_lds<25>(vb, scr, lane, [&](int i, int hh, float s) { put_db(8 + i, hh, s); });
      if constexpr (TWO) {   // two rays per tile: lanes 0 and 16 hold the two rays' shares
        ld_acc += lane_value(ld_acc, 16); lc_acc += lane_value(lc_acc, 16); lo_acc += lane_value(lo_acc, 16);
      }
      if (KR > 0 && lane == 0) { small[40] = ld_acc; small[41] = lc_acc; small[42] = lo_acc; small[43] = __int_as_float(tab_flags);
                                 small[44] = wd_c; small[45] = wc_c; small[46] = wo_c; }
    }
  } else {
    // ===================================================================================================
    // dW role: 8 accumulator blocks per wave, six tiles per step
    // ===================================================================================================
    copy_operands([]() {});
    if constexpr (SPLIT_COPY) copy_transposed();
    // row m of the row-sum block: m = rs * latent slot + object row for the four latent layers, then one
    // "ones" row each for the two plain biases (encoding_shape: group 4, rgb.0: group 5)
    constexpr int rs = RS_ROWS;  // rows per latent slot: a compile-time constant of the instantiation (a run-time stride cost 2.4 us)
    // (WIDE = 2: block A holds slots 0, 1 and the two bias rows, block B slots 2, 3)
    // (WIDE = 3: rows 0..15 = 4 x chain wave + latent slot, flushed every iteration; rows 16, 17 the two bias rows)
    const int nlat_rows = ROWTILE ? 16 : (WIDE == 2 ? 2 : 4) * rs;
    const int rpc_inv = (65536 + rs - 1) / rs;  // m / rs = (m * rpc_inv) >> 16 for m < 32
    const int col_slot = (col * rpc_inv) >> 16;
    const int m_row = col < nlat_rows ? col - col_slot * rs : rs,
              m_grp = col < nlat_rows ? col_slot : 4 + (col - nlat_rows),
              m_grpB = col < nlat_rows ? 2 + col_slot : -1;
    // lane-dependent parts of the LDS read addresses (see consume): swizzled 32 x 32 images (lo / hi row quads), the two PE
    // images, the row one-hot table
    int p_hs_lo, p_hs_hi, p_e1, p_e2, p_r;
    {
      const int i = lane & 15, g16 = lane >> 4, q = i >> 2, pp = i & 3, hh = g16 >> 1;
      const int r = 8 * hh + q, chunk = 4 * (g16 & 1) + pp;
      p_hs_lo = r * 64 + ((chunk ^ ((r >> 1) & 7)) << 3);
      p_hs_hi = (r + 4) * 64 + ((chunk ^ (((r + 4) >> 1) & 7)) << 3);
      p_e1 = r * ST_E1 + (16 * (g16 & 1) + 4 * pp) * 2;
      p_e2 = r * ST_E2 + (16 * (g16 & 1) + 4 * pp) * 2;
      p_r = (m_row * 32 + 8 * h) * 2;
    }
    // Where this lane's column of each owned block goes in the record: every block kind is affine in the output row,
    // idx(o) = i0 + o * st (weights: st = the layer's row length; a bias column: st = 1), i0 < 0 = not a parameter.
    // Computed here, while the chain waves recompute the first forward: the flush then stores accumulators straight to
    // the record (no LDS image, no index table, no gather loop: 7 k -> 2.5 k cycles of every workgroup's tail).
    // the dW wave shares its SIMD with a chain wave; when both have an instruction ready the dW wave goes first: it is
    // the consumer every layer step waits for (measured: -2 % at 2048 x 64, -5 % at 8192 x 128; the reverse: no effect)
    // (raised BEHIND the record-index setup below: those ~1 000 instructions run beside the chain waves' first forward and must
    //  not take issue slots from it)
    int bi0[NACC], bst[NACC], r2i0 = -1, r2st = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) { bi0[i] = -1; bst[i] = 0; }
#define CNR_PIDX8(KIND)                                                                        \
  if (owner8<NDW>(KIND) == dwid) {                                                             \
    constexpr int li = local8<NDW>(KIND);                                                      \
    bi0[li] = block_index(KIND, 0, col);                                                       \
    bst[li] = block_index(KIND, 1, col) - bi0[li];                                             \
  }
    CNR_PIDX8(BK_R0) CNR_PIDX8(BK_T1) CNR_PIDX8(BK_VD_Y) CNR_PIDX8(BK_VD_E0) CNR_PIDX8(BK_VD_E1) CNR_PIDX8(BK_ES)
    CNR_PIDX8(BK_S2) CNR_PIDX8(BK_CAT_Y) CNR_PIDX8(BK_CAT_E0) CNR_PIDX8(BK_CAT_E1) CNR_PIDX8(BK_CAT_E2) CNR_PIDX8(BK_S1)
    CNR_PIDX8(BK_XYZ_E0) CNR_PIDX8(BK_XYZ_E1) CNR_PIDX8(BK_XYZ_E2)
#undef CNR_PIDX8
    if (owner8<NDW>(BK_R0) == dwid) { r2i0 = block_index(BK_R2, 0, col); r2st = block_index(BK_R2, 1, col) - r2i0; }
    asm volatile("" : "+v"(r2i0));               // (keeps the raise below the setup)
    __builtin_amdgcn_s_setprio(3);
    // The role branches on the dW wave's index ONCE, around the whole iteration loop: with the branch inside every layer step
    // the accumulator blocks met at a join after each step, and the compiler copied them between register ranges there (16
    // moves behind the step's last MFMA, in front of the barrier the chain waves wait at).
    // PEDW: this wave runs the PE backward of chain wave dwid's tiles (same lane = same sample column and lane half)
    float w_dbacc[33];
#pragma unroll
    for (int i = 0; i < 33; ++i) w_dbacc[i] = 0.0f;
    auto dw_loop = [&](auto dwi_c) {
    constexpr int DWI = decltype(dwi_c)::value;
    const unsigned char* pcw = chain_base + DWI * K_BYTES;
    // the partner tile's sample position (same index arithmetic as the chain's fetch).  Requested one iteration ahead, right
    // behind the iteration's last barrier: every barrier's release fence waits for the wave's outstanding loads, so a request
    // in front of a layer-step barrier puts its round trip on the workgroup's critical path (~0.3 us per iteration, measured)
    auto pts_of = [&](int tile, float& q0, float& q1, float& q2) {
      const int tl = tile < ntiles ? tile : ntiles - 1, n0 = tl * 32;
      int64_t gs;
      if constexpr (PAD) {
        const int slot = n0 + col, ray_l = slot / SP, sidx = slot % SP;
        gs = ((int64_t)c * R + (ray_l < R ? ray_l : R - 1)) * S + (sidx < S ? sidx : S - 1);
      } else {
        gs = (int64_t)c * N + n0 + col;
      }
      const float* pp = pts + gs * 3;
      q0 = pp[0]; q1 = pp[1]; q2 = pp[2];
    };
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    if constexpr (PEDW) pts_of(blockIdx.x * NCHW + DWI, p0, p1, p2);
    // ROWTILE (the row-sum owner): the object rows of the iteration's four tiles, requested one iteration ahead like the position
    constexpr bool RS_FLUSH = ROWTILE && DWI == owner8<NDW>(BK_RS);
    auto obj_rows = [&](int t0_, int (&o)[4]) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int tile = t0_ + w, tl = tile < ntiles ? tile : ntiles - 1;
        const int ray_l = PAD ? (tl * 32) / SP : tl / (KR > 0 ? KR : 1);
        o[w] = ray_row[(int64_t)c * R + (ray_l < R ? ray_l : R - 1)] - c * rows_per_class;
      }
    };
    int orow[4] = {0, 0, 0, 0};
    if constexpr (RS_FLUSH) obj_rows(blockIdx.x * NCHW, orow);
    for (int t0 = blockIdx.x * NCHW; t0 < ntiles; t0 += tile_step) {
      asm volatile("" ::: "memory");
      P8STAMP_RESET();
      P8STAMP();
      // one layer step over the six tiles; KIND* = the block kinds of the step (X source: 0 = slot X image,
      // 1 = E2 image, 2 = E1 image; col0 = first feature column), RS_GRP >= 0: row sums of the step (dW0)
      auto consume = [&](auto dw_c, auto nx_c, auto k0_c, auto k1_c, auto k2_c, auto k3_c, auto grp_c, auto par_c) {
        constexpr int DW = decltype(dw_c)::value, NX = decltype(nx_c)::value, RS_GRP = decltype(grp_c)::value;
        constexpr int PAR = decltype(par_c)::value * K_PAR;   // which copy of the dPre / input images the step uses
        constexpr int K0 = decltype(k0_c)::value, K1 = decltype(k1_c)::value, K2 = decltype(k2_c)::value,
                      K3 = decltype(k3_c)::value;
        // which of the step's blocks this dW wave owns, and their operand slots (all compile-time: the accumulator
        // and fragment arrays must be indexed by constants)
        constexpr bool OWN0 = NX > 0 && owner8<NDW>(K0) == DW, OWN1 = NX > 1 && owner8<NDW>(K1) == DW,
                       OWN2 = NX > 2 && owner8<NDW>(K2) == DW, OWN3 = NX > 3 && owner8<NDW>(K3) == DW;
        constexpr int SL0 = 0, SL1 = OWN0, SL2 = OWN0 + OWN1, SL3 = OWN0 + OWN1 + OWN2;
        constexpr bool DO_RS = RS_GRP >= 0 && DW == owner8<NDW>(BK_RS);
        if constexpr (!(OWN0 || OWN1 || OWN2 || OWN3 || DO_RS)) return;
        // at most two owned blocks per wave and step; DEPTH tiles in flight (transposing reads are convergent: the
        // compiler keeps them in source order, so the lookahead is spelled out): all four, the registers are there
        constexpr int DEPTH = NCHW;
        // The lane-dependent parts of the LDS addresses (one VGPR per access pattern, formed once before the loop) are made
        // opaque per step, so that nothing is hoisted out of the iteration loop (~30 loop-invariant address registers spilled
        // when they were) and nothing is re-derived from the lane index at the start of a step, where the chain waves wait.
        // The images of the four chain waves lie 83 .. 160 KB into the LDS, beyond a ds instruction's 16-bit offset field: with
        // ONE base per pattern every read needed its own v_add (32 .. 48 per step and wave, seen in the ISA).  TWO bases per
        // pattern -- chain waves 0 | 1 and 2 | 3, each carrying its pair's start -- leave every read a constant below 64 KB that
        // folds into the instruction: ten adds per step.
        constexpr int PAIR_OFF[2] = {L8_CHAIN, L8_CHAIN + 2 * K_BYTES};
        int a_lo[2], a_hi[2], a_e1[2], a_e2[2], a_r[2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          a_lo[pr] = p_hs_lo + PAIR_OFF[pr]; a_hi[pr] = p_hs_hi + PAIR_OFF[pr]; a_e1[pr] = p_e1 + PAIR_OFF[pr];
          a_e2[pr] = p_e2 + PAIR_OFF[pr]; a_r[pr] = p_r + PAIR_OFF[pr];
          asm volatile("" : "+v"(a_lo[pr]), "+v"(a_hi[pr]), "+v"(a_e1[pr]), "+v"(a_e2[pr]), "+v"(a_r[pr]));
        }
        // (w: chain wave of the image, off: the image's offset inside that wave's region)
        auto rd_hs = [&](int w, int off, int sl) {   // = tr_frag_hs(image, sl, lane)
          typedef __attribute__((address_space(3))) s4v* lds_s4;
          const int k = (w & 1) * K_BYTES + off + sl * 1024;
          h8 out;
          out.lo = __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(smem + a_lo[w >> 1] + k)));
          out.hi = __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(smem + a_hi[w >> 1] + k)));
          return out;
        };
        auto rd_pe = [&](int w, int off, int stride, int col0, int sl) {   // = tr_frag(image, stride, col0, sl, lane)
          typedef __attribute__((address_space(3))) s4v* lds_s4;
          const unsigned char* a = smem + (stride == ST_E1 ? a_e1[w >> 1] : a_e2[w >> 1]) + ((w & 1) * K_BYTES + off + 16 * sl * stride + col0 * 2);
          h8 out;
          out.lo = __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(a)));
          out.hi = __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(a + 4 * stride)));
          return out;
        };
        h8 fD[DEPTH][2], fX[DEPTH][2][2];
        u4v fR[DEPTH][2];          // row one-hot operand of the row sums
        auto load_blk = [&](int w, auto kind_c, auto slot_c) {
          constexpr int kind = decltype(kind_c)::value, slot = decltype(slot_c)::value;
          int ximg, stride = 0, col0 = 0;     // (offset of the image inside chain wave w's region)
          if (kind == BK_VD_E0 || kind == BK_VD_E1) { ximg = K_E2; stride = ST_E2; col0 = kind == BK_VD_E1 ? 32 : 0; }
          else if (kind == BK_CAT_E0 || kind == BK_XYZ_E0) { ximg = K_E1; stride = ST_E1; col0 = 0; }
          else if (kind == BK_CAT_E1 || kind == BK_XYZ_E1) { ximg = K_E1; stride = ST_E1; col0 = 32; }
          else if (kind == BK_CAT_E2 || kind == BK_XYZ_E2) { ximg = K_E1; stride = ST_E1; col0 = 64; }
          else ximg = K_X + PAR;
          if (stride == 0) {  // the step's input image (swizzled rows)
            fX[w % DEPTH][slot][0] = rd_hs(w, ximg, 0);
            fX[w % DEPTH][slot][1] = rd_hs(w, ximg, 1);
          } else {            // a PE image
            fX[w % DEPTH][slot][0] = rd_pe(w, ximg, stride, col0, 0);
            fX[w % DEPTH][slot][1] = rd_pe(w, ximg, stride, col0, 1);
          }
        };
        auto mma_blk = [&](int w, auto kind_c, auto slot_c) {
          constexpr int li = local8<NDW>(decltype(kind_c)::value), slot = decltype(slot_c)::value;
          Wacc[li] = MFMA(fD[w % DEPTH][0], fX[w % DEPTH][slot][0], Wacc[li]);
          Wacc[li] = MFMA(fD[w % DEPTH][1], fX[w % DEPTH][slot][1], Wacc[li]);
        };
        auto load_tile = [&](int w) {
          fD[w % DEPTH][0] = rd_hs(w, K_D + PAR, 0);
          fD[w % DEPTH][1] = rd_hs(w, K_D + PAR, 1);
          if constexpr (OWN0) load_blk(w, IC<K0>{}, IC<SL0>{});
          if constexpr (OWN1) load_blk(w, IC<K1>{}, IC<SL1>{});
          if constexpr (OWN2) load_blk(w, IC<K2>{}, IC<SL2>{});
          if constexpr (OWN3) load_blk(w, IC<K3>{}, IC<SL3>{});
          if constexpr (DO_RS && !ROWTILE) {
            const unsigned char* rowoh = smem + a_r[w >> 1] + ((w & 1) * K_BYTES + K_SMALL);   // [m_row][32] halfs: this lane's k = 8 h .. and 16 + 8 h ..
            fR[w % DEPTH][0] = __builtin_bit_cast(u4v, *reinterpret_cast<const h8*>(rowoh));
            fR[w % DEPTH][1] = __builtin_bit_cast(u4v, *reinterpret_cast<const h8*>(rowoh + 32));
          }
        };
#pragma unroll
        for (int w = 0; w < NCHW; ++w) load_tile(w);
        // every read of the step is in flight before the first MFMA: left alone, the scheduler sinks the later tiles' reads
        // between the MFMAs (fewer live registers, which this role has to spare) and the wave then waits out one LDS
        // latency per tile instead of one per step
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int w = 0; w < NCHW; ++w) {
          if constexpr (OWN0) mma_blk(w, IC<K0>{}, IC<SL0>{});
          if constexpr (OWN1) mma_blk(w, IC<K1>{}, IC<SL1>{});
          if constexpr (OWN2) mma_blk(w, IC<K2>{}, IC<SL2>{});
          if constexpr (OWN3) mma_blk(w, IC<K3>{}, IC<SL3>{});
          if constexpr (DO_RS && ROWTILE) {
            // the tile is one ray = one object: row 4 w + slot (latent layers) or 16 / 17 (plain biases) takes the column sum
            const int mrow = RS_GRP < 4 ? 4 * w + RS_GRP : 16 + (RS_GRP - 4);
            const unsigned int on = col == mrow ? 0x3c003c00u : 0u;     // f16 ones
            const h8 ones = __builtin_bit_cast(h8, (u4v){on, on, on, on});
            Wacc[LI_RS] = MFMA(ones, fD[w % DEPTH][0], Wacc[LI_RS]);
            Wacc[LI_RS] = MFMA(ones, fD[w % DEPTH][1], Wacc[LI_RS]);
          } else if constexpr (DO_RS) {
            // RS[m][:] += sum over the tile's samples in m's group of dPre
            constexpr bool BLK_B = WIDE == 2 && (RS_GRP == 2 || RS_GRP == 3);
            constexpr int LI = BLK_B ? LI_RS2 : LI_RS;
            const unsigned int lm = ((BLK_B ? m_grpB : m_grp) == RS_GRP) ? 0xffffffffu : 0u;
            Wacc[LI] = MFMA(__builtin_bit_cast(h8, (u4v)(fR[w % DEPTH][0] & lm)), fD[w % DEPTH][0], Wacc[LI]);
            Wacc[LI] = MFMA(__builtin_bit_cast(h8, (u4v)(fR[w % DEPTH][1] & lm)), fD[w % DEPTH][1], Wacc[LI]);
          }
        }
      };
      if constexpr (KR > 1) { P8SYNC(); }   // the chain waves' composite exchange
#define STEP8(PAR, NX, K0, K1, K2, K3, GRP)                                                                   \
  P8SYNC();                                                                                             \
  consume(IC<DWI>{}, IC<NX>{}, IC<K0>{}, IC<K1>{}, IC<K2>{}, IC<K3>{}, IC<GRP>{}, IC<PAR>{});
      STEP8(0, 1, BK_R0, BK_R0, BK_R0, BK_R0, -1)                            // rgb.2 (rows 16..18 of rgb.0's block)
      STEP8(1, 1, BK_R0, BK_R0, BK_R0, BK_R0, 5)                             // rgb.0
      STEP8(0, 1, BK_T1, BK_T1, BK_T1, BK_T1, 3)                             // texture_layer_1
      STEP8(1, 3, BK_VD_Y, BK_VD_E0, BK_VD_E1, BK_VD_E1, -1)                 // encoding_viewdir
      STEP8(0, 1, BK_ES, BK_ES, BK_ES, BK_ES, 4)                             // encoding_shape
      STEP8(1, 1, BK_S2, BK_S2, BK_S2, BK_S2, 2)                             // shape_layer_2
      STEP8(0, 4, BK_CAT_Y, BK_CAT_E0, BK_CAT_E1, BK_CAT_E2, 1)              // cat_layer
      STEP8(1, 1, BK_S1, BK_S1, BK_S1, BK_S1, 0)                             // shape_layer_1
      STEP8(0, 3, BK_XYZ_E0, BK_XYZ_E1, BK_XYZ_E2, BK_XYZ_E2, -1)            // encoding_xyz
      h8 dq[PEDW ? 9 : 1];
      if constexpr (PEDW) {   // d e2 (in the E2 image since step ES): picked up BEFORE the barrier behind which the partner writes the next tile's E2 features
#pragma unroll
        for (int k = 0; k < 3; ++k) dq[6 + k] = *reinterpret_cast<const h8*>(pcw + K_E2 + lane * 16 + k * 1024);
      }
      P8SYNC();   // done with this iteration's images (the chain waves wait for it before they write the next ones)
#undef STEP8
      if constexpr (RS_FLUSH) {
        // rows 0..15 of the row-sum block = (chain wave, latent slot) column sums of THIS iteration's tiles -> the fixed-point table
        // at the tiles' object rows; this lane half holds rows 4 h + 0..3 (registers 0..3: chain wave h) and 8 + 4 h + 0..3
        // (registers 4..7: chain wave 2 + h)
        long long* tab = rows_fix + ((size_t)(blockIdx.x % cnr_rec::ROWS_FIX_COPIES) * gridDim.y + c) * rows_per_class * 128;
        int nrow[4];
        obj_rows(t0 + tile_step, nrow);
#pragma unroll
        for (int reg = 0; reg < 8; ++reg) {
          const int o = reg < 4 ? (h ? orow[1] : orow[0]) : (h ? orow[3] : orow[2]);
          const float v = Wacc[LI_RS][reg] * inv_gs;
          atomicAdd(reinterpret_cast<unsigned long long*>(tab + (size_t)o * 128 + (reg & 3) * 32 + col),
                    (unsigned long long)__double2ll_rn((double)v * cnr_rec::ROWS_FIX_SCALE));
          Wacc[LI_RS][reg] = 0.0f;
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) orow[w] = nrow[w];
      }
      if constexpr (PEDW) {
        // d e1 (written by the partner between step XYZ's barrier and the one above).  The workgroup's last tile is left to the
        // chain wave: behind the loop this wave has the record to write (~10 k cycles, the longer pole) and the chain wave nothing
        if (t0 + tile_step < ntiles) {
#pragma unroll
          for (int k = 0; k < 6; ++k) dq[k] = *reinterpret_cast<const h8*>(pcw + K_X + lane * 16 + k * 1024);
          // filler work: below the chain wave (priority 1) while it lasts, so that it takes only issue slots the partner's forward
          // leaves empty (at the dW role's priority 3 it pushed the forward back: 40.4 -> 42.0 us at 2048 x 64)
          __builtin_amdgcn_s_setprio(0);
          float n0x = 0.f, n1x = 0.f, n2x = 0.f;
          pts_of(t0 + tile_step + DWI, n0x, n1x, n2x);
          pe_backward_h(dq, p0 * inv_scale, p1 * inv_scale, p2 * inv_scale, w_dbacc);
          p0 = n0x; p1 = n1x; p2 = n2x;
          __builtin_amdgcn_s_setprio(3);
        }
      }
    }
    };
    if (dwid == 0) dw_loop(IC<0>{});
    else if (dwid == 1) dw_loop(IC<1>{});
    else if (dwid == 2) dw_loop(IC<2>{});
    else dw_loop(IC<3>{});
    P8PHASE(3);
    // ---- this wave's blocks -> the workgroup's record, straight from the accumulators -------------------------
    rec_t* rec = records + ((size_t)c * gridDim.x + blockIdx.x) * REC_ENTRIES;
#define CNR_PSTORE8(KIND, NROWS)                                                               \
  if (owner8<NDW>(KIND) == dwid) {                                                             \
    constexpr int li = local8<NDW>(KIND);                                                      \
    if (bi0[li] >= 0) {                                                                        \
      _Pragma("unroll") for (int reg = 0; reg < 16; ++reg) {                                   \
        const int o = acc_row(reg, h);                                                         \
        if (o < (NROWS)) rec[bi0[li] + o * bst[li]] = rec_pack(Wacc[li][reg] * inv_gs);        \
      }                                                                                        \
    }                                                                                          \
  }
    CNR_PSTORE8(BK_R0, 16) CNR_PSTORE8(BK_T1, 32) CNR_PSTORE8(BK_VD_Y, 32) CNR_PSTORE8(BK_VD_E0, 32)
    CNR_PSTORE8(BK_VD_E1, 32) CNR_PSTORE8(BK_ES, 32) CNR_PSTORE8(BK_S2, 32) CNR_PSTORE8(BK_CAT_Y, 32)
    CNR_PSTORE8(BK_CAT_E0, 32) CNR_PSTORE8(BK_CAT_E1, 32) CNR_PSTORE8(BK_CAT_E2, 32) CNR_PSTORE8(BK_S1, 32)
    CNR_PSTORE8(BK_XYZ_E0, 32) CNR_PSTORE8(BK_XYZ_E1, 32) CNR_PSTORE8(BK_XYZ_E2, 32)
#undef CNR_PSTORE8
    P8PHASE(4);
    if (dwid == owner8<NDW>(BK_R0) && r2i0 >= 0) {  // rgb.2 out of rows 16..18 of rgb.0's block
      constexpr int li = local8<NDW>(BK_R0);
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int o = acc_row(reg, h) - 16;
        if (o >= 0 && o < 3) rec[r2i0 + o * r2st] = rec_pack(Wacc[li][reg] * inv_gs);
      }
    }
    if (dwid == owner8<NDW>(BK_RS)) {  // the row-sum block: [m][feature], m = rows_per_class * latent slot + object row | then the two biases
#pragma unroll
      for (int blk = 0; blk < (WIDE == 2 ? 2 : 1); ++blk)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = acc_row(reg, h);
        const float v = (blk ? Wacc[LI_RS2][reg] : Wacc[LI_RS][reg]) * inv_gs;
        if (blk == 0 && m == nlat_rows) rec[OFF_ES_B + col] = rec_pack(v);
        else if (blk == 0 && m == nlat_rows + 1) { if (col < 16) rec[OFF_R0_B + col] = rec_pack(v); }
        else if (!ROWTILE && m < nlat_rows && m - ((m * rpc_inv) >> 16) * rs < rows_per_class) {   // (ROWTILE: flushed every iteration)
          const int slot = 2 * blk + ((m * rpc_inv) >> 16);
          const int i = (m - ((m * rpc_inv) >> 16) * rs) * 128 + slot * 32 + col;  // dbiasrows [row][latent slot][feature]
          rec[TRUNK + 126 + i] = rec_pack(v);
          if (rows_fix)
            atomicAdd(reinterpret_cast<unsigned long long*>(
                          rows_fix + ((size_t)(blockIdx.x % cnr_rec::ROWS_FIX_COPIES) * gridDim.y + c) * rows_per_class * 128 + i),
                      (unsigned long long)__double2ll_rn((double)v * cnr_rec::ROWS_FIX_SCALE));
        }
      }
    }
    P8PHASE(6);
    if constexpr (PEDW) {   // dB partial sums of the partner's tiles -> the partner's E1 image (free: the loop is over), 63 floats
      float* e1f = reinterpret_cast<float*>(chain_base + dwid * K_BYTES + K_E1);
      float* scr = reinterpret_cast<float*>(smem + wv * SUMS_SCRATCH);
      auto put_db = [&](int i, int hh, float s) {
        const int d = i / 3;
        if (!(hh == 1 && d == 10)) e1f[(11 * hh + d) * 3 + (i - 3 * d)] = s;
      };
      float va[17], vb[16];
#pragma unroll
      for (int i = 0; i < 17; ++i) va[i] = w_dbacc[i];
#pragma unroll
      for (int i = 0; i < 16; ++i) vb[i] = w_dbacc[17 + i];
      half_sums_lds<17>(va, scr, lane, [&](int i, int hh, float s) { put_db(i, hh, s); });
      half_sums_lds<16>(vb, scr, lane, [&](int i, int hh, float s) { put_db(17 + i, hh, s); });
    }
  }

  // ========================================= flush ====================================================
  // (the dW waves have written their blocks above; what is left are the chain waves' partial sums)
  rec_t* rec = records + ((size_t)c * gridDim.x + blockIdx.x) * REC_ENTRIES;
  P8PHASE(7);
  __syncthreads();
  P8PHASE(2);
  {
    auto sum_chain = [&](int i) {
      float v = 0.0f;
#pragma unroll
      for (int w = 0; w < NCHW; ++w) v += reinterpret_cast<const float*>(chain_base + w * K_BYTES + K_SMALL)[i];
      return v;
    };
    auto sum_dw = [&](int i) {   // PEDW: the dW waves' shares of dB
      float v = 0.0f;
      if constexpr (PEDW) {
#pragma unroll
        for (int w = 0; w < NCHW; ++w) v += reinterpret_cast<const float*>(chain_base + w * K_BYTES + K_E1)[i];
      }
      return v;
    };
    for (int i = threadIdx.x; i < 63; i += NTHR) { rec[TRUNK + i] = rec_pack((sum_chain(64 + i) + sum_dw(i)) * inv_gs); rec[TRUNK + 63 + i] = 0; }
    for (int i = threadIdx.x; i < 32; i += NTHR) rec[OFF_SG_W + i] = rec_pack(sum_chain(i) * inv_gs);
    if (threadIdx.x == 0) rec[OFF_SG_B] = rec_pack(sum_chain(32) * inv_gs);
    if constexpr (KR > 0) {   // per-block loss partials + the class header, the format of cnr_field_fwd_render
      const int nb = gridDim.x, Cn = gridDim.y;
      if (threadIdx.x < 3) ta.partials[((size_t)c * nb + blockIdx.x) * 3 + threadIdx.x] = sum_chain(40 + threadIdx.x);
      if (blockIdx.x == 0 && threadIdx.x < 4) {
        const float* s0 = reinterpret_cast<const float*>(chain_base + K_SMALL);
        float* hdr = ta.partials + (size_t)Cn * nb * 3 + (size_t)c * 4;
        hdr[threadIdx.x] = threadIdx.x < 3 ? s0[44 + threadIdx.x] : (float)__float_as_int(s0[43]);
      }
    }
  }
  P8PHASE(5);
}
}  // namespace

// launched by cnr_field_bwd_pipe (below) and cnr_field_train: same records
template <int WIDE, int KR, bool TWO = false, bool PAD = false, bool GEO = false>
static int launch_p8(const float* pts, const float* B, const void* packed, const void* packed_lo, const float* biasrows,
                     const int* ray_row, float scale, const float* d_sigma, const float* d_rgb, float grad_scale, int C,
                     int R, int S, int rows_per_class, int blocks, void* workspace, int64_t B_stride, long long* rows_fix,
                     int* clamp_flags, const TrainArgs& ta, void* stream) {
  static cnr::DeviceOnce lds_attr;   // per instantiation, one bit per device
  constexpr int LDS = l8_total(4, WIDE, GEO);
  const int er = cnr::set_max_dynamic_lds(lds_attr, (const void*)field_bwd_pipe8_kernel<4, 4, WIDE, KR, TWO, PAD, GEO>, LDS);
  if (er) return er;
  const int64_t N = (int64_t)R * S;
  hipLaunchKernelGGL((field_bwd_pipe8_kernel<4, 4, WIDE, KR, TWO, PAD, GEO>), dim3((unsigned)blocks, (unsigned)C), dim3(512), LDS,
                     (hipStream_t)stream, pts, B, (const unsigned char*)packed, (const unsigned char*)packed_lo, biasrows,
                     ray_row, 1.0f / scale, d_sigma, d_rgb, grad_scale, (rec_t*)workspace, (int)N, S, R, rows_per_class,
                     B_stride > 0 ? B_stride : (int64_t)63, rows_fix, clamp_flags, ta);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

