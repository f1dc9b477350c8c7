// Shared pieces of the fused backward kernel's translation units (fused_bwd_pipe8*.hip: chain / dW wave pipeline): LDS image
// formats, transposing operand reads, the dW block -> parameter map, the per-workgroup record format and its fixed-order
// reduction.
#pragma once
#include "fused_common.h"
#include "records_common.h"
#include <type_traits>

namespace {   // internal linkage: each backward translation unit gets its own copy (incl. the device table)
using namespace fz;
typedef short s4v __attribute__((ext_vector_type(4)));

enum BlockKind { BK_R2, BK_R0, BK_T1, BK_VD_Y, BK_VD_E0, BK_VD_E1, BK_ES, BK_S2, BK_CAT_Y, BK_CAT_E0, BK_CAT_E1,
                 BK_CAT_E2, BK_S1, BK_XYZ_E0, BK_XYZ_E1, BK_XYZ_E2, NBLOCKS };

constexpr int ST_E1 = 208;  // [32][half0: 48 slots | half1: 48 slots] = 192 B + 16 B pad
constexpr int ST_E2 = 112;  // [32][half0: 24 | half1: 24] = 96 B + 16 B pad
// [sample][feature] image of an accumulator-layout tile held as two f16 fragments, swizzled: unpadded 64-byte rows, the 8-byte
// chunk c of row r stored at
// chunk c ^ ((r >> 1) & 7).  A transposing read's 32-lane half covers 4 consecutive rows x 64 B = all 64 banks once
// (the padded form wraps after 3.5 rows: 2-way), and a ds_write_b64's 16-lane group (16 rows, one logical chunk)
// lands on 16 different bank pairs of the 32 store banks.
constexpr int HSIMG_BYTES = 32 * 64;
__device__ __forceinline__ void stage_hs(unsigned char* img, const h8& f0, const h8& f1, int col, int h) {
  unsigned char* row = img + col * 64;
  const int sw = (col >> 1) & 7;
  *reinterpret_cast<h4*>(row + (((0 + h) ^ sw) << 3)) = f0.lo;
  *reinterpret_cast<h4*>(row + (((2 + h) ^ sw) << 3)) = f0.hi;
  *reinterpret_cast<h4*>(row + (((4 + h) ^ sw) << 3)) = f1.lo;
  *reinterpret_cast<h4*>(row + (((6 + h) ^ sw) << 3)) = f1.hi;
}
__device__ __forceinline__ h8 tr_frag_hs(const unsigned char* img, int s, int lane) {
  const int i = lane & 15, g16 = lane >> 4, q = i >> 2, p = i & 3, hh = g16 >> 1;
  const int r = 16 * s + 8 * hh + q, chunk = 4 * (g16 & 1) + p;
  typedef __attribute__((address_space(3))) s4v* lds_s4;
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + r * 64 + ((chunk ^ ((r >> 1) & 7)) << 3)));
  const s4v hi =
      __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(img + (r + 4) * 64 + ((chunk ^ (((r + 4) >> 1) & 7)) << 3)));
  h8 out;
  out.lo = __builtin_bit_cast(h4, lo);
  out.hi = __builtin_bit_cast(h4, hi);
  return out;
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

using cnr_rec::REC_ENTRIES;
using cnr_rec::rec_t;
using cnr_rec::rec_pack;
using cnr_rec::rec_entry_written;

// sum over the 32 lanes of each wave half with DPP row operations (6 VALU ops; __shfl_xor lowers to
// ds_bpermute, ~80 dependent cycles each): afterwards lane 31 holds the sum of lanes 0..31, lane 63 of 32..63
__device__ __forceinline__ float half_sum_dpp(float v) {
  int x = __builtin_bit_cast(int, v);
#define CNR_DPP_ADD(CTRL, ROWMASK)                                                                     \
  x = __builtin_bit_cast(int, __builtin_bit_cast(float, x) +                                           \
                                  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, CTRL, ROWMASK, 0xf, true)))
  CNR_DPP_ADD(0x111, 0xf);  // row_shr:1
  CNR_DPP_ADD(0x112, 0xf);  // row_shr:2
  CNR_DPP_ADD(0x114, 0xf);  // row_shr:4
  CNR_DPP_ADD(0x118, 0xf);  // row_shr:8   -> lane 15 of every 16-lane row holds the row sum
  CNR_DPP_ADD(0x142, 0xa);  // row_bcast:15 into rows 1 and 3 -> lanes 31 / 63 hold the half sums
#undef CNR_DPP_ADD
  return __builtin_bit_cast(float, x);
}

// Sums over the 32 lanes of each wave half of N per-lane values at once, through a wave-private LDS scratch of HALF_SUMS_STRIDE * N
// floats: every lane stores its N values ([value][lane], rows 68 floats apart: the stores and the 16-byte re-reads are free of
// bank conflicts up to N = 16, two-way beyond), lane t < 2 N then adds the 32 values of (value t >> 1, lane half t & 1) in lane
// order and hands the sum to out(value, half, sum).  N half_sum_dpp calls are N dependent chains of five DPP adds with their
// wait states and one predicated store each (~100 cycles apiece as measured behind the step body's loop: 50 of them 4.8 k cycles);
// this is N stores + 8 reads + 32 adds.  LDS operations of one wave execute in order: no barrier, only the compiler is fenced.
constexpr int HALF_SUMS_STRIDE = 68;
template <int N, class F> __device__ __forceinline__ void half_sums_lds(const float (&v)[N], float* scratch, int lane, F&& out) {
  static_assert(2 * N <= 64, "one (value, half) pair per lane");
#pragma unroll
  for (int i = 0; i < N; ++i) scratch[i * HALF_SUMS_STRIDE + lane] = v[i];
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  if (lane < 2 * N) {
    const f4* row = reinterpret_cast<const f4*>(scratch + (lane >> 1) * HALF_SUMS_STRIDE + (lane & 1) * 32);
    f4 q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = row[k];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s = ((((s + q[k][0]) + q[k][1]) + q[k][2]) + q[k][3]);
    out(lane >> 1, lane & 1, s);
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// ---- 32-lane scans on DPP row operations (a __shfl is a ds_bpermute: ~130 dependent cycles each, a 32-lane scan of five of
// them ~700; these are five VALU instructions).  Each works on both 32-lane halves of the wave independently.
// inclusive prefix sum over lanes 0..i of the half = half_sum_dpp (lane 31 / 63: the half's total)
__device__ __forceinline__ float half_scan_add(float v) { return half_sum_dpp(v); }
// inclusive prefix product
__device__ __forceinline__ float half_scan_mul(float v) {
  int x = __builtin_bit_cast(int, v);
  const int one = 0x3f800000;
#define CNR_DPP_MUL(CTRL, ROWMASK)                                                                     \
  x = __builtin_bit_cast(int, __builtin_bit_cast(float, x) *                                           \
                                  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(one, x, CTRL, ROWMASK, 0xf, false)))
  CNR_DPP_MUL(0x111, 0xf);  // row_shr:1 (lanes without a source keep the neutral element)
  CNR_DPP_MUL(0x112, 0xf);
  CNR_DPP_MUL(0x114, 0xf);
  CNR_DPP_MUL(0x118, 0xf);
  CNR_DPP_MUL(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
#undef CNR_DPP_MUL
  return __builtin_bit_cast(float, x);
}
// value of the lane below (wave_shr:1), lane 0 gets `first`
__device__ __forceinline__ float lane_below(float v, float first) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, first), __builtin_bit_cast(int, v),
                                                               0x138, 0xf, 0xf, false));
}
// inclusive SUFFIX sum over lanes i..31 of lane half 0 (row_shl inside the rows of 16, then row 1's total -- lane 16 --
// onto row 0); lane half 1 is not meaningful
__device__ __forceinline__ float half0_suffix_add(float v, int lane) {
  int x = __builtin_bit_cast(int, v);
#define CNR_DPP_ADDL(CTRL)                                                                             \
  x = __builtin_bit_cast(int, __builtin_bit_cast(float, x) +                                           \
                                  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true)))
  CNR_DPP_ADDL(0x101);  // row_shl:1
  CNR_DPP_ADDL(0x102);
  CNR_DPP_ADDL(0x104);
  CNR_DPP_ADDL(0x108);
#undef CNR_DPP_ADDL
  const float up = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16));
  const float r = __builtin_bit_cast(float, x);
  return (lane & 16) ? r : r + up;
}
__device__ __forceinline__ float lane_value(float v, int l) {  // wave-uniform copy of lane l (l a constant)
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// ---- the same scans over SEGMENTS: ROW16 = false: one segment = the 32 lanes of lane half 0 (as above); ROW16 = true: two
// segments of 16 lanes (DPP rows 0 and 1 of half 0) -- two short rays side by side in one tile.  seg_total: the segment's
// total of an inclusive prefix scan, in every lane of the segment.
template <bool ROW16> __device__ __forceinline__ float seg_scan_add(float v) {
  if constexpr (!ROW16) return half_sum_dpp(v);
  int x = __builtin_bit_cast(int, v);
#define CNR_DPP_ADDR(CTRL)                                                                             \
  x = __builtin_bit_cast(int, __builtin_bit_cast(float, x) +                                           \
                                  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true)))
  CNR_DPP_ADDR(0x111); CNR_DPP_ADDR(0x112); CNR_DPP_ADDR(0x114); CNR_DPP_ADDR(0x118);
#undef CNR_DPP_ADDR
  return __builtin_bit_cast(float, x);
}
template <bool ROW16> __device__ __forceinline__ float seg_scan_mul(float v) {
  if constexpr (!ROW16) return half_scan_mul(v);
  int x = __builtin_bit_cast(int, v);
  const int one = 0x3f800000;
#define CNR_DPP_MULR(CTRL)                                                                             \
  x = __builtin_bit_cast(int, __builtin_bit_cast(float, x) *                                           \
                                  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(one, x, CTRL, 0xf, 0xf, false)))
  CNR_DPP_MULR(0x111); CNR_DPP_MULR(0x112); CNR_DPP_MULR(0x114); CNR_DPP_MULR(0x118);
#undef CNR_DPP_MULR
  return __builtin_bit_cast(float, x);
}
template <bool ROW16> __device__ __forceinline__ float seg_total(float scan, int lane) {
  if constexpr (!ROW16) return lane_value(scan, 31);
  const float a = lane_value(scan, 15), b = lane_value(scan, 31);
  return (lane & 16) ? b : a;
}
template <bool ROW16> __device__ __forceinline__ float seg_suffix_add(float v, int lane) {
  if constexpr (!ROW16) return half0_suffix_add(v, lane);
  int x = __builtin_bit_cast(int, v);
#define CNR_DPP_ADDS(CTRL)                                                                             \
  x = __builtin_bit_cast(int, __builtin_bit_cast(float, x) +                                           \
                                  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true)))
  CNR_DPP_ADDS(0x101); CNR_DPP_ADDS(0x102); CNR_DPP_ADDS(0x104); CNR_DPP_ADDS(0x108);
#undef CNR_DPP_ADDS
  return __builtin_bit_cast(float, x);
}
// lanes 32..63 := lanes 0..31 (v_permlane32_swap, gfx950)
__device__ __forceinline__ float low_half_to_both(float v) {
  const unsigned x = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]);
}

// Fixed-order sum of the per-workgroup records into the (accumulated) outputs.  256 threads = 64 record entries x
// 4 quarters of the workgroup range; each quarter keeps 8 loads in flight, the quarters are combined through LDS in
// a fixed order.  Entries no launch writes (latent-layer biases, padding, unused row sums) are skipped, so the
// workspace needs no clearing.
__global__ __launch_bounds__(256) void reduce_records_kernel(const rec_t* __restrict__ records, int nwg,
                                                             float* __restrict__ dtrunk, float* __restrict__ dB,
                                                             float* __restrict__ dbiasrows, int rows_per_class,
                                                             int64_t st_trunk, int64_t st_B) {
  __shared__ float part[4][64];
  const int c = blockIdx.y;
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + e;
  const bool live = i < REC_ENTRIES && rec_entry_written(i, rows_per_class);
  float s = 0.0f;
  if (live) {
    const int per = (nwg + 3) / 4, w0 = q * per, w1 = min(nwg, w0 + per);
    const rec_t* r = records + (size_t)c * nwg * REC_ENTRIES + i;
    s = cnr_rec::record_range_sum(r, w0, w1);
  }
  part[q][e] = s;
  __syncthreads();
  if (q == 0 && live) {
    const float v = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
    if (i < TRUNK) dtrunk[(size_t)c * st_trunk + i] += v;
    else if (i < TRUNK + 63) atomicAdd(&dB[(size_t)c * st_B + (i - TRUNK)], v);        // two addends per element
    else if (i < TRUNK + 126) atomicAdd(&dB[(size_t)c * st_B + (i - TRUNK - 63)], v);
    else dbiasrows[(size_t)c * rows_per_class * 128 + (i - (TRUNK + 126))] += v;
  }
}

// ---- the pipelined kernel's helpers (fused_bwd_pipe8.hip) ----------------------------------------------------------
template <int V> using IC = std::integral_constant<int, V>;
enum Step { ST_R2, ST_R0, ST_T1, ST_VD, ST_ES, ST_S2, ST_CAT, ST_S1, ST_XYZ, NSTEPS };

// f16 pack of the 8 accumulator registers of k-step s, zeroed where the (post-ReLU, hence non-negative) activation
// is zero.  Two packed-integer VALU ops per register pair -- min(act, 1) per half, then 0 - that = 0xFFFF / 0 -- and an
// AND.  Written with vector types and an OPAQUE constant: with a visible 1 the optimiser rewrites min/negate into a
// compare + select per element (9 instructions and VCC wait states per pair; these masks are on every backward step's
// critical path).  NOT inline-asm instructions: hipcc pads no hazards for an asm statement, and its register allocator
// hands an asm output the register an MFMA issued just before is still reading as its A operand -- seen as run-to-run
// different colour-branch gradients (25 % off) in one scheduling of the 8-wave kernel, while every test passed in the
// others.
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
// 0xFFFF per half where the activation is non-zero.  Only depends on the forward: the backward steps compute the
// mask of the NEXT step before their barrier, so that after the barrier a step is MFMA -> cvt -> and -> store.
__device__ __forceinline__ u4v relu_mask(const h8& act) {
  unsigned int ones = 0x00010001u;
  asm volatile("" : "+s"(ones));  // no instruction: only hides the value from the optimiser
  const u4v o4 = {ones, ones, ones, ones};
  const us8 one = __builtin_bit_cast(us8, o4);
  const us8 t = __builtin_elementwise_min(__builtin_bit_cast(us8, act), one);
  return __builtin_bit_cast(u4v, (us8)((us8)(0) - t));
}
__device__ __forceinline__ h8 pack8_and(const f16v& a, int s, const u4v& m) {
  // (pair by pair on 32-bit words: through an h8 and back the compiler re-assembled every word from its halves -- one v_bfi x, x
  //  per register, 36 per tile in the ISA)
  u4v r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f2 v = {a[8 * s + 2 * j], a[8 * s + 2 * j + 1]};
    r[j] = __builtin_bit_cast(unsigned int, __builtin_convertvector(v, h2)) & m[j];
  }
  return __builtin_bit_cast(h8, r);
}
__device__ __forceinline__ h8 pack8_masked(const f16v& a, int s, const h8& act) { return pack8_and(a, s, relu_mask(act)); }
__device__ __forceinline__ void role_barrier() {  // every wave of the workgroup executes the same number of these
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

}  // namespace
