// Shared by latent.hip (cnr_latent_fwd / cnr_latent_bwd) and fused_fwd.hip (cnr_param_prep): the flat parameter
// row layout of the fused trainer and the per-(object, latent layer) forward block.
#pragma once
#include "cnr_common.h"

namespace cnr {
struct FlatLayout {
  int64_t stride;  // floats per class row
  int64_t latW, latb, shape, tex;
  int L, n_obj;
};
__device__ __forceinline__ void latent_target(int k, int& w_off, int& b_off, int& ld) {
  if (k == 0) { w_off = OFF_S1_W; b_off = OFF_S1_B; ld = 32; }
  else if (k == 1) { w_off = OFF_CAT_W; b_off = OFF_CAT_B; ld = 32 + E1; }
  else if (k == 2) { w_off = OFF_S2_W; b_off = OFF_S2_B; ld = 32; }
  else { w_off = OFF_T1_W; b_off = OFF_T1_B; ld = 32; }
}

// One 256-thread block per (object, slot k): a wave per output for the L-long dot product (coalesced weight rows,
// the 8 outputs of a wave unrolled so that their loads are in flight together), then the 32x32 product with the
// trunk layer: z_k = relu(Wl_k code + bl_k), biasrows[row][k] = Wt_k[:, :32] z_k + bt_k.
// th = this class's parameter row, trunk = its trunk blob (th itself when the trunk sits at offset 0).
__device__ __forceinline__ void latent_fwd_block(const float* __restrict__ th, const float* __restrict__ trunk,
                                                 const FlatLayout& lay, float* __restrict__ zl,
                                                 float* __restrict__ biasrows, int obj, int k, int c) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* code = th + (k == 3 ? lay.tex : lay.shape) + (int64_t)obj * lay.L;
  __shared__ float zs[32];
  // second-phase operands do not depend on the first phase: fetch them now, ahead of the barrier
  int w_off, b_off, ld;
  latent_target(k, w_off, b_off, ld);
  const int o2 = threadIdx.x >> 3, p2 = threadIdx.x & 7;
  float tw[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) tw[q] = trunk[w_off + o2 * ld + p2 * 4 + q];
  const float tb = trunk[b_off + o2];
  // all 4 + 32 loads of a wave (its code slice and the matching slices of its 8 weight rows) are issued before the
  // first use: one memory round trip instead of one per output (L <= 256 unrolled; longer codes take the loop)
  float cv[4], acc[8], lb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) lb[i] = th[lay.latb + k * 32 + wv + 4 * i];
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int l = lane + 64 * j; cv[j] = l < lay.L ? code[l] : 0.0f; }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int o = wv + 4 * i;
    const float* w = th + lay.latW + ((int64_t)k * 32 + o) * lay.L;
    float wl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int l = lane + 64 * j; wl[j] = l < lay.L ? w[l] : 0.0f; }
    float a = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) a = fmaf(cv[j], wl[j], a);
    for (int l = lane + 256; l < lay.L; l += 64) a = fmaf(code[l], w[l], a);
    acc[i] = a;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int o = wv + 4 * i;
    const float a = wave_sum(acc[i]);
    if (lane == 0) zs[o] = fmaxf(a + lb[i], 0.0f);
  }
  __syncthreads();
  {  // 32 outputs x 32 inputs on 256 threads: 8 lanes per output, 4 inputs each, then an 8-lane sum
    const int o = o2, p = p2;
    float br = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) br = fmaf(tw[q], zs[p * 4 + q], br);
    br += xor_lane(br, 1, lane);
    br += xor_lane(br, 2, lane);
    br += xor_lane(br, 4, lane);
    if (p == 0) {
      const int64_t row = (int64_t)c * lay.n_obj + obj;
      zl[(row * 4 + k) * 32 + o] = zs[o];
      biasrows[(row * 4 + k) * 32 + o] = br + tb;
    }
  }
}

// Backward of the latent path for one class (th = its parameter row, z / dbr = its (n_obj,4,32) post-ReLU activations
// and bias-row gradients).  Every block recomputes the small d pre table (n_obj x 128 values) and the code norms into
// `sm` (n_obj * 128 + 2 n_obj floats of LDS), then takes its grid-stride share of the concatenated output space
//   [ d Wt_k[:, :32] and d bt_k : 4*32*33 | d Wl : 4*32*L | d bl : 128 | d shape codes : n_obj*L | d tex codes : n_obj*L ]
// and hands each value to the sink: trunk_add(index in the class row, v) for the first group (skipped when
// with_trunk is false), latent_set(index, v) for the rest -- after prefetch(index) at the start of the element's work,
// where a sink that needs more than the gradient (AdamW: parameter and moments) can put its loads in flight.  reg_scale * code / ||code|| (src/loss.py:5-15) is
// included in the code gradients.
template <class Sink>
__device__ __forceinline__ void latent_bwd_block(const float* __restrict__ th, const FlatLayout& lay,
                                                 const float* __restrict__ z, const float* __restrict__ dbr,
                                                 float reg_scale, float* sm, const Sink& sink, int blk, int nblk,
                                                 bool with_trunk, int n_real = 1 << 30) {
  // n_real: objects the class really has (rows >= n_real are padding of a class with fewer objects than the layout's
  // n_obj: no regulariser there; their bias-row gradients are zero because no ray refers to them)
  const int n_obj = lay.n_obj, L = lay.L;
  float* dpre = sm;
  float* inv_s = sm + n_obj * 128;
  float* inv_t = inv_s + n_obj;
  // (two entries / four code rows at a time with all their loads in flight: a class of 5-15 objects walks these loops several
  //  times, and one memory round trip per pass was 8 us of the tail launch at seven objects; same arithmetic, same order)
  for (int i0 = threadIdx.x; i0 < n_obj * 128; i0 += 2 * 256) {   // d z -> d pre
    float wv_[2][32], zv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = i0 + 256 * u < n_obj * 128 ? i0 + 256 * u : i0;
      const int k = (i >> 5) & 3, j = i & 31;
      int w_off, b_off, ld;
      latent_target(k, w_off, b_off, ld);
#pragma unroll
      for (int o = 0; o < 32; ++o) wv_[u][o] = th[w_off + o * ld + j];
      zv[u] = z[i];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = i0 + 256 * u;
      if (i < n_obj * 128) {
        const int ob = i >> 7, k = (i >> 5) & 3;
        float s = 0.0f;
#pragma unroll
        for (int o = 0; o < 32; ++o) s = fmaf(dbr[(ob * 4 + k) * 32 + o], wv_[u][o], s);
        dpre[i] = zv[u] > 0.0f ? s : 0.0f;
      }
    }
  }
  {  // code norms for the regulariser: one wave per code row
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int t0 = wv; t0 < 2 * n_obj; t0 += 16) {
      float cv[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + 4 * u < 2 * n_obj ? t0 + 4 * u : t0;
        const float* code = th + (t < n_obj ? lay.shape : lay.tex) + (int64_t)(t % n_obj) * L;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int l = lane + 64 * j; cv[u][j] = l < L ? code[l] : 0.0f; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + 4 * u;
        if (t < 2 * n_obj) {     // (wave-uniform)
          const int ob = t % n_obj;
          const float* code = th + (t < n_obj ? lay.shape : lay.tex) + (int64_t)ob * L;
          float s = 0.0f;
#pragma unroll
          for (int j = 0; j < 4; ++j) s = fmaf(cv[u][j], cv[u][j], s);
          for (int l = lane + 256; l < L; l += 64) s = fmaf(code[l], code[l], s);
          s = wave_sum(s);
          if (lane == 0) (t < n_obj ? inv_s : inv_t)[ob] = ob < n_real ? reg_scale / sqrtf(s) : 0.0f;
        }
      }
    }
  }
  __syncthreads();
  const int n0 = 4 * 32 * 33, n1 = n0 + 4 * 32 * L, n2 = n1 + 128, n3 = n2 + n_obj * L, n4 = n3 + n_obj * L;
  for (int t = (with_trunk ? 0 : n0) + blk * 256 + threadIdx.x; t < n4; t += nblk * 256) {
    if (t < n0) {  // added: the field backward wrote the a-part of these weights already
      const int k = t / (32 * 33), r = t % (32 * 33), o = r / 33, j = r % 33;
      int w_off, b_off, ld;
      latent_target(k, w_off, b_off, ld);
      float s = 0.0f;
      for (int ob = 0; ob < n_obj; ++ob) {
        const float d = dbr[(ob * 4 + k) * 32 + o];
        s += j < 32 ? d * z[(ob * 4 + k) * 32 + j] : d;
      }
      sink.trunk_add(j < 32 ? w_off + o * ld + j : b_off + o, s);
    } else if (t < n1) {
      const int i = t - n0, k = i / (32 * L), r = i % (32 * L), o = r / L, l = r % L;
      sink.prefetch(lay.latW + i);
      float s = 0.0f;
      const float* cbase = th + (k == 3 ? lay.tex : lay.shape) + l;
      for (int ob0 = 0; ob0 < n_obj; ob0 += 4) {   // four objects' loads in flight together (same addition order)
        float cv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) cv[u] = ob0 + u < n_obj ? cbase[(int64_t)(ob0 + u) * L] : 0.0f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ob0 + u < n_obj) s = fmaf(dpre[((ob0 + u) * 4 + k) * 32 + o], cv[u], s);
      }
      sink.latent_set(lay.latW + i, s);
    } else if (t < n2) {
      const int i = t - n1;
      sink.prefetch(lay.latb + i);
      float s = 0.0f;
      for (int ob = 0; ob < n_obj; ++ob) s += dpre[ob * 128 + i];
      sink.latent_set(lay.latb + i, s);
    } else if (t < n3) {
      const int i = t - n2, ob = i / L, l = i % L;
      sink.prefetch(lay.shape + i);
      float s = 0.0f;
      const float own = th[lay.shape + i];
#pragma unroll
      for (int k0 = 0; k0 < 96; k0 += 32) {   // 32 weight loads in flight at a time (same additions, same order)
        float w[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) w[u] = th[lay.latW + (int64_t)(k0 + u) * L + l];
#pragma unroll
        for (int u = 0; u < 32; ++u) s = fmaf(dpre[ob * 128 + k0 + u], w[u], s);
      }
      sink.latent_set(lay.shape + i, s + inv_s[ob] * own);
    } else {
      const int i = t - n3, ob = i / L, l = i % L;
      sink.prefetch(lay.tex + i);
      float s = 0.0f, w[32];
      const float own = th[lay.tex + i];
#pragma unroll
      for (int o = 0; o < 32; ++o) w[o] = th[lay.latW + (int64_t)(96 + o) * L + l];
#pragma unroll
      for (int o = 0; o < 32; ++o) s = fmaf(dpre[ob * 128 + 96 + o], w[o], s);
      sink.latent_set(lay.tex + i, s + inv_t[ob] * own);
    }
  }
}
// The same block for the step's last launch (tail.hip), n_obj <= 4 and L <= 256, ONE trip per block, no trunk
// group: latent_bwd_block spends its time in four dependent memory round trips (bias-row sums -> d pre operands ->
// code norms -> the element's own operands), each behind a barrier.  None of those loads depends on another one's
// VALUE, so here all of them are issued before the first barrier and the block pays one round trip.
// Same expressions in the same order as latent_bwd_block: same bits.
//   rows_fix_c: this class's (n_obj,4,32) entries of copy 0 of the fixed-point bias-row table, copy_stride entries
//   between copies; rows_out (block 0 only, may be null): the float rows are also published there.
template <class Sink>
__device__ __forceinline__ void latent_bwd_block_1trip(const float* __restrict__ th, const FlatLayout& lay,
                                                       const float* __restrict__ z,
                                                       const long long* __restrict__ rows_fix_c, int64_t copy_stride,
                                                       int ncopies, double fix_inv_scale, float* __restrict__ rows_out,
                                                       float reg_scale, float* sm, const Sink& sink, int blk,
                                                       int n_real = 1 << 30) {
  const int n_obj = lay.n_obj, L = lay.L, nrow = n_obj * 128;
  float* rows = sm;                 // [n_obj * 128] d biasrows
  float* dpre = rows + nrow;        // [n_obj * 128]
  float* inv_s = dpre + nrow;       // [n_obj]
  float* inv_t = inv_s + n_obj;     // [n_obj]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // ---- phase 0: every global load of the block -------------------------------------------------------------
  long long fx[2][8];
  float thv[2][32], zv[2];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const int i = threadIdx.x + 256 * sl;
    const bool on = i < nrow;
    const int ii = on ? i : 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) fx[sl][k] = (on && k < ncopies) ? rows_fix_c[(int64_t)k * copy_stride + ii] : 0;
    const int kk = (ii >> 5) & 3, j = ii & 31;
    int w_off, b_off, ld;
    latent_target(kk, w_off, b_off, ld);
#pragma unroll
    for (int o = 0; o < 32; ++o) thv[sl][o] = th[w_off + o * ld + j];
    zv[sl] = z[ii];
  }
  float cvn[2][4];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int t = wv + 4 * r;
    const bool on = t < 2 * n_obj;
    const int ob = on ? t % n_obj : 0;
    const float* code = th + ((on && t >= n_obj) ? lay.tex : lay.shape) + (int64_t)ob * L;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int l = lane + 64 * j; cvn[r][j] = (on && l < L) ? code[l] : 0.0f; }
  }
  const int n0 = 4 * 32 * 33, n1 = n0 + 4 * 32 * L, n2 = n1 + 128, n3 = n2 + n_obj * L, n4 = n3 + n_obj * L;
  const int t = n0 + blk * 256 + threadIdx.x;
  float opv[96];  // the element's own operands (which, depends on its group)
  float own = 0.0f;
  if (t < n1) {
    const int i = t - n0, k = i / (32 * L), l = (i % (32 * L)) % L;
    sink.prefetch(lay.latW + i);
    const float* cbase = th + (k == 3 ? lay.tex : lay.shape) + l;
#pragma unroll
    for (int u = 0; u < 4; ++u) opv[u] = u < n_obj ? cbase[(int64_t)u * L] : 0.0f;
  } else if (t < n2) {
    sink.prefetch(lay.latb + (t - n1));
  } else if (t < n3) {
    const int i = t - n2, l = i % L;
    sink.prefetch(lay.shape + i);
#pragma unroll
    for (int ko = 0; ko < 96; ++ko) opv[ko] = th[lay.latW + (int64_t)ko * L + l];
    own = th[lay.shape + i];
  } else if (t < n4) {
    const int i = t - n3, l = i % L;
    sink.prefetch(lay.tex + i);
#pragma unroll
    for (int o = 0; o < 32; ++o) opv[o] = th[lay.latW + (int64_t)(96 + o) * L + l];
    own = th[lay.tex + i];
  }
  // ---- phase 1: bias-row sums as floats ----------------------------------------------------------------------
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const int i = threadIdx.x + 256 * sl;
    if (i < nrow) {
      long long tt = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) tt += fx[sl][k];
      const float v = (float)((double)tt * fix_inv_scale);
      rows[i] = v;
      if (rows_out) rows_out[i] = v;
    }
  }
  __syncthreads();
  // ---- phase 2: d z -> d pre ; code norms ---------------------------------------------------------------------
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const int i = threadIdx.x + 256 * sl;
    if (i < nrow) {
      const int ob = i >> 7, k = (i >> 5) & 3;
      float s = 0.0f;
#pragma unroll
      for (int o = 0; o < 32; ++o) s = fmaf(rows[(ob * 4 + k) * 32 + o], thv[sl][o], s);
      dpre[i] = zv[sl] > 0.0f ? s : 0.0f;
    }
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int tt = wv + 4 * r;
    if (tt < 2 * n_obj) {
      float s = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s = fmaf(cvn[r][j], cvn[r][j], s);
      s = wave_sum(s);
      if (lane == 0) (tt < n_obj ? inv_s : inv_t)[tt % n_obj] = (tt % n_obj) < n_real ? reg_scale / sqrtf(s) : 0.0f;
    }
  }
  __syncthreads();
  // ---- phase 3: this thread's element ---------------------------------------------------------------------------
  if (t < n1) {
    const int i = t - n0, k = i / (32 * L), o = (i % (32 * L)) / L;
    float s = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < n_obj) s = fmaf(dpre[(u * 4 + k) * 32 + o], opv[u], s);
    sink.latent_set(lay.latW + i, s);
  } else if (t < n2) {
    const int i = t - n1;
    float s = 0.0f;
    for (int ob = 0; ob < n_obj; ++ob) s += dpre[ob * 128 + i];
    sink.latent_set(lay.latb + i, s);
  } else if (t < n3) {
    const int i = t - n2, ob = i / L;
    float s = 0.0f;
#pragma unroll
    for (int ko = 0; ko < 96; ++ko) s = fmaf(dpre[ob * 128 + ko], opv[ko], s);
    sink.latent_set(lay.shape + i, s + inv_s[ob] * own);
  } else if (t < n4) {
    const int i = t - n3, ob = i / L;
    float s = 0.0f;
#pragma unroll
    for (int o = 0; o < 32; ++o) s = fmaf(dpre[ob * 128 + 96 + o], opv[o], s);
    sink.latent_set(lay.tex + i, s + inv_t[ob] * own);
  }
}

// ---- the tail launch's latent blocks for classes of MORE than four objects ---------------------------------------------------
// latent_bwd_block lets every block rebuild the whole d pre table (n_obj x 128 dot products of 32) and walk the fixed-point
// table in full: its time grows by ~0.85 us per object (tail launch 10 -> 24 us from 4 to 20 objects).  Here a block builds only
// the d pre entries ITS elements read, from only the table rows those need, with every global load issued before the first
// barrier -- one memory round trip per block whatever the object count (like latent_bwd_block_1trip, which covers <= 4 objects):
//   * a block of 256 consecutive d Wl elements spans max(1, 256 / L) rows (k, o) of the latent weights and reads, per row,
//     d pre[ob][k][o] of every object: n_obj dot products per row.  The thread that owns a row's first element also forms
//     d bl[k][o] = sum over objects of the same entries (there is no separate bias block);
//   * a block of 256 consecutive code-gradient elements spans max(1, 256 / L) objects and reads d pre[ob][.] of those only.
// Blocks are dealt per group (NW = ceil(128 L / 256) weight blocks, then NS = ceil(n_obj L / 256) shape and NS texture blocks),
// so a block never straddles two groups.  Same expressions in the same order as latent_bwd_block: same bits.
__host__ __device__ inline bool latent_local_ok(int L, int n_obj) {
  if (L <= 0 || n_obj <= 0) return false;
  if (!(256 % L == 0 || L % 256 == 0)) return false;      // rows / objects per block: a whole number, block-aligned
  const int per = L >= 256 ? 1 : 256 / L;
  return n_obj <= 128 && per <= 8;     // (up to cnr_rec::ROWS_TILE_MAX objects: object loops in chunks of 32, d pre entries in passes)
}
// floats of dynamic LDS a block of latent_bwd_block_local needs
__host__ __device__ inline int latent_local_lds_floats(int L, int n_obj) {
  const int per = L >= 256 ? 1 : 256 / L;
  const int wblk = 2 * n_obj * 32 + per * n_obj;          // weight block: <= 2 latent slots of table rows + its rows' d pre
  const int cblk = 2 * per * 96 + per;                    // code block: its objects' rows + d pre + norms
  return wblk > cblk ? wblk : cblk;
}
__host__ __device__ inline int latent_local_blocks(int L, int n_obj) {
  return (128 * L + 255) / 256 + 2 * ((n_obj * L + 255) / 256);
}
template <class Sink>
__device__ __forceinline__ void latent_bwd_block_local(const float* __restrict__ th, const FlatLayout& lay,
                                                       const float* __restrict__ z,
                                                       const long long* __restrict__ rows_fix_c, int64_t copy_stride,
                                                       int ncopies, double fix_inv_scale, float reg_scale, float* sm,
                                                       const Sink& sink, int blk, int n_real = 1 << 30) {
  const int n_obj = lay.n_obj, L = lay.L;
  const int NW = (128 * L + 255) / 256, NS = (n_obj * L + 255) / 256;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float* rowsL = sm;                               // table rows this block needs, as floats
  auto row_entry = [&](int idx, long long (&f)[8]) {   // the 8 copies of table entry idx (this class), all in flight
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = k < ncopies ? rows_fix_c[(int64_t)k * copy_stride + idx] : 0;
  };
  auto row_value = [&](const long long (&f)[8]) {
    long long t = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += f[k];
    return (float)((double)t * fix_inv_scale);
  };
  if (blk < NW) {
    // ================================ d Wl (and d bl) ================================
    const int e0 = blk * 256, e = e0 + tid;
    const bool live = e < 128 * L;
    const int p_lo = e0 / L, p_hi = min((e0 + 255) / L, 127), np = p_hi - p_lo + 1;   // rows (k, o) = p / 32, p % 32
    const int k_lo = p_lo >> 5, nk = (p_hi >> 5) - k_lo + 1;                             // <= 2 latent slots
    float* dsel = rowsL + nk * n_obj * 32;            // [np][n_obj] d pre of the block's rows
    // ---- every global load of the block ----
    // (a) table rows (ob, k) for k_lo .. : nk * n_obj * 32 entries, thread r takes entries r, r + 256, ..
    const int nrow = nk * n_obj * 32;
    long long fx[4][8];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      const int r = tid + 256 * sl;
      const int rr = r < nrow ? r : 0;
      const int kk = rr / (n_obj * 32), ob = (rr / 32) % n_obj, o_ = rr & 31;
      row_entry((ob * 4 + k_lo + kk) * 32 + o_, fx[sl]);
    }
    // (b) the d pre entries this thread forms: q = row_local * n_obj + ob, q = tid, tid + 256, ..  (np * n_obj entries; the first
    //     one's operands are requested here with everything else, later passes -- more than 256 / np objects -- fetch their own)
    const int nq = np * n_obj;
    auto dpre_operands = [&](int q, float (&wq_)[32], float& zq_, int& qk_, int& qob_) {
      const int qq = q < nq ? q : 0, qrow = qq / n_obj;
      qob_ = qq - qrow * n_obj;
      const int qp = p_lo + qrow, qo = qp & 31;
      qk_ = qp >> 5;
      int w_off, b_off, ld;
      latent_target(qk_, w_off, b_off, ld);
#pragma unroll
      for (int o_ = 0; o_ < 32; ++o_) wq_[o_] = th[w_off + o_ * ld + qo];
      zq_ = z[(qob_ * 4 + qk_) * 32 + qo];
    };
    float wq[32], zq;
    int qk, qob;
    dpre_operands(tid, wq, zq, qk, qob);
    // (c) this thread's element: code entries of every object at its l (the first 32 objects' here, further chunks below)
    const int ee = live ? e : 0, ep = ee / L, el = ee - ep * L, ek = ep >> 5;
    const float* cbase = th + (ek == 3 ? lay.tex : lay.shape) + el;
    float cv[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) cv[u] = u < n_obj ? cbase[(int64_t)u * L] : 0.0f;
    if (live) sink.prefetch(lay.latW + e);
    // ---- table rows -> floats in LDS ----
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      const int r = tid + 256 * sl;
      if (r < nrow) rowsL[r] = row_value(fx[sl]);
    }
    for (int r = tid + 1024; r < nrow; r += 256) {     // (more than 32 objects' worth of rows: further passes, one round trip each)
      const int kk = r / (n_obj * 32), ob = (r / 32) % n_obj, o_ = r & 31;
      long long f[8];
      row_entry((ob * 4 + k_lo + kk) * 32 + o_, f);
      rowsL[r] = row_value(f);
    }
    __syncthreads();
    // ---- d pre of the block's rows ----
    for (int q = tid; q < nq; q += 256) {
      if (q != tid) dpre_operands(q, wq, zq, qk, qob);
      const float* rr = rowsL + ((qk - k_lo) * n_obj + qob) * 32;
      float s_ = 0.0f;
#pragma unroll
      for (int o_ = 0; o_ < 32; ++o_) s_ = fmaf(rr[o_], wq[o_], s_);
      dsel[q] = zq > 0.0f ? s_ : 0.0f;
    }
    __syncthreads();
    // ---- the element, then (first element of a row) the row's bias ----
    if (live) {
      const float* dp = dsel + (ep - p_lo) * n_obj;
      float s_ = 0.0f;
#pragma unroll
      for (int u = 0; u < 32; ++u)
        if (u < n_obj) s_ = fmaf(dp[u], cv[u], s_);
      for (int u0 = 32; u0 < n_obj; u0 += 32) {   // objects 32 .. : the same chain, 32 code entries in flight at a time
#pragma unroll
        for (int u = 0; u < 32; ++u) cv[u] = u0 + u < n_obj ? cbase[(int64_t)(u0 + u) * L] : 0.0f;
#pragma unroll
        for (int u = 0; u < 32; ++u)
          if (u0 + u < n_obj) s_ = fmaf(dp[u0 + u], cv[u], s_);
      }
      sink.latent_set(lay.latW + e, s_);
      if (el == 0) {
        sink.prefetch(lay.latb + ep);
        float b_ = 0.0f;
        for (int ob = 0; ob < n_obj; ++ob) b_ += dp[ob];
        sink.latent_set(lay.latb + ep, b_);
      }
    }
    return;
  }
  // ================================ code tables ================================
  const bool is_tex = blk >= NW + NS;
  const int bc = blk - NW - (is_tex ? NS : 0);
  const int i0 = bc * 256, i = i0 + tid;
  const bool live = i < n_obj * L;
  const int ob_lo = i0 / L, ob_hi = min((i0 + 255) / L, n_obj - 1), nob = ob_hi - ob_lo + 1;   // objects of this block (<= per)
  const int NK = is_tex ? 32 : 96, K0 = is_tex ? 96 : 0;       // d pre entries per object this group reads: [K0, K0 + NK)
  float* dsel = rowsL + nob * NK;                   // [nob][NK]
  float* inv_n = dsel + nob * NK;                   // [nob] reg_scale / ||code||
  // ---- every global load of the block ----
  // (a) table rows: entries (ob, K0 + j) for the block's objects
  const int nrow = nob * NK;                        // <= 8 * 96 = 768
  long long fx[3][8];
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) {
    const int r = tid + 256 * sl;
    const int rr = r < nrow ? r : 0;
    const int ob = ob_lo + rr / NK, j = rr % NK;
    row_entry(ob * 128 + K0 + j, fx[sl]);
  }
  // (b) code norms: wave wv takes object ob_lo + wv (and + 4): the same loads and sums as latent_bwd_block
  float cvn[2][4];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int t = wv + 4 * r;
    const bool on = t < nob;
    const float* code = th + (is_tex ? lay.tex : lay.shape) + (int64_t)(ob_lo + (on ? t : 0)) * L;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int l = lane + 64 * j; cvn[r][j] = (on && l < L) ? code[l] : 0.0f; }
  }
  // (c) this thread's element: the 96 (32) weights of its l, its own code entry
  const int ii = live ? i : 0, eob = ii / L, el = ii - eob * L;
  float opv[96];
#pragma unroll
  for (int ko = 0; ko < 96; ++ko) opv[ko] = ko < NK ? th[lay.latW + (int64_t)(K0 + ko) * L + el] : 0.0f;
  const float own = th[(is_tex ? lay.tex : lay.shape) + ii];
  if (live) sink.prefetch((is_tex ? lay.tex : lay.shape) + i);
  // (d) the trunk-weight column and activation of the FIRST d pre entry this thread forms (entries tid, tid + 256, ..): with
  //     one object per block (L = 256) that is all of them
  float wq0[32], zq0;
  {
    const int q = tid < nrow ? tid : 0;
    const int jj = K0 + q % NK, k = jj >> 5, j = jj & 31;
    int w_off, b_off, ld;
    latent_target(k, w_off, b_off, ld);
#pragma unroll
    for (int o_ = 0; o_ < 32; ++o_) wq0[o_] = th[w_off + o_ * ld + j];
    zq0 = z[(ob_lo + q / NK) * 128 + jj];
  }
  // ---- table rows -> LDS ----
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) {
    const int r = tid + 256 * sl;
    if (r < nrow) rowsL[r] = row_value(fx[sl]);
  }
  // ---- code norms ----
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int t = wv + 4 * r;
    if (t < nob) {      // (wave-uniform)
      const int ob = ob_lo + t;
      const float* code = th + (is_tex ? lay.tex : lay.shape) + (int64_t)ob * L;
      float s_ = 0.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s_ = fmaf(cvn[r][j], cvn[r][j], s_);
      for (int l = lane + 256; l < L; l += 64) s_ = fmaf(code[l], code[l], s_);
      s_ = wave_sum(s_);
      if (lane == 0) inv_n[t] = ob < n_real ? reg_scale / sqrtf(s_) : 0.0f;
    }
  }
  __syncthreads();
  // ---- d pre entries (ob, K0 + j): dot of the object's slot-k table row with column j of the slot's trunk weights ----
  for (int q = tid; q < nrow; q += 256) {
    const int t = q / NK, jj = K0 + q % NK, k = jj >> 5, j = jj & 31;
    float wq[32], zq;
    if (q == tid) {
#pragma unroll
      for (int o_ = 0; o_ < 32; ++o_) wq[o_] = wq0[o_];
      zq = zq0;
    } else {
      int w_off, b_off, ld;
      latent_target(k, w_off, b_off, ld);
#pragma unroll
      for (int o_ = 0; o_ < 32; ++o_) wq[o_] = th[w_off + o_ * ld + j];
      zq = z[(ob_lo + t) * 128 + jj];
    }
    const float* rr = rowsL + t * NK + (k - (K0 >> 5)) * 32;
    float s_ = 0.0f;
#pragma unroll
    for (int o_ = 0; o_ < 32; ++o_) s_ = fmaf(rr[o_], wq[o_], s_);
    dsel[q] = zq > 0.0f ? s_ : 0.0f;
  }
  __syncthreads();
  if (live) {
    const float* dp = dsel + (eob - ob_lo) * NK;
    float s_ = 0.0f;
#pragma unroll
    for (int ko = 0; ko < 96; ++ko)
      if (ko < NK) s_ = fmaf(dp[ko], opv[ko], s_);
    sink.latent_set((is_tex ? lay.tex : lay.shape) + i, s_ + inv_n[eob - ob_lo] * own);
  }
}

// the trunk-entry term of that first group for ONE trunk index q of a class row (0 when q is not a latent-target
// weight / bias): what latent_bwd_block would have added there.  For consumers that finish the trunk gradient
// themselves (tail.hip).
// which (latent slot k, output o, input j; j = 32: the bias) trunk index q of a class row is, false when the latent path adds
// nothing there
__device__ __forceinline__ bool latent_trunk_index(int q, int& k, int& o, int& j) {
  if (q >= OFF_S1_W && q < OFF_S1_W + 1024) { k = 0; o = (q - OFF_S1_W) >> 5; j = (q - OFF_S1_W) & 31; }
  else if (q >= OFF_S1_B && q < OFF_S1_B + 32) { k = 0; o = q - OFF_S1_B; j = 32; }
  else if (q >= OFF_CAT_W && q < OFF_CAT_W + 32 * (32 + E1)) {
    k = 1; o = (q - OFF_CAT_W) / (32 + E1); j = (q - OFF_CAT_W) % (32 + E1);
    if (j >= 32) return false;
  }
  else if (q >= OFF_CAT_B && q < OFF_CAT_B + 32) { k = 1; o = q - OFF_CAT_B; j = 32; }
  else if (q >= OFF_S2_W && q < OFF_S2_W + 1024) { k = 2; o = (q - OFF_S2_W) >> 5; j = (q - OFF_S2_W) & 31; }
  else if (q >= OFF_S2_B && q < OFF_S2_B + 32) { k = 2; o = q - OFF_S2_B; j = 32; }
  else if (q >= OFF_T1_W && q < OFF_T1_W + 1024) { k = 3; o = (q - OFF_T1_W) >> 5; j = (q - OFF_T1_W) & 31; }
  else if (q >= OFF_T1_B && q < OFF_T1_B + 32) { k = 3; o = q - OFF_T1_B; j = 32; }
  else return false;
  return true;
}
__device__ __forceinline__ float latent_trunk_term(int q, const float* __restrict__ z, const float* __restrict__ dbr,
                                                   int n_obj) {
  int k, o, j;
  if (!latent_trunk_index(q, k, o, j)) return 0.0f;
  float s = 0.0f;
  for (int ob = 0; ob < n_obj; ++ob) {
    const float d = dbr[(ob * 4 + k) * 32 + o];
    s += j < 32 ? d * z[(ob * 4 + k) * 32 + j] : d;
  }
  return s;
}
}  // namespace cnr
