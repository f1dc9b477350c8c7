// Shared by sample.hip (cnr_sample_rays) and fused_fwd.hip (cnr_step_prologue): a2-a5 for ONE ray on one wavefront.
// See sample.hip for what it restates (src/scene_cateogries.py:24-96, 453-546).
#pragma once
#include "cnr_common.h"

namespace cnr_sample {

struct SampleArgs {
  const uint8_t* rgbs; const float* depth; const float* dirs_c; const float* T; const float* u; const float* g;
  uint64_t seed, offset; const int64_t* d_state; int64_t pool_rows; const float* max_bound; int world_frame;
  int C, R, n1, n2; float eps, stop_eps, min_bound;
  float* z; float* pts; float* origins; float* dirs_o; float* gt_rgb; float* gt_depth; uint8_t* depth_mask;
  uint8_t* labels; const int64_t* pool_indices; int n_obj; int* ray_row; const int* perm;
  // max_bound layout: mb_slices <= 1: one value per class (this step's slice); mb_slices > 1: a (C, mb_slices) table
  // over the epoch's slices, indexed by device cursor / R (filled once per reshuffle by cnr_slice_maxdepth)
  int mb_slices;
  // Philox counter of ray r of local class c: ((c * rng_cstride + rng_c0) * rng_R + rng_r0 + r) * 64 + lane -- the
  // index the ray has in the GLOBAL batch when classes (rng_c0 = first global class, rng_cstride = ranks) or rays
  // (rng_R = global rays per class, rng_r0 = this rank's first ray) are sharded over GPUs, so that N ranks draw
  // exactly what one rank would.  All zero: the local index c * R + r.
  int rng_c0, rng_cstride, rng_R, rng_r0;
};


// ---- Philox4x32-10 (perf mode draws) ---------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}
__device__ __forceinline__ void philox4(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), (uint32_t)ctr_hi, (uint32_t)(ctr_hi >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// torch.linspace(0, 1, n + 1)[i] exactly as ATen computes it on CPU and GPU: step = 1/n, lower half
// counted up from 0, upper half counted down from 1 (scene_cateogries.py:54 calls it per bin count).
__device__ __forceinline__ float linspace01(int i, int n) {
  const float step = 1.0f / (float)n;
  const int steps = n + 1;
  return (i < steps / 2) ? step * (float)i : 1.0f - step * (float)(steps - i - 1);
}

// (cnr::xor_lane, cnr_common.h: lane ^ j on DPP moves for j = 1, 2, 4, 8, ds_bpermute for 16 and 32 -- the network below was 54
//  ds_bpermute per ray, ~3 k cycles of the one wave that samples the ray, on the step's first launch's longest job)
// compare-exchange network over the wave's 64 lanes (one element per lane), ascending
__device__ __forceinline__ void bitonic64(float& v, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const float p = cnr::xor_lane(v, j, lane);
      const bool upper = (lane & j) != 0, asc = (lane & k) == 0;
      v = (upper == asc) ? fmaxf(v, p) : fminf(v, p);
    }
  }
}
// compare-exchange helper for the 128-element bitonic network held as (v0: elem lane, v1: elem lane+64)
__device__ __forceinline__ void bitonic128(float& v0, float& v1, int lane) {
#pragma unroll
  for (int k = 2; k <= 128; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j == 64) {
        // partner is the other register of the same lane; k == 128 here: ascending everywhere
        const float lo = fminf(v0, v1), hi = fmaxf(v0, v1);
        v0 = lo; v1 = hi;
      } else {
        const float p0 = cnr::xor_lane(v0, j, lane);
        const float p1 = cnr::xor_lane(v1, j, lane);
        const bool upper = (lane & j) != 0;
        const bool asc0 = ((lane) & k) == 0;          // element index lane
        const bool asc1 = ((lane + 64) & k) == 0;     // element index lane + 64
        v0 = (upper == asc0) ? fmaxf(v0, p0) : fminf(v0, p0);
        v1 = (upper == asc1) ? fmaxf(v1, p1) : fminf(v1, p1);
      }
    }
  }
}

// one wavefront, one ray (ray < C * R)
__device__ __forceinline__ void sample_ray(const SampleArgs& a, int64_t ray, int lane) {
  const uint8_t* __restrict__ rgbs = a.rgbs; const float* __restrict__ depth = a.depth;
  const float* __restrict__ dirs_c = a.dirs_c; const float* __restrict__ T = a.T;
  const float* __restrict__ u = a.u; const float* __restrict__ g = a.g;
  const uint64_t seed = a.seed; uint64_t offset = a.offset;
  const int64_t* __restrict__ d_state = a.d_state; const int64_t pool_rows = a.pool_rows;
  const float* __restrict__ max_bound = a.max_bound; const int world_frame = a.world_frame;
  const int C = a.C, R = a.R, n1 = a.n1, n2 = a.n2; const float eps = a.eps, stop_eps = a.stop_eps, min_bound = a.min_bound;
  float* __restrict__ z = a.z; float* __restrict__ pts = a.pts; float* __restrict__ origins = a.origins;
  float* __restrict__ dirs_o = a.dirs_o; float* __restrict__ gt_rgb = a.gt_rgb; float* __restrict__ gt_depth = a.gt_depth;
  uint8_t* __restrict__ depth_mask = a.depth_mask; uint8_t* __restrict__ labels = a.labels;
  const int64_t* __restrict__ pool_indices = a.pool_indices; const int n_obj = a.n_obj; int* __restrict__ ray_row = a.ray_row;
  const int* __restrict__ perm = a.perm;
  (void)C;
  const int c = (int)(ray / R);
  const int S = n1 + n2;
  const uint64_t rng_ray = a.rng_R > 0 ? ((uint64_t)c * (uint64_t)(a.rng_cstride > 0 ? a.rng_cstride : 1) + (uint64_t)a.rng_c0) * (uint64_t)a.rng_R
                                             + (uint64_t)a.rng_r0 + (uint64_t)(ray - (int64_t)c * R)
                                       : (uint64_t)ray;
  // pool row of this ray: either the slice itself (pool_rows == 0) or row cursor + r of a device-resident
  // (C, pool_rows, ...) pool whose cursor lives on the device (hipGraph replay advances it, no host work)
  int64_t prow = ray;
  if (pool_rows > 0) {
    prow = (int64_t)c * pool_rows + d_state[0] + (ray - (int64_t)c * R);
    if (perm) prow = (int64_t)c * pool_rows + perm[prow];   // epoch shuffle = a new permutation, the pool stays put
    offset += (uint64_t)d_state[1] * 4;
  }

  // ---- a2: origin / direction in object (or world) frame -----------------------------------
  const float* Tm = T + prow * 16;
  float m00 = Tm[0], m01 = Tm[1], m02 = Tm[2], t0 = Tm[3];
  float m10 = Tm[4], m11 = Tm[5], m12 = Tm[6], t1 = Tm[7];
  float m20 = Tm[8], m21 = Tm[9], m22 = Tm[10], t2 = Tm[11];
  const float dx = dirs_c[prow * 3 + 0], dy = dirs_c[prow * 3 + 1], dz = dirs_c[prow * 3 + 2];
  float ox, oy, oz, ex, ey, ez;
  if (world_frame) {
    ox = t0; oy = t1; oz = t2;
    ex = m00 * dx + m01 * dy + m02 * dz;
    ey = m10 * dx + m11 * dy + m12 * dz;
    ez = m20 * dx + m21 * dy + m22 * dz;
  } else {
    // general 3x3 inverse (adjugate / det); bottom row of T is (0,0,0,1) for the sim3 poses of
    // the path (scene_cateogries.py:234-238), so inv(T) = [A^-1 | -A^-1 t].
    const float c00 = m11 * m22 - m12 * m21, c01 = m02 * m21 - m01 * m22, c02 = m01 * m12 - m02 * m11;
    const float c10 = m12 * m20 - m10 * m22, c11 = m00 * m22 - m02 * m20, c12 = m02 * m10 - m00 * m12;
    const float c20 = m10 * m21 - m11 * m20, c21 = m01 * m20 - m00 * m21, c22 = m00 * m11 - m01 * m10;
    const float idet = 1.0f / (m00 * c00 + m01 * c10 + m02 * c20);
    const float i00 = c00 * idet, i01 = c01 * idet, i02 = c02 * idet;
    const float i10 = c10 * idet, i11 = c11 * idet, i12 = c12 * idet;
    const float i20 = c20 * idet, i21 = c21 * idet, i22 = c22 * idet;
    ox = -(i00 * t0 + i01 * t1 + i02 * t2);
    oy = -(i10 * t0 + i11 * t1 + i12 * t2);
    oz = -(i20 * t0 + i21 * t1 + i22 * t2);
    ex = i00 * dx + i01 * dy + i02 * dz;
    ey = i10 * dx + i11 * dy + i12 * dz;
    ez = i20 * dx + i21 * dy + i22 * dz;
  }

  const float d = depth[prow];
  const uint8_t state = rgbs[prow * 4 + 3];
  const bool invalid = d <= min_bound;
  const bool this_obj = (state == 1) && !invalid;
  if (lane == 0) {
    if (origins) { origins[ray * 3 + 0] = ox; origins[ray * 3 + 1] = oy; origins[ray * 3 + 2] = oz; }
    if (dirs_o) { dirs_o[ray * 3 + 0] = ex; dirs_o[ray * 3 + 1] = ey; dirs_o[ray * 3 + 2] = ez; }
    depth_mask[ray] = invalid ? 0 : 1;
    labels[ray] = state;
    if (ray_row) ray_row[ray] = (int)pool_indices[prow] + c * n_obj;   // row of the class-major code tables
  }
  if (lane < 3) gt_rgb[ray * 3 + lane] = (float)rgbs[prow * 4 + lane] / 255.0f;
  if (lane == 3 && gt_depth) gt_depth[ray] = d;

  // ---- a4: sorted, clipped Gaussian offsets for "this object" rays --------------------------
  float g0 = INFINITY, g1 = INFINITY;  // elements lane and lane+64 of the ray's n2 draws
  if (this_obj) {
    if (g) {
      if (lane < n2) g0 = g[ray * n2 + lane];
      if (lane + 64 < n2) g1 = g[ray * n2 + lane + 64];
    } else {
      uint32_t rnd[4];
      philox4(seed, rng_ray * 64 + lane, offset ^ 0x9E3779B97F4A7C15ull, rnd);
      const float sd = eps / 3.0f;
      const float r0 = sqrtf(-2.0f * __logf(1.0f - u01(rnd[0]))), a0 = 6.28318530718f * u01(rnd[1]);
      const float r1 = sqrtf(-2.0f * __logf(1.0f - u01(rnd[2]))), a1 = 6.28318530718f * u01(rnd[3]);
      if (lane < n2) g0 = sd * r0 * __cosf(a0);
      if (lane + 64 < n2) g1 = sd * r1 * __cosf(a1);
    }
    // (up to 64 draws: the second register is all +inf and stays the network's upper half -- the 64-lane network gives the same order)
    if (n2 <= 64) bitonic64(g0, lane); else bitonic128(g0, g1, lane);
    g0 = fminf(fmaxf(g0, -eps), eps);
    g1 = fminf(fmaxf(g1, -eps), eps);
  }

  // ---- a3/a5: per-column z -------------------------------------------------------------------
  float mb;
  if (max_bound) mb = a.mb_slices > 1 ? max_bound[(int64_t)c * a.mb_slices + (int)(d_state[0] / R)] : max_bound[c];
  else {  // max depth of this step's slice of class c, by this wave (same value as cnr_sample_maxdepth: max is exact)
    float m = -INFINITY;
    const int64_t base = pool_rows > 0 ? (int64_t)c * pool_rows + d_state[0] : (int64_t)c * R;
    for (int r = lane; r < R; r += 64)
      m = fmaxf(m, depth[perm ? (int64_t)c * pool_rows + perm[base + r] : base + r]);
    mb = cnr::wave_max(m);
  }
  // (loop is wave-uniform: every lane takes part in the shuffles, only loads/stores are predicated)
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool live = s < S;
    float uu = 0.0f;
    if (u) {
      if (live) uu = u[ray * S + s];
    } else {
      uint32_t rnd[4];
      philox4(seed, rng_ray * 64 + lane, offset + 1 + (s0 >> 6), rnd);
      uu = u01(rnd[0]);
    }
    const int i = s - n1;
    const int isrc = i < 0 ? 0 : i;
    const float ga = __shfl(g0, isrc & 63, 64), gb = __shfl(g1, isrc & 63, 64);
    float zz;
    if (invalid) {
      // stratified(min_bound, max_bound, n1+n2)
      const float range = mb - min_bound;
      zz = (range * linspace01(s, S) + min_bound) + uu * (range / (float)S);
    } else if (s < n1) {
      const float range = (d - eps) - min_bound;
      zz = (range * linspace01(s, n1) + min_bound) + uu * (range / (float)n1);
    } else if (this_obj) {
      zz = d + (isrc < 64 ? ga : gb);
    } else {
      const float lo = d - eps, range = (d + stop_eps) - lo;
      zz = (range * linspace01(i, n2) + lo) + uu * (range / (float)n2);
    }
    if (live) {
      z[ray * S + s] = zz;
      float* p = pts + (ray * S + s) * 3;
      p[0] = ox + ex * zz; p[1] = oy + ey * zz; p[2] = oz + ez * zz;
    }
  }
}
}  // namespace cnr_sample
