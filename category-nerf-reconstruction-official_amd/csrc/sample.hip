// a2-a5: camera->object ray transform + depth-guided sampling, one wavefront per ray.
// Restates src/scene_cateogries.py:24-47 (origin_dirs_O/W), :51-81 (stratified_bins),
// :84-96 (normal_bins_sampling) and :453-546 (sample_3d_points) branch-free per ray: the reference's
// four boolean-mask scatter writes + count_nonzero host syncs become per-ray selects.
// Lane = sample column, so z / pts stores are coalesced; the per-ray ascending sort of the
// around-surface Gaussian draws is an in-register bitonic network over the wave (<= 128 values).
#include "sample_common.h"

namespace {

__global__ void maxdepth_kernel(const float* __restrict__ depth, float* __restrict__ max_bound, int R,
                                const int64_t* __restrict__ d_state, int64_t pool_rows,
                                const int* __restrict__ perm) {
  const int c = blockIdx.x;
  float m = -INFINITY;
  const int64_t base = pool_rows > 0 ? (int64_t)c * pool_rows + d_state[0] : (int64_t)c * R;
  for (int r = threadIdx.x; r < R; r += blockDim.x)
    m = fmaxf(m, depth[perm ? (int64_t)c * pool_rows + perm[base + r] : base + r]);
  __shared__ float sm[16];
  m = cnr::wave_max(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = sm[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) v = fmaxf(v, sm[i]);
    max_bound[c] = v;
  }
}

// max depth of every slice of an epoch: block (s, c) -> out[c][s] = max_r depth[c][perm[c][s R + r]]
__global__ __launch_bounds__(256) void slice_maxdepth_kernel(const float* __restrict__ depth, const int* __restrict__ perm,
                                                             int64_t pool_rows, int R, float* __restrict__ out) {
  const int s = blockIdx.x, c = blockIdx.y, nsl = gridDim.x;
  const int64_t cbase = (int64_t)c * pool_rows, base = cbase + (int64_t)s * R;
  float m = -INFINITY;
  for (int r0 = 0; r0 < R; r0 += 8 * 256) {   // permutation entries together, then depths together
    int64_t at[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = r0 + u * 256 + (int)threadIdx.x;
      at[u] = r < R ? (perm ? cbase + perm[base + r] : base + r) : -1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) m = fmaxf(m, at[u] >= 0 ? depth[at[u]] : -INFINITY);
  }
  __shared__ float sm[4];
  m = cnr::wave_max(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[(int64_t)c * nsl + s] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// mask counts of every slice of an epoch: block s -> out[s][c][0..2] = #(valid depth & label != 0), #(label != 0),
// #(label != 2) over the R rows of slice s of class c (what cnr_render_loss / cnr_field_fwd_render count per step),
// out[s][C][0..2] = 1 where ANY class of the slice has a zero count (the empty-mask rule, src/render_rays.py:67-72)
__global__ __launch_bounds__(256) void slice_maskcounts_kernel(const uint8_t* __restrict__ rgbs,
                                                               const float* __restrict__ depth,
                                                               const int* __restrict__ perm, int64_t pool_rows, int C,
                                                               int R, float min_bound, float* __restrict__ out) {
  const int s = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __shared__ float sm[12];
  float* o = out + (size_t)s * (C + 1) * 4;
  float e0 = 0.f, e1 = 0.f, e2 = 0.f;
  for (int c = 0; c < C; ++c) {
    const int64_t cbase = (int64_t)c * pool_rows, base = cbase + (int64_t)s * R;
    float a = 0.f, b = 0.f, d = 0.f;
    for (int r = threadIdx.x; r < R; r += 256) {
      const int64_t at = perm ? cbase + perm[base + r] : base + r;
      const uint8_t lab = rgbs[at * 4 + 3];
      const bool mo = lab != 0, ms = lab != 2, md = depth[at] > min_bound;
      a += (md && mo) ? 1.f : 0.f; b += mo ? 1.f : 0.f; d += ms ? 1.f : 0.f;
    }
    a = cnr::wave_sum(a); b = cnr::wave_sum(b); d = cnr::wave_sum(d);
    __syncthreads();
    if (lane == 0) { sm[wv] = a; sm[4 + wv] = b; sm[8 + wv] = d; }
    __syncthreads();
    a = (sm[0] + sm[1]) + (sm[2] + sm[3]); b = (sm[4] + sm[5]) + (sm[6] + sm[7]); d = (sm[8] + sm[9]) + (sm[10] + sm[11]);
    if (threadIdx.x == 0) { o[c * 4 + 0] = a; o[c * 4 + 1] = b; o[c * 4 + 2] = d; o[c * 4 + 3] = 0.f; }
    if (a == 0.f) e0 = 1.f;
    if (b == 0.f) e1 = 1.f;
    if (d == 0.f) e2 = 1.f;
  }
  if (threadIdx.x == 0) { o[C * 4 + 0] = e0; o[C * 4 + 1] = e1; o[C * 4 + 2] = e2; o[C * 4 + 3] = 0.f; }
}

__global__ __launch_bounds__(256) void sample_kernel(cnr_sample::SampleArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (ray >= (int64_t)a.C * a.R) return;
  cnr_sample::sample_ray(a, ray, lane);
}
// The epoch shuffle (src/scene_cateogries.py:439-449: a new random order of the pool) as ONE launch: perm[c][i] = P(i), P a keyed
// pseudo-random BIJECTION of [0, n) -- a 6-round balanced Feistel network on the next even power of two above n (round keys
// from Philox4x32 of (seed, epoch, global class id)), cycle-walked back into [0, n) (a value >= n is encrypted again: at most
// four expected trips, the domain is < 4 n).  torch.randperm sorts random keys: a dozen launches and ~100 us of GPU time per
// epoch end for 131 072 rows, which the 63 steps of an epoch each paid 1.5 us for; and every rank had to draw every class's
// permutation from one generator to stay in step -- here a rank computes exactly its own classes'.
__global__ __launch_bounds__(256) void epoch_perm_kernel(int* __restrict__ perm, int64_t n, int half_bits, uint64_t seed,
                                                         uint64_t epoch, const int* __restrict__ class_ids,
                                                         int64_t* __restrict__ state_cursor, int64_t cursor0) {
  const int c = blockIdx.y;
  if (state_cursor && blockIdx.x == 0 && c == 0 && threadIdx.x == 0) state_cursor[0] = cursor0;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t k0[4], k1[4];
  const uint64_t gc = (uint64_t)(class_ids ? class_ids[c] : c);
  cnr_sample::philox4(seed ^ 0x5851F42D4C957F2Dull, epoch, gc * 2 + 0, k0);
  cnr_sample::philox4(seed ^ 0x5851F42D4C957F2Dull, epoch, gc * 2 + 1, k1);
  const uint32_t key[6] = {k0[0], k0[1], k0[2], k0[3], k1[0], k1[1]};
  const uint32_t mask = (1u << half_bits) - 1u;
  uint32_t x = (uint32_t)i;
  do {
    uint32_t l = x >> half_bits, r = x & mask;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      uint32_t f = (r ^ key[t]) * 0x9E3779B1u;
      f ^= f >> 15; f *= 0x85EBCA77u; f ^= f >> 13;
      const uint32_t nl = r;
      r = (l ^ f) & mask;
      l = nl;
    }
    x = (l << half_bits) | r;
  } while ((int64_t)x >= n);
  perm[(size_t)c * n + i] = (int)x;
}
__global__ void advance_kernel(int64_t* state, int64_t add_rows) {
  if (threadIdx.x == 0) { state[0] += add_rows; state[1] += 1; state[2] += 1; }
}
}  // namespace

// device-side step state {pool cursor (rows), rng step, optimiser step}: advanced by a 1-thread kernel so that a
// captured hipGraph of the whole train step replays with no host-side argument patching
extern "C" int cnr_step_advance(int64_t* d_state, int64_t add_rows, void* stream) {
  if (!d_state) return CNR_E_ARG;
  hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_state, add_rows);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_epoch_perm(int* perm, int64_t pool_rows, int C, uint64_t seed, uint64_t epoch, const int* class_ids,
                              int64_t* state_cursor, int64_t cursor0, void* stream) {
  if (!perm || pool_rows <= 0 || pool_rows > (1ll << 30) || C <= 0) return CNR_E_ARG;
  int bits = 2;
  while ((1ll << bits) < pool_rows) bits += 2;           // even: two halves of bits / 2
  hipLaunchKernelGGL(epoch_perm_kernel, dim3((unsigned)((pool_rows + 255) / 256), (unsigned)C), dim3(256), 0, (hipStream_t)stream,
                     perm, pool_rows, bits / 2, seed, epoch, class_ids, state_cursor, cursor0);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_sample_maxdepth(const float* depth, float* max_bound, const int64_t* d_state,
                                   int64_t pool_rows, const int* perm, int C, int R, void* stream) {
  if (!depth || !max_bound || C <= 0 || R <= 0 || pool_rows < 0 || (pool_rows > 0 && !d_state)) return CNR_E_ARG;
  hipLaunchKernelGGL(maxdepth_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, depth, max_bound, R, d_state,
                     pool_rows, perm);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_slice_maxdepth(const float* depth, const int* perm, int64_t pool_rows, int C, int R, int slices,
                                  float* out, void* stream) {
  if (!depth || !out || C <= 0 || R <= 0 || slices <= 0 || pool_rows < (int64_t)slices * R) return CNR_E_ARG;
  hipLaunchKernelGGL(slice_maxdepth_kernel, dim3((unsigned)slices, (unsigned)C), dim3(256), 0, (hipStream_t)stream, depth,
                     perm, pool_rows, R, out);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_slice_maskcounts(const uint8_t* rgbs, const float* depth, const int* perm, int64_t pool_rows, int C, int R,
                                    int slices, float min_bound, float* out, void* stream) {
  if (!rgbs || !depth || !out || C <= 0 || R <= 0 || slices <= 0 || pool_rows < (int64_t)slices * R) return CNR_E_ARG;
  hipLaunchKernelGGL(slice_maskcounts_kernel, dim3((unsigned)slices), dim3(256), 0, (hipStream_t)stream, rgbs, depth, perm,
                     pool_rows, C, R, min_bound, out);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}

extern "C" int cnr_sample_rays(const uint8_t* rgbs, const float* depth, const float* dirs_c, const float* T,
                               const float* u, const float* g, uint64_t seed, uint64_t offset,
                               const int64_t* d_state, int64_t pool_rows,
                               const float* max_bound, int world_frame, int C, int R, int n1, int n2,
                               float eps, float stop_eps, float min_bound, float* z, float* pts,
                               float* origins, float* dirs_o, float* gt_rgb, float* gt_depth,
                               uint8_t* depth_mask, uint8_t* labels, const int64_t* pool_indices, int n_obj,
                               int* ray_row, const int* perm, int max_bound_slices, void* stream) {
  if (!rgbs || !depth || !dirs_c || !T || !z || !pts || !gt_rgb || !depth_mask || !labels) return CNR_E_ARG;
  if (max_bound_slices < 0 || (max_bound_slices > 1 && (pool_rows <= 0 || !d_state || !max_bound))) return CNR_E_ARG;
  if (C <= 0 || R <= 0 || n1 < 0 || n2 <= 0) return CNR_E_ARG;
  if (n2 > 128) return CNR_E_SHAPE;
  if ((u == nullptr) != (g == nullptr)) return CNR_E_ARG;
  if (pool_rows < 0 || (pool_rows > 0 && (!d_state || pool_rows < R))) return CNR_E_ARG;
  if (ray_row && (!pool_indices || n_obj <= 0)) return CNR_E_ARG;
  if (perm && pool_rows == 0) return CNR_E_ARG;
  const int64_t rays = (int64_t)C * R;
  const int waves_per_block = 4;
  const int64_t blocks = (rays + waves_per_block - 1) / waves_per_block;
  cnr_sample::SampleArgs args{rgbs, depth, dirs_c, T, u, g, seed, offset, d_state, pool_rows, max_bound, world_frame,
                              C, R, n1, n2, eps, stop_eps, min_bound, z, pts, origins, dirs_o, gt_rgb, gt_depth,
                              depth_mask, labels, pool_indices, n_obj, ray_row, perm, max_bound_slices, 0, 0, 0, 0};
  hipLaunchKernelGGL(sample_kernel, dim3((unsigned)blocks), dim3(64 * waves_per_block), 0, (hipStream_t)stream, args);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}
