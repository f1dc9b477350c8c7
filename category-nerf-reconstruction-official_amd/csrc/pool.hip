// ray-pool construction from frames (src/scene_cateogries.py:164-260 objects, :262-325
// background) as ONE device gather instead of per-frame Python slicing + torch.cat + index shuffles.
// The reference, per (instance, frame): crop the 2-D bbox out of image / depth / instance mask / cached ray
// directions, flatten row-major over (w, h), derive the pixel state (1 = this object, 2 = unknown (-1), 0 = other),
// attach the frame's pose (T_co = inv(T_wc) T_obj for objects, T_wc for single objects and the background) and the
// object index, concatenate everything and apply one global shuffle.  Here: a crop table
//   crops[k] = {frame slot, instance id, object index, w0, w1, h0, h1, frame index in the instance's list},
//   crop_offset[k] = first row of crop k
// and one thread per OUTPUT row: source row = perm[row] (the shuffle, a gather exactly like x = x[shuffled_idx]),
// crop by binary search in crop_offset, pixel by div/mod in the crop, then 92 B written.  HBM-bound, one pass.
#include "cnr_common.h"

namespace {
struct Crop { int frame, inst_id, obj_index, w0, w1, h0, h1, frame_idx; };

__global__ __launch_bounds__(256) void gather_pool_kernel(
    const uint8_t* __restrict__ images, const float* __restrict__ depth, const int32_t* __restrict__ obj_mask,
    const float* __restrict__ rays_dir, const float* __restrict__ T_crop, const Crop* __restrict__ crops,
    const int64_t* __restrict__ crop_offset, const int64_t* __restrict__ perm, int W, int H, int n_crops, int64_t N,
    uint8_t* __restrict__ rgbs, float* __restrict__ depth_out, float* __restrict__ dirs, float* __restrict__ T_out,
    int64_t* __restrict__ indices, int64_t* __restrict__ frame_out) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= N) return;
  const int64_t src = perm ? perm[row] : row;
  int lo = 0, hi = n_crops - 1;  // last crop with crop_offset[k] <= src
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (crop_offset[mid] <= src) lo = mid; else hi = mid - 1;
  }
  const Crop c = crops[lo];
  const int64_t local = src - crop_offset[lo];
  const int ch = c.h1 - c.h0;
  const int w = c.w0 + (int)(local / ch), h = c.h0 + (int)(local % ch);
  const int64_t pix = ((int64_t)c.frame * W + w) * H + h;
  const int m = obj_mask[pix];
  const uint8_t state = m == c.inst_id ? 1 : (m == -1 ? 2 : 0);
  const uint8_t* ip = images + pix * 3;
  uchar4 px = {ip[0], ip[1], ip[2], state};
  reinterpret_cast<uchar4*>(rgbs)[row] = px;
  depth_out[row] = depth[pix];
  const float* dp = rays_dir + ((int64_t)w * H + h) * 3;
  dirs[row * 3 + 0] = dp[0]; dirs[row * 3 + 1] = dp[1]; dirs[row * 3 + 2] = dp[2];
  if (T_out) {
    const float4* tp = reinterpret_cast<const float4*>(T_crop + (int64_t)lo * 16);
    float4* to = reinterpret_cast<float4*>(T_out + row * 16);
    to[0] = tp[0]; to[1] = tp[1]; to[2] = tp[2]; to[3] = tp[3];
  }
  if (indices) indices[row] = c.obj_index;
  if (frame_out) frame_out[row] = c.frame_idx;
}
}  // namespace

extern "C" int cnr_gather_pool(const uint8_t* images, const float* depth, const int32_t* obj_mask,
                               const float* rays_dir, const float* T_crop, const int32_t* crops,
                               const int64_t* crop_offset, const int64_t* perm, int W, int H, int n_crops, int64_t N,
                               uint8_t* rgbs, float* depth_out, float* dirs, float* T_out, int64_t* indices,
                               int64_t* frame_out, void* stream) {
  if (!images || !depth || !obj_mask || !rays_dir || !crops || !crop_offset || !rgbs || !depth_out || !dirs ||
      W <= 0 || H <= 0 || n_crops <= 0 || N <= 0 || (T_out && !T_crop))
    return CNR_E_ARG;
  if (((uintptr_t)rgbs & 3) != 0 || (T_out && (((uintptr_t)T_out & 15) != 0 || ((uintptr_t)T_crop & 15) != 0)))
    return CNR_E_ALIGN;
  hipLaunchKernelGGL(gather_pool_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, images,
                     depth, obj_mask, rays_dir, T_crop, reinterpret_cast<const Crop*>(crops), crop_offset, perm, W, H,
                     n_crops, N, rgbs, depth_out, dirs, T_out, indices, frame_out);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}
