// library / device probe
#include <string.h>

#include "cnr_common.h"

extern "C" int cnr_version(void) { return 100; }  // 0.1.0

extern "C" int cnr_device_info(int* n_cu, int* lds_bytes, int* gcn_arch_is_gfx950) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return (int)e;
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)prop.sharedMemPerBlock;
  if (gcn_arch_is_gfx950) *gcn_arch_is_gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
  return CNR_OK;
}

// a1: cameraInfo.get_rays_dirs (src/scene_cateogries.py:613-629): dirs[w][h] = ((w - cx) / fx, (h - cy) / fy, 1), indexed
// [w, h], not normalised (z-depth convention).  IEEE subtraction and DIVISION per element (__fsub_rn / __fdiv_rn): bit-equal
// to the reference's torch expression (a tensor / scalar on the GPU through torch multiplies by the reciprocal -- one ulp
// off in a third of the entries).
namespace {
__global__ __launch_bounds__(256) void camera_rays_kernel(float* __restrict__ dirs, int W, int H, float fx, float fy,
                                                          float cx, float cy) {
  const int64_t n = (int64_t)W * H;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int w = (int)(i / H), h = (int)(i - (int64_t)w * H);
    float* o = dirs + i * 3;
    o[0] = __fdiv_rn(__fsub_rn((float)w, cx), fx);
    o[1] = __fdiv_rn(__fsub_rn((float)h, cy), fy);
    o[2] = 1.0f;
  }
}
}  // namespace

extern "C" int cnr_camera_rays(float* dirs, int W, int H, float fx, float fy, float cx, float cy, void* stream) {
  if (!dirs || W <= 0 || H <= 0 || fx == 0.0f || fy == 0.0f) return CNR_E_ARG;
  const int64_t n = (int64_t)W * H;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(camera_rays_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, (hipStream_t)stream,
                     dirs, W, H, fx, fy, cx, cy);
  CNR_LAUNCH_CHECK();
  return CNR_OK;
}
