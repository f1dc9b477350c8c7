// Two waves on one SIMD (workgroup of 8 waves: wave w and w + 4 share SIMD w): waves 0..3 run matrix-core instructions only, waves
// 4..7 vector instructions only.  Do the two streams overlap?  Cycles for each role alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int trips, int mode) {
  const int wv = threadIdx.x >> 6;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
  f16v acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (wv < 4) {
    if (mode & 1)
      for (int t = 0; t < trips; ++t) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
      }
  } else {
    if (mode & 2)
      for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 5) % 16]));
      }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + x[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wv] = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
  const int trips = 2000;
  for (int mode = 1; mode <= 3; ++mode) {
    for (int r = 0; r < 2; ++r) { k<<<256, 512>>>(out, cyc, trips, mode); (void)hipDeviceSynchronize(); }
    long long c[8]; (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("mode %d (%s): MFMA wave %.1f cycles per MFMA | VALU wave %.2f cycles per v_add\n", mode,
           mode == 1 ? "MFMA waves alone" : mode == 2 ? "VALU waves alone" : "both", (double)c[0] / trips / 2, (double)c[4] / trips / 16);
  }
  return 0;
}
