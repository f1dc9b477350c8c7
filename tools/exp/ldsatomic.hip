#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(float* out, long long* cyc, int mode, int iters) {
  __shared__ float tab[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) tab[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float v = (float)threadIdx.x * 1e-3f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (mode == 0) {            // all waves -> same 64 addresses per instruction, 16 instr to distinct rows
#pragma unroll
      for (int r = 0; r < 16; ++r) atomicAdd(&tab[r * 64 + lane], v);
    } else if (mode == 1) {     // wave-private rows
#pragma unroll
      for (int r = 0; r < 16; ++r) atomicAdd(&tab[wv * 1024 + r * 64 + lane], v);
    } else if (mode == 2) {     // plain read-modify-write, wave-private
#pragma unroll
      for (int r = 0; r < 16; ++r) tab[wv * 1024 + r * 64 + lane] += v;
    } else {                    // unsafe fp atomic builtin
#pragma unroll
      for (int r = 0; r < 16; ++r) __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)&tab[r * 64 + lane], v, 0, 0, false);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = tab[threadIdx.x];
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
  for (int mode = 0; mode < 4; ++mode) for (int waves = 1; waves <= 4; waves *= 4) {
    hipLaunchKernelGGL(k, dim3(16), dim3(64 * waves), 0, 0, out, cyc, mode, 100);
    hipDeviceSynchronize();
    long long h[16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d waves %d: %.1f cycles per wave-instruction\n", mode, waves, (double)h[0] / (100 * 16));
  }
  return 0;
}
