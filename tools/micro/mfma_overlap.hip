// Does one wave's vector work run under its own matrix-core instructions on gfx950?  One wave per SIMD-less workgroup (64 threads),
// a loop of { v_mfma_f32_32x32x16_f16 (accumulating in place) ; N independent v_add_f32 }, cycles per trip from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_overlap.hip -o gpurun_out/mfma_overlap && gpurun_out/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int N, int NACC>
__global__ __launch_bounds__(64) void k(float* out, long long* cyc, int trips) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
  f16v acc[NACC];
  for (int q = 0; q < NACC; ++q) for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
  const long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < trips; t += 8) {   // eight repetitions per loop trip: the branch costs ~36 cycles
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < N; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i % 16]) : "v"(x[(i + 5) % 16]));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int q = 0; q < NACC; ++q) for (int i = 0; i < 16; ++i) s += acc[q][i];
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
// the same loop without the matrix instruction: what N vector instructions cost alone
template <int N>
__global__ __launch_bounds__(64) void kv(float* out, long long* cyc, int trips) {
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
  const long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < trips; t += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i % 16]) : "v"(x[(i + 5) % 16]));
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int N, int NACC> void run(float* out, long long* cyc) {
  const int trips = 2000;
  long long c, cv = 0;
  if (N > 0) {
    for (int r = 0; r < 2; ++r) { kv<N><<<256, 64>>>(out, cyc, trips); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&cv, cyc, 8, hipMemcpyDeviceToHost);
  }
  for (int r = 0; r < 2; ++r) { k<N, NACC><<<256, 64>>>(out, cyc, trips); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%d accumulator(s) in turn, %2d independent v_add_f32 behind each MFMA: %5.1f cycles per MFMA  (the %2d v_add alone: %5.1f)\n", NACC, N,
         (double)c / trips / NACC, N, (double)cv / trips);
}
int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 64 * 4); (void)hipMalloc(&cyc, 8);
  run<0, 1>(out, cyc); run<4, 1>(out, cyc); run<6, 1>(out, cyc); run<8, 1>(out, cyc); run<12, 1>(out, cyc); run<16, 1>(out, cyc); run<24, 1>(out, cyc);
  run<0, 2>(out, cyc); run<4, 2>(out, cyc); run<6, 2>(out, cyc); run<8, 2>(out, cyc); run<16, 2>(out, cyc);
  return 0;
}
