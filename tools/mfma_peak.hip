// Measures the sustained rate of v_mfma_f32_32x32x2_f32 on every CU (the exact-fp32 MFMA all kernels here use) and the
// shader clock it runs at, to price "MFMA-bound" kernels against what the chip sustains rather than the data sheet.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(float* out, int iters, long long* clk) {
  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x = threadIdx.x * 1e-3f, y = 1.0f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}
template <int NACC>
void run(int blocks_per_cu, int iters) {
  float* out; long long* clk;
  hipMalloc(&out, 256 * 256 * 8 * sizeof(float) * 4);
  hipMalloc(&clk, 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, out, 10, clk);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, clk);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  double mfmas = (double)grid * 4 * iters * 8 * NACC;
  double tf = mfmas * 4096.0 / (ms * 1e-3) / 1e12;
  printf("acc/wave %d  waves/SIMD %d  time %.3f ms  %.1f TFLOP/s  clock64 ticks %lld (%.1f MHz if ticks = shader cycles)  cycles/MFMA/SIMD %.1f\n",
         NACC, blocks_per_cu, ms, tf, c, c / (ms * 1e3), (double)c / (iters * 8.0 * NACC * blocks_per_cu));
  hipFree(out); hipFree(clk);
}
int main() {
  run<1>(1, 20000); run<2>(1, 20000); run<4>(1, 10000); run<4>(2, 10000); run<2>(4, 10000); run<4>(4, 5000);
  return 0;
}
