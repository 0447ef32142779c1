// Tile order of a persistent streaming kernel: grid-stride (block b takes tiles b, b + G, ...: the resident blocks sweep
// one contiguous window) against blocked (block b takes a contiguous run of tiles: G separate sequential streams).
// 16 KB tiles (a 64-row x 64-channel fp32 tile), 2 reads + 1 write like the fused 1x1 kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_order.hip -o /tmp/stream_order && /tmp/stream_order
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int BLOCKED>
__global__ void __launch_bounds__(256) k_tiles(const f32x4* __restrict__ a, const f32x4* __restrict__ b,
                                               f32x4* __restrict__ o, long long ntiles) {
  const long long G = gridDim.x;
  const long long per = (ntiles + G - 1) / G;
  long long t = BLOCKED ? blockIdx.x * per : blockIdx.x;
  const long long tend = BLOCKED ? (t + per < ntiles ? t + per : ntiles) : ntiles;
  const long long step = BLOCKED ? 1 : G;
  for (; t < tend; t += step) {
    const f32x4* pa = a + t * 1024 + threadIdx.x;
    const f32x4* pb = b + t * 1024 + threadIdx.x;
    f32x4 v[4], w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { v[u] = pa[u * 256]; w[u] = pb[u * 256]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) o[t * 1024 + threadIdx.x + u * 256] = v[u] + w[u];
  }
}
template <int BLOCKED>
void run(long long bytes_each, int grid) {
  f32x4 *a, *b, *o;
  hipMalloc(&a, bytes_each); hipMalloc(&b, bytes_each); hipMalloc(&o, bytes_each);
  hipMemset(a, 0, bytes_each); hipMemset(b, 0, bytes_each);
  const long long ntiles = bytes_each / 16384;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_tiles<BLOCKED>, dim3(grid), dim3(256), 0, 0, a, b, o, ntiles);
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_tiles<BLOCKED>, dim3(grid), dim3(256), 0, 0, a, b, o, ntiles);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s  %5.0f MB/tensor grid %5d : %.2f TB/s\n", BLOCKED ? "blocked    " : "grid-stride", bytes_each / 1e6, grid,
         3.0 * bytes_each * 10 / (ms * 1e-3) / 1e12);
  fflush(stdout);
  hipFree(a); hipFree(b); hipFree(o);
}
int main() {
  for (long long mb : {128ll, 512ll})
    for (int grid : {256, 512, 1024, 2048}) { run<0>(mb << 20, grid); run<1>(mb << 20, grid); }
  return 0;
}
