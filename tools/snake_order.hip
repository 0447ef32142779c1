// Does the 256 MiB Infinity Cache serve a producer -> consumer chain of streaming kernels when the consumer walks the
// tensor in the OPPOSITE direction (most recently written lines first)?  Chain of copy kernels t0 -> t1 -> t2 -> ... over
// tensors of S MB each, every kernel walking ascending ("same") or alternating ascending/descending ("snake").
//   hipcc --offload-arch=gfx950 -O3 tools/snake_order.hip -o tools/snake_order.bin && tools/snake_order.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// persistent blocks, tile = 4 KB per block step (like the 1x1 kernels' row tiles); reads NIN tensors, writes one
template <int NIN>
__global__ void __launch_bounds__(256) k_chain(const f32x4* __restrict__ a, const f32x4* __restrict__ b,
                                               f32x4* __restrict__ o, long long ntiles, int reverse) {
  for (long long t = blockIdx.x; t < ntiles; t += 2 * gridDim.x) {
    long long t0 = t, t1 = t + gridDim.x < ntiles ? t + gridDim.x : t;
    if (reverse) { t0 = ntiles - 1 - t0; t1 = ntiles - 1 - t1; }
    f32x4 v0 = a[t0 * 256 + threadIdx.x], v1 = a[t1 * 256 + threadIdx.x];
    if (NIN == 2) { v0 += b[t0 * 256 + threadIdx.x]; v1 += b[t1 * 256 + threadIdx.x]; }
    o[t0 * 256 + threadIdx.x] = v0 * 1.0001f;
    if (t1 != t0) o[t1 * 256 + threadIdx.x] = v1 * 1.0001f;
  }
}
int main() {
  const int NT = 6;
  for (long long mb : {32ll, 64ll, 128ll, 192ll}) {
    const long long bytes = mb << 20, ntiles = bytes / 4096;
    f32x4* t[NT];
    for (int i = 0; i < NT; ++i) { hipMalloc(&t[i], bytes); hipMemset(t[i], 0, bytes); }
    for (int grid : {1024, 2048}) {
      for (int mode = 0; mode < 3; ++mode) {       // 0 same direction, 1 snake, 2 snake with 2 inputs (x and residual)
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
          hipEventRecord(e0);
          for (int k = 0; k + 1 < NT; ++k) {
            const int rev = mode ? (k & 1) : 0;
            if (mode == 2 && k > 0)
              hipLaunchKernelGGL(k_chain<2>, dim3(grid), dim3(256), 0, 0, t[k], t[k - 1], t[k + 1], ntiles, rev);
            else
              hipLaunchKernelGGL(k_chain<1>, dim3(grid), dim3(256), 0, 0, t[k], t[k], t[k + 1], ntiles, rev);
          }
          hipEventRecord(e1); hipDeviceSynchronize();
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
        }
        const double moved = (mode == 2 ? (2.0 + 3.0 * (NT - 2)) : 2.0 * (NT - 1)) * bytes;
        printf("%4lld MB tensors grid %4d %-10s: %.1f us per kernel, %.2f TB/s algorithmic\n", mb, grid,
               mode == 0 ? "same" : mode == 1 ? "snake" : "snake+res", best * 1e3 / (NT - 1), moved / (best * 1e-3) / 1e12);
      }
    }
    for (int i = 0; i < NT; ++i) hipFree(t[i]);
  }
  return 0;
}
