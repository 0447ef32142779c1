// Do non-temporal loads / stores change the streaming rate on working sets beyond the Infinity Cache?
//   hipcc --offload-arch=gfx950 -O3 tools/stream_nt.hip -o /tmp/stream_nt && /tmp/stream_nt
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NTL, int NTS, int READS>
__global__ void __launch_bounds__(256) k_stream(const f32x4* __restrict__ a, const f32x4* __restrict__ b,
                                                f32x4* __restrict__ o, long long n4) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
    f32x4 v[4], w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long j = i + u * stride < n4 ? i + u * stride : i;
      v[u] = NTL ? __builtin_nontemporal_load(&a[j]) : a[j];
      if (READS == 2) w[u] = NTL ? __builtin_nontemporal_load(&b[j]) : b[j];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long j = i + u * stride;
      if (j < n4) {
        f32x4 r = READS == 2 ? v[u] + w[u] : v[u];
        if (NTS) __builtin_nontemporal_store(r, &o[j]); else o[j] = r;
      }
    }
  }
}
template <int NTL, int NTS, int READS>
void run(long long bytes_each, int grid) {
  f32x4 *a, *b, *o;
  hipMalloc(&a, bytes_each); hipMalloc(&b, bytes_each); hipMalloc(&o, bytes_each);
  hipMemset(a, 0, bytes_each); hipMemset(b, 0, bytes_each);
  long long n4 = bytes_each / 16;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_stream<NTL, NTS, READS>), dim3(grid), dim3(256), 0, 0, a, b, o, n4);
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k_stream<NTL, NTS, READS>), dim3(grid), dim3(256), 0, 0, a, b, o, n4);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("reads %d  nt-load %d  nt-store %d  %5.0f MB/tensor grid %5d : %.2f TB/s\n", READS, NTL, NTS, bytes_each / 1e6, grid,
         (READS + 1.0) * bytes_each * 10 / (ms * 1e-3) / 1e12);
  fflush(stdout);
  hipFree(a); hipFree(b); hipFree(o);
}
int main() {
  for (long long mb : {128ll, 512ll})
    for (int grid : {1024, 2048, 4096}) {
      run<0, 0, 1>(mb << 20, grid); run<0, 1, 1>(mb << 20, grid); run<1, 0, 1>(mb << 20, grid); run<1, 1, 1>(mb << 20, grid);
      run<0, 0, 2>(mb << 20, grid); run<1, 1, 2>(mb << 20, grid);
    }
  return 0;
}
