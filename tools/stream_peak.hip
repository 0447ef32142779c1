// Sustained HBM rates of plain streaming kernels on this chip (16 B per lane, many loads in flight), to price the
// HBM-bound kernels against what streaming code reaches rather than the 8 TB/s data-sheet figure.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_peak.hip -o /tmp/stream_peak && /tmp/stream_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0 copy, 1 read-only sum, 2 write-only, 3 read 2 + write 1 (like a fused elementwise op)
__global__ void __launch_bounds__(256) k_stream(const f32x4* __restrict__ a, const f32x4* __restrict__ b,
                                                f32x4* __restrict__ o, long long n4, float* sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
    f32x4 v[4], w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long j = i + u * stride < n4 ? i + u * stride : i;
      if (MODE != 2) v[u] = a[j];
      if (MODE == 3) w[u] = b[j];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long j = i + u * stride;
      if (MODE == 1) acc += v[u];
      if (j < n4) {
        if (MODE == 0) o[j] = v[u];
        if (MODE == 2) o[j] = acc + (float)u;
        if (MODE == 3) o[j] = v[u] + w[u];
      }
    }
  }
  if (MODE == 1 && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) *sink = 1.f;
}
template <int MODE>
void run(const char* name, long long bytes_each, int grid, double passes) {
  f32x4 *a, *b, *o; float* sink;
  hipMalloc(&a, bytes_each); hipMalloc(&b, bytes_each); hipMalloc(&o, bytes_each); hipMalloc(&sink, 4);
  hipMemset(a, 0, bytes_each); hipMemset(b, 0, bytes_each);
  long long n4 = bytes_each / 16;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_stream<MODE>, dim3(grid), dim3(256), 0, 0, a, b, o, n4, sink);
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_stream<MODE>, dim3(grid), dim3(256), 0, 0, a, b, o, n4, sink);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %5.0f MB/tensor grid %5d : %.2f TB/s\n", name, bytes_each / 1e6, grid, passes * bytes_each * 10 / (ms * 1e-3) / 1e12);
  hipFree(a); hipFree(b); hipFree(o); hipFree(sink);
}
int main() {
  for (long long mb : {128ll, 512ll}) {
    for (int grid : {1024, 4096, 16384}) {
      run<0>("copy (1 read + 1 write)", mb << 20, grid, 2);
      run<1>("read only", mb << 20, grid, 1);
      run<2>("write only", mb << 20, grid, 1);
      run<3>("2 reads + 1 write", mb << 20, grid, 3);
    }
  }
  return 0;
}
