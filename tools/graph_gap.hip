// How long does one DEPENDENT kernel node cost in a replayed hipGraph (and eagerly on a stream)?  N tiny kernels in a
// chain; per-kernel time = wall / N.   hipcc --offload-arch=gfx950 -O3 tools/graph_gap.hip -o /tmp/graph_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void k_wide(float* p, int n) {          // ~20 us of streaming work on the whole chip
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = p[i] * 1.0001f + 1.f;
}
int main() {
  float* d; const int n = 16 << 20;
  hipMalloc(&d, n * sizeof(float)); hipMemset(d, 0, n * sizeof(float));
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int N = 200;
  for (int wide = 0; wide < 2; ++wide) {
    auto launch_all = [&]() {
      for (int i = 0; i < N; ++i) {
        if (wide) hipLaunchKernelGGL(k_wide, dim3(1024), dim3(256), 0, s, d, n);
        else hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d);
      }
    };
    // eager
    launch_all(); hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    launch_all(); hipStreamSynchronize(s);
    double us_eager = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
    // graph
    hipGraph_t g; hipGraphExec_t ex;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    launch_all();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    hipGraphLaunch(ex, s); hipStreamSynchronize(s);
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ex, s);
    hipStreamSynchronize(s);
    double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (5 * N);
    printf("%s kernels: eager %.2f us per kernel, graph replay %.2f us per kernel\n", wide ? "64 MB streaming" : "tiny", us_eager, us_graph);
    fflush(stdout);
  }
  return 0;
}
