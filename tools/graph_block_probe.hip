// Which property of a kernel makes hipGraphLaunch of a LINEAR graph block on the host?  One linear graph of n nodes per variant:
// plain spin | private scratch | 64 KB static LDS | 100 KB dynamic LDS (attribute) | 160-byte kernarg | wide grid.
//   hipcc --offload-arch=gfx950 -O3 tools/graph_block_probe.hip -o tools/graph_block_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ void spin(int ticks) {
  const unsigned long long t0 = wall_clock64();
  while ((long long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_plain(float* p, int ticks) { spin(ticks); }
__global__ void k_scratch(float* p, int ticks) {
  volatile float a[600];
  for (int i = 0; i < 600; ++i) a[i] = p[(i + threadIdx.x) & 63];
  spin(ticks);
  float t = 0; for (int i = 0; i < 600; i += 7) t += a[(i * 13 + threadIdx.x) % 600];
  if (t == 12345.f) p[0] = t;
}
__global__ void k_lds(float* p, int ticks) { __shared__ float sm[16384]; sm[threadIdx.x] = p[threadIdx.x & 63]; __syncthreads(); spin(ticks); if (sm[(threadIdx.x + 1) & 63] == 12345.f) p[0] = 1; }
__global__ void k_dyn(float* p, int ticks) { extern __shared__ float dsm[]; dsm[threadIdx.x] = p[threadIdx.x & 63]; __syncthreads(); spin(ticks); if (dsm[(threadIdx.x + 1) & 63] == 12345.f) p[0] = 1; }
struct Big { float* p; int ticks; int pad[36]; };
__global__ void k_bigarg(Big b) { spin(b.ticks); if (b.pad[5] == 12345) b.p[0] = 1; }
__global__ void k_wide(float* p, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = p[i] * 1.0001f + 1.f;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 200; const int ticks = (argc > 2 ? atoi(argv[2]) : 8) * 100;
  float* d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipFuncSetAttribute((const void*)k_dyn, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  Big b{}; b.p = d; b.ticks = ticks;
  for (int v = 0; v < 7; ++v) {
    const char* name[] = {"plain", "scratch", "static LDS 64K", "dynamic LDS 100K", "kernarg 160 B", "grid 2048 x 256", "mixed plain/scratch"};
    auto one = [&](int k) {
      const int vv = v == 6 ? (k & 1) : v;
      switch (vv) {
        case 0: hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, s, d, ticks); break;
        case 1: hipLaunchKernelGGL(k_scratch, dim3(1), dim3(64), 0, s, d, ticks); break;
        case 2: hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, s, d, ticks); break;
        case 3: hipLaunchKernelGGL(k_dyn, dim3(1), dim3(64), 100 * 1024, s, d, ticks); break;
        case 4: hipLaunchKernelGGL(k_bigarg, dim3(1), dim3(64), 0, s, b); break;
        case 5: hipLaunchKernelGGL(k_plain, dim3(2048), dim3(256), 0, s, d, ticks); break;
      }
    };
    for (int k = 0; k < 4; ++k) one(k);
    hipStreamSynchronize(s);
    hipGraph_t g; hipGraphExec_t ex;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int k = 0; k < n; ++k) one(k);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    for (int w = 0; w < 2; ++w) hipGraphLaunch(ex, s);
    hipStreamSynchronize(s);
    const int reps = 6; double host = 0; const double t0 = now_us();
    for (int r = 0; r < reps; ++r) { const double a = now_us(); hipGraphLaunch(ex, s); host += now_us() - a; }
    hipStreamSynchronize(s);
    printf("%-22s n=%d: host %8.1f us per launch, wall %8.1f us per launch  (%s)\n", name[v], n, host / reps, (now_us() - t0) / reps,
           hipGetErrorString(hipGetLastError()));
    fflush(stdout);
    if (v == 0) {                      // what precedes the launch on the stream: an eager kernel | an event wait | a second stream's graph
      hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
      hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
      for (int pre = 0; pre < 3; ++pre) {
        double hostg = 0, hostp = 0; const double t1 = now_us();
        for (int r = 0; r < reps; ++r) {
          double a = now_us();
          if (pre == 0) hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, s, d, ticks);
          if (pre == 1) { hipEventRecord(ev, s2); hipStreamWaitEvent(s, ev, 0); }
          if (pre == 2) { hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, s2, d, ticks); hipEventRecord(ev, s2); hipStreamWaitEvent(s, ev, 0); }
          hostp += now_us() - a; a = now_us();
          hipGraphLaunch(ex, s); hostg += now_us() - a;
        }
        hipStreamSynchronize(s);
        const char* pn[] = {"eager kernel before", "idle-stream event wait before", "busy-stream event wait before"};
        printf("  %-32s: pre %6.1f us, graph launch host %8.1f us, wall %8.1f us\n", pn[pre], hostp / reps, hostg / reps, (now_us() - t1) / reps);
        fflush(stdout);
      }
    }
    hipGraphExecDestroy(ex); hipGraphDestroy(g);
  }
  {   // kernels that STREAM memory (a 64 MB read-modify-write each, ~25 us): is the host cost of a launch a function of what the device is doing?
    float* big; hipMalloc(&big, 64 << 20); hipMemset(big, 0, 64 << 20);
    hipGraph_t g; hipGraphExec_t ex;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int k = 0; k < n; ++k) hipLaunchKernelGGL(k_wide, dim3(2048), dim3(256), 0, s, big, 16 << 20);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    for (int w = 0; w < 2; ++w) hipGraphLaunch(ex, s);
    hipStreamSynchronize(s);
    const int reps = 6; double host = 0; const double t0 = now_us();
    for (int r = 0; r < reps; ++r) { const double a = now_us(); hipGraphLaunch(ex, s); host += now_us() - a; }
    const double t1 = now_us();
    hipStreamSynchronize(s);
    printf("streaming kernels      n=%d: host %8.1f us per launch (all launches returned after %.1f us), wall %8.1f us per launch\n", n, host / reps, t1 - t0, (now_us() - t0) / reps);
    // eager, for comparison
    const double t2 = now_us();
    for (int k = 0; k < n; ++k) hipLaunchKernelGGL(k_wide, dim3(2048), dim3(256), 0, s, big, 16 << 20);
    const double t3 = now_us();
    hipStreamSynchronize(s);
    printf("streaming kernels eager n=%d: host %8.1f us for all, wall %8.1f us\n", n, t3 - t2, now_us() - t2);
  }
  {   // two DIFFERENT linear graphs alternating on one stream, and the same on two streams
    hipGraph_t g[2]; hipGraphExec_t ex[2];
    for (int q = 0; q < 2; ++q) {
      hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
      for (int k = 0; k < n; ++k) hipLaunchKernelGGL(k_plain, dim3(1 + q), dim3(64), 0, s, d, ticks);
      hipStreamEndCapture(s, &g[q]); hipGraphInstantiate(&ex[q], g[q], nullptr, nullptr, 0);
    }
    for (int w = 0; w < 2; ++w) { hipGraphLaunch(ex[0], s); hipGraphLaunch(ex[1], s); }
    hipStreamSynchronize(s);
    const int reps = 6; double host[2] = {0, 0}; const double t0 = now_us();
    for (int r = 0; r < reps; ++r) for (int q = 0; q < 2; ++q) { const double a = now_us(); hipGraphLaunch(ex[q], s); host[q] += now_us() - a; }
    hipStreamSynchronize(s);
    printf("two graphs alternating on one stream: host %8.1f / %8.1f us per launch, wall %8.1f us per pair\n", host[0] / reps, host[1] / reps, (now_us() - t0) / reps);
  }
  return 0;
}
