// What does hipGraphLaunch cost on the HOST, per kernel node, for the shapes of graph a train step can be captured as?
//   A  one graph: root -> L parallel chains of n kernels (captured with L forked streams) -> join   (what runtime.cpp replays)
//   B  L linear graphs (one chain each) launched on L streams, fork / join with eager events
//   C  the same launches eagerly on L streams
// Every kernel spins for d microseconds (wall clock), so that host back-pressure (the launch call blocking on the device)
// shows as host time growing with d.  The first kernel of every chain stamps its start time: the stagger between chains
// is how far the host's enqueue order delays a branch.
//   hipcc --offload-arch=gfx950 -O3 tools/graph_launch_cost.hip -o tools/graph_launch_cost.bin ; ./tools/graph_launch_cost.bin [L] [n] [d_us]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_spin(unsigned long long* stamp, int ticks) {
  const unsigned long long t0 = wall_clock64();
  if (stamp && threadIdx.x == 0) *stamp = t0;
  while ((long long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 3, n = argc > 2 ? atoi(argv[2]) : 100;
  const double d_us = argc > 3 ? atof(argv[3]) : 4.0;
  const int ticks = (int)(d_us * 100.0);                       // wall_clock64: 100 MHz
  unsigned long long* st; hipMalloc(&st, 64 * 8); hipMemset(st, 0, 64 * 8);
  std::vector<hipStream_t> s(L + 1);
  for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  std::vector<hipEvent_t> ef(L), ej(L);
  hipEvent_t fork; hipEventCreateWithFlags(&fork, hipEventDisableTiming);
  for (int i = 0; i < L; ++i) { hipEventCreateWithFlags(&ef[i], hipEventDisableTiming); hipEventCreateWithFlags(&ej[i], hipEventDisableTiming); }
  auto chain = [&](int i, hipStream_t q) {
    for (int k = 0; k < n; ++k) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, q, k == 0 ? st + 1 + i : nullptr, ticks);
  };
  auto whole = [&]() {                                          // root on s[0], chains on s[1..L], join on s[0]
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[0], st, ticks);
    hipEventRecord(fork, s[0]);
    for (int i = 0; i < L; ++i) { hipStreamWaitEvent(s[1 + i], fork, 0); chain(i, s[1 + i]); hipEventRecord(ej[i], s[1 + i]); }
    for (int i = 0; i < L; ++i) hipStreamWaitEvent(s[0], ej[i], 0);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[0], st + 1 + L, ticks);
  };
  auto report = [&](const char* name, double host, double wall, int reps) {
    std::vector<unsigned long long> h(2 + L); hipMemcpy(h.data(), st, (2 + L) * 8, hipMemcpyDeviceToHost);
    printf("%-28s host %7.1f us / launch (%.2f us per node)  wall %7.1f us   chain starts after root:", name, host / reps,
           host / reps / (L * n + 2), wall / reps);
    for (int i = 0; i < L; ++i) printf(" %6.1f", (double)(long long)(h[1 + i] - h[0]) / 100.0);
    printf("  join %6.1f us\n", (double)(long long)(h[1 + L] - h[0]) / 100.0);
    fflush(stdout);
  };
  const int reps = 20;
  // ---- C: eager
  for (int w = 0; w < 3; ++w) whole();
  hipDeviceSynchronize();
  double t0 = now_us(), host = 0;
  for (int r = 0; r < reps; ++r) { const double a = now_us(); whole(); host += now_us() - a; }
  hipDeviceSynchronize();
  report("C eager, L streams", host, now_us() - t0, reps);
  // ---- A: one graph
  hipGraph_t g; hipGraphExec_t ex;
  hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal); whole(); hipStreamEndCapture(s[0], &g);
  hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  for (int w = 0; w < 3; ++w) hipGraphLaunch(ex, s[0]);
  hipDeviceSynchronize();
  t0 = now_us(); host = 0;
  for (int r = 0; r < reps; ++r) { const double a = now_us(); hipGraphLaunch(ex, s[0]); host += now_us() - a; }
  hipDeviceSynchronize();
  report("A one forked graph", host, now_us() - t0, reps);
  // ---- B: L linear graphs
  std::vector<hipGraphExec_t> exl(L);
  for (int i = 0; i < L; ++i) {
    hipGraph_t gi; hipStreamBeginCapture(s[1 + i], hipStreamCaptureModeThreadLocal); chain(i, s[1 + i]); hipStreamEndCapture(s[1 + i], &gi);
    hipGraphInstantiate(&exl[i], gi, nullptr, nullptr, 0);
  }
  auto split = [&]() {
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[0], st, ticks);
    hipEventRecord(fork, s[0]);
    for (int i = 0; i < L; ++i) { hipStreamWaitEvent(s[1 + i], fork, 0); hipGraphLaunch(exl[i], s[1 + i]); hipEventRecord(ej[i], s[1 + i]); }
    for (int i = 0; i < L; ++i) hipStreamWaitEvent(s[0], ej[i], 0);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[0], st + 1 + L, ticks);
  };
  for (int w = 0; w < 3; ++w) split();
  hipDeviceSynchronize();
  t0 = now_us(); host = 0;
  for (int r = 0; r < reps; ++r) { const double a = now_us(); split(); host += now_us() - a; }
  hipDeviceSynchronize();
  report("B L linear graphs + events", host, now_us() - t0, reps);
  // ---- D: one linear graph of L*n nodes on one stream (the packet-copy fast path, if there is one)
  hipGraph_t gl; hipGraphExec_t exlin;
  hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < L; ++i) chain(i, s[0]);
  hipStreamEndCapture(s[0], &gl); hipGraphInstantiate(&exlin, gl, nullptr, nullptr, 0);
  for (int w = 0; w < 3; ++w) hipGraphLaunch(exlin, s[0]);
  hipDeviceSynchronize();
  t0 = now_us(); host = 0;
  for (int r = 0; r < reps; ++r) { const double a = now_us(); hipGraphLaunch(exlin, s[0]); host += now_us() - a; }
  hipDeviceSynchronize();
  report("D one linear graph", host, now_us() - t0, reps);
  printf("last hip error: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
