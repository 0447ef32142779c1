// Times the backward pair of a 64 -> 64 1x1 convolution (k_gemm_dual<64, MODE> float32-MFMA / k_gemm_dual_s<MODE>
// split-bf16) alone on one stream and reports their difference:   dual_probe [batch] [H]
// (MVAE_SPLIT_DUAL is read once per process: the two variants are run by two processes, see the shell line in DESIGN.)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../multiscale_variational_autoencoder_amd/csrc/kernels.h"
using namespace mvae;
static float* dev_rand(size_t n, float scale, unsigned seed, bool relu = false) {
  std::vector<float> h(n);
  srand(seed);
  for (auto& v : h) { v = ((rand() / (float)RAND_MAX) - 0.5f) * 2.f * scale; if (relu && v < 0) v = 0; }
  float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  return d;
}
int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 512, H = argc > 2 ? atoi(argv[2]) : 32;
  const int64_t M = (int64_t)nb * H * H, C = 64;
  float* X = dev_rand(M * C, 1.f, 1); float* aux = dev_rand(M * C, 1.f, 2, true); float* res = dev_rand(M * C, 1.f, 3);
  float* W = dev_rand(C * C, 0.1f, 4); float* gate = dev_rand(nb * C, 0.5f, 5, true);
  float *Y, *dW, *db, *dot;
  hipMalloc(&Y, M * C * 4); hipMalloc(&dW, C * C * 4); hipMalloc(&db, C * 4); hipMalloc(&dot, nb * C * 4);
  GradSlots sl;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 1; mode <= 2; ++mode) {
    auto run = [&]() {
      if (mode == 1) launch_gemm_dual_mfma(X, W, aux, gate, nullptr, Y, dW, db, dot, M, (int64_t)H * H, 64, sl, 1, 0, 0);
      else launch_gemm_dual_mfma(X, W, aux, nullptr, res, Y, dW, db, nullptr, M, (int64_t)H * H, 64, sl, 1, 0, 0);
    };
    for (int k = 0; k < 3; ++k) run();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int k = 0; k < 20; ++k) run();
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemset(dW, 0, C * C * 4); hipMemset(db, 0, C * 4); hipMemset(dot, 0, nb * C * 4);
    run(); hipDeviceSynchronize();
    std::vector<float> hy(1 << 16), hw(C * C), hb(C), hd(nb * C);
    hipMemcpy(hy.data(), Y, hy.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hw.data(), dW, C * C * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hb.data(), db, C * 4, hipMemcpyDeviceToHost); hipMemcpy(hd.data(), dot, nb * C * 4, hipMemcpyDeviceToHost);
    double sy = 0, sw = 0, sb = 0, sd = 0;
    for (float v : hy) sy += std::fabs(v); for (float v : hw) sw += std::fabs(v); for (float v : hb) sb += std::fabs(v);
    for (float v : hd) sd += std::fabs(v);
    const double passes = mode == 1 ? 3 : 4;
    printf("M 2^%.0f mode %d: %.1f us  (%.0f GB/s algorithmic)   checksums |Y| %.6e |dW| %.6e |db| %.6e |dot| %.6e\n", std::log2((double)M), mode,
           ms * 50, passes * M * C * 4 / (ms * 50e-6) / 1e9, sy, sw, sb, mode == 1 ? sd : 0.0);
  }
  return 0;
}
