// Times launch_conv_taps_mfma on the headline 5x5 stride-2 geometry outside the engine -- a perf-debug driver, links the
// built kernels_mfma.o:   conv_probe <ci> <co> <transposed 0|1> <batch>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../multiscale_variational_autoencoder_amd/csrc/kernels.h"
namespace mvae { bool launch_conv_taps_mfma(bool transposed, const float* in, const float* w, const float* bias, float* out,
                           const ConvGeom& g, hipStream_t s); }
int main(int argc, char** argv) {
  using namespace mvae;
  const int ci = argc > 1 ? atoi(argv[1]) : 64, co = argc > 2 ? atoi(argv[2]) : 32, tr = argc > 3 ? atoi(argv[3]) : 0;
  const int nb = argc > 4 ? atoi(argv[4]) : 512;
  ConvGeom g{nb, 32, 32, ci, 16, 16, co, 5, 5, 2, 2, 1, 1};
  float *big, *small, *w, *b;
  hipMalloc(&big, (size_t)nb * 32 * 32 * ci * 4); hipMalloc(&small, (size_t)nb * 16 * 16 * co * 4);
  hipMalloc(&w, 25 * ci * co * 4); hipMalloc(&b, 256 * 4);
  hipMemset(big, 0, (size_t)nb * 32 * 32 * ci * 4); hipMemset(small, 0, (size_t)nb * 16 * 16 * co * 4);
  hipMemset(w, 0, 25 * ci * co * 4); hipMemset(b, 0, 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  {
    for (int k = 0; k < 3; ++k) launch_conv_taps_mfma(tr, tr ? small : big, w, b, tr ? big : small, g, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int k = 0; k < 20; ++k) launch_conv_taps_mfma(tr, tr ? small : big, w, b, tr ? big : small, g, 0);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("B %d ci %d co %d T %d  %.1f us/launch\n", nb, ci, co, tr, ms * 1000 / 20); fflush(stdout);
  }
  return 0;
}
