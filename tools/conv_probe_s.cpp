// Times the split-bf16 5x5 kernels (kernels_split.hip) next to the float32-MFMA ones (kernels_mfma.hip) on the headline
// geometry, and reports their difference on random data:   conv_probe_s <batch>
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../multiscale_variational_autoencoder_amd/csrc/kernels.h"
namespace mvae { bool launch_conv_wgrad_taprow_f32(const float* big, const float* small, float* dW, float* db, const ConvGeom& g, hipStream_t s); }
namespace mvae { bool launch_conv_taps_mfma(bool transposed, const float* in, const float* w, const float* bias, float* out,
                           const ConvGeom& g, hipStream_t s); }
using namespace mvae;
static float* dev_rand(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  srand(seed);
  for (auto& v : h) v = ((rand() / (float)RAND_MAX) - 0.5f) * 2.f * scale;
  float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  return d;
}
int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 512;
  const int IH = argc > 2 ? atoi(argv[2]) : 32, IW = argc > 3 ? atoi(argv[3]) : IH;       // even sizes
  for (int cfg = 0; cfg < 4; ++cfg) {
    const int ci = cfg < 2 ? 32 : 64, co = cfg < 2 ? 64 : 32, tr = cfg & 1;
    ConvGeom g{nb, IH, IW, ci, IH / 2, IW / 2, co, 5, 5, 2, 2, 1, 1};
    const size_t nbig = (size_t)nb * IH * IW * ci, nsm = (size_t)nb * (IH / 2) * (IW / 2) * co;
    float* big = dev_rand(nbig, 1.f, 1); float* small = dev_rand(nsm, 1.f, 2);
    float* w = dev_rand(25 * ci * co, 0.05f, 3); float* b = dev_rand(256, 0.1f, 4);
    float *o1, *o2; const size_t nout = tr ? nbig : nsm;
    hipMalloc(&o1, nout * 4); hipMalloc(&o2, nout * 4);
    void* planes; hipMalloc(&planes, split_planes_bytes(g));
    launch_split_weights(w, planes, g, 0);
    const float* in = tr ? small : big;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    for (int which = 0; which < 2; ++which) {
      auto run = [&]() { if (which) launch_conv_taps_split(tr, in, planes, b, o2, g, 0); else launch_conv_taps_mfma(tr, in, w, b, o1, g, 0); };
      for (int k = 0; k < 3; ++k) run();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int k = 0; k < 20; ++k) run();
      hipEventRecord(e1); hipDeviceSynchronize();
      hipEventElapsedTime(&ms[which], e0, e1);
    }
    std::vector<float> h1(nout), h2(nout);
    hipMemcpy(h1.data(), o1, nout * 4, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), o2, nout * 4, hipMemcpyDeviceToHost);
    double num = 0, den = 0, mx = 0;
    for (size_t i = 0; i < nout; ++i) { const double d = (double)h1[i] - h2[i]; num += d * d; den += (double)h1[i] * h1[i]; if (std::fabs(d) > mx) mx = std::fabs(d); }
    const double gf = 2.0 * nb * (IH / 2) * (IW / 2) * co * 25 * ci / 1e9;
    printf("B %d ci %d co %d T %d : f32-MFMA %.1f us (%.0f TF)  split %.1f us (%.0f TF)   rel diff %.2e max %.2e\n", nb, ci, co, tr,
           ms[0] * 50, gf / (ms[0] * 50e-6) / 1e3, ms[1] * 50, gf / (ms[1] * 50e-6) / 1e3, std::sqrt(num / den), mx);
    fflush(stdout);
    if (!tr) {      // weight gradient of the same layer: float32-MFMA kernel next to the split-bf16 one
      float *g1, *g2, *b1, *b2;
      const size_t nw = (size_t)25 * ci * co;
      hipMalloc(&g1, nw * 4); hipMalloc(&g2, nw * 4); hipMalloc(&b1, 256 * 4); hipMalloc(&b2, 256 * 4);
      float wms[2];
      for (int which = 0; which < 2; ++which) {
        float* gw = which ? g2 : g1; float* gb = which ? b2 : b1;
        auto run = [&]() { if (which) launch_conv_wgrad_split(big, small, gw, gb, g, 0); else launch_conv_wgrad_taprow_f32(big, small, gw, gb, g, 0); };
        for (int k = 0; k < 2; ++k) run();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int k = 0; k < 10; ++k) run();
        hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&wms[which], e0, e1);
        hipMemset(gw, 0, nw * 4); hipMemset(gb, 0, 256 * 4);
        run(); hipDeviceSynchronize();
      }
      std::vector<float> w1(nw), w2(nw), c1(co), c2(co);
      hipMemcpy(w1.data(), g1, nw * 4, hipMemcpyDeviceToHost); hipMemcpy(w2.data(), g2, nw * 4, hipMemcpyDeviceToHost);
      hipMemcpy(c1.data(), b1, co * 4, hipMemcpyDeviceToHost); hipMemcpy(c2.data(), b2, co * 4, hipMemcpyDeviceToHost);
      double n2 = 0, d2 = 0, nb = 0, dbb = 0;
      for (size_t i = 0; i < nw; ++i) { const double d = (double)w1[i] - w2[i]; n2 += d * d; d2 += (double)w1[i] * w1[i]; }
      for (int i = 0; i < co; ++i) { const double d = (double)c1[i] - c2[i]; nb += d * d; dbb += (double)c1[i] * c1[i]; }
      printf("      wgrad ci %d co %d : f32-MFMA %.1f us (%.0f TF)  split %.1f us (%.0f TF)   rel diff dW %.2e db %.2e\n", ci, co, wms[0] * 100,
             gf / (wms[0] * 100e-6) / 1e3, wms[1] * 100, gf / (wms[1] * 100e-6) / 1e3, std::sqrt(n2 / d2), std::sqrt(nb / (dbb + 1e-30)));
      hipFree(g1); hipFree(g2); hipFree(b1); hipFree(b2);
    }
    hipFree(big); hipFree(small); hipFree(w); hipFree(b); hipFree(o1); hipFree(o2); hipFree(planes);
  }
  return 0;
}
