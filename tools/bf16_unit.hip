// Stand-alone check of kernels_bf16.hip against a double-precision CPU reference (bf16-rounded operands), one kernel at a
// time.  Build (on the GPU box or here): hipcc --offload-arch=gfx950 -O2 tools/bf16_unit.hip csrc/kernels_bf16.o
// csrc/kernels_generic.o csrc/dispatch.o ... -o bf16_unit   (see tools/run_bf16_unit.sh)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../multiscale_variational_autoencoder_amd/csrc/kernels.h"
using namespace mvae;
namespace mvae {
bool launch16_pw(bool transposed, const void* in, const float* w, const float* bias, const float* gate, const void* residual,
                 void* out, int64_t M, int64_t rows_per_image, int K, int N, int act, hipStream_t s, bool out_f32 = false,
                 const float* pivot = nullptr, float* st1 = nullptr, float* st2 = nullptr, int nslots = 1, int64_t slot_stride = 0);
bool launch16_dual(const void* X, const float* W, const void* aux, const float* gate, const void* residual, void* Y,
                   float* dW, float* db, float* dot_out, int64_t M, int64_t rows_per_image, int C, GradSlots sl, hipStream_t s, bool embed_mask = false);
bool launch16_taps(bool transposed, const void* in, const float* w, const float* bias, void* out, const ConvGeom& g,
                   hipStream_t s, const float* w2 = nullptr, const float* bias2 = nullptr, void* out2 = nullptr,
                   bool* chained = nullptr);
bool launch16_wgrad(const void* big, const void* small, float* dW, float* db, const ConvGeom& g, GradSlots sl, hipStream_t s,
                    float* db_big = nullptr, bool* db_big_done = nullptr);
}
static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float b2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static float rb(float f) { return b2f(f2b(f)); }
static float rnd() { return (float)rand() / RAND_MAX * 2.f - 1.f; }
template <class T> T* dev(const std::vector<T>& v) { T* p; hipMalloc(&p, v.size() * sizeof(T)); hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice); return p; }
template <class T> std::vector<T> host(const T* p, size_t n) { std::vector<T> v(n); hipMemcpy(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost); return v; }
static double relerr(const std::vector<double>& ref, const std::vector<double>& got) {
  double a = 0, b = 0; for (size_t i = 0; i < ref.size(); ++i) { a += (ref[i] - got[i]) * (ref[i] - got[i]); b += ref[i] * ref[i]; } return std::sqrt(a / (b + 1e-300));
}
static std::vector<double> b2d(const std::vector<uint16_t>& v) { std::vector<double> o(v.size()); for (size_t i = 0; i < v.size(); ++i) o[i] = b2f(v[i]); return o; }
static std::vector<double> f2d(const std::vector<float>& v) { return std::vector<double>(v.begin(), v.end()); }

static void test_pw(int K, int N, bool WT, bool gate, bool res, int act) {
  const int B = 3, HW = 96; const int64_t M = (int64_t)B * HW;
  std::vector<uint16_t> X(M * K), R(M * N); std::vector<float> W(K * N), bias(N), G(B * K);
  for (auto& v : X) v = f2b(rnd()); for (auto& v : R) v = f2b(rnd()); for (auto& v : W) v = rnd() * 0.2f; for (auto& v : bias) v = rnd(); for (auto& v : G) v = 0.5f + 0.5f * rnd();
  std::vector<double> ref(M * N);
  for (int64_t m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
    double a = 0; for (int k = 0; k < K; ++k) { float x = b2f(X[m * K + k]); if (gate) x = rb(x * G[(m / HW) * K + k]); a += (double)x * rb(WT ? W[n * K + k] : W[k * N + n]); }
    a += bias[n]; if (act == ACT_RELU) a = a > 0 ? a : 0; if (res) a += b2f(R[m * N + n]); ref[m * N + n] = a;
  }
  auto dX = dev(X); auto dR = dev(R); auto dWt = dev(W); auto db = dev(bias); auto dG = dev(G); uint16_t* dY; hipMalloc(&dY, M * N * 2); hipMemset(dY, 0, M * N * 2);
  bool ok = launch16_pw(WT, dX, dWt, db, gate ? dG : nullptr, res ? dR : nullptr, dY, M, HW, K, N, act, nullptr);
  hipDeviceSynchronize();
  printf("pw K=%d N=%d WT=%d gate=%d res=%d act=%d: launched=%d rel=%.3e\n", K, N, WT, gate, res, act, ok, relerr(ref, b2d(host(dY, M * N))));
}
static void test_dual(int C, int mode) {
  const int B = 3, HW = 96; const int64_t M = (int64_t)B * HW;
  std::vector<uint16_t> X(M * C), A(M * C), R(M * C); std::vector<float> W(C * C), G(B * C);
  for (auto& v : X) v = f2b(rnd()); for (auto& v : A) v = f2b(rnd() + 0.3f); for (auto& v : R) v = f2b(rnd()); for (auto& v : W) v = rnd() * 0.2f; for (auto& v : G) v = 0.5f + 0.5f * rnd();
  std::vector<double> Y(M * C), dW(C * C, 0.0), dbv(C, 0.0), dot(B * C, 0.0);
  for (int64_t m = 0; m < M; ++m) for (int ci = 0; ci < C; ++ci) {
    double a = 0, ae = 0; for (int co = 0; co < C; ++co) { a += (double)b2f(X[m * C + co]) * rb(W[ci * C + co]); ae += (double)b2f(X[m * C + co]) * W[ci * C + co]; }
    if (mode == 2) a += b2f(R[m * C + ci]);
    Y[m * C + ci] = a;   // (mode 1: the kernel rounds one bit shorter and carries the mask of aux in the LSB: ~4e-3 rel)
    dot[(m / HW) * C + ci] += ae * b2f(A[m * C + ci]);     // dg = sum_hw (X . W^T with the float32 W) * aux
  }
  for (int64_t m = 0; m < M; ++m) for (int ci = 0; ci < C; ++ci) { const double av = (double)b2f(A[m * C + ci]) * (mode == 1 ? G[(m / HW) * C + ci] : 1.0);
    for (int co = 0; co < C; ++co) dW[ci * C + co] += av * b2f(X[m * C + co]); }
  for (int64_t m = 0; m < M; ++m) for (int co = 0; co < C; ++co) dbv[co] += b2f(X[m * C + co]);
  auto dX = dev(X); auto dA = dev(A); auto dR = dev(R); auto dWt = dev(W); auto dG = dev(G);
  uint16_t* dY; hipMalloc(&dY, M * C * 2); float *gW, *gb, *gd; hipMalloc(&gW, C * C * 4); hipMalloc(&gb, C * 4); hipMalloc(&gd, B * C * 4);
  hipMemset(gW, 0, C * C * 4); hipMemset(gb, 0, C * 4); hipMemset(gd, 0, B * C * 4);
  GradSlots sl;
  bool ok = launch16_dual(dX, dWt, dA, mode == 1 ? dG : nullptr, mode == 2 ? dR : nullptr, dY, gW, gb, mode == 1 ? gd : nullptr, M, HW, C, sl, nullptr);
  hipDeviceSynchronize();
  printf("dual C=%d mode=%d: launched=%d Y rel=%.3e dW rel=%.3e db rel=%.3e", C, mode, ok, relerr(Y, b2d(host(dY, M * C))), relerr(dW, f2d(host(gW, C * C))), relerr(dbv, f2d(host(gb, C))));
  if (mode == 1) printf(" dot rel=%.3e", relerr(dot, f2d(host(gd, B * C))));
  printf("\n");
}
static void conv_ref(const std::vector<uint16_t>& big, const std::vector<float>& W, const ConvGeom& g, std::vector<double>& small) {
  small.assign((size_t)g.B * g.OH * g.OW * g.CO, 0.0);
  for (int b = 0; b < g.B; ++b) for (int oh = 0; oh < g.OH; ++oh) for (int ow = 0; ow < g.OW; ++ow) for (int kh = 0; kh < g.KH; ++kh) for (int kw = 0; kw < g.KW; ++kw) {
    int y = oh * g.SH + kh - g.PT, x = ow * g.SW + kw - g.PL; if (y < 0 || y >= g.IH || x < 0 || x >= g.IW) continue;
    for (int ci = 0; ci < g.CI; ++ci) { double v = b2f(big[(((size_t)b * g.IH + y) * g.IW + x) * g.CI + ci]);
      for (int co = 0; co < g.CO; ++co) small[(((size_t)b * g.OH + oh) * g.OW + ow) * g.CO + co] += v * rb(W[((kh * g.KW + kw) * g.CI + ci) * g.CO + co]); } }
}
static void test_taps(int CI, int CO, int IH, int IW) {
  ConvGeom g{}; g.B = 2; g.IH = IH; g.IW = IW; g.CI = CI; g.CO = CO; g.KH = g.KW = 5; g.SH = g.SW = 2; g.OH = (IH + 1) / 2; g.OW = (IW + 1) / 2;
  int tot = (g.OH - 1) * 2 + 5 - IH; g.PT = tot > 0 ? tot / 2 : 0; tot = (g.OW - 1) * 2 + 5 - IW; g.PL = tot > 0 ? tot / 2 : 0;
  const size_t nb = (size_t)g.B * IH * IW * CI, ns = (size_t)g.B * g.OH * g.OW * CO;
  std::vector<uint16_t> big(nb), small(ns); std::vector<float> W(25 * CI * CO), bias(CO), biasT(CI);
  for (auto& v : big) v = f2b(rnd()); for (auto& v : small) v = f2b(rnd()); for (auto& v : W) v = rnd() * 0.1f; for (auto& v : bias) v = rnd(); for (auto& v : biasT) v = rnd();
  // F-form
  std::vector<double> ref; conv_ref(big, W, g, ref); for (size_t i = 0; i < ns; ++i) ref[i] += bias[i % CO];
  auto dB = dev(big); auto dS = dev(small); auto dWt = dev(W); auto db = dev(bias); auto dbT = dev(biasT);
  uint16_t *oS, *oB; hipMalloc(&oS, ns * 2); hipMalloc(&oB, nb * 2); hipMemset(oS, 0, ns * 2); hipMemset(oB, 0, nb * 2);
  bool ok = launch16_taps(false, dB, dWt, db, oS, g, nullptr); hipDeviceSynchronize();
  printf("taps F %d->%d %dx%d: launched=%d rel=%.3e\n", CI, CO, IH, IW, ok, relerr(ref, b2d(host(oS, ns))));
  // T-form: big[b,y,x,ci] = biasT[ci] + sum small[...] W
  std::vector<double> rt(nb, 0.0);
  for (int b = 0; b < g.B; ++b) for (int oh = 0; oh < g.OH; ++oh) for (int ow = 0; ow < g.OW; ++ow) for (int kh = 0; kh < 5; ++kh) for (int kw = 0; kw < 5; ++kw) {
    int y = oh * 2 + kh - g.PT, x = ow * 2 + kw - g.PL; if (y < 0 || y >= IH || x < 0 || x >= IW) continue;
    for (int co = 0; co < CO; ++co) { double v = b2f(small[(((size_t)b * g.OH + oh) * g.OW + ow) * CO + co]);
      for (int ci = 0; ci < CI; ++ci) rt[(((size_t)b * IH + y) * IW + x) * CI + ci] += v * rb(W[((kh * 5 + kw) * CI + ci) * CO + co]); } }
  for (size_t i = 0; i < nb; ++i) rt[i] += biasT[i % CI];
  ok = launch16_taps(true, dS, dWt, dbT, oB, g, nullptr); hipDeviceSynchronize();
  printf("taps T %d->%d %dx%d: launched=%d rel=%.3e\n", CO, CI, IH, IW, ok, relerr(rt, b2d(host(oB, nb))));
  // wgrad
  std::vector<double> rw(25 * CI * CO, 0.0), rbias(CO, 0.0);
  for (int b = 0; b < g.B; ++b) for (int oh = 0; oh < g.OH; ++oh) for (int ow = 0; ow < g.OW; ++ow) {
    for (int co = 0; co < CO; ++co) rbias[co] += b2f(small[(((size_t)b * g.OH + oh) * g.OW + ow) * CO + co]);
    for (int kh = 0; kh < 5; ++kh) for (int kw = 0; kw < 5; ++kw) { int y = oh * 2 + kh - g.PT, x = ow * 2 + kw - g.PL; if (y < 0 || y >= IH || x < 0 || x >= IW) continue;
      for (int ci = 0; ci < CI; ++ci) { double v = b2f(big[(((size_t)b * IH + y) * IW + x) * CI + ci]);
        for (int co = 0; co < CO; ++co) rw[((kh * 5 + kw) * CI + ci) * CO + co] += v * b2f(small[(((size_t)b * g.OH + oh) * g.OW + ow) * CO + co]); } } }
  float *gW, *gb; hipMalloc(&gW, rw.size() * 4); hipMalloc(&gb, CO * 4); hipMemset(gW, 0, rw.size() * 4); hipMemset(gb, 0, CO * 4);
  float* gbb; hipMalloc(&gbb, CI * 4); hipMemset(gbb, 0, CI * 4); bool bdone = false;      // column sums of `big` (a convT's bias gradient)
  GradSlots sl; ok = launch16_wgrad(dB, dS, gW, gb, g, sl, nullptr, gbb, &bdone); hipDeviceSynchronize();
  printf("wgrad 5x5 %d,%d: launched=%d dW rel=%.3e db rel=%.3e\n", CI, CO, ok, relerr(rw, f2d(host(gW, rw.size()))), relerr(rbias, f2d(host(gb, CO))));
  if (bdone) {
    std::vector<double> rbb(CI, 0.0);
    for (size_t i = 0; i < nb; ++i) rbb[i % CI] += b2f(big[i]);
    printf("wgrad 5x5 %d,%d %dx%d big column sums: launched=1 db rel=%.3e\n", CI, CO, IH, IW, relerr(rbb, f2d(host(gbb, CI))));
  }
}
// one-pass column statistics about a pivot (k_colstat4<2> + k_bn2d_finalize), float32 input: the adverse case of a column
// whose mean is `off` standard deviations away from zero (E[x^2] - E[x]^2 would lose log2(off^2) bits; the pivot is a sample)
static void test_colstat(int64_t M, int C, float off) {
  std::vector<float> x((size_t)M * C); for (auto& v : x) v = off + 0.5f * rnd();
  std::vector<double> mu(C, 0.0), var(C, 0.0);
  for (int64_t m = 0; m < M; ++m) for (int c = 0; c < C; ++c) mu[c] += x[m * C + c];
  for (int c = 0; c < C; ++c) mu[c] /= (double)M;
  for (int64_t m = 0; m < M; ++m) for (int c = 0; c < C; ++c) { const double d = x[m * C + c] - mu[c]; var[c] += d * d; }
  for (int c = 0; c < C; ++c) var[c] /= (double)M;
  auto dx = dev(x); const int ns = 16;
  std::vector<float> ones(C, 1.f), zeros(C, 0.f); auto dg = dev(ones); auto dbt = dev(zeros);
  float *st, *o; hipMalloc(&st, 2 * ns * C * 4); hipMemset(st, 0, 2 * ns * C * 4); hipMalloc(&o, 8 * C * 4);
  bool ok = launch_colstat_opt(2, dx, nullptr, 0, 0.f, st, ns, C, M, C, nullptr, false, st + ns * C);
  launch_bn2d_finalize(st, st + ns * C, dg, dbt, dbt, dbt, o, o + C, o + 2 * C, o + 3 * C, o + 4 * C, o + 5 * C, M, C, 1e-4f, 1, ns, nullptr, dx);
  hipDeviceSynchronize();
  auto got = host(o, 8 * C);
  std::vector<double> gm(got.begin() + 4 * C, got.begin() + 5 * C), gv(got.begin() + 5 * C, got.begin() + 6 * C);
  // the mean is compared on the scale of the standard deviation (what the normalisation sees), the variance relatively
  double em = 0; for (int c = 0; c < C; ++c) em = std::max(em, std::fabs(gm[c] - mu[c]) / std::sqrt(var[c]));
  printf("colstat one-pass M=%lld C=%d offset=%g: launched=%d mean rel=%.3e var rel=%.3e\n", (long long)M, C, off, ok, em, relerr(var, gv));
}
static void test_wgrad_pw(int CI, int CO) {
  ConvGeom g{}; g.B = 3; g.IH = g.OH = 8; g.IW = g.OW = 12; g.CI = CI; g.CO = CO; g.KH = g.KW = g.SH = g.SW = 1;
  const int64_t M = (int64_t)g.B * 96; std::vector<uint16_t> big(M * CI), small(M * CO);
  for (auto& v : big) v = f2b(rnd()); for (auto& v : small) v = f2b(rnd());
  std::vector<double> rw(CI * CO, 0.0), rbias(CO, 0.0);
  for (int64_t m = 0; m < M; ++m) { for (int co = 0; co < CO; ++co) rbias[co] += b2f(small[m * CO + co]);
    for (int ci = 0; ci < CI; ++ci) for (int co = 0; co < CO; ++co) rw[ci * CO + co] += (double)b2f(big[m * CI + ci]) * b2f(small[m * CO + co]); }
  auto dB = dev(big); auto dS = dev(small); float *gW, *gb; hipMalloc(&gW, CI * CO * 4); hipMalloc(&gb, CO * 4); hipMemset(gW, 0, CI * CO * 4); hipMemset(gb, 0, CO * 4);
  GradSlots sl; bool ok = launch16_wgrad(dB, dS, gW, gb, g, sl, nullptr); hipDeviceSynchronize();
  printf("wgrad 1x1 %d,%d: launched=%d dW rel=%.3e db rel=%.3e\n", CI, CO, ok, relerr(rw, f2d(host(gW, CI * CO))), relerr(rbias, f2d(host(gb, CO))));
}
// timing only (no reference): the 5 x 5 stride-2 family at the sizes of a C256-nb step.   bf16_unit.bin time [B]
static uint16_t* dev_pattern(size_t n) {
  std::vector<uint16_t> pat(1 << 20); for (auto& v : pat) v = f2b(rnd());
  uint16_t* p; hipMalloc(&p, n * 2);
  for (size_t o = 0; o < n; o += pat.size()) hipMemcpy(p + o, pat.data(), std::min(pat.size(), n - o) * 2, hipMemcpyHostToDevice);
  return p;
}
static void time_taps(int CI, int CO, int B, int IH, int IW) {
  ConvGeom g{}; g.B = B; g.IH = IH; g.IW = IW; g.CI = CI; g.CO = CO; g.KH = g.KW = 5; g.SH = g.SW = 2; g.OH = IH / 2; g.OW = IW / 2; g.PT = g.PL = 1;
  const size_t nb = (size_t)B * IH * IW * CI, ns = (size_t)B * g.OH * g.OW * CO;
  uint16_t *big = dev_pattern(nb), *small = dev_pattern(ns), *oS, *oB; hipMalloc(&oS, ns * 2); hipMalloc(&oB, nb * 2);
  std::vector<float> W(25 * CI * CO), bias(64, 0.f); for (auto& v : W) v = rnd() * 0.1f;
  auto dWt = dev(W); auto db = dev(bias); float *gW, *gb; hipMalloc(&gW, W.size() * 4); hipMalloc(&gb, 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flops = 2.0 * B * g.OH * g.OW * 25.0 * CI * CO;
  for (int which = 0; which < 3; ++which) {
    const int reps = 8; float ms = 0;
    for (int r = -2; r < reps; ++r) {
      if (r == 0) hipEventRecord(e0, nullptr);
      if (which == 0) launch16_taps(false, big, dWt, db, oS, g, nullptr);
      else if (which == 1) launch16_taps(true, small, dWt, db, oB, g, nullptr);
      else { GradSlots sl; launch16_wgrad(big, small, gW, gb, g, sl, nullptr); }
    }
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    const char* nm[] = {"F (conv)", "T (convT)", "wgrad"};
    printf("time %-9s %d<->%d B=%d %dx%d: %8.1f us  %6.1f TFLOP/s\n", nm[which], CI, CO, B, IH, IW, ms * 1000 / reps, flops / (ms / reps * 1e-3) / 1e12);
  }
  hipFree(big); hipFree(small); hipFree(oS); hipFree(oB); hipFree(gW); hipFree(gb);
}
int main(int argc, char** argv) {
  srand(1);
  if (argc > 1 && !strcmp(argv[1], "time")) {
    const int B = argc > 2 ? atoi(argv[2]) : 64;
    for (int sz = 256; sz >= 32; sz /= 2) { time_taps(32, 64, B, sz, sz); time_taps(64, 32, B, sz, sz); }
    printf("last hip error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
  }
  test_pw(64, 64, false, false, false, ACT_RELU); test_pw(64, 64, false, true, true, ACT_NONE); test_pw(32, 32, false, true, true, ACT_NONE);
  test_pw(64, 32, false, false, false, ACT_NONE); test_pw(32, 64, true, false, false, ACT_NONE); test_pw(64, 32, true, false, false, ACT_NONE); test_pw(32, 64, false, false, false, ACT_NONE);
  test_dual(64, 1); test_dual(64, 2); test_dual(32, 1); test_dual(32, 2);
  test_taps(32, 64, 16, 16); test_taps(64, 32, 16, 16); test_taps(32, 64, 12, 20);
  test_taps(32, 64, 8, 64); test_taps(64, 32, 8, 64); test_taps(32, 64, 6, 128); test_taps(64, 32, 10, 128);   // OW = 32 / 64: k16_wgrad_seg, k16_taps_fr<.., 32>
  test_wgrad_pw(64, 32); test_wgrad_pw(32, 64);
  test_colstat(200000, 32, 0.f); test_colstat(200000, 32, 100.f); test_colstat(65536, 64, -30.f);
  printf("last hip error: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
