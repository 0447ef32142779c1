// dispatch.hip -- chooses, per op and shape, between the shape-generic kernels (kernels_generic.hip) and
// the LDS/MFMA-tiled gfx950 kernels (kernels_mfma.hip).  Both are device paths; there is no CPU fallback.
#include "kernels.h"
#include "prof.h"
#include <cstdio>

namespace mvae {

Profiler& profiler() {
  static Profiler p;
  return p;
}

static bool g_det_mode = false;
bool det_mode() { return g_det_mode; }
void set_det_mode(bool on) { g_det_mode = on; }

static int g_fused_stats[3] = {0, 0, 0};
void fused_launch_note(bool fwd, int B, int grid) {
  ++g_fused_stats[fwd ? 0 : 1];
  const int ipb = (B + grid - 1) / grid;
  if (ipb > g_fused_stats[2]) g_fused_stats[2] = ipb;
}
void fused_launch_stats(int out[3]) { for (int i = 0; i < 3; ++i) out[i] = g_fused_stats[i]; }

static inline double f4(double n) { return 4.0 * n; }
// tag with the row count (log2 bucket) so the profile separates the big launches from the launch-bound ones
static const char* tagm(const char* base, double rows) {
  static thread_local char buf[64];
  if (!profiler().on) return base;
  int lg = 0;
  while ((1ll << lg) < (long long)rows) ++lg;
  snprintf(buf, sizeof(buf), "%s@2^%d", base, lg);
  return buf;
}

void launch_conv_f_generic(const float*, const float*, const float*, const float*, float*, ConvGeom, PreOp, int,
                           hipStream_t);
void launch_conv_t_generic(const float*, const float*, const float*, const float*, float*, ConvGeom, hipStream_t);
void launch_conv_wgrad_generic(const float*, const float*, float*, ConvGeom, PreOp, hipStream_t);
void launch_dw_fwd_generic(const float*, const float*, const float*, float*, int, int, int, int, hipStream_t);
void launch_dw_bwd_data_generic(const float*, const float*, const float*, float*, int, int, int, int, hipStream_t);
void launch_dw_wgrad_generic(const float*, const float*, float*, float*, int, int, int, int, hipStream_t);
void launch_mn_dt1pre_generic(float*, const float*, const float*, const float*, int, int64_t, int, float,
                              hipStream_t);
void launch_gemm_nn_generic(const float*, const float*, const float*, float*, float*, int, int, int, int,
                            hipStream_t);
void launch_gemm_nt_generic(const float*, const float*, float*, int, int, int, const float*, int, hipStream_t);
void launch_gemm_tn_generic(const float*, const float*, float*, float*, int, int, int, const float*, const float*,
                            const float*, hipStream_t);
bool launch_conv1x1_mfma(bool transposed, const float* in, const float* w, const float* bias, const float* residual,
                         float* out, const ConvGeom& g, PreOp pre, int act, const float* dot_src, float* dot_out,
                         hipStream_t s);
bool launch_gemm_tn_opt(const float*, const float*, float*, float*, int, int, int, const float*, const float*,
                        const float*, hipStream_t);
bool launch_gemm_nn_opt(const float*, const float*, const float*, float*, float*, int, int, int, int, hipStream_t);
bool launch_gemm_nt_opt(const float*, const float*, float*, int, int, int, const float*, int, hipStream_t);
bool launch_dw_fwd_opt(const float*, const float*, const float*, float*, int, int, int, int, hipStream_t);
bool launch_dw_bwd_data_opt(const float*, const float*, const float*, float*, int, int, int, int, hipStream_t);
bool launch_dw_wgrad_opt(const float*, const float*, float*, float*, int, int, int, int, hipStream_t);
bool launch_conv_taps_mfma(bool transposed, const float* in, const float* w, const float* bias, float* out,
                           const ConvGeom& g, hipStream_t s);
bool launch_conv_wgrad_mfma(const float* big, const float* small, float* dW, float* db, const ConvGeom& g, PreOp pre, GradSlots sl,
                            hipStream_t s);

void launch_conv_f(const float* big, const float* w, const float* bias, const float* residual, float* small,
                   ConvGeom g, PreOp pre, int act, hipStream_t s) {
  double nb = (double)g.B * g.IH * g.IW * g.CI, ns = (double)g.B * g.OH * g.OW * g.CO, nw = (double)g.KH * g.KW * g.CI * g.CO;
  ProfScope ps(tagm(g.KH * g.KW == 1 ? "conv1x1_f" : "convkxk_f", (double)g.B * g.OH * g.OW), f4(nb + ns * (residual ? 2 : 1) + nw), 2.0 * ns * g.KH * g.KW * g.CI, s);
  if (launch_conv1x1_mfma(false, big, w, bias, residual, small, g, pre, act, nullptr, nullptr, s)) return;
  if (g.KH * g.KW > 1 && !residual && !pre.gate && !pre.scale && act == ACT_NONE &&
      launch_conv_taps_mfma(false, big, w, bias, small, g, s)) return;
  launch_conv_f_generic(big, w, bias, residual, small, g, pre, act, s);
}
void launch_conv_t(const float* small, const float* w, const float* bias, const float* residual, float* big,
                   ConvGeom g, hipStream_t s) {
  double nb = (double)g.B * g.IH * g.IW * g.CI, ns = (double)g.B * g.OH * g.OW * g.CO, nw = (double)g.KH * g.KW * g.CI * g.CO;
  ProfScope ps(tagm(g.KH * g.KW == 1 ? "conv1x1_t" : "convkxk_t", (double)g.B * g.IH * g.IW), f4(ns + nb * (residual ? 2 : 1) + nw), 2.0 * ns * g.KH * g.KW * g.CI, s);
  PreOp none{nullptr, nullptr, nullptr};
  if (launch_conv1x1_mfma(true, small, w, bias, residual, big, g, none, ACT_NONE, nullptr, nullptr, s)) return;
  if (g.KH * g.KW > 1 && !residual && launch_conv_taps_mfma(true, small, w, bias, big, g, s)) return;
  launch_conv_t_generic(small, w, bias, residual, big, g, s);
}
void launch_conv_wgrad(const float* big, const float* small, float* dW, float* db, ConvGeom g, PreOp pre, GradSlots sl,
                       hipStream_t s) {
  double nb = (double)g.B * g.IH * g.IW * g.CI, ns = (double)g.B * g.OH * g.OW * g.CO, nw = (double)g.KH * g.KW * g.CI * g.CO;
  ProfScope ps(tagm(g.KH * g.KW == 1 ? "conv1x1_wgrad" : "convkxk_wgrad", (double)g.B * g.OH * g.OW), f4(nb + ns + nw), 2.0 * ns * g.KH * g.KW * g.CI, s);
  if (launch_conv_wgrad_mfma(big, small, dW, db, g, pre, sl, s)) return;
  launch_conv_wgrad_generic(big, small, dW, g, pre, s);
  if (db) launch_colsum(small, db, (int64_t)g.B * g.OH * g.OW, g.CO, s);
}
void launch_dw_fwd(const float* in, const float* w, const float* b, float* out, int B, int H, int W, int C,
                   hipStream_t s) {
  double n = (double)B * H * W * C;
  ProfScope ps(tagm("dw_fwd", n), f4(2 * n), 18.0 * n, s);
  if (launch_dw_fwd_opt(in, w, b, out, B, H, W, C, s)) return;
  launch_dw_fwd_generic(in, w, b, out, B, H, W, C, s);
}
void launch_dw_bwd_data(const float* dy, const float* w, const float* mask_src, float* dx, int B, int H, int W,
                        int C, hipStream_t s) {
  double n = (double)B * H * W * C;
  ProfScope ps(tagm("dw_bwd_data", n), f4(3 * n), 18.0 * n, s);
  if (launch_dw_bwd_data_opt(dy, w, mask_src, dx, B, H, W, C, s)) return;
  launch_dw_bwd_data_generic(dy, w, mask_src, dx, B, H, W, C, s);
}
void launch_dw_wgrad(const float* in, const float* dy, float* dW, float* db, int B, int H, int W, int C,
                     hipStream_t s) {
  double n = (double)B * H * W * C;
  ProfScope ps(tagm("dw_wgrad", n), f4(2 * n), 20.0 * n, s);
  if (launch_dw_wgrad_opt(in, dy, dW, db, B, H, W, C, s)) return;
  launch_dw_wgrad_generic(in, dy, dW, db, B, H, W, C, s);
}
void launch_mn_dt1pre(float* d, const float* t1, const float* g, const float* dgap, int B, int64_t HW, int C,
                      float inv_hw, hipStream_t s) {
  double n = (double)B * HW * C;
  ProfScope ps(tagm("mn_dt1pre", n), f4(3 * n), 3.0 * n, s);
  launch_mn_dt1pre_generic(d, t1, g, dgap, B, HW, C, inv_hw, s);
}
void launch_gemm_nn(const float* a, const float* w, const float* bias, float* out, float* out_lin, int B, int K,
                    int N, int act, hipStream_t s) {
  ProfScope ps(tagm("gemm_nn", (double)K * N), f4((double)B * K + (double)K * N + (double)B * N), 2.0 * B * K * N, s);
  if (launch_gemm_nn_opt(a, w, bias, out, out_lin, B, K, N, act, s)) return;
  launch_gemm_nn_generic(a, w, bias, out, out_lin, B, K, N, act, s);
}
void launch_gemm_nt(const float* a, const float* w, float* out, int B, int K, int N, const float* hs_lin,
                    int accumulate, hipStream_t s) {
  ProfScope ps(tagm("gemm_nt", (double)K * N), f4((double)B * K + (double)K * N + (double)B * N), 2.0 * B * K * N, s);
  if (launch_gemm_nt_opt(a, w, out, B, K, N, hs_lin, accumulate, s)) return;
  launch_gemm_nt_generic(a, w, out, B, K, N, hs_lin, accumulate, s);
}
void launch_gemm_tn(const float* a, const float* g, float* dW, float* db, int B, int K, int N, const float* a_scale,
                    const float* a_shift, const float* hs_lin, hipStream_t s) {
  ProfScope ps(tagm("gemm_tn", (double)K * N), f4((double)B * K + (double)K * N + (double)B * N), 2.0 * B * K * N, s);
  if (launch_gemm_tn_opt(a, g, dW, db, B, K, N, a_scale, a_shift, hs_lin, s)) return;
  launch_gemm_tn_generic(a, g, dW, db, B, K, N, a_scale, a_shift, hs_lin, s);
}

// big = convT(small) for a 1x1 layer, and dot_out[b,c] = sum_hw big * dot_src in the same pass when the MFMA path
// applies; otherwise the two separate launches.
void launch_conv_t_dot(const float* small, const float* w, float* big, const float* dot_src, float* dot_out,
                       ConvGeom g, hipStream_t s) {
  const int64_t HW = (int64_t)g.IH * g.IW;
  {
    double nb = (double)g.B * HW * g.CI, ns = (double)g.B * HW * g.CO;
    ProfScope ps(tagm("conv1x1_t_dot", (double)g.B * HW), f4(ns + 2 * nb), 2.0 * ns * g.CI, s);
    PreOp none{nullptr, nullptr, nullptr};
    launch_zero(dot_out, (int64_t)g.B * g.CI, s);
    if (launch_conv1x1_mfma(true, small, w, nullptr, nullptr, big, g, none, ACT_NONE, dot_src, dot_out, s)) return;
  }
  launch_conv_t(small, w, nullptr, nullptr, big, g, s);
  launch_spatial_dot(big, dot_src, dot_out, g.B, HW, g.CI, s);
}

}  // namespace mvae
