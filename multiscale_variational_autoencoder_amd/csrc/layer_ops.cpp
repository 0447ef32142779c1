// layer_ops.cpp -- stateless C-ABI operators (include/mvae_hip.h, "layer operators") from which the Python facade assembles
// the reference's remaining blocks: attention_block / self_attention_block (layer_blocks.py:654-783), the excite / inhibit
// masks and block (:191-412), resnet_block with strides and the BatchNormalization variants (:521-537, 847-853, 884-886).
// The reference composes Keras layers in Python; this library offers the same granularity as device operators -- every
// byte of arithmetic runs in a HIP kernel, the host only sequences them.  The convolutions are the hot path's launchers.
#include "../../include/mvae_hip.h"
#include "kernels.h"

using namespace mvae;

namespace {
void same_pad1(int n, int k, int s, int* out, int* before) {
  *out = (n + s - 1) / s;
  int total = (*out - 1) * s + k - n;
  if (total < 0) total = 0;
  *before = total / 2;
}
ConvGeom conv_geom(int B, int H, int W, int C, int F, int kh, int kw, int sh, int sw) {
  int oh, ow, pt, pl;
  same_pad1(H, kh, sh, &oh, &pt);
  same_pad1(W, kw, sw, &ow, &pl);
  return ConvGeom{B, H, W, C, oh, ow, F, kh, kw, sh, sw, pt, pl};
}
int done() { return hipGetLastError() == hipSuccess ? MVAE_OK : MVAE_E_HIP; }
// stateless entry points never inherit a handle's deterministic-reduction mode (a process-global launcher switch)
bool use_device(int device) {
  if (hipSetDevice(device) != hipSuccess) return false;
  set_det_mode(false);
  return true;
}
}  // namespace

extern "C" {

int mvae_conv2d_forward(int32_t device, const float* x, int32_t B, int32_t H, int32_t W, int32_t C, const float* w, const float* b,
                        int32_t F, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t relu, float* y, void* stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || F <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0) return MVAE_E_INVALID;
  if (!use_device(device)) return MVAE_E_HIP;
  PreOp none{nullptr, nullptr, nullptr};
  launch_conv_f(x, w, b, nullptr, y, conv_geom(B, H, W, C, F, kh, kw, sh, sw), none, relu ? ACT_RELU : ACT_NONE,
                static_cast<hipStream_t>(stream));
  return done();
}

int mvae_conv2d_backward(int32_t device, const float* x, const float* dpre, int32_t B, int32_t H, int32_t W, int32_t C,
                         const float* w, int32_t F, int32_t kh, int32_t kw, int32_t sh, int32_t sw, float* dx, float* dw, float* db,
                         void* stream) {
  if (!x || !dpre || !w || !dw || B <= 0 || H <= 0 || W <= 0 || C <= 0 || F <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0)
    return MVAE_E_INVALID;
  if (!use_device(device)) return MVAE_E_HIP;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const ConvGeom g = conv_geom(B, H, W, C, F, kh, kw, sh, sw);
  PreOp none{nullptr, nullptr, nullptr};
  GradSlots direct;
  launch_conv_wgrad(x, dpre, dw, db, g, none, direct, s);
  if (dx) launch_conv_t(dpre, w, nullptr, nullptr, dx, g, s);
  return done();
}

int mvae_depthwise3x3_forward(int32_t device, const float* x, int32_t B, int32_t H, int32_t W, int32_t C, const float* w,
                              const float* b, float* y, void* stream) {
  if (!x || !w || !b || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return MVAE_E_INVALID;
  if (!use_device(device)) return MVAE_E_HIP;
  launch_dw_fwd(x, w, b, y, B, H, W, C, static_cast<hipStream_t>(stream));          // 3x3, SAME, bias, ReLU
  return done();
}

int mvae_depthwise3x3_backward(int32_t device, const float* x, const float* y, const float* dy, int32_t B, int32_t H, int32_t W,
                               int32_t C, const float* w, float* dx, float* dw, float* db, float* work, void* stream) {
  if (!x || !y || !dy || !w || !dx || !dw || !db || !work || B <= 0 || H <= 0 || W <= 0 || C <= 0) return MVAE_E_INVALID;
  if (!use_device(device)) return MVAE_E_HIP;
  hipStream_t s = static_cast<hipStream_t>(stream);
  launch_relu_bwd(dy, y, work, (int64_t)B * H * W * C, s);                            // through the ReLU
  launch_dw_wgrad(x, work, dw, db, B, H, W, C, s);
  launch_dw_bwd_plain(work, w, dx, B, H, W, C, s);
  return done();
}

int mvae_activation_forward(int32_t device, int32_t act, const float* x, float* y, int64_t n, float param, void* stream) {
  if (!x || !y || n <= 0 || act < 0 || act > LAYER_ACT_ATTENUATE) return MVAE_E_INVALID;
  if (!use_device(device)) return MVAE_E_HIP;
  launch_act_fwd(act, x, y, n, param, static_cast<hipStream_t>(stream));
  return done();
}

int mvae_activation_backward(int32_t device, int32_t act, const float* y, const float* dy, float* dx, int64_t n, float param,
                             void* stream) {
  if (!y || !dy || !dx || n <= 0 || act < 0 || act > LAYER_ACT_ATTENUATE) return MVAE_E_INVALID;
  if (!use_device(device)) return MVAE_E_HIP;
  launch_act_bwd(act, y, dy, dx, n, param, static_cast<hipStream_t>(stream));
  return done();
}

int mvae_eltwise(int32_t device, int32_t op, const float* a, const float* b, float* out, int64_t n, void* stream) {
  if (!a || !b || !out || n <= 0 || op < 0 || op > 2) return MVAE_E_INVALID;
  if (!use_device(device)) return MVAE_E_HIP;
  launch_eltwise(op, a, b, out, n, static_cast<hipStream_t>(stream));
  return done();
}

int mvae_scale_channels_forward(int32_t device, const float* x, const float* m, float* y, int32_t B, int64_t HW, int32_t C,
                                void* stream) {
  if (!x || !m || !y || B <= 0 || HW <= 0 || C <= 0) return MVAE_E_INVALID;
  if (!use_device(device)) return MVAE_E_HIP;
  launch_scale_channels(x, m, y, B, HW, C, static_cast<hipStream_t>(stream));
  return done();
}

int mvae_scale_channels_backward(int32_t device, const float* x, const float* m, const float* dy, float* dx, float* dm, int32_t B,
                                 int64_t HW, int32_t C, void* stream) {
  if (!x || !m || !dy || B <= 0 || HW <= 0 || C <= 0 || (!dx && !dm)) return MVAE_E_INVALID;
  if (!use_device(device)) return MVAE_E_HIP;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dx) launch_scale_channels(dy, m, dx, B, HW, C, s);
  if (dm) launch_scale_channels_bwd_m(x, dy, dm, B, HW, C, s);
  return done();
}

int mvae_global_maxpool_forward(int32_t device, const float* x, float* y, int32_t* idx, int32_t B, int64_t HW, int32_t C, void* stream) {
  if (!x || !y || !idx || B <= 0 || HW <= 0 || HW >= (1LL << 31) || C <= 0) return MVAE_E_INVALID;
  if (!use_device(device)) return MVAE_E_HIP;
  launch_gmax_fwd(x, y, idx, B, HW, C, static_cast<hipStream_t>(stream));
  return done();
}

int mvae_global_maxpool_backward(int32_t device, const float* dy, const int32_t* idx, float* dx, int32_t B, int64_t HW, int32_t C,
                                 void* stream) {
  if (!dy || !idx || !dx || B <= 0 || HW <= 0 || C <= 0) return MVAE_E_INVALID;
  if (!use_device(device)) return MVAE_E_HIP;
  launch_gmax_bwd(dy, idx, dx, B, HW, C, static_cast<hipStream_t>(stream));
  return done();
}

int mvae_maxpool_same_forward(int32_t device, const float* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ph, int32_t pw,
                              int32_t sh, int32_t sw, float* y, int32_t* idx, void* stream) {
  if (!x || !y || !idx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ph <= 0 || pw <= 0 || sh <= 0 || sw <= 0) return MVAE_E_INVALID;
  if (!use_device(device)) return MVAE_E_HIP;
  int oh, ow, pt, pl;
  same_pad1(H, ph, sh, &oh, &pt);
  same_pad1(W, pw, sw, &ow, &pl);
  launch_maxpool_fwd(x, y, idx, B, H, W, C, oh, ow, ph, pw, sh, sw, pt, pl, static_cast<hipStream_t>(stream));
  return done();
}

int mvae_maxpool_same_backward(int32_t device, const float* dy, const int32_t* idx, int32_t B, int32_t H, int32_t W, int32_t C,
                               int32_t ph, int32_t pw, int32_t sh, int32_t sw, float* dx, void* stream) {
  if (!dy || !idx || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ph <= 0 || pw <= 0 || sh <= 0 || sw <= 0) return MVAE_E_INVALID;
  if (!use_device(device)) return MVAE_E_HIP;
  int oh, ow, pt, pl;
  same_pad1(H, ph, sh, &oh, &pt);
  same_pad1(W, pw, sw, &ow, &pl);
  launch_maxpool_bwd(dy, idx, dx, B, H, W, C, oh, ow, ph, pw, sh, sw, pt, pl, static_cast<hipStream_t>(stream));
  return done();
}

int mvae_batchnorm_forward(int32_t device, const float* x, int64_t M, int32_t C, const float* gamma, const float* beta, float eps,
                           int32_t training, const float* moving_mean, const float* moving_var, float* mean, float* invstd,
                           float* batch_var, float* y, void* stream) {
  if (!x || !gamma || !beta || !mean || !invstd || !y || M <= 0 || C <= 0) return MVAE_E_INVALID;
  if (!training && (!moving_mean || !moving_var)) return MVAE_E_INVALID;
  if (!use_device(device)) return MVAE_E_HIP;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (training) launch_bn_rows_stats(x, mean, invstd, batch_var, M, C, eps, s);
  else launch_bn_rows_from_moving(moving_mean, moving_var, mean, invstd, C, eps, s);
  launch_bn_rows_apply(x, mean, invstd, gamma, beta, y, M, C, s);
  return done();
}

int mvae_batchnorm_backward(int32_t device, const float* x, const float* dy, int64_t M, int32_t C, const float* gamma,
                            const float* mean, const float* invstd, int32_t training, float* dx, float* dgamma, float* dbeta,
                            float* work, void* stream) {
  if (!x || !dy || !gamma || !mean || !invstd || !dx || !dgamma || !dbeta || !work || M <= 0 || C <= 0) return MVAE_E_INVALID;
  if (!use_device(device)) return MVAE_E_HIP;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float *s1 = work, *s2 = work + C;                       // this call's column sums (dgamma / dbeta may hold earlier terms)
  launch_zero(work, 2 * (int64_t)C, s);
  launch_bn_rows_bwd_sums(x, dy, mean, invstd, s2, s1, M, C, s);
  launch_bn_rows_bwd_apply(x, dy, mean, invstd, gamma, s1, s2, dx, training ? 1 : 0, M, C, s);
  launch_eltwise(0, dgamma, s2, dgamma, C, s);
  launch_eltwise(0, dbeta, s1, dbeta, C, s);
  return done();
}

int mvae_attention_core_forward(int32_t device, const float* theta, const float* phi, const float* g, int32_t B, int64_t HW, int32_t F,
                                float* scores, float* out, void* stream) {
  if (!theta || !phi || !g || !scores || !out || B <= 0 || HW <= 0 || F <= 0 || F > 64) return MVAE_E_INVALID;
  if (!use_device(device)) return MVAE_E_HIP;
  if (!launch_attention_core_fwd(theta, phi, g, scores, out, B, HW, F, static_cast<hipStream_t>(stream))) return MVAE_E_INVALID;
  return done();
}

int mvae_attention_core_backward(int32_t device, const float* theta, const float* phi, const float* g, const float* scores,
                                 const float* dout, int32_t B, int64_t HW, int32_t F, float* dtheta, float* dphi, float* dg,
                                 float* work, void* stream) {
  if (!theta || !phi || !g || !scores || !dout || !dtheta || !dphi || !dg || !work || B <= 0 || HW <= 0 || F <= 0 || F > 64)
    return MVAE_E_INVALID;
  if (!use_device(device)) return MVAE_E_HIP;
  if (!launch_attention_core_bwd(theta, phi, g, scores, dout, dtheta, dphi, dg, work, B, HW, F, static_cast<hipStream_t>(stream)))
    return MVAE_E_INVALID;
  return done();
}

}  // extern "C"
