// kernels.h -- launcher prototypes of the gfx950 kernels behind libmvae_hip.so.
// Everything is float32, NHWC.  All launchers enqueue on `s` and never synchronise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvae {

// Geometry of a strided 'SAME' convolution in "F-form coordinates": the BIG side [B,IH,IW,CI] is the
// conv input, the SMALL side [B,OH,OW,CO] its output; weights are [KH,KW,CI,CO].  A Keras Conv2D uses
// it as is (kernel HWIO); a Keras Conv2DTranspose is the adjoint map small->big and its kernel
// (kh,kw,out,in) is the very same [KH,KW,CI(big),CO(small)] array (layer_blocks.py:946-951).
struct ConvGeom {
  int B, IH, IW, CI, OH, OW, CO, KH, KW, SH, SW, PT, PL;
};

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_ELU = 2, ACT_HSIG = 3, ACT_TANH = 4 };

struct PreOp {               // transform applied to the BIG-side operand when it is read
  const float* gate;         // [B,CI]  multiply (squeeze-excite gate), or null
  const float* scale;        // [CI]    per-channel affine (folded BatchNorm), or null
  const float* shift;        // [CI]
};

// ---- deterministic-reduction mode (MVAE_DETERMINISTIC=1, read by mvae_create) ----
// Float atomics make a sum depend on the order in which blocks retire.  In this mode no accumulator cell receives more than
// one atomic add: every kernel that reduces into gradient / statistic slots gets ONE SLOT PER BLOCK (kDetSlots copies, folded in
// fixed order by k_slot_sum / the finalize kernels), split-K kernels run unsplit, the squeeze-excite gate gradient and the
// global average pool are recomputed by single-pass kernels, and the 5x5 weight gradients take the slotted kernel.  Two runs
// of the same step are then bit-identical (tests/test_deterministic_gpu.py).  Float32 activations only.
constexpr int kDetSlots = 1024;
bool det_mode();
void set_det_mode(bool on);

// ---- RNG (Philox4x32-10) ----
// the seed is read from DEVICE memory so that a captured hipGraph stays valid from step to step
void launch_set_u64(uint64_t* dst, uint64_t v, hipStream_t s);
void launch_stamp(uint64_t* dst, hipStream_t s);
void launch_seed_next(uint64_t* p, hipStream_t s);
void launch_zero(float* p, int64_t n, hipStream_t s);
// Hyper-parameters live in a small DEVICE block (like the seed), written by a one-thread kernel ahead of the graph
// launch: one captured hipGraph then serves every learning rate / loss factor (a step-decay schedule used to
// instantiate a new graph per distinct lr).
enum { HP_RF_OVER_B = 0, HP_KF_OVER_B = 1, HP_LR = 2, HP_CLIP = 3, HP_GRAD_SCALE = 4, HP_COUNT = 8 };
void launch_set_f3(float* dst, float a, float b, float c, int n, hipStream_t s);
void launch_gather_rows(const float* src, const int64_t* idx, float* dst, int64_t n, int64_t row_elems, hipStream_t s);
void launch_rng_normal(float* out, int64_t n, float stddev, const uint64_t* seed, uint32_t stream_id, hipStream_t s);
void launch_rng_step(float* eps, int64_t n_eps, float std_eps, float* noise, int64_t n_noise, float* keep, int64_t n_keep,
                     float p_drop, float* zero_buf, int64_t n_zero, const uint64_t* seed, hipStream_t s);
void launch_rng_keepmask(float* out, int64_t n, float p_drop, const uint64_t* seed, uint32_t stream_id, hipStream_t s);

// ---- input transform (multiscale_vae.py:129-160, 292-315) ----
void launch_prep(const float* x, const float* noise, const float* keep, float* out, int B, int H, int W, int C,
                 float v0, float v1, float noise_std, float keep_scale, hipStream_t s);
void launch_blur_split(const float* in, float* band, float* down, int B, int H, int W, int C, hipStream_t s);
// stand-alone Laplacian pyramid level (layer_blocks.py:40-72): down = (G (*) in)[::2, ::2]; diff = in - up2(down)
// false = shape not covered (C > 8 or >= 2^31 pixels)
// normalise: `in` is the raw image in [v0, v1], normalised to [-1, 1] as it is read
bool launch_lap_level(const float* in, float* diff, float* down, int B, int H, int W, int C, const float* gauss9,
                      bool normalise, float v0, float v1, hipStream_t s);
void launch_denorm_clip(const float* in, float* out, int64_t n, float v0, float v1, hipStream_t s);
// cat[b,y,x,:] = [up2(coarse)[b,y,x,:C], fine[b,y,x,:C]]   (Concatenate of layer_blocks.py:148-150); false: C > 8
bool launch_lap_concat(const float* coarse, const float* fine, float* cat, int B, int H, int W, int C, hipStream_t s);
// shape-generic convolution with any activation (ACT_TANH included): small = act(conv(big) + bias) + residual
void launch_conv_f_any(const float* big, const float* w, const float* bias, const float* residual, float* small,
                       ConvGeom g, int act, hipStream_t s);
// out = up2(coarse) + fine; final_level: out = clip(denormalise(.)) instead (layer_blocks.py:123-131, 137-171)
bool launch_lap_merge(const float* coarse, const float* fine, float* out, int B, int H, int W, int C, bool final_level,
                      float v0, float v1, hipStream_t s);

// ---- convolutions ----
// small = act(conv(pre(big)) + bias) + residual
void launch_conv_f(const float* big, const float* w, const float* bias, const float* residual, float* small,
                   ConvGeom g, PreOp pre, int act, hipStream_t s);
// big = convT(small) + bias + residual      (conv backward-data / Conv2DTranspose forward)
void launch_conv_t(const float* small, const float* w, const float* bias, const float* residual, float* big,
                   ConvGeom g, hipStream_t s);
// dW[kh,kw,ci,co] += sum pre(big) * small ; db[co] += sum small (db nullable)   (both pre-zeroed by the caller)
// Gradient slots: float atomics into a footprint of a few KB from hundreds of blocks run far below the chip's atomic
// rate (MI355X_MICROARCH.md, Global float atomics: "contention"), so small weight gradients are accumulated into
// n copies of the gradient arena (block b -> copy b % n) which k_slot_sum folds into the arena at the end of backward.
struct GradSlots {
  float* base = nullptr;        // n copies, `stride` floats apart, each laid out like the gradient arena
  const float* gbase = nullptr; // the gradient arena itself
  int64_t stride = 0;
  int n = 0;                    // 0 = accumulate straight into the arena
  float* at(float* g) const { return (n && g) ? base + (g - gbase) : g; }
  int count() const { return n ? n : 1; }
};
void launch_conv_wgrad(const float* big, const float* small, float* dW, float* db, ConvGeom g, PreOp pre, GradSlots sl,
                       hipStream_t s);
// big = convT_1x1(small) and dot_out[b,c] = sum_hw big * dot_src, fused when the shape allows
void launch_conv_t_dot(const float* small, const float* w, float* big, const float* dot_src, float* dot_out,
                       ConvGeom g, hipStream_t s);
// MobileNetV3 backward pair of a 1x1 C->C conv in one pass (kernels_mfma.hip: k_gemm_dual); false = not covered
bool launch_gemm_dual_mfma(const float* X, const float* W, const float* aux, const float* gate, const float* residual,
                           float* Y, float* dW, float* db, float* dot_out, int64_t M, int64_t rows_per_image, int C,
                           GradSlots slots, int dslots, int64_t dstride, hipStream_t s);
// ELU backward in place: d *= (y > 0 ? 1 : y + 1)
void launch_elu_bwd(float* d, const float* y, int64_t n, hipStream_t s);

// depthwise 3x3 stride 1 SAME + bias + ReLU (layer_blocks.py:604-614)
void launch_dw_fwd(const float* in, const float* w, const float* b, float* out, int B, int H, int W, int C,
                   hipStream_t s);
// dx = dwT(dy) * (mask_src > 0)
void launch_dw_bwd_data(const float* dy, const float* w, const float* mask_src, float* dx, int B, int H, int W,
                        int C, hipStream_t s);
// dW[3,3,C] += sum in_shifted * dy ; db[C] += sum dy
void launch_dw_wgrad(const float* in, const float* dy, float* dW, float* db, int B, int H, int W, int C,
                     hipStream_t s);

// ---- reductions over pixels ----
// out[b,c] = scale * sum_hw x[b,hw,c]
void launch_spatial_sum(const float* x, float* out, int B, int64_t HW, int C, float scale, hipStream_t s);
// out[b,c] = sum_hw a*b
void launch_spatial_dot(const float* a, const float* b, float* out, int B, int64_t HW, int C, hipStream_t s);
// out[c] += sum_m x[m,c]
void launch_colsum(const float* x, float* out, int64_t M, int C, hipStream_t s);
// vectorised column statistics into slot copies (kernels_opt.hip); mode 0: sum, mode 1: squared deviations from
// mean = inv_m * (sum of the msl slot copies of msum).  false = shape not covered.
bool launch_colstat_opt(int mode, const float* x, const float* msum, int msl, float inv_m, float* out, int nslots,
                        int64_t slot_stride, int64_t M, int C, hipStream_t s, bool bf = false, float* out2 = nullptr);
constexpr int kStatSlots = 16;
// out[c] += sum_m (x[m,c]-mean[c])^2
void launch_colsqdev(const float* x, const float* mean, float* out, int64_t M, int C, hipStream_t s);
// out0[c] += sum_m d ; out1[c] += sum_m d * (x-mean)*invstd
void launch_bn_bwd_reduce(const float* d, const float* x, const float* mean, const float* invstd, float* sum_d,
                          float* sum_dx, int64_t M, int C, hipStream_t s);

// ---- MobileNetV3 / squeeze-excite pieces (layer_blocks.py:418-462, 556-648) ----
// d = (d * g[b,c] + dgap[b,c] * inv_hw) * (t1 > 0)       in place
void launch_mn_dt1pre(float* d, const float* t1, const float* g, const float* dgap, int B, int64_t HW, int C,
                      float inv_hw, hipStream_t s);
// out[b,n] = act(sum_k a[b,k] w[k,n] + bias[n])
void launch_gemm_nn(const float* a, const float* w, const float* bias, float* out, float* out_lin, int B, int K,
                    int N, int act, hipStream_t s);
// out[b,k] (+)= sum_n a'[b,n] w[k,n]; hs_lin != null: a' = a * hsig'(hs_lin[b,n])
void launch_gemm_nt(const float* a, const float* w, float* out, int B, int K, int N, const float* hs_lin,
                    int accumulate, hipStream_t s);
// dW[k,n] += sum_b a[b,k] g[b,n] (a optionally affine: a*a_scale[k]+a_shift[k]) ; db[n] += sum_b g[b,n]
// g optionally multiplied by hsig'(hs_lin)
void launch_gemm_tn(const float* a, const float* g, float* dW, float* db, int B, int K, int N,
                    const float* a_scale, const float* a_shift, const float* hs_lin, hipStream_t s);
// BatchNorm over the batch axis of [B,C]; training: batch statistics (written to stat_mean/var), else moving.
void launch_bn1d_fwd(const float* x, const float* gamma, const float* beta, const float* mov_mean,
                     const float* mov_var, float* xhat, float* invstd, float* y, float* stat_mean, float* stat_var,
                     int B, int C, float eps, int training, hipStream_t s);
// dv = relu'(x) * invstd * (dy*gamma - mean(dy*gamma) - xhat*mean(dy*gamma*xhat)); dgamma += ; dbeta +=
void launch_bn1d_bwd(const float* dy, const float* xhat, const float* invstd, const float* gamma,
                     const float* relu_src, float* dx, float* dgamma, float* dbeta, int B, int C, hipStream_t s);

// ---- decoder BatchNorm (multiscale_vae.py:420-421) helpers ----
// training: mean = sum/M, var=sqdev/M -> scale/shift, stats out; inference: from moving stats
void launch_bn2d_finalize(const float* sum, const float* sqdev, const float* gamma, const float* beta,
                          const float* mov_mean, const float* mov_var, float* mean, float* invstd, float* scale,
                          float* shift, float* stat_mean, float* stat_var, int64_t M, int C, float eps, int training, int nslots,
                          hipStream_t s, const float* pivot = nullptr);
void launch_scale_vec(float* v, float a, int n, hipStream_t s);
// dx = scale_c * (d - sum_d/M - xhat * sum_dx/M)   in place on d ; dgamma += sum_dx ; dbeta += sum_d
void launch_bn2d_bwd_apply(float* d, const float* x, const float* mean, const float* invstd, const float* gamma,
                           const float* sum_d, const float* sum_dx, float* dgamma, float* dbeta, int64_t M, int C,
                           hipStream_t s);

// ---- latent head (multiscale_vae.py:358-383, 485-488) ----
// z = mu + exp(lv) * eps ; losses[b, kl_col] = KL_s ; also strided copies into the concatenated outputs
void launch_sample_kl(const float* mu, const float* lv, const float* eps, int eps_stride, int eps_off, float* z,
                      float* kl_out, int kl_stride, int kl_col, int B, int Z, hipStream_t s);
// dmu = dz + kf/B * mu ; dlv = dz * eps * exp(lv) + kf/B * 0.5 * (exp(lv) - 1)
void launch_sample_kl_bwd(const float* dz, const float* mu, const float* lv, const float* eps, int eps_stride,
                          int eps_off, float* dmu, float* dlv, const float* hp, int B, int Z, hipStream_t s);
void launch_copy_cols(const float* src, int src_stride, int src_off, float* dst, int dst_stride, int dst_off, int B,
                      int n, hipStream_t s);

// ---- merge + loss (multiscale_vae.py:204-224, 86-94, 453-481) ----
// fine_out = up2(coarse) + fine_in ; if recon != null also recon = clip(denorm(fine_out))
void launch_upsample_add(const float* coarse, const float* fine_in, float* fine_out, float* recon, int B, int H,
                         int W, int C, float v0, float v1, hipStream_t s);
// coarse_grad = up2^T(fine_grad)
void launch_upsample_bwd(const float* fine_grad, float* coarse_grad, int B, int h, int w, int C, hipStream_t s);
// per image: r, r_exp -> losses[b,0..1]; signs of the channel-mean terms -> sgn[b, 2*C]
// also losses[b,2] = sum_s losses[b,3+s] (total KL)
void launch_loss_fwd(const float* y, const float* recon, float* losses, int loss_stride, int nscales, float* sgn,
                     int B, int H, int W, int C, hipStream_t s, float* scratch = nullptr, int64_t scratch_elems = 0);
// du = clipmask(m) * (v1-v0)/2 * rf/B * (-sign(y-recon)/N - 0.5*(sgn_ch/(C*HW) + incrop*sgn_cc/(C*ncrop)))
void launch_loss_bwd(const float* y, const float* recon, const float* merged, const float* sgn, float* du, int B,
                     int H, int W, int C, float v0, float v1, const float* hp, hipStream_t s);
// metrics[0] += B ; metrics[1+j] += sum_b losses[b,j]
void launch_metrics(const float* losses, int ncol, int B, float* metrics, hipStream_t s);

// ---- optimiser (multiscale_vae.py:497-499) ----
// first / count: the contiguous run of chunks that make up this chunk's tensor (for the deterministic norm)
struct ChunkDesc { int64_t offset; int32_t len; int32_t tensor; int32_t reg; int32_t pad; int32_t first; int32_t count; };
// zero / fold the gradient-slot copies of the listed (single-chunk) tensors
void launch_slot_zero(const ChunkDesc* chunks, int nchunks, float* slots, int64_t stride, int n, hipStream_t s);
void launch_slot_sum(const ChunkDesc* chunks, int nchunks, float* g, const float* slots, int64_t stride, int n,
                     hipStream_t s);
// g = g*grad_scale + reg'(w) ; norms[chunk] = sum over the chunk of g^2 (plain store: the per-tensor norm is then summed
// in a fixed order by the apply pass, so every data-parallel replica computes bit-identical clip factors)
void launch_opt_prepare(const float* w, float* g, const ChunkDesc* chunks, int nchunks, float* norms,
                        const float* hp, hipStream_t s);
// clip per tensor ; a += g^2 ; w -= lr * g / (sqrt(a) + 1e-7)
void launch_opt_apply(float* w, const float* g, float* a, const ChunkDesc* chunks, int nchunks, const float* norms,
                      const float* hp, bool clip, hipStream_t s);
// reg[0] += sum 0.01|w| or 0.01 w^2
void launch_reg_loss(const float* w, const ChunkDesc* chunks, int nchunks, float* out, hipStream_t s);
// moving = moving*mom + stat*stat_scale*(1-mom)*corr ; corr = n/(n-1), n = B*per_image when per_image > 0
struct StateDesc { int64_t offset; int32_t len; float momentum; float per_image; int32_t pad; };
void launch_state_update(float* state, const float* stats, const StateDesc* descs, int ndesc, const float* hp,
                         int B, hipStream_t s);
void set_gauss_constants(const float* g9);
void launch_bn2d_mean(const float* sum, const float* mov_mean, float* mean, int64_t M, int C, int training,
                      hipStream_t s);


// ---- edge layers (kernels_edge.hip); return false when the shape is not covered ----
bool launch_convbase_fwd(const float* in, const float* W, const float* bias, float* out, int B, int H, int Wd, int CI,
                         int CO, hipStream_t s, bool bf = false);
bool launch_convbase_wgrad(const float* in, const float* dy, const float* y, float* dW, float* db, int B, int H, int Wd,
                           int CI, int CO, GradSlots sl, hipStream_t s, bool bf = false);
bool launch_head_fwd(const float* x, const float* scale, const float* shift, const float* W, const float* bias,
                     float* y, int64_t M, int dc, int C, hipStream_t s, bool bf = false);
// squeeze-excite branch in two forward / two backward launches (kernels_se.hip); part: [se_max_blocks(B)][2][C] floats
int se_max_blocks(int max_batch);
bool launch_se_forward(const float* gap, const float* W0, const float* b0, const float* gamma, const float* beta,
                       const float* mov_mean, const float* mov_var, const float* W1, const float* b1, float* s0,
                       float* xhat, float* invstd, float* ulin, float* g, float* stat_mean, float* stat_var,
                       float* part, int B, int C, float eps, int training, hipStream_t s);
bool launch_se_backward(const float* dg, const float* ulin, const float* xhat, const float* invstd, const float* gamma,
                        const float* beta, const float* s0, const float* gap, const float* W1, const float* W0,
                        float* ds1, float* dgap, float* dW1, float* db1, float* dgamma, float* dbeta, float* dW0,
                        float* db0, float* part, int B, int C, GradSlots sl, int dslots, int64_t dstride, hipStream_t s);
// the MobileNetV3 block's 1x1 convs in the block-tiled form (kernels_mfma.hip: k_conv0_tile): conv0 Y = relu(X.W + b);
// conv2 (gate != null) Y = (X * gate[image]).W + b + residual.  false = shape not covered
bool launch_conv0_tile(const float* X, const float* W, const float* bias, const float* gate, const float* residual,
                       float* Y, int64_t M, int64_t rows_per_image, int C, hipStream_t s);
int head_slots();
// S: [head_slots()][2][dc] floats, zeroed by the caller; adds dW, db (gradient slots), dgamma, dbeta
bool launch_head_bwd(const float* x, const float* dy, const float* W, const float* gamma, const float* scale,
                     const float* shift, const float* mean, const float* invstd, float* S, float* dW, float* db,
                     float* dgamma, float* dbeta, float* dout, int64_t M, int dc, int C, GradSlots sl, hipStream_t s,
                     bool bf = false);
// sliding-row depthwise kernels (kernels_dw.hip); false = shape not covered
// true where the whole-image kernels (k_dw_fwd_img / k_dw_bwd_img, feature maps <= 16 wide) take the launch
bool dw_uses_img(bool backward, bool mask_in_lsb, int B, int H, int W, int C);
// bf (here and below): the activation tensors are bfloat16 (act16.h); the pointers then address bf16 elements
bool launch_dw_fwd_gap(const float* in, const float* w, const float* b, float* out, float* gap, int B, int H, int W,
                       int C, hipStream_t s, bool bf = false);
constexpr int kDwMaxBlocks = 1024;     // partial-sum scratch: kDwMaxBlocks * 10 * C floats
// mask_in_lsb: dt2 carries the ReLU mask (t1 > 0) in its mantissa LSB (k_gemm_dual conv2 pair) and t1 is not read
bool launch_dw_bwd_fused(const float* dt2, const float* t1, const float* t0, const float* w, const float* gate,
                         const float* dgap, float* dt0, float* dW, float* db, GradSlots sl, bool mask_in_lsb, int B, int H,
                         int W, int C, hipStream_t s, bool bf = false);
// float32: conv2 of one MobileNetV3 block chained with conv0 of the next (kernels_mfma.hip: k_conv2_chain); false = not covered
bool launch_conv2_chain(const float* X, const float* W, const float* bias, const float* gate, const float* residual,
                        float* Y, const float* wm, const float* biasm, float* Ym, bool mid_transposed, const float* W2,
                        const float* bias2, float* Y2, int64_t M, int64_t rows_per_image, int C, hipStream_t s);
// bf16: conv2 of one MobileNetV3 block chained with conv0 of the next (k16_pw_chain), optionally through the 1x1
// convolution 64 -> 32 between them (wm / biasm / outm; mid_transposed = Conv2DTranspose form); false = shape not covered
bool launch16_pw_chain(const void* in, const float* w, const float* bias, const float* gate, const void* residual, void* out,
                       const float* wm, const float* biasm, void* outm, bool mid_transposed,
                       const float* w2, const float* bias2, void* out2, int64_t M, int64_t rows_per_image, int C,
                       hipStream_t s);
// bf16: that depthwise backward fused with conv0's backward pair (kernels_bf16.hip: k16_dw_bwd_conv0) -- da = dt0 . W0^T +
// dout, dW0 += a^T dt0, db0; dt0 never stored.  C = 64, W % 32 == 0, mask in the LSB of dt2.  false = shape not covered.
bool launch16_dw_bwd_conv0(const void* dt2, const void* t0, const float* w, const float* gate, const float* dgap,
                           const float* W0, const void* a_in, const void* dout, void* da, float* dW, float* db, float* dW0,
                           float* db0, GradSlots sl, int B, int H, int W, int C, hipStream_t s);
// Dense layers around the latent (kernels_dense.hip); false = shape not covered
bool launch_dense_mu_lv(const float* x, const float* Wmu, const float* bmu, const float* Wlv, const float* blv,
                        float* mu, float* lv, int B, int K, int Z, hipStream_t s, bool bf = false);
bool launch_dense_dz(const float* dy, const float* W, float* dz, int B, int Z, int N, hipStream_t s, bool bf = false);
bool launch_dense_dflat(const float* dmu, const float* dlv, const float* Wmu, const float* Wlv, float* out, int B, int K,
                        int Z, hipStream_t s, bool bf = false);
bool launch_dense_expand(const float* z, const float* W, const float* bias, float* out, int B, int Z, int N,
                         hipStream_t s, bool bf = false);
bool launch_dense_wgrad_mu_lv(const float* flat, const float* dmu, const float* dlv, float* dWmu, float* dWlv,
                              float* dbmu, float* dblv, int B, int K, int Z, hipStream_t s, bool bf = false);
// squeeze-excite backward pair in one launch: dW += a'^T g' (+ db) and dx = g' W^T   (kernels_opt.hip)
void launch_se_pair(const float* a, const float* g, const float* W, float* dW, float* db, float* dx, int B, int K, int N,
                    const float* a_scale, const float* a_shift, const float* hs_lin, hipStream_t s);
bool launch_dense_wgrad_dec(const float* z, const float* dy, float* dW, float* db, int B, int Z, int N, hipStream_t s,
                            bool bf = false);
void launch_add_vec2(float* o0, const float* a0, float* o1, const float* a1, int n, hipStream_t s);

// ---- operators of the remaining block library (kernels_layers.hip; include/mvae_hip.h: MVAE_LAYER_ACT_*) ----
enum { LAYER_ACT_LINEAR = 0, LAYER_ACT_RELU = 1, LAYER_ACT_SIGMOID = 2, LAYER_ACT_TANH = 3, LAYER_ACT_ATTENUATE = 4 };
void launch_act_fwd(int act, const float* x, float* y, int64_t n, float m, hipStream_t s);
void launch_act_bwd(int act, const float* y, const float* dy, float* dx, int64_t n, float m, hipStream_t s);   // dx = dy * act'(y)
void launch_eltwise(int op, const float* a, const float* b, float* out, int64_t n, hipStream_t s);             // 0 +, 1 -, 2 *
void launch_scale_channels(const float* x, const float* m, float* y, int B, int64_t HW, int C, hipStream_t s);  // y = x * m[b, c]
void launch_scale_channels_bwd_m(const float* x, const float* dy, float* dm, int B, int64_t HW, int C, hipStream_t s);
void launch_gmax_fwd(const float* x, float* y, int* idx, int B, int64_t HW, int C, hipStream_t s);
void launch_gmax_bwd(const float* dy, const int* idx, float* dx, int B, int64_t HW, int C, hipStream_t s);
void launch_maxpool_fwd(const float* x, float* y, int* idx, int B, int H, int W, int C, int OH, int OW, int ph, int pw, int sh,
                        int sw, int pt, int pl, hipStream_t s);
void launch_maxpool_bwd(const float* dy, const int* idx, float* dx, int B, int H, int W, int C, int OH, int OW, int ph, int pw,
                        int sh, int sw, int pt, int pl, hipStream_t s);
void launch_bn_rows_stats(const float* x, float* mean, float* invstd, float* var_out, int64_t M, int C, float eps, hipStream_t s);
void launch_bn_rows_from_moving(const float* mm, const float* mv, float* mean, float* invstd, int C, float eps, hipStream_t s);
void launch_bn_rows_apply(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, float* y,
                          int64_t M, int C, hipStream_t s);
void launch_bn_rows_bwd_sums(const float* x, const float* dy, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                             int64_t M, int C, hipStream_t s);                       // dgamma / dbeta += this launch's column sums
void launch_bn_rows_bwd_apply(const float* x, const float* dy, const float* mean, const float* invstd, const float* gamma,
                              const float* sum_dy, const float* sum_dyx, float* dx, int training, int64_t M, int C, hipStream_t s);
bool launch_attention_core_fwd(const float* theta, const float* phi, const float* g, float* scores, float* out, int B, int64_t HW,
                               int F, hipStream_t s);
bool launch_attention_core_bwd(const float* theta, const float* phi, const float* g, const float* scores, const float* dout,
                               float* dtheta, float* dphi, float* dg, float* work, int B, int64_t HW, int F, hipStream_t s);

// ---- glue of the stand-alone block library (kernels_blocks.hip) ----
void launch_relu_bwd(const float* dy, const float* y, float* out, int64_t n, hipStream_t s);       // out = dy * (y > 0)
void launch_relu_add(float* y, const float* r, bool relu, int64_t n, hipStream_t s);             // y = [relu](y) (+ r)
void launch_add2(const float* a, const float* b, float* out, int64_t n, hipStream_t s);
void launch_dw_bwd_plain(const float* dy, const float* w, float* dx, int B, int H, int W, int C, hipStream_t s);

// ---- float32 k x k convolutions as split-bf16 products on the bf16 matrix cores (kernels_split.hip) ----
// x = x1 + x2 + x3 (three exact bf16 pieces), six bf16 MFMAs per product: float32 accuracy at 2.7x the f32-MFMA rate.
bool split_conv_covers(const ConvGeom& g);                     // 5x5-like layers between 32 and 64 channels
int64_t split_planes_bytes(const ConvGeom& g);                 // workspace for one layer's weight planes (both forms)
void launch_split_weights(const float* W, void* planes, const ConvGeom& g, hipStream_t s);   // once per layer and step
bool split_selftest();                 // first call: run the kernels next to a self-checking VALU kernel (see kernels_split.hip)
int k16_erratum_count();               // the same count with the bfloat16-storage 5x5 kernels as the neighbour (-1: not run)
int split_conv_erratum_count();        // wrong values the v_pk_fma_f32 form of the self-test's check kernel returned (-1: not run)
// backward pair of a 64 -> 64 1x1 convolution with split products (kernels_split.hip); false = not covered / switched off
bool launch_gemm_dual_split(const float* X, const float* W, const float* aux, const float* gate, const float* residual,
                            float* Y, float* dW, float* db, float* dot_out, int64_t M, int64_t rows_per_image, int C,
                            GradSlots sl, int dslots, int64_t dstride, int cap, hipStream_t s);
const char* gemm_dual_split_kernel(bool gated, int64_t M, int64_t rows_per_image, int C);   // its kernel name, or nullptr
bool launch_gemm_dual_stats(const float* X, const float* W, const float* aux, const float* gate, unsigned* mask, float* dW,
                            float* db, float* dot_out, int64_t M, int64_t rows_per_image, int C, GradSlots sl, int dslots,
                            int64_t dstride, int cap, hipStream_t s);                      // conv2 pair without dt2 (+ mask words)
// whole MobileNetV3 backward behind the squeeze-excite step in one pass (kernels_fused.hip): dt2 recomputed from dout
const char* mn_bwd_split_kernel(int B, int H, int W, int C);
bool launch_mn_bwd_split(const float* dout, const unsigned* mask, const float* t0, const float* w, const float* gate,
                         const float* dgap, const float* W2, const float* W0, const float* a_in, float* da, float* dW, float* db,
                         float* dW0, float* db0, GradSlots sl, int B, int H, int W, int C, hipStream_t s);
// depthwise backward + conv0 backward pair in one pass (kernels_fused.hip): C = 64, W = 32, H even, mask in the LSB of dt2
const char* dw_bwd_conv0_split_kernel(int B, int H, int W, int C);                         // kernel name, or nullptr
bool launch_dw_bwd_conv0_split(const float* dt2, const float* t0, const float* w, const float* gate, const float* dgap,
                               const float* W0, const float* a_in, const float* dout, float* da, float* dW, float* db,
                               float* dW0, float* db0, GradSlots sl, int B, int H, int W, int C, hipStream_t s);
// conv2 of a block + conv0 and depthwise stage of the next one in one pass (kernels_fused_fwd.hip)
const char* mn_fwd_chain_split_kernel(int B, int H, int W, int C);
bool launch_mn_fwd_chain_split(const float* t1, const float* gate, const float* x, const float* W2, const float* b2,
                               const float* W0n, const float* b0n, const float* wdn, const float* bdn, float* y, float* t0n,
                               float* t1n, float* gapn, int B, int H, int W, int C, hipStream_t s);
bool launch_mn_fwd_first_split(const float* x, const float* W0, const float* b0, const float* wd, const float* bd, float* t0,
                               float* t1, float* gap, int B, int H, int W, int C, hipStream_t s);   // conv0 + depthwise stage
bool mn_fwd_first_split_on();
// process-wide launch geometry of the image-resident fused kernels (diagnostic: tests assert that a case really put
// several images on a block): launches of the forward / backward kernels, largest ceil(B / grid) seen
void fused_launch_note(bool fwd, int B, int grid);
void fused_launch_stats(int out[3]);
int split_conv_status();               // 0 switched off (MVAE_SPLIT_CONV=0), 1 in use, 2 disabled by the self-test on this board
// 5 x 5 weight gradient (32 <-> 64 channels) with split-bf16 products (kernels_split_wgrad.hip); false = not covered / off
bool launch_conv_wgrad_split(const float* big, const float* small, float* dW, float* db, const ConvGeom& g, hipStream_t s);
bool launch_conv_taps_split(bool transposed, const float* in, const void* planes, const float* bias, float* out,
                            const ConvGeom& g, hipStream_t s);

}  // namespace mvae
