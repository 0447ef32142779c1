/*
 * mvae_hip.h -- C ABI of libmvae_hip.so, the MI355X (gfx950) implementation of the multiscale-VAE
 * train-step hot path.
 *
 * The reference (NikolasMarkou/multiscale_variational_autoencoder) has NO native / FFI boundary: its
 * arithmetic is dispatched by Keras 2.4.3 into TensorFlow 2.3.1 when
 *     mvae/multiscale_vae.py:550-557   self._model_trainable.fit(x, x, ...)            (train step)
 *     mvae/multiscale_vae.py:238-257   encoder / decoder keras.Model.predict           (inference)
 * run the static graph that mvae/multiscale_vae.py:73-288 and mvae/layer_blocks.py:418-462,556-648,
 * 893-1050 describe.  This header is therefore the boundary the build defines for that one path
 * (SURVEY.md section 8(b)); each entry point names the reference lines whose work it replaces.  The
 * Python facade `multiscale_variational_autoencoder_amd.MultiscaleVAE` (same constructor / compile /
 * train surface as mvae.MultiscaleVAE) binds exactly these symbols through ctypes; INTEGRATION.md shows
 * the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++ / torch / HIP types in any signature (streams are `void*`
 *     holding a hipStream_t; NULL = the default stream).
 *   - every function returns 0 on success and a negative MVAE_E_* code on failure; nothing throws or
 *     aborts across the ABI; mvae_last_error() gives the text.
 *   - all device buffers are CALLER-owned (the Python host allocates them with torch so the flat
 *     gradient arena can be handed to torch.distributed/RCCL as one tensor); the handle owns only host
 *     metadata.  All tensors are float32, images NHWC (reference: Keras channels_last).
 *   - all work is enqueued asynchronously on the given stream; no hidden synchronisation.
 *   - a handle is bound to one device and is not thread-safe (data parallel = one process per GPU).
 */
#ifndef MVAE_HIP_H
#define MVAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVAE_ABI_VERSION 4
#define MVAE_MAX_LEVELS 16
#define MVAE_MAX_BLOCKS 16
#define MVAE_NAME_CAP 96

#define MVAE_OK 0
#define MVAE_E_INVALID (-1)   /* bad argument / configuration (Python facade raises ValueError) */
#define MVAE_E_STATE (-2)     /* call order: not bound, backward without training forward, ...  */
#define MVAE_E_HIP (-3)       /* a HIP runtime call or kernel launch failed                      */
#define MVAE_E_NOMEM (-4)     /* caller-provided workspace too small                             */

#define MVAE_ACT_F32 0         /* activations, saved tensors and activation gradients in float32           */
#define MVAE_ACT_BF16 1        /* ... stored as bfloat16 (wide [M,c] tensors only); parameters, gradients,  */
                               /* optimiser state, BatchNorm statistics, latents and losses stay float32     */

#define MVAE_REG_NONE 0
#define MVAE_REG_L1 1         /* keras "l1": 0.01 * sum |w|   (layer_blocks.py:14)              */
#define MVAE_REG_L2 2         /* keras "l2": 0.01 * sum w^2   (multiscale_vae.py:63-64)         */

/* Constructor arguments of MultiscaleVAE (multiscale_vae.py:12-26), flattened. */
typedef struct mvae_config {
  int32_t abi_version;                       /* MVAE_ABI_VERSION */
  int32_t input_h, input_w, input_c;         /* input_dims (HxWxC, channels_index = 2) */
  int32_t levels;                            /* len(z_dims), >= 2 */
  int32_t z_dims[MVAE_MAX_LEVELS];
  int32_t enc_n;                             /* len(encoder["filters"]) */
  int32_t enc_filters[MVAE_MAX_BLOCKS];
  int32_t enc_kh[MVAE_MAX_BLOCKS], enc_kw[MVAE_MAX_BLOCKS];
  int32_t enc_sh[MVAE_MAX_BLOCKS], enc_sw[MVAE_MAX_BLOCKS];
  int32_t dec_n;                             /* decoder dict (already reversed by the caller when None) */
  int32_t dec_filters[MVAE_MAX_BLOCKS];
  int32_t dec_kh[MVAE_MAX_BLOCKS], dec_kw[MVAE_MAX_BLOCKS];
  int32_t dec_sh[MVAE_MAX_BLOCKS], dec_sw[MVAE_MAX_BLOCKS];
  float min_value, max_value;                /* value range, multiscale_vae.py:65-66 */
  float sample_std;                          /* stddev of the sampling epsilon, :67 */
  int32_t max_batch;                         /* largest batch any later call will pass */
  int32_t act_dtype;                         /* MVAE_ACT_F32 / MVAE_ACT_BF16 (BASELINE configs 4-5: bf16) */
} mvae_config;

typedef struct mvae_handle mvae_handle;

/* One batch through the graph.  Device pointers; nullable members are optional. */
typedef struct mvae_step_io {
  const float* x;          /* [B,H,W,C] in [min_value,max_value]                                   */
  int32_t batch;           /* B <= max_batch                                                       */
  int32_t training;        /* 1: noise + dropout + batch-statistics BN (model_trainable.fit);      */
                           /* 0: inference (model.predict): no noise/dropout, moving statistics    */
  const float* eps;        /* [B,sum z] epsilon ALREADY scaled by sample_std, or NULL -> Philox    */
  const float* noise;      /* [B,H,W,C] standard normal for GaussianNoise, or NULL -> Philox       */
  const float* keep_mask;  /* [B,C] 1/0 keep mask of SpatialDropout2D(0.1), or NULL -> Philox      */
  uint64_t seed;           /* Philox seed for whatever is not injected                             */
  float* recon;            /* out [B,H,W,C] reconstruction in [min,max] (nullable)                 */
  float* mu;               /* out [B,sum z] (nullable)                                             */
  float* log_var;          /* out [B,sum z] (nullable)                                             */
  float* z;                /* out [B,sum z] sampled latents (nullable)                             */
  float* losses;           /* out [B, 3+levels]: vae_r_loss, vae_r_experimental_loss, kl,          */
                           /*     kl_scale_0.. (multiscale_vae.py:453-488) (nullable)              */
} mvae_step_io;

/* ---- life cycle: replaces MultiscaleVAE.__init__/_build (multiscale_vae.py:12-288); host only ---- */
int mvae_create(const mvae_config* cfg, mvae_handle** out);
void mvae_destroy(mvae_handle* h);
const char* mvae_last_error(const mvae_handle* h);      /* h may be NULL: error of the last failed create */
int mvae_abi_version(void);
/* 1 when the library was built with -DMVAE_DEBUG_BUILD (timing-diagnostic switches such as MVAE_DEBUG_ONLY_SCALE are
 * compiled in: results may then be garbage on request); 0 for the release build, which contains none of them. */
int mvae_debug_build(void);
/* Deterministic-reduction mode: MVAE_DETERMINISTIC=1 in the environment of mvae_create.  Every float reduction of the step
 * then has a fixed order (one gradient / statistic slot per block folded in order, no split-K, single-pass gate gradients):
 * repeated runs are bit-identical.  Float32 activations; needs 1024 gradient-arena copies of workspace.  1 = this handle
 * runs in that mode. */
int mvae_deterministic(const mvae_handle* h);
/* The float32 5x5 stride-2 convolutions run as split-bf16 products on the bf16 matrix cores (csrc/kernels_split.hip:
 * float32 accuracy, 2.7x the float32-MFMA rate).  0 = switched off (MVAE_SPLIT_CONV=0), 1 = in use, 2 = disabled for this
 * process by the hardware self-test the first mvae_bind runs (a board on which a kernel running beside them returned
 * wrong values keeps the float32-MFMA kernels). */
int mvae_split_conv_status(void);
/* What the self-test measured about the hardware erratum the build works around (no packed-float32 instructions in any
 * kernel): the number of wrong values its check kernel returned when written with v_pk_fma_f32; -1 = self-test not run. */
int mvae_split_conv_erratum(void);
/* Both measurements of that self-test: wrong values of the packed-float32 check kernel beside the split-bf16 float32
 * kernels and beside the bfloat16-storage kernels (k16_taps).  A non-zero count means: on this board, code of ANY library
 * that contains v_pk_*_f32 instructions must not run concurrently with that kernel family -- RCCL's float32 reduce kernels
 * do contain them (profiles/round4_rccl_packed_f32_scan.json), which is why the two-phase gradient exchange that overlaps
 * the all-reduce with the backward pass (MVAE_DP_OVERLAP) is refused by the Python engine while a count is non-zero. */
int mvae_packed_f32_hazard(int32_t* beside_split, int32_t* beside_bf16);
/* Launch geometry of the image-resident fused MobileNetV3 kernels (k_mn_fwd_chain_s / k_dw_bwd_conv0_s: one block per CU
 * walks whole images) since the process started: launches of the forward / backward kernel and the largest number of
 * images one block walked.  Diagnostic: the parity tests assert that their multi-image cases really exercised that path. */
/* MVAE_STAMPS=1 in the environment of mvae_create: the step's graphs carry one-thread kernels that store the 100 MHz device
 * clock at the forks, joins and chain ends (ids in csrc/runtime.cpp, stamp()).  Copies the first n (<= 128) stamps to the
 * host after a device synchronise; MVAE_E_STATE when the handle was created without the switch.  Diagnostic. */
int mvae_stamps(const mvae_handle* h, uint64_t* out, int32_t n);
int mvae_fused_launch_stats(int32_t* fwd, int32_t* bwd, int32_t* max_images_per_block);

/* ---- tables: the layer/variable inventory Keras builds ---- */
int64_t mvae_param_count(const mvae_handle* h);         /* number of trainable tensors            */
int64_t mvae_param_elems(const mvae_handle* h);         /* floats in the parameter arena (padded) */
int mvae_param_info(const mvae_handle* h, int64_t i, char* name, int32_t name_cap,
                    int64_t shape[4], int32_t* ndim, int64_t* offset, int32_t* reg);
int64_t mvae_state_count(const mvae_handle* h);         /* BatchNorm moving mean / variance tensors */
int64_t mvae_state_elems(const mvae_handle* h);
int mvae_state_info(const mvae_handle* h, int64_t i, char* name, int32_t name_cap,
                    int64_t* elems, int64_t* offset);
int64_t mvae_latent_dim(const mvae_handle* h);          /* sum(z_dims) */
int64_t mvae_reduce_elems(const mvae_handle* h);        /* floats in the reduce arena:                     */
                                                        /* [grads P | BN batch statistics S | metrics]      */
int64_t mvae_metrics_offset(const mvae_handle* h);      /* offset of the metrics block in the reduce arena:  */
                                                        /* [count, sum r, sum r_exp, sum kl, sum kl_s..]     */
int64_t mvae_workspace_bytes(const mvae_handle* h);     /* activations + scratch for max_batch               */

/* ---- binding caller-owned device memory ---- */
int mvae_bind(mvae_handle* h, int32_t device, float* params, float* reduce_arena, float* accum,
              float* state, void* workspace, int64_t workspace_bytes);

/* ---- the hot path ---- */
/* forward: input_transform + encoders + sampling + decoders + merge + losses
 *          (multiscale_vae.py:129-160, 292-315, 319-433, 204-224, 453-488). */
int mvae_forward(mvae_handle* h, const mvae_step_io* io, void* stream);
/* backward of mean_B(r_factor * r_exp + kl_factor * kl) w.r.t. every trainable tensor; fills the
 * gradient part of the reduce arena (regularisers are added by mvae_apply_adagrad).  What
 * keras.Model.fit derives by autodiff from compile()'s vae_loss (multiscale_vae.py:491-504). */
int mvae_backward(mvae_handle* h, float r_factor, float kl_factor, void* stream);
/* The same pass in two halves, for data-parallel callers that overlap the gradient exchange with the backward pass:
 * phase 0 = loss, decoder halves and the Dense-layer gradients -- after it the leading mvae_reduce_split() floats of the
 * reduce arena (the Dense weights: 95 % of the gradient bytes of the 256x256 configuration) are final and may be
 * all-reduced while phase 1 (encoder halves, conv_base, gradient-slot fold) runs; then the rest of the arena. */
int mvae_backward_phase(mvae_handle* h, int32_t phase, float r_factor, float kl_factor, void* stream);
int64_t mvae_reduce_split(const mvae_handle* h);

/* hipGraph bookkeeping of the replayable calls: how many launch sequences are captured (one per call kind and batch), and
 * how many calls wanted a graph but ran eagerly because the stream could not be captured (the legacy default stream).
 * A production caller expects eager_fallbacks == 0. */
int mvae_graph_stats(const mvae_handle* h, int32_t* captured, int32_t* eager_fallbacks);
/* g = grad_scale * g + d(reg)/dw ; per-variable clipnorm ; Adagrad (a0 = 0.1 set by the host) ;
 * BN moving statistics update from the (reduced) batch statistics (multiscale_vae.py:497-499). */
int mvae_apply_adagrad(mvae_handle* h, float lr, float clip_norm, float grad_scale, void* stream);
/* forward + backward + apply on one device (keras train_on_batch). */
int mvae_train_step(mvae_handle* h, const mvae_step_io* io, float r_factor, float kl_factor, float lr,
                    float clip_norm, void* stream);
/* ---- data-parallel exchange: the batch shards over one process per GPU, every rank holds a parameter
 *      replica, and ONE all-reduce(sum, float32) of the reduce arena [gradients | BatchNorm batch statistics | metrics] per
 *      step is the only exchange.  The reference has no counterpart (a single-device keras fit, multiscale_vae.py:550-557);
 *      these entry points bind RCCL (librccl.so, looked up at run time: a copy already loaded in the process is reused)
 *      so that a C caller needs nothing else.  The Python facade keeps calling torch.distributed.all_reduce on the same
 *      arena by default (same RCCL underneath).
 *      unique_id: rank 0 creates the 128-byte ncclUniqueId and hands it to the other ranks out of band (file, socket, MPI).
 *      comm_init: collective over all ranks; the handle must be bound (the communicator lives on its device).
 *      allreduce: in-place sum over reduce-arena floats [offset, offset + count) on `stream`; count < 0 = to the end.
 *      train_step_dp = forward + backward + allreduce(whole arena) + apply_adagrad(grad_scale = 1 / nranks). ---- */
#define MVAE_COMM_ID_BYTES 128
int mvae_comm_unique_id(char id[MVAE_COMM_ID_BYTES]);
int mvae_comm_init(mvae_handle* h, const char id[MVAE_COMM_ID_BYTES], int32_t rank, int32_t nranks);
int mvae_comm_destroy(mvae_handle* h);
int mvae_comm_size(const mvae_handle* h);                /* ranks of the handle's communicator, 0 = none */
int mvae_allreduce(mvae_handle* h, int64_t offset, int64_t count, void* stream);
int mvae_train_step_dp(mvae_handle* h, const mvae_step_io* io, float r_factor, float kl_factor, float lr,
                       float clip_norm, void* stream);
/* sum of the Keras regularisation losses (added to the reported `loss`) -> 1 float on the device. */
int mvae_reg_loss(mvae_handle* h, float* out_dev, void* stream);
/* decoder model (multiscale_vae.py:247-257): z [B,sum z] -> recon [B,H,W,C], inference mode. */
int mvae_decode(mvae_handle* h, const float* z, int32_t batch, float* recon, void* stream);

/* ---- input pipeline of train() (multiscale_vae.py:550-557: fit(x, x, batch_size, shuffle=True)): gather one batch
 *      from an HBM-resident dataset.  dst[i, :] = src[idx[i], :]; src [N,row_elems], idx [n] int64 (device),
 *      dst [n,row_elems]; stateless. ---- */
int mvae_gather_rows(int32_t device, const float* src, const int64_t* idx, int64_t n, int64_t row_elems, float* dst,
                     void* stream);

/* ---- stand-alone Laplacian pyramid.  Replaces the Keras models built by
 *      mvae/layer_blocks.py:23-99 (laplacian_transform_split) and :107-185 (laplacian_transform_merge, trainable=False),
 *      whose behaviour the reference pins in tests/test_layer_blocks.py:118-190.  Stateless; all pointers are device
 *      memory except `gauss9` (host: the 3x3 kernel of layer_blocks.gaussian_kernel, row-major) and the pointer arrays.
 *      split:  x [B,H,W,C] in [min,max] -> out[i] [B,H/2^i,W/2^i,C]: level i < levels-1 holds n_i - up2(n_{i+1}), the
 *              last level holds n_{levels-1}; n_0 = normalised x, n_{i+1} = (G (*) n_i)[::2, ::2].
 *              work: >= B*H*W*C*4/3 floats.       H, W must be multiples of 2^(levels-1).
 *      merge:  in[i] as produced by split -> out [B,H,W,C] = clip(denormalise(sum)), work: >= 2*B*H*W*C floats. ---- */
int mvae_laplacian_split(int32_t device, const float* x, int32_t batch, int32_t H, int32_t W, int32_t C, int32_t levels,
                         float min_value, float max_value, const float* gauss9, float* const* out, float* work,
                         void* stream);
int mvae_laplacian_merge(int32_t device, const float* const* in, int32_t batch, int32_t H, int32_t W, int32_t C,
                         int32_t levels, float min_value, float max_value, float* out, float* work, void* stream);

/* laplacian_transform_merge(trainable=True), forward (layer_blocks.py:137-171): per level i < levels-1, coarse to fine,
 *      x = Concatenate([UpSampling2D(2, bilinear)(out), in[i]]) -> Conv2D(filters, 3x3, SAME, relu, bias) ->
 *      Conv2D(C, 1x1, tanh, no bias);  out = x + in[i];  result = clip(denormalise(out)).
 *      w3[i] [3,3,2C,filters], b3[i] [filters], w1[i] [1,1,filters,C]: device pointers for levels 0 .. levels-2 (host
 *      arrays of pointers).  work: >= B*H*W*(2C + filters + 2C) floats.  The weights are the caller's: the reference
 *      only ever builds this model, it has no training loop for it (and neither does this library). ---- */
int mvae_laplacian_merge_mix(int32_t device, const float* const* in, int32_t batch, int32_t H, int32_t W, int32_t C,
                             int32_t levels, int32_t filters, const float* const* w3, const float* const* b3,
                             const float* const* w1, float min_value, float max_value, float* out, float* work,
                             void* stream);

/* ---- stand-alone blocks of the reference's block library: forward and backward, stateless; all
 *      tensors NHWC float32 device memory, weights in the Keras layouts, gradients are ADDED to dw / db (zero them first).
 *      mobilenetV2_block (layer_blocks.py:468-550; use_batchnorm=False, dropout 0 -- the reference's defaults):
 *        t0 = x.W0 + b0 [B,H,W,F];  t1 = relu(dw3x3(t0, Wd) + bd);  u = relu(t1.W2 + b2) [B,H,W,C];  y = u + x
 *        W0 [1,1,C,F], Wd [3,3,F,1], W2 [1,1,F,C].  backward work: >= B*H*W*(C + 2F) floats.
 *      resnet_block (layer_blocks.py:789-887; strides (1,1), use_batchnorm=False, dropout 0; relu != 0: activation "relu",
 *      else "linear"):  x0 = act(conv(x, W0) + b0);  y = act(conv(x0, W1) + b1 + skip),  skip = x if C == F else x.Ws + bs
 *        W0 [kh,kw,C,F], W1 [kh,kw,F,F], Ws [1,1,C,F] (NULL when C == F; `skip` buffer [B,H,W,F] likewise).
 *        backward work: >= B*H*W*(2F + C) floats. ---- */
int mvae_mnv2_forward(int32_t device, const float* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t F, const float* w0,
                      const float* b0, const float* wd, const float* bd, const float* w2, const float* b2, float* t0,
                      float* t1, float* u, float* y, void* stream);
int mvae_mnv2_backward(int32_t device, const float* x, const float* t0, const float* t1, const float* u, const float* dy,
                       int32_t B, int32_t H, int32_t W, int32_t C, int32_t F, const float* w0, const float* wd,
                       const float* w2, float* dx, float* dw0, float* db0, float* dwd, float* dbd, float* dw2, float* db2,
                       float* work, void* stream);
int mvae_resnet_forward(int32_t device, const float* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t F, int32_t kh,
                        int32_t kw, int32_t relu, const float* w0, const float* b0, const float* w1, const float* b1,
                        const float* ws, const float* bs, float* x0, float* skip, float* y, void* stream);
int mvae_resnet_backward(int32_t device, const float* x, const float* x0, const float* y, const float* dy, int32_t B, int32_t H,
                         int32_t W, int32_t C, int32_t F, int32_t kh, int32_t kw, int32_t relu, const float* w0,
                         const float* w1, const float* ws, float* dx, float* dw0, float* db0, float* dw1, float* db1,
                         float* dws, float* dbs, float* work, void* stream);

/* ---- layer operators (the rest of the block library): the device operators from which the Python
 *      facade assembles attention_block / self_attention_block (layer_blocks.py:654-783), attenuate_activation and the
 *      excite / inhibit masks and block (:191-412), resnet_block with strides (:847-853) and the BatchNormalization variants
 *      of resnet_block / mobilenetV2_block (:521-537, 884-886).  The reference composes Keras layers in Python; these are
 *      the same layers as stateless device calls (NHWC float32 device memory, the caller's buffers, weights in the Keras
 *      layouts).  Weight / bias gradients are ADDED to dw / db (zero them first).
 *      conv2d:     y = [relu](conv_SAME(x, w) + b), w [kh,kw,C,F], TF SAME padding, strides (sh, sw); a Dense layer is the
 *                  1x1 case on a [B,1,1,C] tensor.  backward takes dpre = the gradient at the PRE-activation output
 *                  (mvae_activation_backward gives it) and writes dx (nullable), dw += , db += (db nullable).
 *      activation: MVAE_LAYER_ACT_*; ATTENUATE = (tanh(param * x) + 1) / 2 (attenuate_activation, :191-198).  backward is
 *                  written on the OUTPUT y: dx = dy * act'(y).
 *      eltwise:    op 0 add, 1 subtract, 2 multiply.
 *      scale_channels: keras Multiply([x [B,H,W,C], m [B,C]]): y = x * m[b,c]; dx = dy * m; dm[b,c] = sum_hw dy * x.
 *      global_maxpool: GlobalMaxPool2D with the position of the (first) maximum; backward scatters dy there.
 *      maxpool_same:   MaxPooling2D(pool (ph,pw), strides (sh,sw), "same"); idx = y * W + x of the window's first maximum.
 *      batchnorm:  keras BatchNormalization over the rows of x [M,C]; training = batch statistics (biased variance),
 *                  else the given moving statistics; mean / invstd [C] are outputs kept for the backward; batch_var nullable.
 *                  backward: work >= 2 C floats.
 *      attention_core (:716-728, as written): S[b,i,j] = sum_p theta[b,p,i] phi[b,p,j]; scores = softmax_j S [B,F,F];
 *                  out[b,j,p] = sum_i scores[b,i,j] g[b,p,i], i.e. the [B,F,HW] buffer the reference then RESHAPES to
 *                  [B,H,W,F]; F <= 64.  backward: dout in that same [B,F,HW] layout; work >= B F F floats. ---- */
#define MVAE_LAYER_ACT_LINEAR 0
#define MVAE_LAYER_ACT_RELU 1
#define MVAE_LAYER_ACT_SIGMOID 2
#define MVAE_LAYER_ACT_TANH 3
#define MVAE_LAYER_ACT_ATTENUATE 4
int mvae_conv2d_forward(int32_t device, const float* x, int32_t B, int32_t H, int32_t W, int32_t C, const float* w, const float* b,
                        int32_t F, int32_t kh, int32_t kw, int32_t sh, int32_t sw, int32_t relu, float* y, void* stream);
int mvae_conv2d_backward(int32_t device, const float* x, const float* dpre, int32_t B, int32_t H, int32_t W, int32_t C,
                         const float* w, int32_t F, int32_t kh, int32_t kw, int32_t sh, int32_t sw, float* dx, float* dw, float* db,
                         void* stream);
/*      depthwise3x3: keras DepthwiseConv2D(3x3, strides 1, 'same', relu), w [3,3,C,1]; backward: work >= B H W C floats. */
int mvae_depthwise3x3_forward(int32_t device, const float* x, int32_t B, int32_t H, int32_t W, int32_t C, const float* w,
                              const float* b, float* y, void* stream);
int mvae_depthwise3x3_backward(int32_t device, const float* x, const float* y, const float* dy, int32_t B, int32_t H, int32_t W,
                               int32_t C, const float* w, float* dx, float* dw, float* db, float* work, void* stream);
int mvae_activation_forward(int32_t device, int32_t act, const float* x, float* y, int64_t n, float param, void* stream);
int mvae_activation_backward(int32_t device, int32_t act, const float* y, const float* dy, float* dx, int64_t n, float param,
                             void* stream);
int mvae_eltwise(int32_t device, int32_t op, const float* a, const float* b, float* out, int64_t n, void* stream);
int mvae_scale_channels_forward(int32_t device, const float* x, const float* m, float* y, int32_t B, int64_t HW, int32_t C,
                                void* stream);
int mvae_scale_channels_backward(int32_t device, const float* x, const float* m, const float* dy, float* dx, float* dm, int32_t B,
                                 int64_t HW, int32_t C, void* stream);
int mvae_global_maxpool_forward(int32_t device, const float* x, float* y, int32_t* idx, int32_t B, int64_t HW, int32_t C,
                                void* stream);
int mvae_global_maxpool_backward(int32_t device, const float* dy, const int32_t* idx, float* dx, int32_t B, int64_t HW, int32_t C,
                                 void* stream);
int mvae_maxpool_same_forward(int32_t device, const float* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ph, int32_t pw,
                              int32_t sh, int32_t sw, float* y, int32_t* idx, void* stream);
int mvae_maxpool_same_backward(int32_t device, const float* dy, const int32_t* idx, int32_t B, int32_t H, int32_t W, int32_t C,
                               int32_t ph, int32_t pw, int32_t sh, int32_t sw, float* dx, void* stream);
int mvae_batchnorm_forward(int32_t device, const float* x, int64_t M, int32_t C, const float* gamma, const float* beta, float eps,
                           int32_t training, const float* moving_mean, const float* moving_var, float* mean, float* invstd,
                           float* batch_var, float* y, void* stream);
int mvae_batchnorm_backward(int32_t device, const float* x, const float* dy, int64_t M, int32_t C, const float* gamma,
                            const float* mean, const float* invstd, int32_t training, float* dx, float* dgamma, float* dbeta,
                            float* work, void* stream);
int mvae_attention_core_forward(int32_t device, const float* theta, const float* phi, const float* g, int32_t B, int64_t HW,
                                int32_t F, float* scores, float* out, void* stream);
int mvae_attention_core_backward(int32_t device, const float* theta, const float* phi, const float* g, const float* scores,
                                 const float* dout, int32_t B, int64_t HW, int32_t F, float* dtheta, float* dphi, float* dg,
                                 float* work, void* stream);

/* ---- diagnostics (process-global): per-launch HIP-event timing on the launch stream, used by bench.py for
 *      the per-kernel roofline line.  report writes a JSON object {tag: {count, ms, bytes, flops}} (algorithmic
 *      bytes / flops summed over the launches), returns its length, and clears the records; it synchronises. ---- */
int mvae_profile_enable(int32_t on);
int64_t mvae_profile_report(char* buf, int64_t cap);

/* ---- debugging / parity: look up a saved intermediate of the last forward by name ---- */
int mvae_tensor_lookup(const mvae_handle* h, const char* name, float** ptr, int64_t* elems_per_image);
/* ... and its storage type (MVAE_ACT_F32 / MVAE_ACT_BF16: the wide tensors of a bfloat16 scale) */
int mvae_tensor_lookup2(const mvae_handle* h, const char* name, void** ptr, int64_t* elems_per_image, int32_t* dtype);
/* storage type a pyramid scale runs in: with MVAE_ACT_BF16 a scale whose shapes the bf16 kernels do not cover (the
 * 8x8 / 4x4 tops of a deep pyramid) stays float32; the scales only meet in the float32 3-channel pyramid / merge */
int mvae_scale_dtype(const mvae_handle* h, int32_t scale);

#ifdef __cplusplus
}
#endif
#endif /* MVAE_HIP_H */
