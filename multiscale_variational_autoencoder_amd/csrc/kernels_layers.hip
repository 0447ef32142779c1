// kernels_layers.hip -- device operators of the reference's remaining block library:
// attention_block / self_attention_block (layer_blocks.py:654-783), the excite / inhibit masks (:191-412), resnet_block with
// strides (:847-853) and the BatchNormalization variants (:521-537, 884-886).  The convolutions of those blocks run on the
// hot path's own launchers (kernels.h); what is specific to them lives here: the channel-attention core, the elementwise
// activations with their derivatives, global / windowed max pooling with argmax, channel scaling with its two gradients,
// and a per-channel BatchNorm over rows.  All tensors NHWC float32; every kernel is shape-generic.
#include "kernels.h"
#include "prof.h"

namespace mvae {

namespace {
constexpr int kBlk = 256;
inline unsigned grid_of(int64_t n) { int64_t g = (n + kBlk - 1) / kBlk; return (unsigned)(g < 1 ? 1 : (g > 65535 * 16 ? 65535 * 16 : g)); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// activations (Keras: relu, sigmoid, tanh; attenuate_activation(x, m) = (tanh(m x) + 1) / 2, layer_blocks.py:191-198).
// The derivative is written in terms of the OUTPUT y, which is what the blocks keep.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_fwd(int act, float x, float m) {
  switch (act) {
    case LAYER_ACT_RELU: return x > 0.f ? x : 0.f;
    case LAYER_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
    case LAYER_ACT_TANH: return tanhf(x);
    case LAYER_ACT_ATTENUATE: return 0.5f * (tanhf(m * x) + 1.f);
    default: return x;
  }
}
__device__ __forceinline__ float act_grad(int act, float y, float m) {
  switch (act) {
    case LAYER_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case LAYER_ACT_SIGMOID: return y * (1.f - y);
    case LAYER_ACT_TANH: return 1.f - y * y;
    case LAYER_ACT_ATTENUATE: return 2.f * m * y * (1.f - y);            // t = 2y - 1: (m / 2)(1 - t^2) = 2 m y (1 - y)
    default: return 1.f;
  }
}
__global__ void __launch_bounds__(256) k_act_fwd(int act, const float* __restrict__ x, float* __restrict__ y, int64_t n, float m) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) y[i] = act_fwd(act, x[i], m);
}
__global__ void __launch_bounds__(256) k_act_bwd(int act, const float* __restrict__ y, const float* __restrict__ dy,
                                                 float* __restrict__ dx, int64_t n, float m) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk)
    dx[i] = dy[i] * act_grad(act, y[i], m);
}
void launch_act_fwd(int act, const float* x, float* y, int64_t n, float m, hipStream_t s) {
  hipLaunchKernelGGL(k_act_fwd, dim3(grid_of(n)), dim3(kBlk), 0, s, act, x, y, n, m);
}
void launch_act_bwd(int act, const float* y, const float* dy, float* dx, int64_t n, float m, hipStream_t s) {
  hipLaunchKernelGGL(k_act_bwd, dim3(grid_of(n)), dim3(kBlk), 0, s, act, y, dy, dx, n, m);
}

// out = a (op) b, elementwise: 0 add, 1 subtract, 2 multiply
__global__ void __launch_bounds__(256) k_eltwise(int op, const float* __restrict__ a, const float* __restrict__ b,
                                                 float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const float x = a[i], y = b[i];
    out[i] = op == 0 ? x + y : (op == 1 ? x - y : x * y);
  }
}
void launch_eltwise(int op, const float* a, const float* b, float* out, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_eltwise, dim3(grid_of(n)), dim3(kBlk), 0, s, op, a, b, out, n);
}

// ---------------------------------------------------------------------------------------------------------------------
// keras Multiply([x [B,H,W,C], m [B,C]]) (the mask broadcast of excite_inhibit_block, layer_blocks.py:372-376):
// y = x * m[b, c];  dx = dy * m;  dm[b, c] = sum_hw dy * x
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_scale_ch(const float* __restrict__ x, const float* __restrict__ m, float* __restrict__ y,
                                                  int64_t HW, int C, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const int64_t b = i / (HW * C);
    y[i] = x[i] * m[b * C + (int)(i % C)];
  }
}
// one block per (image, channel chunk of 64): lanes along channels, rows strided over the block's 4 waves
__global__ void __launch_bounds__(256) k_scale_ch_bwd_m(const float* __restrict__ x, const float* __restrict__ dy,
                                                        float* __restrict__ dm, int64_t HW, int C) {
  __shared__ float red[4][64];
  const int b = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  float acc = 0.f;
  if (c < C)
    for (int64_t p = w; p < HW; p += 4) {
      const int64_t o = ((int64_t)b * HW + p) * C + c;
      acc += dy[o] * x[o];
    }
  red[w][threadIdx.x & 63] = acc;
  __syncthreads();
  if (w == 0 && c < C) dm[(int64_t)b * C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
void launch_scale_channels(const float* x, const float* m, float* y, int B, int64_t HW, int C, hipStream_t s) {
  const int64_t n = (int64_t)B * HW * C;
  hipLaunchKernelGGL(k_scale_ch, dim3(grid_of(n)), dim3(kBlk), 0, s, x, m, y, HW, C, n);
}
void launch_scale_channels_bwd_m(const float* x, const float* dy, float* dm, int B, int64_t HW, int C, hipStream_t s) {
  hipLaunchKernelGGL(k_scale_ch_bwd_m, dim3(B, (C + 63) / 64), dim3(256), 0, s, x, dy, dm, HW, C);
}

// ---------------------------------------------------------------------------------------------------------------------
// GlobalMaxPool2D (layer_blocks.py:320, 331, 338): y[b, c] = max_hw x, idx = first position of the maximum;
// backward: dx = 0 except dx[b, idx[b, c], c] = dy[b, c]
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gmax_fwd(const float* __restrict__ x, float* __restrict__ y, int* __restrict__ idx,
                                                  int64_t HW, int C) {
  __shared__ float rv[4][64];
  __shared__ int ri[4][64];
  const int b = blockIdx.x, l = threadIdx.x & 63, c = blockIdx.y * 64 + l, w = threadIdx.x >> 6;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  if (c < C)
    for (int64_t p = w; p < HW; p += 4) {
      const float v = x[((int64_t)b * HW + p) * C + c];
      if (v > best) { best = v; bi = (int)p; }               // strictly greater: the first maximum of this wave's rows
    }
  rv[w][l] = best; ri[w][l] = bi;
  __syncthreads();
  if (w == 0 && c < C) {
#pragma unroll
    for (int q = 1; q < 4; ++q)
      if (rv[q][l] > best || (rv[q][l] == best && ri[q][l] < bi)) { best = rv[q][l]; bi = ri[q][l]; }
    y[(int64_t)b * C + c] = best;
    idx[(int64_t)b * C + c] = bi;
  }
}
__global__ void __launch_bounds__(256) k_gmax_bwd(const float* __restrict__ dy, const int* __restrict__ idx, float* __restrict__ dx,
                                                  int64_t HW, int C, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const int c = (int)(i % C);
    const int64_t p = (i / C) % HW, b = i / (HW * C);
    dx[i] = idx[b * C + c] == (int)p ? dy[b * C + c] : 0.f;
  }
}
void launch_gmax_fwd(const float* x, float* y, int* idx, int B, int64_t HW, int C, hipStream_t s) {
  hipLaunchKernelGGL(k_gmax_fwd, dim3(B, (C + 63) / 64), dim3(256), 0, s, x, y, idx, HW, C);
}
void launch_gmax_bwd(const float* dy, const int* idx, float* dx, int B, int64_t HW, int C, hipStream_t s) {
  const int64_t n = (int64_t)B * HW * C;
  hipLaunchKernelGGL(k_gmax_bwd, dim3(grid_of(n)), dim3(kBlk), 0, s, dy, idx, dx, HW, C, n);
}

// ---------------------------------------------------------------------------------------------------------------------
// MaxPooling2D(pool (ph, pw), strides (sh, sw), padding "same") -- the skip path of a strided resnet_block
// (layer_blocks.py:847-853: pool = strides + 1).  TF SAME: OH = ceil(H / sh), pad_total = max((OH - 1) sh + ph - H, 0),
// pad_before = pad_total / 2; padded positions never win.  idx = flat input pixel (y * W + x) of the window's first maximum.
// backward: windows overlap when pool > stride, so dx is gathered: every input pixel sums the dy of the windows it won.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_maxpool_fwd(const float* __restrict__ x, float* __restrict__ y, int* __restrict__ idx,
                                                     int H, int W, int C, int OH, int OW, int ph, int pw, int sh, int sw, int pt,
                                                     int pl, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const int c = (int)(i % C);
    const int64_t q = i / C;
    const int ox = (int)(q % OW), oy = (int)((q / OW) % OH);
    const int64_t b = q / ((int64_t)OW * OH);
    float best = -INFINITY;
    int bi = -1;
    for (int a = 0; a < ph; ++a) {
      const int yy = oy * sh + a - pt;
      if (yy < 0 || yy >= H) continue;
      for (int e = 0; e < pw; ++e) {
        const int xx = ox * sw + e - pl;
        if (xx < 0 || xx >= W) continue;
        const float v = x[((b * H + yy) * W + xx) * C + c];
        if (v > best) { best = v; bi = yy * W + xx; }
      }
    }
    y[i] = best;
    idx[i] = bi;
  }
}
__global__ void __launch_bounds__(256) k_maxpool_bwd(const float* __restrict__ dy, const int* __restrict__ idx, float* __restrict__ dx,
                                                     int H, int W, int C, int OH, int OW, int ph, int pw, int sh, int sw, int pt,
                                                     int pl, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const int c = (int)(i % C);
    const int64_t q = i / C;
    const int xx = (int)(q % W), yy = (int)((q / W) % H);
    const int64_t b = q / ((int64_t)W * H);
    const int me = yy * W + xx;
    float acc = 0.f;
    // windows (oy, ox) that contain (yy, xx): oy * sh - pt <= yy < oy * sh - pt + ph
    const int oy_hi = (yy + pt) / sh, ox_hi = (xx + pl) / sw;
    for (int oy = oy_hi; oy >= 0 && oy * sh - pt + ph > yy; --oy) {
      if (oy >= OH) continue;
      for (int ox = ox_hi; ox >= 0 && ox * sw - pl + pw > xx; --ox) {
        if (ox >= OW) continue;
        const int64_t o = ((b * OH + oy) * OW + ox) * C + c;
        if (idx[o] == me) acc += dy[o];
      }
    }
    dx[i] = acc;
  }
}
void launch_maxpool_fwd(const float* x, float* y, int* idx, int B, int H, int W, int C, int OH, int OW, int ph, int pw, int sh,
                        int sw, int pt, int pl, hipStream_t s) {
  const int64_t n = (int64_t)B * OH * OW * C;
  hipLaunchKernelGGL(k_maxpool_fwd, dim3(grid_of(n)), dim3(kBlk), 0, s, x, y, idx, H, W, C, OH, OW, ph, pw, sh, sw, pt, pl, n);
}
void launch_maxpool_bwd(const float* dy, const int* idx, float* dx, int B, int H, int W, int C, int OH, int OW, int ph, int pw,
                        int sh, int sw, int pt, int pl, hipStream_t s) {
  const int64_t n = (int64_t)B * H * W * C;
  hipLaunchKernelGGL(k_maxpool_bwd, dim3(grid_of(n)), dim3(kBlk), 0, s, dy, idx, dx, H, W, C, OH, OW, ph, pw, sh, sw, pt, pl, n);
}

// ---------------------------------------------------------------------------------------------------------------------
// BatchNormalization over the rows of x [M, C] (keras default axis -1: per channel over batch and pixels).
//   training:  mean / biased variance of the batch (two passes), y = gamma (x - mean) invstd + beta
//   inference: the given moving statistics
// backward (training): dx = gamma invstd (dy - mean(dy) - xhat mean(dy xhat)); dgamma = sum dy xhat; dbeta = sum dy
// backward (inference): dx = gamma invstd dy.
// One block per 64 channels; its 4 waves stride the rows (column sums in a fixed order: deterministic).
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_bn_stats(const float* __restrict__ x, float* __restrict__ mean, float* __restrict__ invstd,
                                                  float* __restrict__ var_out, int64_t M, int C, float eps) {
  __shared__ float red[4][64];
  const int l = threadIdx.x & 63, c = blockIdx.x * 64 + l, w = threadIdx.x >> 6;
  float acc = 0.f;
  if (c < C)
    for (int64_t r = w; r < M; r += 4) acc += x[r * C + c];
  red[w][l] = acc;
  __syncthreads();
  const float mu = (red[0][l] + red[1][l] + red[2][l] + red[3][l]) / (float)M;
  __syncthreads();
  acc = 0.f;
  if (c < C)
    for (int64_t r = w; r < M; r += 4) { const float d = x[r * C + c] - mu; acc += d * d; }
  red[w][l] = acc;
  __syncthreads();
  if (w == 0 && c < C) {
    const float var = (red[0][l] + red[1][l] + red[2][l] + red[3][l]) / (float)M;
    mean[c] = mu;
    invstd[c] = rsqrtf(var + eps);
    if (var_out) var_out[c] = var;
  }
}
__global__ void __launch_bounds__(256) k_bn_apply(const float* __restrict__ x, const float* __restrict__ mean,
                                                  const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float* __restrict__ y, int C, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const int c = (int)(i % C);
    y[i] = (x[i] - mean[c]) * invstd[c] * gamma[c] + beta[c];
  }
}
__global__ void __launch_bounds__(256) k_bn_bwd_sums(const float* __restrict__ x, const float* __restrict__ dy,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t M, int C) {
  __shared__ float r1[4][64], r2[4][64];
  const int l = threadIdx.x & 63, c = blockIdx.x * 64 + l, w = threadIdx.x >> 6;
  float a1 = 0.f, a2 = 0.f;
  if (c < C) {
    const float mu = mean[c], is = invstd[c];
    for (int64_t r = w; r < M; r += 4) {
      const float d = dy[r * C + c];
      a1 += d;
      a2 += d * (x[r * C + c] - mu) * is;
    }
  }
  r1[w][l] = a1; r2[w][l] = a2;
  __syncthreads();
  if (w == 0 && c < C) {
    dbeta[c] += r1[0][l] + r1[1][l] + r1[2][l] + r1[3][l];
    dgamma[c] += r2[0][l] + r2[1][l] + r2[2][l] + r2[3][l];
  }
}
// sums: this launch's column sums (sum dy, sum dy xhat), NOT the accumulated dgamma / dbeta
__global__ void __launch_bounds__(256) k_bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ dy,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ sum_dy,
                                                      const float* __restrict__ sum_dyx, float* __restrict__ dx, float inv_m,
                                                      int training, int C, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const int c = (int)(i % C);
    const float gi = gamma[c] * invstd[c];
    if (training) {
      const float xh = (x[i] - mean[c]) * invstd[c];
      dx[i] = gi * (dy[i] - sum_dy[c] * inv_m - xh * sum_dyx[c] * inv_m);
    } else {
      dx[i] = gi * dy[i];
    }
  }
}
__global__ void __launch_bounds__(256) k_bn_from_moving(const float* __restrict__ mm, const float* __restrict__ mv,
                                                        float* __restrict__ mean, float* __restrict__ invstd, int C, float eps) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) { mean[c] = mm[c]; invstd[c] = rsqrtf(mv[c] + eps); }
}
void launch_bn_rows_from_moving(const float* mm, const float* mv, float* mean, float* invstd, int C, float eps, hipStream_t s) {
  hipLaunchKernelGGL(k_bn_from_moving, dim3((C + 255) / 256), dim3(256), 0, s, mm, mv, mean, invstd, C, eps);
}
void launch_bn_rows_stats(const float* x, float* mean, float* invstd, float* var_out, int64_t M, int C, float eps, hipStream_t s) {
  hipLaunchKernelGGL(k_bn_stats, dim3((C + 63) / 64), dim3(256), 0, s, x, mean, invstd, var_out, M, C, eps);
}
void launch_bn_rows_apply(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta, float* y,
                          int64_t M, int C, hipStream_t s) {
  const int64_t n = M * C;
  hipLaunchKernelGGL(k_bn_apply, dim3(grid_of(n)), dim3(kBlk), 0, s, x, mean, invstd, gamma, beta, y, C, n);
}
void launch_bn_rows_bwd_sums(const float* x, const float* dy, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                             int64_t M, int C, hipStream_t s) {
  hipLaunchKernelGGL(k_bn_bwd_sums, dim3((C + 63) / 64), dim3(256), 0, s, x, dy, mean, invstd, dgamma, dbeta, M, C);
}
void launch_bn_rows_bwd_apply(const float* x, const float* dy, const float* mean, const float* invstd, const float* gamma,
                              const float* sum_dy, const float* sum_dyx, float* dx, int training, int64_t M, int C, hipStream_t s) {
  const int64_t n = M * C;
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3(grid_of(n)), dim3(kBlk), 0, s, x, dy, mean, invstd, gamma, sum_dy, sum_dyx, dx,
                     1.0f / (float)M, training, C, n);
}

// ---------------------------------------------------------------------------------------------------------------------
// The attention core of attention_block (layer_blocks.py:716-728) AS THE REFERENCE WRITES IT.  With theta, phi, g
// [B, HW, F] (the three convolutions, flattened over pixels):
//     S[b, i, j] = sum_p theta[b, p, i] phi[b, p, j]          Dot(axes=(1, 2)) of (HW, F) with the permuted (F, HW): F x F
//     A = softmax over j                                       Softmax() on the last axis
//     O[b, j, p] = sum_i A[b, i, j] g[b, p, i]                 Dot(axes=(1, 2)) of (F, F) with (HW, F): (F, HW)
// and the (F, HW) result is RESHAPED (not transposed) to (H, W, F): the output buffer is O row-major.  It is a channel
// attention: the contraction runs over the pixels, the attention map is F x F per image -- nothing of size (HW)^2 exists.
//   scores : one block per (image, 1024-pixel chunk); thread (i, j-quad) accumulates over the chunk's pixels from LDS rows;
//            one float atomic per entry and chunk into S (zeroed by the launcher)
//   softmax: one wave per (image, i) row
//   apply  : thread = pixel p (lanes along p: the stores out[j * HW + p] are coalesced), A^T in LDS
// backward:  dA[i, j] = sum_p dO[j, p] g[p, i];  dg[p, i] = sum_j A[i, j] dO[j, p]
//            dS = A (dA - rowsum(A dA));  dtheta[p, i] = sum_j dS[i, j] phi[p, j];  dphi[p, j] = sum_i dS[i, j] theta[p, i]
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kAttnChunk = 512;       // pixels per block of the two reductions over pixels
constexpr int kAttnMaxF = 64;

// S[b] += X^T Y over the block's pixel chunk, X, Y [B, HW, F] (YT = false) or Y given as [B, F, HW] (YT = true: dO)
//   scores: X = theta, Y = phi         -> S[i][j]
//   dA    : X = g,     Y = dO^T (YT)   -> dA[i][j] = sum_p g[p][i] dO[j][p]
template <bool YT>
__global__ void __launch_bounds__(256) k_attn_xty(const float* __restrict__ X, const float* __restrict__ Y, float* __restrict__ S,
                                                  int64_t HW, int F) {
  __shared__ float sx[16][kAttnMaxF], sy[16][kAttnMaxF + 1];
  const int b = blockIdx.y;
  const int64_t p0 = (int64_t)blockIdx.x * kAttnChunk;
  const int64_t p1 = p0 + kAttnChunk < HW ? p0 + kAttnChunk : HW;
  // thread t owns entries e = t, t + 256, ... of the F x F matrix
  float acc[kAttnMaxF * kAttnMaxF / 256];
#pragma unroll
  for (int u = 0; u < kAttnMaxF * kAttnMaxF / 256; ++u) acc[u] = 0.f;
  const int FF = F * F;
  for (int64_t pb = p0; pb < p1; pb += 16) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < 16 * F; idx += 256) {
      const int r = idx / F, c = idx % F;
      const int64_t p = pb + r;
      sx[r][c] = p < p1 ? X[((int64_t)b * HW + p) * F + c] : 0.f;
      sy[r][c] = p < p1 ? (YT ? Y[((int64_t)b * F + c) * HW + p] : Y[((int64_t)b * HW + p) * F + c]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kAttnMaxF * kAttnMaxF / 256; ++u) {
      const int e = threadIdx.x + u * 256;
      if (e < FF) {
        const int i = e / F, j = e % F;
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a += sx[r][i] * sy[r][j];
        acc[u] += a;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < kAttnMaxF * kAttnMaxF / 256; ++u) {
    const int e = threadIdx.x + u * 256;
    if (e < FF) atomicAdd(&S[(int64_t)b * FF + e], acc[u]);
  }
}
// rows of [R, F]: in-place softmax (FWD) or dS = A (dA - sum_j A dA) written over dA (BWD); one wave per row
template <bool BWD>
__global__ void __launch_bounds__(256) k_attn_softmax(float* __restrict__ S, const float* __restrict__ A, int64_t R, int F) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int l = threadIdx.x & 63;
  if (row >= R) return;
  float* s = S + row * F;
  if (!BWD) {
    float v = l < F ? s[l] : -INFINITY, m = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const float e = l < F ? expf(v - m) : 0.f;
    const float z = wave_sum(e);
    if (l < F) s[l] = e / z;
  } else {
    const float a = l < F ? A[row * F + l] : 0.f, d = l < F ? s[l] : 0.f;
    const float dot = wave_sum(a * d);
    if (l < F) s[l] = a * (d - dot);
  }
}
// out[b, j, p] = sum_i M[b, i, j] X[b, p, i]   (TR = false: forward O from (A, g))
// out[b, p, i] = sum_j M[b, i, j] Y[b, j, p]   (TR = true : dg from (A, dO);  Y given [B, F, HW])
// thread = pixel; the F x F matrix in LDS
template <bool TR, int FT>
__global__ void __launch_bounds__(256) k_attn_apply(const float* __restrict__ Mx, const float* __restrict__ X, float* __restrict__ out,
                                                    int64_t HW, int F) {
  __shared__ float sm[kAttnMaxF * kAttnMaxF];
  const int b = blockIdx.y;
  for (int idx = threadIdx.x; idx < F * F; idx += 256) sm[idx] = Mx[(int64_t)b * F * F + idx];
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  if (!TR) {
    float xv[FT];
#pragma unroll
    for (int i = 0; i < FT; ++i) xv[i] = i < F ? X[((int64_t)b * HW + p) * F + i] : 0.f;
    for (int j = 0; j < F; ++j) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < FT; ++i) a += (i < F ? sm[i * F + j] : 0.f) * xv[i];
      out[((int64_t)b * F + j) * HW + p] = a;
    }
  } else {
    float yv[FT];
#pragma unroll
    for (int j = 0; j < FT; ++j) yv[j] = j < F ? X[((int64_t)b * F + j) * HW + p] : 0.f;
    for (int i = 0; i < F; ++i) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < FT; ++j) a += (j < F ? sm[i * F + j] : 0.f) * yv[j];
      out[((int64_t)b * HW + p) * F + i] = a;
    }
  }
}
// out[b, p, :] = M[b] (or M[b]^T) applied to the pixel's vector:  TRM = false: out[p][i] = sum_j M[i][j] X[p][j] (dtheta from
// dS and phi);  TRM = true: out[p][j] = sum_i M[i][j] X[p][i] (dphi from dS and theta)
template <bool TRM, int FT>
__global__ void __launch_bounds__(256) k_attn_rows(const float* __restrict__ Mx, const float* __restrict__ X, float* __restrict__ out,
                                                   int64_t HW, int F) {
  __shared__ float sm[kAttnMaxF * kAttnMaxF];
  const int b = blockIdx.y;
  for (int idx = threadIdx.x; idx < F * F; idx += 256) sm[idx] = Mx[(int64_t)b * F * F + idx];
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  float xv[FT];
#pragma unroll
  for (int i = 0; i < FT; ++i) xv[i] = i < F ? X[((int64_t)b * HW + p) * F + i] : 0.f;
  for (int o = 0; o < F; ++o) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < FT; ++k) a += (k < F ? (TRM ? sm[k * F + o] : sm[o * F + k]) : 0.f) * xv[k];
    out[((int64_t)b * HW + p) * F + o] = a;
  }
}

bool launch_attention_core_fwd(const float* theta, const float* phi, const float* g, float* scores, float* out, int B, int64_t HW,
                               int F, hipStream_t s) {
  if (F < 1 || F > kAttnMaxF || B < 1 || HW < 1) return false;
  launch_zero(scores, (int64_t)B * F * F, s);
  const dim3 gc((unsigned)((HW + kAttnChunk - 1) / kAttnChunk), B), gp((unsigned)((HW + 255) / 256), B);
  hipLaunchKernelGGL(k_attn_xty<false>, gc, dim3(256), 0, s, theta, phi, scores, HW, F);
  hipLaunchKernelGGL(k_attn_softmax<false>, dim3((unsigned)(((int64_t)B * F + 3) / 4)), dim3(256), 0, s, scores, nullptr, (int64_t)B * F, F);
  if (F <= 32) hipLaunchKernelGGL((k_attn_apply<false, 32>), gp, dim3(256), 0, s, scores, g, out, HW, F);
  else hipLaunchKernelGGL((k_attn_apply<false, 64>), gp, dim3(256), 0, s, scores, g, out, HW, F);
  return true;
}
// work: B * F * F floats
bool launch_attention_core_bwd(const float* theta, const float* phi, const float* g, const float* scores, const float* dout,
                               float* dtheta, float* dphi, float* dg, float* work, int B, int64_t HW, int F, hipStream_t s) {
  if (F < 1 || F > kAttnMaxF || B < 1 || HW < 1) return false;
  const dim3 gc((unsigned)((HW + kAttnChunk - 1) / kAttnChunk), B), gp((unsigned)((HW + 255) / 256), B);
  launch_zero(work, (int64_t)B * F * F, s);
  hipLaunchKernelGGL(k_attn_xty<true>, gc, dim3(256), 0, s, g, dout, work, HW, F);                     // dA
  if (F <= 32) hipLaunchKernelGGL((k_attn_apply<true, 32>), gp, dim3(256), 0, s, scores, dout, dg, HW, F);      // dg
  else hipLaunchKernelGGL((k_attn_apply<true, 64>), gp, dim3(256), 0, s, scores, dout, dg, HW, F);
  hipLaunchKernelGGL(k_attn_softmax<true>, dim3((unsigned)(((int64_t)B * F + 3) / 4)), dim3(256), 0, s, work, scores, (int64_t)B * F, F);   // dS
  if (F <= 32) {
    hipLaunchKernelGGL((k_attn_rows<false, 32>), gp, dim3(256), 0, s, work, phi, dtheta, HW, F);
    hipLaunchKernelGGL((k_attn_rows<true, 32>), gp, dim3(256), 0, s, work, theta, dphi, HW, F);
  } else {
    hipLaunchKernelGGL((k_attn_rows<false, 64>), gp, dim3(256), 0, s, work, phi, dtheta, HW, F);
    hipLaunchKernelGGL((k_attn_rows<true, 64>), gp, dim3(256), 0, s, work, theta, dphi, HW, F);
  }
  return true;
}

}  // namespace mvae
