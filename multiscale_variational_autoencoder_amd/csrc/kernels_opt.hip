// kernels_opt.hip -- tuned VALU kernels for the ops that are not MFMA-shaped: the skinny Dense layers around
// the latent (K or N = z <= 32 against 512..524288), squeeze-excite weight gradients, and the depthwise 3x3
// convolution (forward / backward-data vectorised 16 B per lane, weight gradient with a sliding 3x3 register
// window).  Reference: mvae/multiscale_vae.py:358-370,402-406 (Dense mu / log_var / decoder Dense),
// mvae/layer_blocks.py:440-456 (SE Dense), :604-614 (DepthwiseConv2D).
#include "kernels.h"
#include "act16.h"
#include "prof.h"

namespace mvae {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float hsig_grad_o(float u) { return (u >= -2.5f && u <= 2.5f) ? 0.2f : 0.f; }
__device__ __forceinline__ float act_apply_o(float v, int act) {
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == ACT_ELU) return v > 0.f ? v : expm1f(v);
  if (act == ACT_HSIG) return fminf(fmaxf(0.2f * v + 0.5f, 0.f), 1.f);
  return v;
}
__device__ __forceinline__ float wave_sum_o(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// -------------------------------------------------------------------------------------------------
// dW[k,n] += sum_b a'[b,k] g'[b,n] ; db[n] += sum_b g'[b,n]   (small K x N; the batch axis is split over
// blockIdx.y so that a 64x64 problem still fills the chip; float atomics, dW/db pre-zeroed)
// -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gemm_tn_split(const float* __restrict__ a, const float* __restrict__ g,
                                                       float* __restrict__ dW, float* __restrict__ db, int B, int K,
                                                       int N, const float* __restrict__ a_scale,
                                                       const float* __restrict__ a_shift,
                                                       const float* __restrict__ hs_lin, int bpc) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)K * N) return;
  const int j = (int)(i % N), k = (int)(i / N);
  const float sc = a_scale ? a_scale[k] : 1.f, sf = a_scale ? a_shift[k] : 0.f;
  const int b0 = blockIdx.y * bpc, b1 = min(B, b0 + bpc);
  float acc = 0.f, accb = 0.f;
#pragma unroll 8
  for (int b = b0; b < b1; ++b) {
    float gv = g[(int64_t)b * N + j];
    if (hs_lin) gv *= hsig_grad_o(hs_lin[(int64_t)b * N + j]);
    acc += (a[(int64_t)b * K + k] * sc + sf) * gv;
    accb += gv;
  }
  atomicAdd(&dW[i], acc);
  if (k == 0 && db) atomicAdd(&db[j], accb);
}

// -------------------------------------------------------------------------------------------------
// Outer products against a WIDE matrix: out (+)= sum_b wide[b, w] * small[b, j], j < NS <= 32.
//   LAYOUT 0: out[w][j]  (Dense mu/log_var weight gradient: wide = flattened activations)
//   LAYOUT 1: out[j][w]  (decoder Dense weight gradient:   wide = upstream gradient)
// one thread per wide column (coalesced), the small rows are broadcast from LDS; batch split over blockIdx.y.
// -------------------------------------------------------------------------------------------------
template <int LAYOUT>
__global__ void __launch_bounds__(256) k_outer_wide(const float* __restrict__ wide, const float* __restrict__ small,
                                                    float* __restrict__ out, float* __restrict__ dbw,
                                                    float* __restrict__ dbs, int B, int Wd, int NS, int bpc) {
  __shared__ float ssm[64 * 32];
  const int b0 = blockIdx.y * bpc, b1 = min(B, b0 + bpc);
  for (int t = threadIdx.x; t < (b1 - b0) * NS; t += 256) ssm[t] = small[(int64_t)b0 * NS + t];
  __syncthreads();
  const int w = blockIdx.x * 256 + threadIdx.x;
  float acc[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) acc[j] = 0.f;
  float accw = 0.f;
  if (w < Wd) {
    for (int b = b0; b < b1; ++b) {
      const float v = wide[(int64_t)b * Wd + w];
      accw += v;
      const float* sp = ssm + (b - b0) * NS;
#pragma unroll
      for (int j = 0; j < 32; ++j)
        if (j < NS) acc[j] += v * sp[j];
    }
#pragma unroll
    for (int j = 0; j < 32; ++j)
      if (j < NS) atomicAdd(LAYOUT == 0 ? &out[(int64_t)w * NS + j] : &out[(int64_t)j * Wd + w], acc[j]);
    if (dbw) atomicAdd(&dbw[w], accw);
  }
  if (dbs && blockIdx.x == 0 && threadIdx.x < NS) {
    float t = 0.f;
    for (int b = 0; b < b1 - b0; ++b) t += ssm[b * NS + threadIdx.x];
    atomicAdd(&dbs[threadIdx.x], t);
  }
}

// -------------------------------------------------------------------------------------------------
// Skinny row products, one block per batch row, NS <= 32 outputs per row, long reduction axis L:
//   NTF = false: out[b, j] = act(bias[j] + sum_l x[b,l] * W[l*NS + j])       (Dense mu / log_var forward)
//   NTF = true : out[b, j] =            sum_l x[b,l] * W[j*L + l]            (decoder Dense backward-data)
// -------------------------------------------------------------------------------------------------
template <bool NTF>
__global__ void __launch_bounds__(256) k_rowdot(const float* __restrict__ x, const float* __restrict__ W,
                                                const float* __restrict__ bias, float* __restrict__ out, int L, int NS,
                                                int act) {
  __shared__ float red[4][32];
  const int b = blockIdx.x;
  const float* xp = x + (int64_t)b * L;
  float acc[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) acc[j] = 0.f;
  for (int l = threadIdx.x; l < L; l += 256) {
    const float xv = xp[l];
#pragma unroll
    for (int j = 0; j < 32; ++j)
      if (j < NS) acc[j] += xv * (NTF ? W[(int64_t)j * L + l] : W[(int64_t)l * NS + j]);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    if (j < NS) {
      float t = wave_sum_o(acc[j]);
      if (lane == 0) red[wave][j] = t;
    }
  }
  __syncthreads();
  if (threadIdx.x < NS) {
    float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (!NTF && bias) t += bias[threadIdx.x];
    out[(int64_t)b * NS + threadIdx.x] = act_apply_o(t, act);
  }
}

// -------------------------------------------------------------------------------------------------
// depthwise 3x3, stride 1, SAME, kernel [3][3][C]; 4 channels (16 B) per lane
// -------------------------------------------------------------------------------------------------
template <bool BWD>
__global__ void __launch_bounds__(256) k_dw_v4(const f32x4* __restrict__ in, const f32x4* __restrict__ w,
                                               const f32x4* __restrict__ bias, const f32x4* __restrict__ mask_src,
                                               f32x4* __restrict__ out, int B, int H, int W, int C4) {
  const int64_t n = (int64_t)B * H * W * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    int64_t p = i / C4;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int64_t b = p / H;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (!BWD) acc = bias[c4];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int yy = BWD ? y - (a - 1) : y + a - 1;
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int xx = BWD ? x - (e - 1) : x + e - 1;
        if (xx < 0 || xx >= W) continue;
        acc += w[(a * 3 + e) * C4 + c4] * in[((b * H + yy) * W + xx) * C4 + c4];
      }
    }
    f32x4 r;
    if (BWD) {
      const f32x4 m = mask_src[i];
#pragma unroll
      for (int q = 0; q < 4; ++q) r[q] = m[q] > 0.f ? acc[q] : 0.f;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) r[q] = acc[q] > 0.f ? acc[q] : 0.f;
    }
    out[i] = r;
  }
}

// depthwise weight gradient: dW[a][e][c] += sum dy[b,y,x,c] * in[b,y+a-1,x+e-1,c] ; db[c] += sum dy.
// block = C channels x (256 / C) row groups, walks whole images; each thread slides a 3x3 register window of
// `in` along x (4 loads per pixel instead of 10); one set of atomics per block.
__global__ void __launch_bounds__(256) k_dw_wgrad_slide(const float* __restrict__ in, const float* __restrict__ dy,
                                                        float* __restrict__ dW, float* __restrict__ db, int B, int H,
                                                        int W, int C) {
  __shared__ float sh[10 * 256];
  const int c = threadIdx.x % C, g = threadIdx.x / C, rg = 256 / C;
  float acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.f;
  float accb = 0.f;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const float* ib = in + (int64_t)b * H * W * C + c;
    const float* db_ = dy + (int64_t)b * H * W * C + c;
    for (int y = g; y < H; y += rg) {
      float L[3], Mi[3], R[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int yy = y + a - 1;
        const bool ok = yy >= 0 && yy < H;
        L[a] = 0.f;
        Mi[a] = ok ? ib[((int64_t)yy * W + 0) * C] : 0.f;
        R[a] = (ok && W > 1) ? ib[((int64_t)yy * W + 1) * C] : 0.f;
      }
      for (int x = 0; x < W; ++x) {
        const float d = db_[((int64_t)y * W + x) * C];
        accb += d;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          acc[a * 3 + 0] += d * L[a];
          acc[a * 3 + 1] += d * Mi[a];
          acc[a * 3 + 2] += d * R[a];
          L[a] = Mi[a];
          Mi[a] = R[a];
          const int yy = y + a - 1;
          R[a] = (x + 2 < W && yy >= 0 && yy < H) ? ib[((int64_t)yy * W + x + 2) * C] : 0.f;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) sh[k * 256 + threadIdx.x] = acc[k];
  sh[9 * 256 + threadIdx.x] = accb;
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      float t = 0.f;
      for (int r = 0; r < rg; ++r) t += sh[k * 256 + r * C + c];
      if (k < 9) atomicAdd(&dW[k * C + c], t);
      else atomicAdd(&db[c], t);
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Squeeze-excite backward pair in ONE launch (the two products share g' and are independent):
//   blocks [0, tn_blocks):  dW[k,n] += sum_b a'[b,k] g'[b,n] ; db[n] += sum_b g'[b,n]     (batch split in `chunks`)
//   remaining blocks     :  dx[b,k]  = sum_n g'[b,n] W[k,n]
// with a' = a*a_scale+a_shift (folded BatchNorm output) and g' = g * hsig'(hs_lin) when hs_lin != null.
// -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_se_pair(const float* __restrict__ a, const float* __restrict__ g,
                                                 const float* __restrict__ W, float* __restrict__ dW,
                                                 float* __restrict__ db, float* __restrict__ dx, int B, int K, int N,
                                                 const float* __restrict__ a_scale, const float* __restrict__ a_shift,
                                                 const float* __restrict__ hs_lin, int bpc, int chunks, int tn_blocks) {
  if ((int)blockIdx.x < tn_blocks) {
    const int bx = blockIdx.x / chunks, by = blockIdx.x % chunks;
    const int64_t i = (int64_t)bx * 256 + threadIdx.x;
    if (i >= (int64_t)K * N) return;
    const int j = (int)(i % N), k = (int)(i / N);
    const float sc = a_scale ? a_scale[k] : 1.f, sf = a_scale ? a_shift[k] : 0.f;
    const int b0 = by * bpc, b1 = min(B, b0 + bpc);
    float acc = 0.f, accb = 0.f;
#pragma unroll 8
    for (int b = b0; b < b1; ++b) {
      float gv = g[(int64_t)b * N + j];
      if (hs_lin) gv *= hsig_grad_o(hs_lin[(int64_t)b * N + j]);
      acc += (a[(int64_t)b * K + k] * sc + sf) * gv;
      accb += gv;
    }
    atomicAdd(&dW[i], acc);
    if (k == 0 && db) atomicAdd(&db[j], accb);
  } else {
    const int64_t i = (int64_t)(blockIdx.x - tn_blocks) * 256 + threadIdx.x;
    if (i >= (int64_t)B * K) return;
    const int k = (int)(i % K);
    const int64_t b = i / K;
    const float* gp = g + b * N;
    const float* wp = W + (int64_t)k * N;
    float acc = 0.f;
    if (hs_lin) {
      const float* hp = hs_lin + b * N;
#pragma unroll 8
      for (int j = 0; j < N; ++j) acc += gp[j] * hsig_grad_o(hp[j]) * wp[j];
    } else {
#pragma unroll 8
      for (int j = 0; j < N; ++j) acc += gp[j] * wp[j];
    }
    dx[i] = acc;
  }
}
void launch_se_pair(const float* a, const float* g, const float* W, float* dW, float* db, float* dx, int B, int K, int N,
                    const float* a_scale, const float* a_shift, const float* hs_lin, hipStream_t s) {
  int chunks = B >= 256 ? 16 : (B >= 32 ? 4 : 1);
  int bpc = (B + chunks - 1) / chunks;
  chunks = (B + bpc - 1) / bpc;
  const int tn_blocks = (int)(((int64_t)K * N + 255) / 256) * chunks;
  const int nt_blocks = (int)(((int64_t)B * K + 255) / 256);
  hipLaunchKernelGGL(k_se_pair, dim3(tn_blocks + nt_blocks), dim3(256), 0, s, a, g, W, dW, db, dx, B, K, N, a_scale,
                     a_shift, hs_lin, bpc, chunks, tn_blocks);
}

// ---- launchers: return false when the shape is not covered ------------------------------------------------
bool launch_gemm_tn_opt(const float* a, const float* g, float* dW, float* db, int B, int K, int N,
                        const float* a_scale, const float* a_shift, const float* hs_lin, hipStream_t s) {
  const bool plain = !a_scale && !hs_lin;
  if (plain && N <= 32 && K >= 256) {          // wide = a [B,K], small = g [B,N] -> dW[K][N]; db over small
    int bpc = B >= 512 ? 64 : (B >= 64 ? 32 : B);
    if (bpc < 1) bpc = 1;
    hipLaunchKernelGGL(k_outer_wide<0>, dim3((K + 255) / 256, (B + bpc - 1) / bpc), dim3(256), 0, s, a, g, dW,
                       (float*)nullptr, db, B, K, N, bpc);
    return true;
  }
  if (plain && K <= 32 && N >= 256) {          // wide = g [B,N], small = a [B,K] -> dW[K][N]; db over wide
    int bpc = B >= 512 ? 64 : (B >= 64 ? 32 : B);
    if (bpc < 1) bpc = 1;
    hipLaunchKernelGGL(k_outer_wide<1>, dim3((N + 255) / 256, (B + bpc - 1) / bpc), dim3(256), 0, s, g, a, dW, db,
                       (float*)nullptr, B, N, K, bpc);
    return true;
  }
  int chunks = B >= 256 ? 16 : (B >= 32 ? 4 : 1);
  int bpc = (B + chunks - 1) / chunks;
  hipLaunchKernelGGL(k_gemm_tn_split, dim3((unsigned)(((int64_t)K * N + 255) / 256), (B + bpc - 1) / bpc), dim3(256), 0,
                     s, a, g, dW, db, B, K, N, a_scale, a_shift, hs_lin, bpc);
  return true;
}

bool launch_gemm_nn_opt(const float* a, const float* w, const float* bias, float* out, float* out_lin, int B, int K,
                        int N, int act, hipStream_t s) {
  if (out_lin || N > 32 || K < 256) return false;
  hipLaunchKernelGGL(k_rowdot<false>, dim3(B), dim3(256), 0, s, a, w, bias, out, K, N, act);
  return true;
}

bool launch_gemm_nt_opt(const float* a, const float* w, float* out, int B, int K, int N, const float* hs_lin,
                        int accumulate, hipStream_t s) {
  if (hs_lin || accumulate || K > 32 || N < 256) return false;
  hipLaunchKernelGGL(k_rowdot<true>, dim3(B), dim3(256), 0, s, a, w, (const float*)nullptr, out, N, K, (int)ACT_NONE);
  return true;
}

static inline int grid_v4(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 256 * 32 ? 256 * 32 : g));
}
bool launch_dw_fwd_opt(const float* in, const float* w, const float* b, float* out, int B, int H, int W, int C,
                       hipStream_t s) {
  if (C % 4) return false;
  int64_t n = (int64_t)B * H * W * (C / 4);
  hipLaunchKernelGGL(k_dw_v4<false>, dim3(grid_v4(n)), dim3(256), 0, s, (const f32x4*)in, (const f32x4*)w,
                     (const f32x4*)b, (const f32x4*)nullptr, (f32x4*)out, B, H, W, C / 4);
  return true;
}
bool launch_dw_bwd_data_opt(const float* dy, const float* w, const float* mask_src, float* dx, int B, int H, int W,
                            int C, hipStream_t s) {
  if (C % 4) return false;
  int64_t n = (int64_t)B * H * W * (C / 4);
  hipLaunchKernelGGL(k_dw_v4<true>, dim3(grid_v4(n)), dim3(256), 0, s, (const f32x4*)dy, (const f32x4*)w,
                     (const f32x4*)nullptr, (const f32x4*)mask_src, (f32x4*)dx, B, H, W, C / 4);
  return true;
}
bool launch_dw_wgrad_opt(const float* in, const float* dy, float* dW, float* db, int B, int H, int W, int C,
                         hipStream_t s) {
  if (C > 256 || (256 % C) != 0) return false;
  int grid = B < 512 ? B : 512;
  hipLaunchKernelGGL(k_dw_wgrad_slide, dim3(grid), dim3(256), 0, s, in, dy, dW, db, B, H, W, C);
  return true;
}

// ------------------------------------------------------------------------------------------------
// Column statistics of a [M, C] tensor, 16 bytes per lane, 8 independent loads in flight per thread.
//   MODE 0: out[c] += sum_m x[m,c]            MODE 1: out[c] += sum_m (x[m,c] - mean[c])^2,
// mean[c] = inv_m * sum over the `msl` slot copies of a MODE 0 result.
//   MODE 2 (round 4): both moments in ONE pass about a pivot: d = x[m,c] - x[0,c]; out[c] += sum d, out2[c] += sum d^2.  The
// finalize kernel turns them into mean = x[0,c] + S1 / M and M var = S2 - S1^2 / M: the pivot is a sample of the column, so
// (mean - pivot)^2 is of the order of the variance and the subtraction loses a few bits at most (a pivot of 0, i.e.
// E[x^2] - E[x]^2, would lose log2(mean^2 / var) of them).  Saves a full read pass of the decoder's last activation.  The block's result leaves as one atomic set into
// slot (block % nslots) (kernels.h: GradSlots -- a few hundred blocks adding into ONE 128-byte line serialise).
// ------------------------------------------------------------------------------------------------
template <int MODE, typename T>
__global__ void __launch_bounds__(256) k_colstat4(const V4<T> x, const float* __restrict__ msum, int msl,
                                                  float inv_m, float* __restrict__ out, int nslots, int64_t slot_stride,
                                                  int64_t M, int C4, int64_t rpb, float* __restrict__ out2) {
  __shared__ f32x4 red[4][64];
  const int c4 = threadIdx.x % C4, rl = threadIdx.x / C4, nr = 256 / C4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 mean = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 1) {
    for (int k = 0; k < msl; ++k) mean += reinterpret_cast<const f32x4*>(msum + (int64_t)k * C4 * 4)[c4];
    mean *= inv_m;
  }
  if (MODE == 2) mean = V4<T>::cv(x.ld(c4));              // the pivot: row 0 of the column
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
  const int64_t m0 = (int64_t)blockIdx.x * rpb;
  int64_t m1 = m0 + rpb;
  if (m1 > M) m1 = M;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t mb = m0 + rl; mb < m1; mb += 8 * nr) {
    typename V4<T>::raw vr[8];
    float mk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t m = mb + u * nr;
      mk[u] = m < m1 ? 1.f : 0.f;
      vr[u] = x.ld((m < m1 ? m : m0) * C4 + c4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const f32x4 v = V4<T>::cv(vr[u]);
      if (MODE == 0) {
        acc += v * mk[u];
      } else if (MODE == 1) {
        const f32x4 d = v - mean;
        acc += d * d * mk[u];
      } else {
        const f32x4 d = (v - mean) * mk[u];
        acc += d;
        acc2 += d * d;
      }
    }
  }
  for (int off = C4; off < 64; off <<= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += __shfl_xor(acc[e], off, 64);
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (threadIdx.x < C4 * 4) {
    const int cc = threadIdx.x >> 2, e = threadIdx.x & 3;
    const float t = red[0][cc][e] + red[1][cc][e] + red[2][cc][e] + red[3][cc][e];
    atomicAdd(out + (int64_t)(blockIdx.x % nslots) * slot_stride + threadIdx.x, t);
  }
  if (MODE == 2) {
    for (int off = C4; off < 64; off <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc2[e] += __shfl_xor(acc2[e], off, 64);
    }
    __syncthreads();
    red[wave][lane] = acc2;
    __syncthreads();
    if (threadIdx.x < C4 * 4) {
      const int cc = threadIdx.x >> 2, e = threadIdx.x & 3;
      const float t = red[0][cc][e] + red[1][cc][e] + red[2][cc][e] + red[3][cc][e];
      atomicAdd(out2 + (int64_t)(blockIdx.x % nslots) * slot_stride + threadIdx.x, t);
    }
  }
}

// false = shape not covered (C not a power-of-two multiple of 4 up to 256)
bool launch_colstat_opt(int mode, const float* x, const float* msum, int msl, float inv_m, float* out, int nslots,
                        int64_t slot_stride, int64_t M, int C, hipStream_t s, bool bf, float* out2) {
  if (mode == 2 && !out2) return false;
  if (C < 4 || C > 256 || (C & (C - 1))) return false;
  const int C4 = C / 4, nr = 256 / C4;
  int64_t rpb = 8 * nr;                                  // one unrolled trip per thread at least
  while ((M + rpb - 1) / rpb > 1024) rpb *= 2;
  const unsigned grid = (unsigned)((M + rpb - 1) / rpb);
  ProfScope ps("col_reduce", (bf ? 2.0 : 4.0) * M * C, 0.0, s);
  const int ns = nslots < 1 ? 1 : nslots;
  if (bf) {
    const V4<bf16_t> xv((const bf16_t*)x);
    if (mode == 0) hipLaunchKernelGGL((k_colstat4<0, bf16_t>), dim3(grid), dim3(256), 0, s, xv, msum, msl, inv_m, out, ns, slot_stride, M, C4, rpb, nullptr);
    else if (mode == 1) hipLaunchKernelGGL((k_colstat4<1, bf16_t>), dim3(grid), dim3(256), 0, s, xv, msum, msl, inv_m, out, ns, slot_stride, M, C4, rpb, nullptr);
    else hipLaunchKernelGGL((k_colstat4<2, bf16_t>), dim3(grid), dim3(256), 0, s, xv, msum, msl, inv_m, out, ns, slot_stride, M, C4, rpb, out2);
  } else {
    const V4<float> xv(x);
    if (mode == 0) hipLaunchKernelGGL((k_colstat4<0, float>), dim3(grid), dim3(256), 0, s, xv, msum, msl, inv_m, out, ns, slot_stride, M, C4, rpb, nullptr);
    else if (mode == 1) hipLaunchKernelGGL((k_colstat4<1, float>), dim3(grid), dim3(256), 0, s, xv, msum, msl, inv_m, out, ns, slot_stride, M, C4, rpb, nullptr);
    else hipLaunchKernelGGL((k_colstat4<2, float>), dim3(grid), dim3(256), 0, s, xv, msum, msl, inv_m, out, ns, slot_stride, M, C4, rpb, out2);
  }
  return true;
}

}  // namespace mvae
