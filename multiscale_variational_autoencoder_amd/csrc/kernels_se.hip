// kernels_se.hip -- the squeeze-excite branch of the MobileNetV3 block (layer_blocks.py:418-462) as two forward and
// two backward launches (it was 3 + 3: Dense, BatchNorm-1D, Dense each on their own).
//
//   forward :  s0 = relu(gap W0 + b0) ; s1 = BN_batch(s0) ; u = s1 W1 + b1 ; g = hard_sigmoid(u)
//   backward:  du = dg * hsig'(u) ; dW1, db1 ; ds1 = du W1^T ; BatchNorm backward (batch statistics) -> dv (through the
//              ReLU) ; dW0, db0 ; dgap = dv W0^T
//
// The tensors are tiny ([B, c], c = 32/64): the cost is launch count and dependent memory round trips, not bytes.  All
// four kernels partition the BATCH ROWS over blocks; the only cross-row couplings are the BatchNorm statistics:
//   * forward : kernel 1 leaves per-block (mean, M2) of its rows; kernel 2 merges them with the pairwise-update formula
//               (Chan et al.), i.e. a two-pass-accurate variance with no atomics and no extra pass;
//   * backward: kernel 1 leaves per-block column sums of ds1 and ds1*xhat; kernel 2 adds them up before it applies the
//               BatchNorm backward to its rows.
// Weight gradients leave as one atomic set per block into the gradient slots (kernels.h: GradSlots).
#include "kernels.h"
#include "prof.h"

namespace mvae {

namespace {
constexpr int kSeRowsMax = 64;                       // rows per block: 16 (B <= 1024) or 64; LDS tiles are [ROWS][C]

__device__ __forceinline__ float se_hsig(float v) { return fminf(fmaxf(0.2f * v + 0.5f, 0.f), 1.f); }
__device__ __forceinline__ float se_hsig_grad(float u) { return (u >= -2.5f && u <= 2.5f) ? 0.2f : 0.f; }

// Every global load of these kernels is issued up front as one unrolled batch (weights, per-block partials, the
// block's rows): the kernels are chains of dependent memory round trips, so each trip saved is ~1 us of the ~5.
//
// wcol[k] = W[k*C + j]: the Dense weight column of thread j, one batch of C coalesced loads
template <int C>
__device__ __forceinline__ void se_load_wcol(const float* __restrict__ W, int j, float (&wcol)[C]) {
#pragma unroll
  for (int k = 0; k < C; ++k) wcol[k] = W[k * C + j];
}
// out[r][j] = bias[j] + sum_k sA[r][k] * wcol[k] for the block's rows; thread (j = t % C, rq = t / C) owns rows
// rq, rq + G, ...
template <int C, int RPT>
__device__ __forceinline__ void se_fc_rows(const float* sA, const float (&wcol)[C], float bias, int rq, float (&acc)[RPT]) {
  constexpr int G = 256 / C;
#pragma unroll
  for (int q = 0; q < RPT; ++q) acc[q] = bias;
#pragma unroll
  for (int k0 = 0; k0 < C; k0 += 4) {
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(&sA[(rq + q * G) * C + k0]);
      acc[q] += a.x * wcol[k0] + a.y * wcol[k0 + 1] + a.z * wcol[k0 + 2] + a.w * wcol[k0 + 3];
    }
  }
}
// W[C][C] -> sW with row pitch C + 1: the C*C/256 loads of a thread as one batch, then the LDS stores
template <int C>
__device__ __forceinline__ void se_stage_w(const float* __restrict__ W, float* sW) {
  float wv[C * C / 256];
#pragma unroll
  for (int u = 0; u < C * C / 256; ++u) wv[u] = W[threadIdx.x + u * 256];
#pragma unroll
  for (int u = 0; u < C * C / 256; ++u) {
    const int idx = threadIdx.x + u * 256;
    sW[(idx / C) * (C + 1) + idx % C] = wv[u];
  }
}
// the per-block partials part[i][which][j], i = rq, rq + G, ... (<= kMaxPart of them), as one batch of loads
constexpr int kSeMaxBlocks = 64;
template <int C>
__device__ __forceinline__ void se_load_part(const float* __restrict__ part, int nblk, int which, int j, int rq,
                                             float (&v)[kSeMaxBlocks / (256 / C)]) {
  constexpr int G = 256 / C;
#pragma unroll
  for (int t = 0; t < kSeMaxBlocks / G; ++t) {
    const int i = rq + t * G;
    v[t] = part[((int64_t)(i < nblk ? i : 0) * 2 + which) * C + j];
  }
}
}  // namespace

// ---- forward 1: s0 = relu(gap W0 + b0) for the block's rows; part[blk] = {column mean, column M2} of those rows ------
template <int C, int ROWS>
__global__ void __launch_bounds__(256) k_se_fwd1(const float* __restrict__ gap, const float* __restrict__ W0,
                                                 const float* __restrict__ b0, float* __restrict__ s0,
                                                 float* __restrict__ part, int B, int RB) {
  constexpr int G = 256 / C, RPT = ROWS / G;
  __shared__ __attribute__((aligned(16))) float sA[ROWS * C];
  __shared__ float red[G][C];
  __shared__ float smean[C];
  const int j = threadIdx.x % C, rq = threadIdx.x / C;
  const int r0 = blockIdx.x * RB, nrows = min(RB, B - r0);
  float wcol[C];
  se_load_wcol<C>(W0, j, wcol);
  const float bias = b0[j];
  for (int idx = threadIdx.x; idx < ROWS * C / 4; idx += 256) {
    const int r = idx / (C / 4);
    float4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < nrows) v = reinterpret_cast<const float4*>(gap + (int64_t)r0 * C)[idx];
    reinterpret_cast<float4*>(sA)[idx] = v;
  }
  __syncthreads();
  float acc[RPT];
  se_fc_rows<C, RPT>(sA, wcol, bias, rq, acc);
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = rq + q * G;
    acc[q] = fmaxf(acc[q], 0.f);
    if (r < nrows) {
      s0[(int64_t)(r0 + r) * C + j] = acc[q];
      sum += acc[q];
    }
  }
  red[rq][j] = sum;
  __syncthreads();
  if (rq == 0) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) t += red[g][j];
    smean[j] = t / (float)nrows;
  }
  __syncthreads();
  const float mu = smean[j];
  float m2 = 0.f;
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = rq + q * G;
    const float d = acc[q] - mu;
    if (r < nrows) m2 += d * d;
  }
  __syncthreads();
  red[rq][j] = m2;
  __syncthreads();
  if (rq == 0) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) t += red[g][j];
    part[((int64_t)blockIdx.x * 2 + 0) * C + j] = mu;
    part[((int64_t)blockIdx.x * 2 + 1) * C + j] = t;
  }
}

// ---- forward 2: merge the per-block statistics, BatchNorm, Dense + hard_sigmoid for the block's rows -------------------
template <int C, int ROWS>
__global__ void __launch_bounds__(256) k_se_fwd2(const float* __restrict__ s0, const float* __restrict__ part, int nblk,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 const float* __restrict__ mov_mean, const float* __restrict__ mov_var,
                                                 const float* __restrict__ W1, const float* __restrict__ b1,
                                                 float* __restrict__ xhat, float* __restrict__ invstd,
                                                 float* __restrict__ ulin, float* __restrict__ gout,
                                                 float* __restrict__ stat_mean, float* __restrict__ stat_var, int B,
                                                 int RB, float eps, int training) {
  constexpr int G = 256 / C, RPT = ROWS / G;
  __shared__ __attribute__((aligned(16))) float sA[ROWS * C];
  __shared__ float red[G][C];
  __shared__ float smean[C], sinv[C];
  const int j = threadIdx.x % C, rq = threadIdx.x / C;
  const int r0 = blockIdx.x * RB, nrows = min(RB, B - r0);
  // all global loads first: weight column, this block's rows of s0, the per-block statistics
  float wcol[C];
  se_load_wcol<C>(W1, j, wcol);
  const float bias = b1[j], gm = gamma[j], bt = beta[j];
  float xs[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = rq + q * G;
    xs[q] = s0[(int64_t)(r0 + (r < nrows ? r : 0)) * C + j];
  }
  if (training) {
    // mean = sum n_i mean_i / B ; M2 = sum (M2_i + n_i (mean_i - mean)^2)   (pairwise merge of per-block statistics)
    float pm[kSeMaxBlocks / G], p2[kSeMaxBlocks / G];
    se_load_part<C>(part, nblk, 0, j, rq, pm);
    se_load_part<C>(part, nblk, 1, j, rq, p2);
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < kSeMaxBlocks / G; ++u) {
      const int i = rq + u * G;
      if (i < nblk) t += (float)min(RB, B - i * RB) * pm[u];
    }
    red[rq][j] = t;
    __syncthreads();
    if (rq == 0) {
      float tt = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) tt += red[g][j];
      smean[j] = tt / (float)B;
    }
    __syncthreads();
    const float mu = smean[j];
    t = 0.f;
#pragma unroll
    for (int u = 0; u < kSeMaxBlocks / G; ++u) {
      const int i = rq + u * G;
      const float d = pm[u] - mu;
      if (i < nblk) t += p2[u] + (float)min(RB, B - i * RB) * d * d;
    }
    __syncthreads();
    red[rq][j] = t;
    __syncthreads();
    if (rq == 0) {
      float tt = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) tt += red[g][j];
      const float var = tt / (float)B;
      sinv[j] = rsqrtf(var + eps);
      if (blockIdx.x == 0) { stat_mean[j] = mu; stat_var[j] = var; }
    }
  } else if (rq == 0) {
    smean[j] = mov_mean[j];
    sinv[j] = rsqrtf(mov_var[j] + eps);
  }
  __syncthreads();
  const float mu = smean[j], inv = sinv[j];
  if (blockIdx.x == 0 && rq == 0) invstd[j] = inv;
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = rq + q * G;
    float s1 = 0.f;
    if (r < nrows) {
      const float xh = (xs[q] - mu) * inv;
      xhat[(int64_t)(r0 + r) * C + j] = xh;
      s1 = xh * gm + bt;
    }
    sA[r * C + j] = s1;
  }
  __syncthreads();
  float acc[RPT];
  se_fc_rows<C, RPT>(sA, wcol, bias, rq, acc);
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = rq + q * G;
    if (r < nrows) {
      ulin[(int64_t)(r0 + r) * C + j] = acc[q];
      gout[(int64_t)(r0 + r) * C + j] = se_hsig(acc[q]);
    }
  }
}

// Shared tail of the two backward kernels for the block's rows, with the row tiles sV = gradient at the Dense output
// [rows][C] and sX = the Dense input [rows][C] in LDS and W staged as sW[k][j] (row pitch C+1: conflict-free for lanes
// along k):   dx[r][k] = sum_j sV[r][j] W[k][j]   (returned in registers for the caller's rows rq + q*G, column k = j)
//             dW[k][j] += sum_r sX[r][k] sV[r][j] ;  db[j] += sum_r sV[r][j]          (one atomic set per block)
template <int C, int RPT>
__device__ __forceinline__ void se_dense_bwd(const float* sV, const float* sX, const float* sW, int nrows, int j, int rq,
                                             float (&dx)[RPT], float* __restrict__ dW, float* __restrict__ db) {
  constexpr int G = 256 / C;
#pragma unroll
  for (int q = 0; q < RPT; ++q) dx[q] = 0.f;
#pragma unroll 4
  for (int n0 = 0; n0 < C; n0 += 4) {
    float w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = sW[j * (C + 1) + n0 + e];          // W[k = j][n0 + e]
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(&sV[(rq + q * G) * C + n0]);
      dx[q] += v.x * w[0] + v.y * w[1] + v.z * w[2] + v.w * w[3];
    }
  }
  // weight gradient: thread (j, kq = rq) owns dW[k = kq + G*q][j], q < C/G
  float gw[C / G];
#pragma unroll
  for (int q = 0; q < C / G; ++q) gw[q] = 0.f;
  float gb = 0.f;
  for (int r = 0; r < nrows; ++r) {
    const float v = sV[r * C + j];
    gb += v;
#pragma unroll
    for (int q = 0; q < C / G; ++q) gw[q] += sX[r * C + rq + G * q] * v;
  }
#pragma unroll
  for (int q = 0; q < C / G; ++q) atomicAdd(&dW[(rq + G * q) * C + j], gw[q]);
  if (rq == 0) atomicAdd(&db[j], gb);
}

// ---- backward 1: du = dg hsig'(u); dW1, db1; ds1 = du W1^T; per-block column sums of ds1 and ds1 * xhat --------------
template <int C, int ROWS>
__global__ void __launch_bounds__(256) k_se_bwd1(const float* __restrict__ dg, const float* __restrict__ ulin,
                                                 const float* __restrict__ xhat, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, const float* __restrict__ W1,
                                                 float* __restrict__ ds1, float* __restrict__ dW1,
                                                 float* __restrict__ db1, float* __restrict__ part, int B, int RB,
                                                 int nslots, int64_t slot_stride, int dslots, int64_t dstride) {
  constexpr int G = 256 / C, RPT = ROWS / G;
  __shared__ __attribute__((aligned(16))) float sV[ROWS * C];
  __shared__ __attribute__((aligned(16))) float sX[ROWS * C];
  __shared__ float sW[C * (C + 1)];
  __shared__ float red[2][G][C];
  const int j = threadIdx.x % C, rq = threadIdx.x / C;
  const int r0 = blockIdx.x * RB, nrows = min(RB, B - r0);
  const float gm = gamma[j], bt = beta[j];
  float xh[RPT], rdg[RPT], rul[RPT];                  // the block's rows, raw (clamped), one batch of loads
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = rq + q * G;
    const int64_t o = (int64_t)(r0 + (r < nrows ? r : 0)) * C + j;
    // the gate gradient arrives in `dslots` (<= 64) partial copies: eight raw, clamped loads per round trip (one copy per
    // trip made this kernel 72 us on 256x256 feature maps, where dslots = 64)
    float gsum = 0.f;
    for (int k0 = 0; k0 < dslots; k0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = dg[o + (int64_t)min(k0 + u, dslots - 1) * dstride];
#pragma unroll
      for (int u = 0; u < 8; ++u) gsum += k0 + u < dslots ? v[u] : 0.f;
    }
    rdg[q] = gsum; rul[q] = ulin[o]; xh[q] = xhat[o];
  }
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = rq + q * G;
    const bool ok = r < nrows;
    xh[q] = ok ? xh[q] : 0.f;
    sV[r * C + j] = ok ? rdg[q] * se_hsig_grad(rul[q]) : 0.f;
    sX[r * C + j] = ok ? xh[q] * gm + bt : 0.f;
  }
  se_stage_w<C>(W1, sW);
  __syncthreads();
  const int64_t slot = (int64_t)(blockIdx.x % nslots) * slot_stride;
  float dx[RPT];
  se_dense_bwd<C, RPT>(sV, sX, sW, nrows, j, rq, dx, dW1 + slot, db1 + slot);
  float p1 = 0.f, p2 = 0.f;
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = rq + q * G;
    if (r < nrows) {
      ds1[(int64_t)(r0 + r) * C + j] = dx[q];
      p1 += dx[q];
      p2 += dx[q] * xh[q];
    }
  }
  red[0][rq][j] = p1;
  red[1][rq][j] = p2;
  __syncthreads();
  if (rq < 2) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) t += red[rq][g][j];
    part[((int64_t)blockIdx.x * 2 + rq) * C + j] = t;
  }
}

// ---- backward 2: BatchNorm backward with the merged sums, ReLU mask; dW0, db0; dgap = dv W0^T; dgamma, dbeta ------------
template <int C, int ROWS>
__global__ void __launch_bounds__(256) k_se_bwd2(const float* __restrict__ ds1, const float* __restrict__ part, int nblk,
                                                 const float* __restrict__ xhat, const float* __restrict__ invstd,
                                                 const float* __restrict__ gamma, const float* __restrict__ s0,
                                                 const float* __restrict__ gap, const float* __restrict__ W0,
                                                 float* __restrict__ dgap, float* __restrict__ dW0,
                                                 float* __restrict__ db0, float* __restrict__ dgamma,
                                                 float* __restrict__ dbeta, int B, int RB, int nslots,
                                                 int64_t slot_stride) {
  constexpr int G = 256 / C, RPT = ROWS / G;
  __shared__ __attribute__((aligned(16))) float sV[ROWS * C];
  __shared__ __attribute__((aligned(16))) float sX[ROWS * C];
  __shared__ float sW[C * (C + 1)];
  __shared__ float red[2][G][C];
  __shared__ float stot[2][C];
  const int j = threadIdx.x % C, rq = threadIdx.x / C;
  const int r0 = blockIdx.x * RB, nrows = min(RB, B - r0);
  float rds[RPT], rxh[RPT], rs0[RPT], rgp[RPT];                // the block's rows, raw, issued before any barrier
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = rq + q * G;
    const int64_t o = (int64_t)(r0 + (r < nrows ? r : 0)) * C + j;
    rds[q] = ds1[o]; rxh[q] = xhat[o]; rs0[q] = s0[o]; rgp[q] = gap[o];
  }
  float p1 = 0.f, p2 = 0.f;
  {
    float v1[kSeMaxBlocks / G], v2[kSeMaxBlocks / G];
    se_load_part<C>(part, nblk, 0, j, rq, v1);
    se_load_part<C>(part, nblk, 1, j, rq, v2);
#pragma unroll
    for (int u = 0; u < kSeMaxBlocks / G; ++u) {
      if (rq + u * G < nblk) { p1 += v1[u]; p2 += v2[u]; }
    }
  }
  red[0][rq][j] = p1;
  red[1][rq][j] = p2;
  se_stage_w<C>(W0, sW);
  __syncthreads();
  if (rq < 2) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) t += red[rq][g][j];
    stot[rq][j] = t;
    if (blockIdx.x == 0) {
      if (rq == 0) dbeta[j] += t;          // sum ds1
      else dgamma[j] += t;                 // sum ds1 * xhat
    }
  }
  __syncthreads();
  const float inv_b = 1.0f / (float)B;
  const float md = stot[0][j] * inv_b, mdx = stot[1][j] * inv_b, gi = gamma[j] * invstd[j];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = rq + q * G;
    float dv = 0.f, gp = 0.f;
    if (r < nrows) {
      const float v = gi * (rds[q] - md - rxh[q] * mdx);
      dv = rs0[q] > 0.f ? v : 0.f;
      gp = rgp[q];
    }
    sV[r * C + j] = dv;
    sX[r * C + j] = gp;
  }
  __syncthreads();
  const int64_t slot = (int64_t)(blockIdx.x % nslots) * slot_stride;
  float dx[RPT];
  se_dense_bwd<C, RPT>(sV, sX, sW, nrows, j, rq, dx, dW0 + slot, db0 + slot);
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int r = rq + q * G;
    if (r < nrows) dgap[(int64_t)(r0 + r) * C + j] = dx[q];
  }
}

// rows per block: 16 keeps each block's Dense work at a few hundred FMAs per thread; at most 64 blocks to merge
static inline int se_rows_per_block(int B) { return (B + 15) / 16 <= 64 ? 16 : kSeRowsMax; }
int se_max_blocks(int max_batch) {
  const int rb = se_rows_per_block(max_batch);
  return (max_batch + rb - 1) / rb;
}

#define MVAE_SE_DISPATCH(KERNEL, ...)                                                                       \
  do {                                                                                                      \
    if (C == 64 && RB == 16) hipLaunchKernelGGL((KERNEL<64, 16>), dim3(nblk), dim3(256), 0, s, __VA_ARGS__);  \
    else if (C == 64) hipLaunchKernelGGL((KERNEL<64, 64>), dim3(nblk), dim3(256), 0, s, __VA_ARGS__);         \
    else if (RB == 16) hipLaunchKernelGGL((KERNEL<32, 16>), dim3(nblk), dim3(256), 0, s, __VA_ARGS__);        \
    else hipLaunchKernelGGL((KERNEL<32, 64>), dim3(nblk), dim3(256), 0, s, __VA_ARGS__);                      \
  } while (0)

// false = shape not covered (channels other than 32 / 64, or more rows than 64 blocks of 64)
bool launch_se_forward(const float* gap, const float* W0, const float* b0, const float* gamma, const float* beta,
                       const float* mov_mean, const float* mov_var, const float* W1, const float* b1, float* s0,
                       float* xhat, float* invstd, float* ulin, float* g, float* stat_mean, float* stat_var,
                       float* part, int B, int C, float eps, int training, hipStream_t s) {
  if ((C != 32 && C != 64) || B > 64 * kSeRowsMax) return false;
  const int RB = se_rows_per_block(B), nblk = (B + RB - 1) / RB;
  ProfScope ps("se_fwd", 4.0 * (5.0 * B * C + 2.0 * C * C), 4.0 * B * C * C, s);
  MVAE_SE_DISPATCH(k_se_fwd1, gap, W0, b0, s0, part, B, RB);
  MVAE_SE_DISPATCH(k_se_fwd2, s0, part, nblk, gamma, beta, mov_mean, mov_var, W1, b1, xhat, invstd, ulin, g, stat_mean,
                   stat_var, B, RB, eps, training);
  return true;
}

bool launch_se_backward(const float* dg, const float* ulin, const float* xhat, const float* invstd, const float* gamma,
                        const float* beta, const float* s0, const float* gap, const float* W1, const float* W0,
                        float* ds1, float* dgap, float* dW1, float* db1, float* dgamma, float* dbeta, float* dW0,
                        float* db0, float* part, int B, int C, GradSlots sl, int dslots, int64_t dstride,
                        hipStream_t s) {
  if ((C != 32 && C != 64) || B > 64 * kSeRowsMax) return false;
  const int RB = se_rows_per_block(B), nblk = (B + RB - 1) / RB;
  ProfScope ps("se_bwd", 4.0 * (8.0 * B * C + 2.0 * C * C), 8.0 * B * C * C, s);
  MVAE_SE_DISPATCH(k_se_bwd1, dg, ulin, xhat, gamma, beta, W1, ds1, sl.at(dW1), sl.at(db1), part, B, RB, sl.count(),
                   sl.stride, dslots < 1 ? 1 : dslots, dstride);
  MVAE_SE_DISPATCH(k_se_bwd2, ds1, part, nblk, xhat, invstd, gamma, s0, gap, W0, dgap, sl.at(dW0), sl.at(db0), dgamma,
                   dbeta, B, RB, sl.count(), sl.stride);
  return true;
}
#undef MVAE_SE_DISPATCH

}  // namespace mvae
