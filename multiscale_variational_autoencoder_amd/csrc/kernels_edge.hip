// kernels_edge.hip -- the two "edge" layers of every per-scale VAE, whose channel counts (3 <-> 32) fit neither
// the MFMA tiles nor the generic kernels well, fused so that each makes the minimum number of passes over its
// [B*H*W, 32] tensor:
//   * encoder conv_base: Conv2D(3x3, C->32, SAME) + ELU (multiscale_vae.py:333-341): forward, and a weight
//     gradient that applies ELU' on the fly (its input is data: no backward-data);
//   * decoder head: BatchNormalization(0.999, 1e-4) -> Conv2D(1x1, 32->C) (multiscale_vae.py:420-431): the
//     forward folds BN into the conv read; the backward is ONE reduction pass (BN sums, dW, db) and ONE apply pass.
#include "kernels.h"
#include "act16.h"

namespace mvae {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// -------------------------------------------------------------------------------------------------
// conv_base forward: out[m, co] = elu(b[co] + sum_{tap,ci} in[gather(m,tap), ci] * W[tap][ci][co]),  CO = 32
// block = 32 pixels x 8 lanes; a lane produces 4 output channels (one 16-byte store, 128 B per pixel)
// -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_convbase_fwd(const float* __restrict__ in, const float* __restrict__ W,
                                                      const float* __restrict__ bias, float* __restrict__ out, int B,
                                                      int H, int Wd, int CI) {
  __shared__ __attribute__((aligned(16))) float sW[9 * 4 * 32];
  for (int t = threadIdx.x; t < 9 * CI * 32; t += 256) sW[t] = W[t];
  __syncthreads();
  const int c4 = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const f32x4 b4 = reinterpret_cast<const f32x4*>(bias)[c4];
  const int64_t M = (int64_t)B * H * Wd;
  for (int64_t p = (int64_t)blockIdx.x * 32 + pl; p < M; p += (int64_t)gridDim.x * 32) {
    const int x = (int)(p % Wd);
    const int64_t q = p / Wd;
    const int y = (int)(q % H);
    const int64_t b = q / H;
    f32x4 acc = b4;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int yy = y + a - 1;
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int xx = x + e - 1;
        if (xx < 0 || xx >= Wd) continue;
        const float* ip = in + ((b * H + yy) * Wd + xx) * CI;
        const float* wp = sW + (a * 3 + e) * CI * 32 + c4 * 4;
        for (int ci = 0; ci < CI; ++ci) acc += ip[ci] * *reinterpret_cast<const f32x4*>(wp + ci * 32);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = acc[k] > 0.f ? acc[k] : expm1f(acc[k]);
    reinterpret_cast<f32x4*>(out)[p * 8 + c4] = acc;
  }
}

// conv_base weight gradient with ELU' fused: dpre = dy * (y > 0 ? 1 : y + 1);
//   dW[tap][ci][co] += sum_m in[gather(m,tap), ci] * dpre[m, co] ;  db[co] += sum_m dpre[m, co]
// block = 32 output channels x 8 pixel lanes; 9*CI + 1 accumulators per thread (CI <= 4)
__global__ void __launch_bounds__(256) k_convbase_wgrad(const float* __restrict__ in, const float* __restrict__ dy,
                                                        const float* __restrict__ y, float* __restrict__ dW,
                                                        float* __restrict__ db, int B, int H, int Wd, int CI,
                                                        int64_t ppb) {
  __shared__ float red[37 * 256];
  const int co = threadIdx.x & 31, pl = threadIdx.x >> 5;
  float acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.f;
  float accb = 0.f;
  const int64_t M = (int64_t)B * H * Wd;
  const int64_t p0 = (int64_t)blockIdx.x * ppb;
  int64_t p1 = p0 + ppb;
  if (p1 > M) p1 = M;
  for (int64_t p = p0 + pl; p < p1; p += 8) {
    const float yv = y[p * 32 + co];
    const float d = dy[p * 32 + co] * (yv > 0.f ? 1.0f : yv + 1.0f);
    accb += d;
    const int x = (int)(p % Wd);
    const int64_t q = p / Wd;
    const int yy0 = (int)(q % H);
    const int64_t b = q / H;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int yy = yy0 + a - 1;
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int xx = x + e - 1;
        if (xx < 0 || xx >= Wd) continue;
        const float* ip = in + ((b * H + yy) * Wd + xx) * CI;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
          if (ci < CI) acc[(a * 3 + e) * 4 + ci] += ip[ci] * d;
      }
    }
  }
  // reduce the 8 pixel lanes through LDS (static register indices only: no scratch)
#pragma unroll
  for (int k = 0; k < 36; ++k) red[k * 256 + threadIdx.x] = acc[k];
  red[36 * 256 + threadIdx.x] = accb;
  __syncthreads();
  if (pl == 0) {
#pragma unroll
    for (int k = 0; k < 37; ++k) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) t += red[k * 256 + r * 32 + co];
      if (k < 36) {
        const int tap = k >> 2, ci = k & 3;
        if (ci < CI) atomicAdd(&dW[(tap * CI + ci) * 32 + co], t);
      } else {
        atomicAdd(&db[co], t);
      }
    }
  }
}

typedef float f32x16e __attribute__((ext_vector_type(16)));
// exp(v) - 1 for v <= 0 to ~1e-7 ABSOLUTE error (fp32 resolution of the O(1) activations it feeds): cubic near zero
// (truncation < v^4/24 = 3.4e-8 at -0.03), hardware exp below.  libm's expm1f is ~40 instructions per element and made
// this 3 -> 32 channel kernel VALU-bound (16 calls per lane per 32-pixel tile).
__device__ __forceinline__ float elu_neg(float v) {
  return v > -0.03f ? v * (1.0f + v * (0.5f + v * 0.16666667f)) : __expf(v) - 1.0f;
}
// MFMA form of the conv_base forward for 9*CI <= 32: out[32 pixels][32 co] = patch[32 pixels][9*CI] . W[9*CI][32],
// two patch elements per v_mfma_f32_32x32x2_f32 (lane half h takes element 2s + h).  A wave owns 32 consecutive
// pixels; W (one column per lane) stays in registers; the patch loads are raw + clamped, issued ahead of the MFMAs;
// D leaves straight from the accumulator layout (each store instruction = two whole 128-byte pixels) after bias + ELU.
template <int CI, typename T>
__global__ void __launch_bounds__(256) k_convbase_fwd_mfma(const float* __restrict__ in, const float* __restrict__ W,
                                                           const float* __restrict__ bias, T* __restrict__ out,
                                                           int B, int H, int Wd, int ntiles) {
  constexpr int KP = 9 * CI, NS = (KP + 1) / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  float wreg[NS];
  int dy_[NS], dx_[NS], dc_[NS];
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    const int k = 2 * st + h;
    const bool kv = k < KP;
    const int tap = kv ? k / CI : 0;
    dy_[st] = kv ? tap / 3 - 1 : -100000;                 // an invalid k never passes the bounds test
    dx_[st] = tap % 3 - 1;
    dc_[st] = kv ? k % CI : 0;
    wreg[st] = kv ? W[k * 32 + i] : 0.f;
  }
  const float bz = bias[i];
  const uint32_t M = (uint32_t)B * H * Wd;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    const uint32_t p = (uint32_t)tile * 32 + i;
    const bool pv = p < M;
    const uint32_t pc = pv ? p : 0;
    const uint32_t q = pc / (uint32_t)Wd;
    const int x = (int)(pc - q * (uint32_t)Wd);
    const uint32_t b = q / (uint32_t)H;
    const int y = (int)(q - b * (uint32_t)H);
    float av[NS], am[NS];
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const int yy = y + dy_[st], xx = x + dx_[st];
      const bool ok = pv && yy >= 0 && yy < H && xx >= 0 && xx < Wd;
      const int64_t src = ok ? (((int64_t)b * H + yy) * Wd + xx) * CI + dc_[st] : 0;
      av[st] = in[src];
      am[st] = ok ? 1.f : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x16e acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int st = 0; st < NS; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st] * am[st], wreg[st], acc, 0, 0, 0);
    const int64_t left = (int64_t)M - (int64_t)tile * 32 - 4 * h;
    const int lim = left < 32 ? (int)left : 32;
    T* po = out + ((int64_t)tile * 32 + 4 * h) * 32 + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rc = (r & 3) + 8 * (r >> 2);
      float v = acc[r] + bz;
      v = v > 0.f ? v : elu_neg(v);
      if (rc < lim) st1(po, rc * 32, v);
    }
  }
}

// MFMA form of the conv_base weight gradient for 9*CI <= 32 (C <= 3): the [27 x 32] gradient is ONE 32x32 tile,
// D[i = patch element k][j = co] += patch[m][k] * dpre[m][co], two pixels m per v_mfma_f32_32x32x2_f32.
// A wave walks whole image rows; lane i decodes its patch element (a, e, ci) once.  4 waves reduce through LDS.
template <typename T>
__global__ void __launch_bounds__(256) k_convbase_wgrad_mfma(const float* __restrict__ in, const T* __restrict__ dy,
                                                             const T* __restrict__ y, float* __restrict__ dW,
                                                             float* __restrict__ db, int B, int H, int Wd, int CI,
                                                             int rows_per_wave, int nslots, int64_t slot_stride) {
  __shared__ float red[4][16][64];
  __shared__ float redb[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int KP = 9 * CI;
  const bool kv = i < KP;
  const int tap = kv ? i / CI : 0, ci = kv ? i % CI : 0, a = tap / 3, e = tap % 3;
  f32x16e acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  const int64_t nrows = (int64_t)B * H;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * rows_per_wave;
  for (int64_t rr = r0; rr < r0 + rows_per_wave && rr < nrows; ++rr) {
    const int yy0 = (int)(rr % H);
    const int64_t b = rr / H;
    const int yy = yy0 + a - 1;
    const bool yok = kv && yy >= 0 && yy < H;
    const float* irow = in + ((b * H + (yok ? yy : 0)) * Wd) * CI + ci;
    const T* drow = dy + rr * Wd * 32 + i;
    const T* yrow = y + rr * Wd * 32 + i;
    for (int x = 0; x < Wd; x += 8) {             // 4 pixel pairs per trip: 12 raw (clamped) loads, then the MFMAs
      float av[4], dv[4], yv[4], am[4], dm[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int xm = x + 2 * u + h;               // this lane half's pixel
        const bool mv = xm < Wd;
        const int xx = xm + e - 1;
        const bool aok = mv && yok && xx >= 0 && xx < Wd;
        am[u] = aok ? 1.f : 0.f;
        dm[u] = mv ? 1.f : 0.f;
        av[u] = irow[(aok ? xx : 0) * CI];
        const int xc = mv ? xm : 0;
        yv[u] = ld1(yrow, xc * 32);
        dv[u] = ld1(drow, xc * 32);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d = dv[u] * (yv[u] > 0.f ? 1.0f : yv[u] + 1.0f) * dm[u];
        bsum += d;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u] * am[u], d, acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  bsum += __shfl_xor(bsum, 32, 64);
  if (h == 0) redb[wave][i] = bsum;
  __syncthreads();
  const int64_t gslot = (int64_t)(blockIdx.x % nslots) * slot_stride;     // gradient slot (kernels.h: GradSlots)
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = (r & 3) + 8 * (r >> 2) + 4 * h;          // patch element (row of D), column = co = i
      if (k < KP) atomicAdd(&dW[gslot + k * 32 + i], red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane]);
    }
    if (h == 0) atomicAdd(&db[gslot + i], redb[0][i] + redb[1][i] + redb[2][i] + redb[3][i]);
  }
}

// -------------------------------------------------------------------------------------------------
// decoder head.  DC4 = dc/4 lanes per pixel (dc = BatchNorm channels, a power of two >= 4), C <= 8 outputs.
// forward: y[m,o] = b[o] + sum_c (x[m,c] * scale[c] + shift[c]) * W[c][o]
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_head_fwd(const V4<T> x, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, const float* __restrict__ W,
                                                  const float* __restrict__ bias, float* __restrict__ yout, int64_t M,
                                                  int dc4, int C) {
  const int c4 = threadIdx.x % dc4, pl = threadIdx.x / dc4, ppb = 256 / dc4;
  const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[c4], sh = reinterpret_cast<const f32x4*>(shift)[c4];
  float w[4][8];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int o = 0; o < 8; ++o) w[e][o] = o < C ? W[(c4 * 4 + e) * C + o] : 0.f;
  for (int64_t p = (int64_t)blockIdx.x * ppb + pl; p < M; p += (int64_t)gridDim.x * ppb) {
    const f32x4 v = x[p * dc4 + c4] * sc + sh;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = v[0] * w[0][o] + v[1] * w[1][o] + v[2] * w[2][o] + v[3] * w[3][o];
    for (int off = 1; off < dc4; off <<= 1) {
#pragma unroll
      for (int o = 0; o < 8; ++o)
        if (o < C) acc[o] += __shfl_xor(acc[o], off, 64);
    }
    if (c4 == 0) {
#pragma unroll
      for (int o = 0; o < 8; ++o)
        if (o < C) yout[p * C + o] = acc[o] + bias[o];
    }
  }
}

// backward pass 1 (reduction): with dxbn[m,c] = sum_o dy[m,o] W[c][o], xhat = (x - mean) * invstd,
//   S1[c] += dxbn ; S2[c] += dxbn * xhat ; dW[c][o] += (x*scale+shift) * dy[m,o] ; db[o] += dy[m,o]
// C (output channels) is a template parameter so every per-output loop is straight-line code; pixels are taken four at
// a time with raw clamped loads ahead of the arithmetic.  The per-thread partial sums are folded over the pixel lanes
// with wave shuffles, over the 4 waves through LDS, and leave as ONE atomic set per block into slot (block % nslots):
// S = [nslots][2][dc] (summed by the apply pass), dW/db through the gradient slots (kernels.h: GradSlots).
constexpr int kHeadSlots = 16;
template <int C, typename T>
__global__ void __launch_bounds__(256) k_head_bwd_reduce(const V4<T> x, const float* __restrict__ dy,
                                                         const float* __restrict__ W, const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, float* __restrict__ S,
                                                         float* __restrict__ dW, float* __restrict__ db, int64_t M,
                                                         int dc4, int64_t ppb, int nslots, int64_t slot_stride, int nhead) {
  constexpr int NQ = 8 + 5 * C;                       // s1[4] s2[4] gw[4][C] gb[C]
  __shared__ float red[4][NQ][64];
  const int c4 = threadIdx.x % dc4, pl = threadIdx.x / dc4, npl = 256 / dc4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[c4], sh = reinterpret_cast<const f32x4*>(shift)[c4];
  const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[c4], is = reinterpret_cast<const f32x4*>(invstd)[c4];
  float w[4][C];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int o = 0; o < C; ++o) w[e][o] = W[(c4 * 4 + e) * C + o];
  float q[NQ];
#pragma unroll
  for (int k = 0; k < NQ; ++k) q[k] = 0.f;
  const int64_t p0 = (int64_t)blockIdx.x * ppb;
  int64_t p1 = p0 + ppb;
  if (p1 > M) p1 = M;
  for (int64_t pb = p0 + pl; pb < p1; pb += 4 * npl) {
    typename V4<T>::raw xr[4];
    f32x4 xv[4];
    float d[4][C], msk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t p = pb + u * npl;
      const bool ok = p < p1;
      const int64_t pc = ok ? p : p0;
      msk[u] = ok ? 1.f : 0.f;
      xr[u] = x.ld(pc * dc4 + c4);
#pragma unroll
      for (int o = 0; o < C; ++o) d[u][o] = dy[pc * C + o];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      xv[u] = V4<T>::cv(xr[u]);
#pragma unroll
      for (int o = 0; o < C; ++o) d[u][o] *= msk[u];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float dx = 0.f;
#pragma unroll
        for (int o = 0; o < C; ++o) dx += d[u][o] * w[e][o];
        const float xh = (xv[u][e] - mu[e]) * is[e], xb = xv[u][e] * sc[e] + sh[e];
        q[e] += dx;
        q[4 + e] += dx * xh;
#pragma unroll
        for (int o = 0; o < C; ++o) q[8 + e * C + o] += xb * d[u][o];
      }
#pragma unroll
      for (int o = 0; o < C; ++o) q[8 + 4 * C + o] += d[u][o];
    }
  }
  // fold the pixel lanes of the wave (lanes with equal lane % dc4), then the 4 waves through LDS
  for (int off = dc4; off < 64; off <<= 1) {
#pragma unroll
    for (int k = 0; k < NQ; ++k) q[k] += __shfl_xor(q[k], off, 64);
  }
#pragma unroll
  for (int k = 0; k < NQ; ++k) red[wave][k][lane] = q[k];
  __syncthreads();
  float* Ss = S + (int64_t)(blockIdx.x % nhead) * 2 * dc4 * 4;
  const int64_t gslot = (int64_t)(blockIdx.x % nslots) * slot_stride;
  float* dWs = dW + gslot;
  float* dbs = db + gslot;
  for (int u = threadIdx.x; u < NQ * dc4; u += 256) {
    const int k = u / dc4, cc = u % dc4;
    if (k >= 8 + 4 * C && cc != 0) continue;          // db: once per pixel, take channel group 0's copy
    const float t = red[0][k][cc] + red[1][k][cc] + red[2][k][cc] + red[3][k][cc];
    if (k < 4) atomicAdd(&Ss[cc * 4 + k], t);
    else if (k < 8) atomicAdd(&Ss[dc4 * 4 + cc * 4 + (k - 4)], t);
    else if (k < 8 + 4 * C) atomicAdd(&dWs[(cc * 4 + (k - 8) / C) * C + (k - 8) % C], t);
    else atomicAdd(&dbs[k - 8 - 4 * C], t);
  }
}

// backward pass 2 (apply): d[m,c] = gamma*invstd * (dxbn - S1/M - xhat * S2/M); block 0 also adds the BatchNorm
// parameter gradients dgamma += S2, dbeta += S1
template <int C, typename T, typename TD>
__global__ void __launch_bounds__(256) k_head_bwd_apply(const V4<T> x, const float* __restrict__ dy,
                                                        const float* __restrict__ W, const float* __restrict__ gamma,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, const float* __restrict__ S,
                                                        const V4<TD> dout, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, int64_t M, int dc4, int nslots) {
  const int c4 = threadIdx.x % dc4, pl = threadIdx.x / dc4, ppb = 256 / dc4;
  const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[c4], is = reinterpret_cast<const f32x4*>(invstd)[c4];
  const f32x4 gm = reinterpret_cast<const f32x4*>(gamma)[c4];
  const float inv_m = 1.0f / (float)M;
  f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < nslots; ++k) {
    t1 += reinterpret_cast<const f32x4*>(S + (int64_t)k * 2 * dc4 * 4)[c4];
    t2 += reinterpret_cast<const f32x4*>(S + (int64_t)k * 2 * dc4 * 4 + dc4 * 4)[c4];
  }
  if (blockIdx.x == 0 && pl == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dgamma[c4 * 4 + e] += t2[e];
      dbeta[c4 * 4 + e] += t1[e];
    }
  }
  const f32x4 m1 = t1 * inv_m, m2 = t2 * inv_m;
  float w[4][C];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int o = 0; o < C; ++o) w[e][o] = W[(c4 * 4 + e) * C + o];
  const int64_t stride = (int64_t)gridDim.x * ppb;
  for (int64_t pb = (int64_t)blockIdx.x * ppb + pl; pb < M; pb += 4 * stride) {
    typename V4<T>::raw xr[4];
    f32x4 xv[4];
    float d[4][C];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t p = pb + u * stride;
      const int64_t pc = p < M ? p : pb;
      xr[u] = x.ld(pc * dc4 + c4);
#pragma unroll
      for (int o = 0; o < C; ++o) d[u][o] = dy[pc * C + o];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      xv[u] = V4<T>::cv(xr[u]);
      f32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float dx = 0.f;
#pragma unroll
        for (int o = 0; o < C; ++o) dx += d[u][o] * w[e][o];
        const float xh = (xv[u][e] - mu[e]) * is[e];
        r[e] = gm[e] * is[e] * (dx - m1[e] - xh * m2[e]);
      }
      const int64_t p = pb + u * stride;
      if (p < M) dout.st(p * dc4 + c4, r);
    }
  }
}

// ---- launchers (false = shape not covered, caller uses the generic kernels) -----------------------------------
static inline int cap_grid(int64_t g) { return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g)); }

template <typename T>
static bool run_convbase_fwd(const float* in, const float* W, const float* bias, T* out, int B, int H, int Wd, int CI,
                             int CO, hipStream_t s) {
  if (CO != 32 || CI > 4) return false;
  int64_t M = (int64_t)B * H * Wd;
  if (M < (1ll << 31) - 64 && (CI == 3 || CI == 1)) {
    const int ntiles = (int)((M + 31) / 32);
    const int grid = cap_grid((ntiles + 3) / 4);
    if (CI == 3) hipLaunchKernelGGL((k_convbase_fwd_mfma<3, T>), dim3(grid), dim3(256), 0, s, in, W, bias, out, B, H, Wd, ntiles);
    else hipLaunchKernelGGL((k_convbase_fwd_mfma<1, T>), dim3(grid), dim3(256), 0, s, in, W, bias, out, B, H, Wd, ntiles);
    return true;
  }
  return false;
}
bool launch_convbase_fwd(const float* in, const float* W, const float* bias, float* out, int B, int H, int Wd, int CI,
                         int CO, hipStream_t s, bool bf) {
  if (bf) return run_convbase_fwd<bf16_t>(in, W, bias, (bf16_t*)out, B, H, Wd, CI, CO, s);
  if (run_convbase_fwd<float>(in, W, bias, out, B, H, Wd, CI, CO, s)) return true;
  if (CO != 32 || CI > 4) return false;
  int64_t M = (int64_t)B * H * Wd;
  hipLaunchKernelGGL(k_convbase_fwd, dim3(cap_grid((M + 31) / 32)), dim3(256), 0, s, in, W, bias, out, B, H, Wd, CI);
  return true;
}
bool launch_convbase_wgrad(const float* in, const float* dy, const float* y, float* dW, float* db, int B, int H, int Wd,
                           int CI, int CO, GradSlots sl, hipStream_t s, bool bf) {
  if (CO != 32 || CI > 4) return false;
  if (9 * CI <= 32) {
    const int64_t nrows = (int64_t)B * H;
    int rpw = 1;
    while (nrows / rpw > 4096) rpw *= 2;            // <= 4096 waves: bounds the float-atomic traffic
    const int64_t waves = (nrows + rpw - 1) / rpw;
    if (bf)
      hipLaunchKernelGGL(k_convbase_wgrad_mfma<bf16_t>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, in,
                         (const bf16_t*)dy, (const bf16_t*)y, sl.at(dW), sl.at(db), B, H, Wd, CI, rpw, sl.count(), sl.stride);
    else
      hipLaunchKernelGGL(k_convbase_wgrad_mfma<float>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, in, dy, y, sl.at(dW),
                         sl.at(db), B, H, Wd, CI, rpw, sl.count(), sl.stride);
    return true;
  }
  if (bf) return false;
  int64_t M = (int64_t)B * H * Wd;
  int64_t ppb = 512;
  while ((M + ppb - 1) / ppb > 1024) ppb *= 2;
  hipLaunchKernelGGL(k_convbase_wgrad, dim3((unsigned)((M + ppb - 1) / ppb)), dim3(256), 0, s, in, dy, y, dW, db, B, H,
                     Wd, CI, ppb);
  return true;
}
static bool head_ok(int dc, int C) {
  if (C > 8 || dc < 4 || dc > 256 || (dc & (dc - 1))) return false;
  return true;
}
bool launch_head_fwd(const float* x, const float* scale, const float* shift, const float* W, const float* bias,
                     float* y, int64_t M, int dc, int C, hipStream_t s, bool bf) {
  if (!head_ok(dc, C)) return false;
  const int dc4 = dc / 4, ppb = 256 / dc4;
  if (bf)
    hipLaunchKernelGGL(k_head_fwd<bf16_t>, dim3(cap_grid((M + ppb - 1) / ppb)), dim3(256), 0, s, V4<bf16_t>((const bf16_t*)x),
                       scale, shift, W, bias, y, M, dc4, C);
  else
    hipLaunchKernelGGL(k_head_fwd<float>, dim3(cap_grid((M + ppb - 1) / ppb)), dim3(256), 0, s, V4<float>(x), scale, shift, W,
                       bias, y, M, dc4, C);
  return true;
}
int head_slots() { return det_mode() ? kDetSlots : kHeadSlots; }
template <int C, typename T, typename TD>
static void run_head_bwd(const T* x, const float* dy, const float* W, const float* gamma, const float* scale,
                         const float* shift, const float* mean, const float* invstd, float* S, float* dW, float* db,
                         float* dgamma, float* dbeta, TD* dout, int64_t M, int dc, GradSlots sl, hipStream_t s) {
  const int dc4 = dc / 4, npl = 256 / dc4;
  int64_t ppb = 16 * npl;                                  // >= 16 pixels per thread
  while ((M + ppb - 1) / ppb > (det_mode() ? kDetSlots : 2048)) ppb *= 2;
  hipLaunchKernelGGL((k_head_bwd_reduce<C, T>), dim3((unsigned)((M + ppb - 1) / ppb)), dim3(256), 0, s, V4<T>(x), dy, W,
                     scale, shift, mean, invstd, S, sl.at(dW), sl.at(db), M, dc4, ppb, sl.count(), sl.stride, head_slots());
  hipLaunchKernelGGL((k_head_bwd_apply<C, T, TD>), dim3(cap_grid((M + 4 * npl - 1) / (4 * npl))), dim3(256), 0, s,
                     V4<T>(x), dy, W, gamma, mean, invstd, S, V4<TD>(dout), dgamma, dbeta, M, dc4, head_slots());
}
template <typename T, typename TD>
static bool run_head_bwd_c(const T* x, const float* dy, const float* W, const float* gamma, const float* scale,
                           const float* shift, const float* mean, const float* invstd, float* S, float* dW, float* db,
                           float* dgamma, float* dbeta, TD* dout, int64_t M, int dc, int C, GradSlots sl, hipStream_t s) {
  if (C == 3) run_head_bwd<3, T, TD>(x, dy, W, gamma, scale, shift, mean, invstd, S, dW, db, dgamma, dbeta, dout, M, dc, sl, s);
  else if (C == 1) run_head_bwd<1, T, TD>(x, dy, W, gamma, scale, shift, mean, invstd, S, dW, db, dgamma, dbeta, dout, M, dc, sl, s);
  else if (C == 4) run_head_bwd<4, T, TD>(x, dy, W, gamma, scale, shift, mean, invstd, S, dW, db, dgamma, dbeta, dout, M, dc, sl, s);
  else return false;
  return true;
}
// S: [head_slots()][2][dc] floats, zeroed by the caller.  Adds dW, db (through the gradient slots), dgamma, dbeta.
bool launch_head_bwd(const float* x, const float* dy, const float* W, const float* gamma, const float* scale,
                     const float* shift, const float* mean, const float* invstd, float* S, float* dW, float* db,
                     float* dgamma, float* dbeta, float* dout, int64_t M, int dc, int C, GradSlots sl, hipStream_t s, bool bf) {
  if (!head_ok(dc, C)) return false;
  // the reduce pass folds S into slot (block % kHeadSlots) and dW/db into gradient slot (block % sl.count())
  // bf: the BatchNorm input x stays float32 (see store_tile_t32 in kernels_bf16.hip), the gradient leaves as bfloat16
  if (bf)
    return run_head_bwd_c<float, bf16_t>(x, dy, W, gamma, scale, shift, mean, invstd, S, dW, db, dgamma, dbeta,
                                         (bf16_t*)dout, M, dc, C, sl, s);
  return run_head_bwd_c<float, float>(x, dy, W, gamma, scale, shift, mean, invstd, S, dW, db, dgamma, dbeta, dout, M, dc, C, sl, s);
}

}  // namespace mvae
