// kernels_dense.hip -- the Dense layers around the latent (multiscale_vae.py:358-370 Flatten -> Dense mu / log_var,
// :402-406 decoder Dense): skinny products, K or N = z <= 32 against 512 .. 524288.
//   k_skinny_mfma   out[b, j] += sum_k A[b,k] * Wcol_j[k]     split-K over waves on v_mfma_f32_32x32x2_f32;
//                   mu and log_var are produced by ONE launch (their weights are two column blocks of the B operand)
//   k_dense_bwd2    dflat[b,k] = sum_j dmu[b,j] Wmu[k,j] + dlv[b,j] Wlv[k,j]   (one pass, 16 batch rows per block)
//   k_dense_expand  out[b,n]   = bias[n] + sum_k z[b,k] W[k,n]                  (16-byte stores, 4 batch rows per block)
//   k_outer_wide2   dWmu[k,j], dWlv[k,j] += sum_b flat[b,k] * {dmu,dlv}[b,j]    (flat is read once for both)
#include "kernels.h"
#include "act16.h"

namespace mvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave = 32 batch rows x (up to) 32 output columns x KC reduction indices; lane half h owns [h*KC/2, (h+1)*KC/2).
// NT = false: Wx stored [K][Nx] (Dense kernel as in Keras);  NT = true: Wx stored [Nx][K] (its transpose use).
template <bool NT, typename T>
__global__ void __launch_bounds__(256) k_skinny_mfma(const T* __restrict__ A, const float* __restrict__ W1,
                                                     const float* __restrict__ W2, const float* __restrict__ bias1,
                                                     const float* __restrict__ bias2, float* __restrict__ out1,
                                                     float* __restrict__ out2, int B, int K, int N1, int N2, int KC,
                                                     int nchunks) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int rt = (int)(gw / nchunks), kc = (int)(gw % nchunks);
  if (rt * 32 >= B) return;
  const int i = lane & 31, h = lane >> 5;
  const int row = rt * 32 + i;
  const bool rv = row < B;
  const int half = KC / 2;
  const int kbeg = kc * KC + h * half;
  // this lane's B-operand column
  const float* Wc = nullptr;
  int jx = 0, Nx = 0;
  if (i < N1) { Wc = W1; jx = i; Nx = N1; }
  else if (i < N1 + N2) { Wc = W2; jx = i - N1; Nx = N2; }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const V4<T> arow(A + (int64_t)(rv ? row : 0) * K);
  for (int q = 0; q < half; q += 4) {
    const int k = kbeg + q;
    f32x4 a4 = {0.f, 0.f, 0.f, 0.f}, b4 = {0.f, 0.f, 0.f, 0.f};
    if (k < K) {                                           // K % 4 == 0: a float4 is inside or outside as a whole
      if (rv) a4 = arow[k / 4];
      if (Wc) {
        if (NT) b4 = *reinterpret_cast<const f32x4*>(Wc + (int64_t)jx * K + k);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) b4[e] = Wc[(int64_t)(k + e) * Nx + jx];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc, 0, 0, 0);
  }
  if (!Wc) return;
  float* out = (i < N1) ? out1 : out2;
  const float bv = (kc == 0) ? ((i < N1) ? (bias1 ? bias1[jx] : 0.f) : (bias2 ? bias2[jx] : 0.f)) : 0.f;
  // every output element lives in exactly one (lane, register): the k = 0 chunk adds the bias
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int orow = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (orow < B) atomicAdd(&out[(int64_t)orow * Nx + jx], acc[r] + bv);
  }
}

// The same product for K % 256 == 0, built for latency instead of a k loop: a block = 32 batch rows x 256 reduction
// indices, wave w takes 64 of them (lane half h: 32), issues ALL its loads as one batch (8 x 16 B of its A row, the 32
// matching B values), then runs its 32 MFMAs; the 4 waves' tiles are summed through LDS and leave as ONE atomic set per
// block.  The looped kernel above waited out a full memory round trip per 4 MFMAs (B = 512, K = 8192: 32 trips, 38 us).
template <bool NT, typename T>
__global__ void __launch_bounds__(256) k_skinny_mfma256(const T* __restrict__ A, const float* __restrict__ W1,
                                                        const float* __restrict__ W2, const float* __restrict__ bias1,
                                                        const float* __restrict__ bias2, float* __restrict__ out1,
                                                        float* __restrict__ out2, int B, int K, int N1, int N2,
                                                        int ngroups) {
  __shared__ float red[3][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rt = blockIdx.x / ngroups, kg = blockIdx.x % ngroups;
  const int i = lane & 31, h = lane >> 5;
  const int row = rt * 32 + i;
  const int rowc = row < B ? row : B - 1;                  // clamped: the extra rows are never written
  const int k0 = kg * 256 + wave * 64 + h * 32;
  // this lane's B-operand column; lanes past N1 + N2 read column 0 of W1 and are never written either
  const bool two = i >= N1 && i < N1 + N2;
  const float* Wc = two ? W2 : W1;
  const int Nx = two ? N2 : N1;
  const int jx = two ? i - N1 : (i < N1 ? i : 0);
  f32x4 a4[8];
  float bq[32];
  const V4<T> ap(A + (int64_t)rowc * K + k0);
#pragma unroll
  for (int q = 0; q < 8; ++q) a4[q] = ap[q];
  if (NT) {
    const f32x4* bp = reinterpret_cast<const f32x4*>(Wc + (int64_t)jx * K + k0);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 v = bp[q];
      bq[q * 4] = v[0]; bq[q * 4 + 1] = v[1]; bq[q * 4 + 2] = v[2]; bq[q * 4 + 3] = v[3];
    }
  } else {
    const float* bp = Wc + (int64_t)k0 * Nx + jx;
#pragma unroll
    for (int e = 0; e < 32; ++e) bq[e] = bp[(int64_t)e * Nx];
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q][e], bq[q * 4 + e], acc, 0, 0, 0);
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave != 0 || i >= N1 + N2) return;
  float* out = two ? out2 : out1;
  const float* bias = two ? bias2 : bias1;
  const float bv = (kg == 0 && bias) ? bias[jx] : 0.f;     // every output element: one (lane, register); chunk 0 adds the bias
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int orow = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    const float v = acc[r] + red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + bv;
    if (orow < B) atomicAdd(&out[(int64_t)orow * Nx + jx], v);
  }
}

// RB batch rows per block: the two [K, Z] weight matrices are re-read B / RB times (4 rows per block made that 16 times,
// 1 GB of L2 traffic, for the 524288 x 16 heads of the 256 x 256 configuration: 268 us)
constexpr int kDflatRows = 16;
template <typename T>
__global__ void __launch_bounds__(256) k_dense_bwd2(const float* __restrict__ g1, const float* __restrict__ g2,
                                                    const float* __restrict__ W1, const float* __restrict__ W2,
                                                    T* __restrict__ out, int B, int K, int Z) {
  constexpr int RB = kDflatRows;
  __shared__ float sg[RB][64];
  const int b0 = blockIdx.y * RB;
  for (int t = threadIdx.x; t < RB * 2 * Z; t += 256) {
    const int r = t / (2 * Z), j = t % (2 * Z);
    const int b = b0 + r;
    sg[r][j] = b < B ? (j < Z ? g1[(int64_t)b * Z + j] : g2[(int64_t)b * Z + j - Z]) : 0.f;
  }
  __syncthreads();
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  float acc[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) acc[r] = 0.f;
  const float* w1 = W1 + (int64_t)k * Z;
  const float* w2 = W2 + (int64_t)k * Z;
  if (Z == 16) {                        // the notebook's z: both weight rows as one batch of 16-byte loads
    f32x4 a4[4], c4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { a4[q] = reinterpret_cast<const f32x4*>(w1)[q]; c4[q] = reinterpret_cast<const f32x4*>(w2)[q]; }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] += sg[r][q * 4 + e] * a4[q][e] + sg[r][16 + q * 4 + e] * c4[q][e];
  } else {
    for (int j = 0; j < Z; ++j) {
      const float a = w1[j], c = w2[j];
#pragma unroll
      for (int r = 0; r < RB; ++r) acc[r] += sg[r][j] * a + sg[r][Z + j] * c;
    }
  }
#pragma unroll
  for (int r = 0; r < RB; ++r)
    if (b0 + r < B) st1(out, (int64_t)(b0 + r) * K + k, acc[r]);
}

template <typename T>
__global__ void __launch_bounds__(256) k_dense_expand(const float* __restrict__ z, const f32x4* __restrict__ W,
                                                      const f32x4* __restrict__ bias, const V4<T> out, int B,
                                                      int Z, int N4) {
  __shared__ float sz[4][32];
  const int b0 = blockIdx.y * 4;
  for (int t = threadIdx.x; t < 4 * Z; t += 256) {
    const int r = t / Z, j = t % Z;
    sz[r][j] = (b0 + r < B) ? z[(int64_t)(b0 + r) * Z + j] : 0.f;
  }
  __syncthreads();
  const int n4 = blockIdx.x * 256 + threadIdx.x;
  if (n4 >= N4) return;
  const f32x4 bv = bias[n4];
  f32x4 acc[4] = {bv, bv, bv, bv};
  for (int k = 0; k < Z; ++k) {
    const f32x4 w = W[(int64_t)k * N4 + n4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += w * sz[r][k];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (b0 + r < B) out.st((int64_t)(b0 + r) * N4 + n4, acc[r]);
}

// dW1[w][j] += sum_b wide[b,w] s1[b,j] ; dW2[w][j] += sum_b wide[b,w] s2[b,j] ; db1[j] += sum_b s1 ; db2 likewise
// one thread per wide column; the batch range of the block is walked in LDS-staged chunks of 64 rows.  With
// gridDim.y == 1 a thread owns its outputs: plain stores into the zeroed arena, no atomics (deterministic).
__global__ void __launch_bounds__(256) k_outer_wide2(const float* __restrict__ wide, const float* __restrict__ s1,
                                                     const float* __restrict__ s2, float* __restrict__ dW1,
                                                     float* __restrict__ dW2, float* __restrict__ db1,
                                                     float* __restrict__ db2, int B, int Wd, int Z, int bpc) {
  __shared__ float ssm[64 * 32];
  const int b0 = blockIdx.y * bpc, b1 = min(B, b0 + bpc);
  const int w = blockIdx.x * 256 + threadIdx.x;
  float acc[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) acc[j] = 0.f;
  float bsum = 0.f;
  for (int c0 = b0; c0 < b1; c0 += 64) {
    const int c1 = min(b1, c0 + 64);
    __syncthreads();
    for (int t = threadIdx.x; t < (c1 - c0) * 2 * Z; t += 256) {
      const int r = t / (2 * Z), j = t % (2 * Z);
      ssm[t] = j < Z ? s1[(int64_t)(c0 + r) * Z + j] : s2[(int64_t)(c0 + r) * Z + j - Z];
    }
    __syncthreads();
    if (w < Wd) {
      for (int b = c0; b < c1; ++b) {
        const float v = wide[(int64_t)b * Wd + w];
        const float* sp = ssm + (b - c0) * 2 * Z;
#pragma unroll
        for (int j = 0; j < 32; ++j)
          if (j < 2 * Z) acc[j] += v * sp[j];
      }
    }
    if (blockIdx.x == 0 && threadIdx.x < 2 * Z)
      for (int b = 0; b < c1 - c0; ++b) bsum += ssm[b * 2 * Z + threadIdx.x];
  }
  if (w < Wd) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      if (j < 2 * Z) {
        float* dst = j < Z ? &dW1[(int64_t)w * Z + j] : &dW2[(int64_t)w * Z + j - Z];
        if (gridDim.y == 1) *dst += acc[j];
        else atomicAdd(dst, acc[j]);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 2 * Z)
    atomicAdd(threadIdx.x < Z ? &db1[threadIdx.x] : &db2[threadIdx.x - Z], bsum);
}

// MFMA form of the same gradient: a wave owns K-slice [k0, k0+32) and all 2Z <= 32 columns; D[i = k][j] accumulates
// over the batch two rows per instruction (lane half h takes row b + h).  Operands are channel-on-lane, so the
// loads are 128-byte row segments; the wave owns its outputs: plain stores, deterministic.
template <typename T>
__global__ void __launch_bounds__(256) k_dense_wgrad_mfma(const T* __restrict__ flat, const float* __restrict__ g1,
                                                          const float* __restrict__ g2, float* __restrict__ dW1,
                                                          float* __restrict__ dW2, float* __restrict__ db1,
                                                          float* __restrict__ db2, int B, int K, int Z) {
  // block = one K-slice of 32; its 4 waves split the batch, partial tiles are summed through LDS
  __shared__ float red[4][16][64];
  __shared__ float redb[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k0 = blockIdx.x * 32;
  const int i = lane & 31, h = lane >> 5;
  const bool kv = k0 + i < K;
  const float* gsrc = i < Z ? g1 : g2;
  const int jx = i < Z ? i : i - Z;
  const bool jv = i < 2 * Z;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  const int per = ((B + 7) / 8) * 2;                 // rows per wave, even
  const int bb = wave * per, be = min(B, bb + per);
  // unconditional (clamped) loads, masked by multiplication: the compiler can then issue a whole unrolled batch of
  // loads ahead of its MFMAs instead of one exec-masked load -> wait -> MFMA chain per row pair
  const T* pa = flat + (kv ? k0 + i : 0);
  const float* pg = gsrc + (jv ? jx : 0);
  const float ma = kv ? 1.f : 0.f, mg = jv ? 1.f : 0.f;
  for (int b = bb; b < be; b += 16) {                // 8 row pairs per trip, all 16 loads issued before the MFMAs
    float a[8], g[8], rm[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = b + 2 * u + h;
      rm[u] = row < be ? 1.f : 0.f;
      const int rc = row < be ? row : bb;
      a[u] = ld1(pa, (int64_t)rc * K);
      g[u] = pg[(int64_t)rc * Z];
    }
    __builtin_amdgcn_sched_barrier(0);               // raw loads above, every use below
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float gv = g[u] * (mg * rm[u]);
      bsum += gv;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u] * (ma * rm[u]), gv, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  bsum += __shfl_xor(bsum, 32, 64);
  if (h == 0) redb[wave][i] = bsum;
  __syncthreads();
  if (wave == 0 && jv) {
    float* dW = i < Z ? dW1 : dW2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float t = red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane];
      if (k < K) dW[(int64_t)k * Z + jx] += t;
    }
    if (blockIdx.x == 0 && h == 0) (i < Z ? db1 : db2)[jx] += redb[0][i] + redb[1][i] + redb[2][i] + redb[3][i];
  }
}

// decoder Dense weight gradient on the MFMA: dW[k][n] += sum_b z[b,k] dy[b,n] (k < Z <= 32), db[n] += sum_b dy[b,n].
// block = one 32-wide slice of N; its 4 waves split the batch; D[i = k][j = n]; plain stores (the block owns its slice).
template <typename T>
__global__ void __launch_bounds__(256) k_dense_wgrad_mfma_t(const float* __restrict__ z, const T* __restrict__ dy,
                                                            float* __restrict__ dW, float* __restrict__ db, int B,
                                                            int Z, int N) {
  __shared__ float red[4][16][64];
  __shared__ float redb[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 32;
  const int i = lane & 31, h = lane >> 5;
  const bool kv = i < Z, nv = n0 + i < N;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  const int per = ((B + 7) / 8) * 2;
  const int bb = wave * per, be = min(B, bb + per);
  const float* pa = z + (kv ? i : 0);
  const T* pg = dy + (nv ? n0 + i : 0);
  const float ma = kv ? 1.f : 0.f, mg = nv ? 1.f : 0.f;
  for (int b = bb; b < be; b += 16) {                // clamped loads + multiplicative masks: see k_dense_wgrad_mfma
    float a[8], g[8], rm[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = b + 2 * u + h;
      rm[u] = row < be ? 1.f : 0.f;
      const int rc = row < be ? row : bb;
      a[u] = pa[(int64_t)rc * Z];
      g[u] = ld1(pg, (int64_t)rc * N);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float gv = g[u] * (mg * rm[u]);
      bsum += gv;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u] * (ma * rm[u]), gv, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  bsum += __shfl_xor(bsum, 32, 64);
  if (h == 0) redb[wave][i] = bsum;
  __syncthreads();
  if (wave == 0 && nv) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
      if (k < Z) dW[(int64_t)k * N + n0 + i] += red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane];
    }
    if (h == 0) db[n0 + i] += redb[0][i] + redb[1][i] + redb[2][i] + redb[3][i];
  }
}

// ---- launchers; false = shape not covered ---------------------------------------------------------------------
template <typename T>
static void run_skinny(bool nt, const T* A, const float* W1, const float* W2, const float* b1, const float* b2,
                       float* o1, float* o2, int B, int K, int N1, int N2, hipStream_t s) {
  launch_zero(o1, (int64_t)B * N1, s);
  if (N2) launch_zero(o2, (int64_t)B * N2, s);
  if (K % 256 == 0 && N1 >= 1 && !det_mode()) {
    const int ngroups = K / 256;
    dim3 grid((unsigned)(((B + 31) / 32) * ngroups));
    if (nt) hipLaunchKernelGGL((k_skinny_mfma256<true, T>), grid, dim3(256), 0, s, A, W1, W2, b1, b2, o1, o2, B, K, N1, N2, ngroups);
    else hipLaunchKernelGGL((k_skinny_mfma256<false, T>), grid, dim3(256), 0, s, A, W1, W2, b1, b2, o1, o2, B, K, N1, N2, ngroups);
    return;
  }
  int KC = 256;
  while (KC > 8 && (int64_t)((B + 31) / 32) * ((K + KC - 1) / KC) < 512 && KC > 32) KC /= 2;   // >= ~512 waves
  if (det_mode()) KC = (K + 7) / 8 * 8;              // deterministic mode: no split-K, every output gets one add onto zero
  const int nchunks = (K + KC - 1) / KC;
  const int64_t waves = (int64_t)((B + 31) / 32) * nchunks;
  dim3 grid((unsigned)((waves + 3) / 4));
  if (nt) hipLaunchKernelGGL((k_skinny_mfma<true, T>), grid, dim3(256), 0, s, A, W1, W2, b1, b2, o1, o2, B, K, N1, N2, KC, nchunks);
  else hipLaunchKernelGGL((k_skinny_mfma<false, T>), grid, dim3(256), 0, s, A, W1, W2, b1, b2, o1, o2, B, K, N1, N2, KC, nchunks);
}

// mu = x Wmu + bmu ; lv = x Wlv + blv   (x [B,K], W [K,Z])
bool launch_dense_mu_lv(const float* x, const float* Wmu, const float* bmu, const float* Wlv, const float* blv,
                        float* mu, float* lv, int B, int K, int Z, hipStream_t s, bool bf) {
  if (2 * Z > 32 || K < 256 || (K % 4)) return false;
  if (bf) run_skinny<bf16_t>(false, (const bf16_t*)x, Wmu, Wlv, bmu, blv, mu, lv, B, K, Z, Z, s);
  else run_skinny<float>(false, x, Wmu, Wlv, bmu, blv, mu, lv, B, K, Z, Z, s);
  return true;
}
// dz[b,k] = sum_n dy[b,n] W[k,n]   (W [Z,N])
bool launch_dense_dz(const float* dy, const float* W, float* dz, int B, int Z, int N, hipStream_t s, bool bf) {
  if (Z > 32 || N < 256 || (N % 4)) return false;
  if (bf) run_skinny<bf16_t>(true, (const bf16_t*)dy, W, nullptr, nullptr, nullptr, dz, nullptr, B, N, Z, 0, s);
  else run_skinny<float>(true, dy, W, nullptr, nullptr, nullptr, dz, nullptr, B, N, Z, 0, s);
  return true;
}
bool launch_dense_dflat(const float* dmu, const float* dlv, const float* Wmu, const float* Wlv, float* out, int B, int K,
                        int Z, hipStream_t s, bool bf) {
  if (2 * Z > 64) return false;
  const dim3 grid((K + 255) / 256, (B + kDflatRows - 1) / kDflatRows);
  if (bf) hipLaunchKernelGGL(k_dense_bwd2<bf16_t>, grid, dim3(256), 0, s, dmu, dlv, Wmu, Wlv, (bf16_t*)out, B, K, Z);
  else hipLaunchKernelGGL(k_dense_bwd2<float>, grid, dim3(256), 0, s, dmu, dlv, Wmu, Wlv, out, B, K, Z);
  return true;
}
bool launch_dense_expand(const float* z, const float* W, const float* bias, float* out, int B, int Z, int N,
                         hipStream_t s, bool bf) {
  if (Z > 32 || (N % 4)) return false;
  const dim3 grid((N / 4 + 255) / 256, (B + 3) / 4);
  if (bf) hipLaunchKernelGGL(k_dense_expand<bf16_t>, grid, dim3(256), 0, s, z, (const f32x4*)W, (const f32x4*)bias,
                             V4<bf16_t>((bf16_t*)out), B, Z, N / 4);
  else hipLaunchKernelGGL(k_dense_expand<float>, grid, dim3(256), 0, s, z, (const f32x4*)W, (const f32x4*)bias,
                          V4<float>(out), B, Z, N / 4);
  return true;
}
bool launch_dense_wgrad_mu_lv(const float* flat, const float* dmu, const float* dlv, float* dWmu, float* dWlv,
                              float* dbmu, float* dblv, int B, int K, int Z, hipStream_t s, bool bf) {
  if (2 * Z > 32) return false;
  if (K >= 64) {
    if (bf) hipLaunchKernelGGL(k_dense_wgrad_mfma<bf16_t>, dim3((K + 31) / 32), dim3(256), 0, s, (const bf16_t*)flat, dmu, dlv,
                               dWmu, dWlv, dbmu, dblv, B, K, Z);
    else hipLaunchKernelGGL(k_dense_wgrad_mfma<float>, dim3((K + 31) / 32), dim3(256), 0, s, flat, dmu, dlv, dWmu, dWlv, dbmu,
                            dblv, B, K, Z);
    return true;
  }
  if (bf) return false;
  int bpc = B >= 512 ? 64 : (B >= 64 ? 32 : B);
  if (bpc < 1) bpc = 1;
  hipLaunchKernelGGL(k_outer_wide2, dim3((K + 255) / 256, (B + bpc - 1) / bpc), dim3(256), 0, s, flat, dmu, dlv, dWmu,
                     dWlv, dbmu, dblv, B, K, Z, bpc);
  return true;
}

// dW[k][n] += sum_b z[b,k] dy[b,n] ; db[n] += sum_b dy[b,n]     (decoder Dense, multiscale_vae.py:402-406)
bool launch_dense_wgrad_dec(const float* z, const float* dy, float* dW, float* db, int B, int Z, int N, hipStream_t s,
                            bool bf) {
  if (Z > 32 || N < 256) return false;
  if (bf) hipLaunchKernelGGL(k_dense_wgrad_mfma_t<bf16_t>, dim3((N + 31) / 32), dim3(256), 0, s, z, (const bf16_t*)dy, dW, db, B, Z, N);
  else hipLaunchKernelGGL(k_dense_wgrad_mfma_t<float>, dim3((N + 31) / 32), dim3(256), 0, s, z, dy, dW, db, B, Z, N);
  return true;
}

}  // namespace mvae
