// kernels_generic.hip -- shape-generic gfx950 kernels for every op of the multiscale-VAE train step.
// These are the any-shape path (odd channel counts, tiny pyramid tops); the LDS/MFMA-tiled kernels in
// kernels_mfma.hip take over the wide-channel shapes.  Reference semantics: mvae/multiscale_vae.py and
// mvae/layer_blocks.py (line cites at each kernel).  Wave = 64 lanes; blocks are multiples of 64.
#include "kernels.h"
#include "prof.h"

namespace mvae {

static constexpr int kBlock = 256;
static constexpr int kMaxGrid = 256 * 16;   // 256 CUs x 16 blocks: grid-stride beyond that

static inline int grid_for(int64_t n, int block = kBlock) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > kMaxGrid) g = kMaxGrid;
  return (int)g;
}

#define GRID_STRIDE(i, n)                                                                  \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n);                \
       i += (int64_t)gridDim.x * blockDim.x)
// the same loop with the index type I of a kernel templated on it: uint32_t when the element count fits (the per-element
// div / mod of the index then cost a few instructions instead of ~100 each for int64), int64_t otherwise
#define GRID_STRIDE_T(I, i, n)                                                             \
  for (I i = (I)blockIdx.x * (I)blockDim.x + (I)threadIdx.x; i < (I)(n); i += (I)gridDim.x * (I)blockDim.x)
#define LAUNCH_IDX(n, KERNEL, ...)                                                         \
  do {                                                                                     \
    if ((n) < (int64_t)0x7FFFFFFF) hipLaunchKernelGGL(KERNEL<uint32_t>, __VA_ARGS__);      \
    else hipLaunchKernelGGL(KERNEL<int64_t>, __VA_ARGS__);                                 \
  } while (0)

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == ACT_ELU) return v > 0.f ? v : expm1f(v);
  if (act == ACT_HSIG) return fminf(fmaxf(0.2f * v + 0.5f, 0.f), 1.f);   // keras<=2.x hard_sigmoid
  if (act == ACT_TANH) return tanhf(v);
  return v;
}
__device__ __forceinline__ float hsig_grad(float u) { return (u >= -2.5f && u <= 2.5f) ? 0.2f : 0.f; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block-wide sum; result valid in thread 0.  blockDim.x <= 1024, multiple of 64.
__device__ __forceinline__ float block_sum(float v, float* sh /* >= 16 floats */) {
  v = wave_sum(v);
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG (timed runs generate noise / dropout / epsilon on the device; parity runs
// inject them, because TF's random streams cannot be reproduced -- SURVEY.md 7, hard part 8)
// ------------------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox(uint64_t ctr, uint32_t stream_id, uint64_t seed) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = stream_id, c3 = 0x5EED5EEDu;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u01(uint32_t v) { return ((float)(v >> 8) + 0.5f) * (1.0f / 16777216.0f); }

__global__ void k_rng_normal(float* out, int64_t n, float stddev, const uint64_t* seed_ptr, uint32_t sid) {
  const uint64_t seed = *seed_ptr;
  int64_t n4 = (n + 3) / 4;
  GRID_STRIDE(i, n4) {
    U4 r = philox((uint64_t)i, sid, seed);
    float a0 = sqrtf(-2.f * __logf(u01(r.x))), a1 = sqrtf(-2.f * __logf(u01(r.z)));
    float t0 = 6.2831853071795865f * u01(r.y), t1 = 6.2831853071795865f * u01(r.w);
    float v[4] = {a0 * __cosf(t0), a0 * __sinf(t0), a1 * __cosf(t1), a1 * __sinf(t1)};
    for (int j = 0; j < 4; ++j)
      if (i * 4 + j < n) out[i * 4 + j] = v[j] * stddev;
  }
}
__global__ void k_rng_keepmask(float* out, int64_t n, float p_drop, const uint64_t* seed_ptr, uint32_t sid) {
  const uint64_t seed = *seed_ptr;
  int64_t n4 = (n + 3) / 4;
  GRID_STRIDE(i, n4) {
    U4 r = philox((uint64_t)i, sid, seed);
    uint32_t v[4] = {r.x, r.y, r.z, r.w};
    for (int j = 0; j < 4; ++j)
      if (i * 4 + j < n) out[i * 4 + j] = u01(v[j]) >= p_drop ? 1.f : 0.f;
  }
}
// The random draws of a training step in ONE launch (round 4; three launches + a zero fill sat serially at the head of every
// forward pass): eps (stream 1), input noise (stream 2), dropout keep mask (stream 3) -- element for element what
// k_rng_normal / k_rng_keepmask produce -- and the zeroed metrics vector.
__global__ void k_rng_step(float* eps, int64_t n_eps, float std_eps, float* noise, int64_t n_noise, float* keep, int64_t n_keep,
                           float p_drop, float* zero_buf, int64_t n_zero, const uint64_t* seed_ptr) {
  const uint64_t seed = *seed_ptr;
  const int64_t q0 = (n_eps + 3) / 4, q1 = q0 + (n_noise + 3) / 4, q2 = q1 + (n_keep + 3) / 4, q3 = q2 + (n_zero + 3) / 4;
  GRID_STRIDE(t, q3) {
    if (t < q1) {
      const bool first = t < q0;
      const int64_t i = first ? t : t - q0, n = first ? n_eps : n_noise;
      float* out = first ? eps : noise;
      const float sd = first ? std_eps : 1.0f;
      U4 r = philox((uint64_t)i, first ? 1u : 2u, seed);
      float a0 = sqrtf(-2.f * __logf(u01(r.x))), a1 = sqrtf(-2.f * __logf(u01(r.z)));
      float t0 = 6.2831853071795865f * u01(r.y), t1 = 6.2831853071795865f * u01(r.w);
      float v[4] = {a0 * __cosf(t0), a0 * __sinf(t0), a1 * __cosf(t1), a1 * __sinf(t1)};
      for (int j = 0; j < 4; ++j)
        if (i * 4 + j < n) out[i * 4 + j] = v[j] * sd;
    } else if (t < q2) {
      const int64_t i = t - q1;
      U4 r = philox((uint64_t)i, 3u, seed);
      uint32_t v[4] = {r.x, r.y, r.z, r.w};
      for (int j = 0; j < 4; ++j)
        if (i * 4 + j < n_keep) keep[i * 4 + j] = u01(v[j]) >= p_drop ? 1.f : 0.f;
    } else {
      const int64_t i = t - q2;
      for (int j = 0; j < 4; ++j)
        if (i * 4 + j < n_zero) zero_buf[i * 4 + j] = 0.f;
    }
  }
}
void launch_rng_step(float* eps, int64_t n_eps, float std_eps, float* noise, int64_t n_noise, float* keep, int64_t n_keep,
                     float p_drop, float* zero_buf, int64_t n_zero, const uint64_t* seed, hipStream_t s) {
  ProfScope ps("rng", (double)(4.0 * (n_eps + n_noise + n_keep)), 0.0, s);
  const int64_t q = (n_eps + 3) / 4 + (n_noise + 3) / 4 + (n_keep + 3) / 4 + (n_zero + 3) / 4;
  if (q <= 0) return;
  hipLaunchKernelGGL(k_rng_step, dim3(grid_for(q)), dim3(kBlock), 0, s, eps, n_eps, std_eps, noise, n_noise, keep, n_keep, p_drop,
                     zero_buf, n_zero, seed);
}
void launch_rng_normal(float* out, int64_t n, float stddev, const uint64_t* seed, uint32_t sid, hipStream_t s) {
  ProfScope ps("rng", (double)(4.0*n), 0.0, s);
  if (n <= 0) return;
  hipLaunchKernelGGL(k_rng_normal, dim3(grid_for((n + 3) / 4)), dim3(kBlock), 0, s, out, n, stddev, seed, sid);
}
// zero fill as a KERNEL: hipMemsetAsync nodes inside a captured hipGraph were observed to lose their ordering
// against neighbouring kernel nodes on replay (ROCm 7.2), a kernel node keeps plain kernel->kernel edges.
__global__ void k_zero(float* p, int64_t n) {
  GRID_STRIDE(i, n) p[i] = 0.f;
}
void launch_zero(float* p, int64_t n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_zero, dim3(grid_for(n)), dim3(kBlock), 0, s, p, n);
}
__global__ void k_set_u64(uint64_t* dst, uint64_t v) { *dst = v; }
__global__ void k_seed_next(uint64_t* p) { *p += 1; }
void launch_seed_next(uint64_t* p, hipStream_t s) { hipLaunchKernelGGL(k_seed_next, dim3(1), dim3(1), 0, s, p); }
// the constant-frequency (100 MHz) device clock -> *dst (diagnostic time stamps inside replayed graphs: runtime.cpp stamp())
__global__ void k_stamp(uint64_t* dst) { *dst = wall_clock64(); }
void launch_stamp(uint64_t* dst, hipStream_t s) { hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, s, dst); }
void launch_set_u64(uint64_t* dst, uint64_t v, hipStream_t s) { hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, s, dst, v); }
__global__ void k_set_f3(float* dst, float a, float b, float c, int n) {
  if (n > 0) dst[0] = a;
  if (n > 1) dst[1] = b;
  if (n > 2) dst[2] = c;
}
// dst[i, :] = src[idx[i], :]   rows of row4 16-byte items (the train() input pipeline: batches are gathered on the
// device from an HBM-resident dataset by a device permutation, multiscale_vae.py:550-557 `shuffle=True`)
__global__ void __launch_bounds__(256) k_gather_rows(const float4* __restrict__ src, const int64_t* __restrict__ idx,
                                                     float4* __restrict__ dst, int64_t n, int64_t row4) {
  const int64_t total = n * row4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / row4, c = i - r * row4;
    dst[i] = src[idx[r] * row4 + c];
  }
}
__global__ void __launch_bounds__(256) k_gather_rows1(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                      float* __restrict__ dst, int64_t n, int64_t row) {
  const int64_t total = n * row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / row, c = i - r * row;
    dst[i] = src[idx[r] * row + c];
  }
}
void launch_gather_rows(const float* src, const int64_t* idx, float* dst, int64_t n, int64_t row_elems, hipStream_t s) {
  if (n <= 0 || row_elems <= 0) return;
  if (row_elems % 4 == 0)
    hipLaunchKernelGGL(k_gather_rows, dim3(grid_for(n * row_elems / 4)), dim3(256), 0, s, (const float4*)src, idx,
                       (float4*)dst, n, row_elems / 4);
  else
    hipLaunchKernelGGL(k_gather_rows1, dim3(grid_for(n * row_elems)), dim3(256), 0, s, src, idx, dst, n, row_elems);
}
void launch_set_f3(float* dst, float a, float b, float c, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_set_f3, dim3(1), dim3(1), 0, s, dst, a, b, c, n);
}
void launch_rng_keepmask(float* out, int64_t n, float p_drop, const uint64_t* seed, uint32_t sid, hipStream_t s) {
  ProfScope ps("rng", (double)(4.0*n), 0.0, s);
  if (n <= 0) return;
  hipLaunchKernelGGL(k_rng_keepmask, dim3(grid_for((n + 3) / 4)), dim3(kBlock), 0, s, out, n, p_drop, seed, sid);
}

// ------------------------------------------------------------------------------------------------
// input transform: normalize (:79-84), GaussianNoise (:139-142), SpatialDropout2D (:144-147)
// ------------------------------------------------------------------------------------------------
template <typename I>
__global__ void k_prep(const float* __restrict__ x, const float* __restrict__ noise, const float* __restrict__ keep,
                       float* __restrict__ out, int64_t n, int64_t per_img, int C, float v0, float inv_range2,
                       float noise_std, float keep_scale) {
  GRID_STRIDE_T(I, i, n) {
    float v = (x[i] - v0) * inv_range2 - 1.0f;
    if (noise) v += noise[i] * noise_std;
    if (keep) {
      I b = i / (I)per_img;
      int c = (int)(i % (I)C);
      v *= keep[b * C + c] * keep_scale;
    }
    out[i] = v;
  }
}
void launch_prep(const float* x, const float* noise, const float* keep, float* out, int B, int H, int W, int C,
                 float v0, float v1, float noise_std, float keep_scale, hipStream_t s) {
  ProfScope ps("pyramid", (double)(8.0*B*H*W*C), 0.0, s);
  int64_t per = (int64_t)H * W * C, n = per * B;
  LAUNCH_IDX(n, k_prep, dim3(grid_for(n)), dim3(kBlock), 0, s, x, noise, keep, out, n, per, C, v0, 2.0f / (v1 - v0),
             noise_std, keep_scale);
}

// _downsample_upsample (:292-315): f = G (*) in (3x3, zero SAME pad); band = in - f; down = f[::2, ::2]
__constant__ float c_gauss[9];   // set once by mvae_bind from layer_blocks.gaussian_kernel((3,3),(2,2))
void set_gauss_constants(const float* g9) { (void)hipMemcpyToSymbol(HIP_SYMBOL(c_gauss), g9, 9 * sizeof(float)); }

__global__ void k_blur_split(const float* __restrict__ in, float* __restrict__ band, float* __restrict__ down, int B,
                             int H, int W, int C, int DH, int DW) {
  int64_t n = (int64_t)B * H * W * C;
  GRID_STRIDE(i, n) {
    int c = (int)(i % C);
    int64_t p = i / C;
    int xw = (int)(p % W);
    p /= W;
    int yh = (int)(p % H);
    int64_t b = p / H;
    float f = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      int yy = yh + a - 1;
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        int xx = xw + e - 1;
        if (xx < 0 || xx >= W) continue;
        f += c_gauss[a * 3 + e] * in[((b * H + yy) * W + xx) * C + c];
      }
    }
    band[i] = in[i] - f;
    if (!(yh & 1) && !(xw & 1)) {
      int dy = yh >> 1, dx = xw >> 1;
      if (dy < DH && dx < DW) down[((b * DH + dy) * DW + dx) * C + c] = f;
    }
  }
}
// the same with one thread per pixel (all C <= 4 channels) and 32-bit index arithmetic
template <int C>
__global__ void __launch_bounds__(256) k_blur_split_px(const float* __restrict__ in, float* __restrict__ band,
                                                       float* __restrict__ down, unsigned B, unsigned H, unsigned W,
                                                       unsigned DH, unsigned DW) {
  const unsigned n = B * H * W;
  for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < n; p += gridDim.x * 256u) {
    const unsigned xw = p % W, q = p / W, yh = q % H, b = q / H;
    float f[C];
#pragma unroll
    for (int c = 0; c < C; ++c) f[c] = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int yy = (int)yh + a - 1;
      if (yy < 0 || yy >= (int)H) continue;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int xx = (int)xw + e - 1;
        if (xx < 0 || xx >= (int)W) continue;
        const float* ip = in + ((size_t)(b * H + (unsigned)yy) * W + (unsigned)xx) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) f[c] += c_gauss[a * 3 + e] * ip[c];
      }
    }
    const float* ip = in + (size_t)p * C;
    float* bp = band + (size_t)p * C;
#pragma unroll
    for (int c = 0; c < C; ++c) bp[c] = ip[c] - f[c];
    if (!(yh & 1) && !(xw & 1) && (yh >> 1) < DH && (xw >> 1) < DW) {
      float* dp = down + ((size_t)(b * DH + (yh >> 1)) * DW + (xw >> 1)) * C;
#pragma unroll
      for (int c = 0; c < C; ++c) dp[c] = f[c];
    }
  }
}
void launch_blur_split(const float* in, float* band, float* down, int B, int H, int W, int C, hipStream_t s) {
  ProfScope ps("pyramid", (double)(10.0*B*H*W*C), 0.0, s);
  int64_t n = (int64_t)B * H * W * C;
  if (C >= 1 && C <= 4 && (int64_t)B * H * W < (1ll << 31)) {
    const unsigned gf = (unsigned)grid_for((int64_t)B * H * W);
    switch (C) {
#define MVAE_BS(C_)                                                                                                   \
  case C_:                                                                                                            \
    hipLaunchKernelGGL(k_blur_split_px<C_>, dim3(gf), dim3(256), 0, s, in, band, down, (unsigned)B, (unsigned)H,      \
                       (unsigned)W, (unsigned)(H / 2), (unsigned)(W / 2));                                            \
    return;
      MVAE_BS(1) MVAE_BS(2) MVAE_BS(3) MVAE_BS(4)
#undef MVAE_BS
    }
  }
  hipLaunchKernelGGL(k_blur_split, dim3(grid_for(n)), dim3(kBlock), 0, s, in, band, down, B, H, W, C, H / 2, W / 2);
}

// ------------------------------------------------------------------------------------------------
// stand-alone Laplacian pyramid (layer_blocks.py:23-185): unlike the model's own input transform
// above, a level here is  down = (G (*) in)[::2, ::2]  and  diff = in - up2(down), with up2 = the x2 bilinear
// upsample of the merge below -- so merge(split(x)) reproduces x exactly up to rounding.
// ------------------------------------------------------------------------------------------------
struct Gauss9 { float w[9]; };
// one thread per output PIXEL (all C <= 8 channels), 32-bit index arithmetic (pixels < 2^31: launcher): the
// per-element form spent its time in three 64-bit divisions per float
// NORM: `in` is the raw image, normalised on the fly (v - v0) * k - 1 (level 0: saves writing and re-reading it)
template <int C, bool NORM>
__global__ void __launch_bounds__(256) k_lap_down(const float* __restrict__ in, float* __restrict__ down, unsigned B,
                                                  unsigned H, unsigned W, Gauss9 g, float v0, float k2) {
  const unsigned h = H / 2, w = W / 2, n = B * h * w;
  for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < n; p += gridDim.x * 256u) {
    const unsigned xo = p % w, q = p / w, yo = q % h, b = q / h;
    const int x = (int)xo * 2, y = (int)yo * 2;
    float f[C];
#pragma unroll
    for (int c = 0; c < C; ++c) f[c] = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int yy = y + a - 1;
      if (yy < 0 || yy >= (int)H) continue;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int xx = x + e - 1;
        if (xx < 0 || xx >= (int)W) continue;
        const float* ip = in + ((size_t)(b * H + (unsigned)yy) * W + (unsigned)xx) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) f[c] += g.w[a * 3 + e] * (NORM ? (ip[c] - v0) * k2 - 1.0f : ip[c]);
      }
    }
    float* op = down + (size_t)p * C;
#pragma unroll
    for (int c = 0; c < C; ++c) op[c] = f[c];
  }
}
template <int C, bool NORM>
__global__ void __launch_bounds__(256) k_lap_diff(const float* __restrict__ in, const float* __restrict__ coarse,
                                                  float* __restrict__ diff, unsigned B, unsigned H, unsigned W, float v0,
                                                  float k2) {
  const unsigned h = H / 2, w = W / 2, n = B * H * W;
  for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < n; p += gridDim.x * 256u) {
    const unsigned x = p % W, q = p / W, y = q % H, b = q / H;
    const unsigned iy = y >> 1, ix = x >> 1;
    const unsigned y2 = (y & 1) ? min(iy + 1, h - 1) : (iy ? iy - 1 : 0u);
    const unsigned x2 = (x & 1) ? min(ix + 1, w - 1) : (ix ? ix - 1 : 0u);
    const float* cp = coarse + (size_t)b * h * w * C;
    const float *p00 = cp + ((size_t)iy * w + ix) * C, *p01 = cp + ((size_t)iy * w + x2) * C;
    const float *p10 = cp + ((size_t)y2 * w + ix) * C, *p11 = cp + ((size_t)y2 * w + x2) * C;
    const float* ip = in + (size_t)p * C;
    float* op = diff + (size_t)p * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float up = 0.75f * (0.75f * p00[c] + 0.25f * p01[c]) + 0.25f * (0.75f * p10[c] + 0.25f * p11[c]);   // as k_upsample_add
      op[c] = (NORM ? (ip[c] - v0) * k2 - 1.0f : ip[c]) - up;
    }
  }
}
__global__ void k_denorm_clip(const float* __restrict__ in, float* __restrict__ out, int64_t n, float v0, float v1) {
  GRID_STRIDE(i, n) {
    float v = (in[i] + 1.0f) * (v1 - v0) * 0.5f + v0;
    out[i] = fminf(fmaxf(v, v0), v1);
  }
}
// merge step (layer_blocks.py:137-171): fine_out = up2(coarse) + fine_in; FINAL: only clip(denormalise(.)) is written
// MODE 0: out = sum;  1 (FINAL): out = clip(denormalise(sum));  2: out = sum and out2 = clip(denormalise(sum))
template <int C, int MODE>
__global__ void __launch_bounds__(256) k_lap_merge(const float* __restrict__ coarse, const float* __restrict__ fine,
                                                   float* __restrict__ out, float* __restrict__ out2, unsigned B,
                                                   unsigned H, unsigned W, float v0, float v1) {
  const unsigned h = H / 2, w = W / 2, n = B * H * W;
  for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < n; p += gridDim.x * 256u) {
    const unsigned x = p % W, q = p / W, y = q % H, b = q / H;
    const unsigned iy = y >> 1, ix = x >> 1;
    const unsigned y2 = (y & 1) ? min(iy + 1, h - 1) : (iy ? iy - 1 : 0u);
    const unsigned x2 = (x & 1) ? min(ix + 1, w - 1) : (ix ? ix - 1 : 0u);
    const float* cp = coarse + (size_t)b * h * w * C;
    const float *p00 = cp + ((size_t)iy * w + ix) * C, *p01 = cp + ((size_t)iy * w + x2) * C;
    const float *p10 = cp + ((size_t)y2 * w + ix) * C, *p11 = cp + ((size_t)y2 * w + x2) * C;
    const float* ip = fine + (size_t)p * C;
    float* op = out + (size_t)p * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float up = 0.75f * (0.75f * p00[c] + 0.25f * p01[c]) + 0.25f * (0.75f * p10[c] + 0.25f * p11[c]);
      const float m = up + ip[c];
      const float v = fminf(fmaxf((m + 1.0f) * (v1 - v0) * 0.5f + v0, v0), v1);
      if (MODE == 1) op[c] = v;
      else op[c] = m;
      if (MODE == 2) out2[(size_t)p * C + c] = v;
    }
  }
}
// Concatenate([UpSampling2D(2, bilinear)(coarse), fine]) of the trainable merge (layer_blocks.py:141-150)
template <int C>
__global__ void __launch_bounds__(256) k_lap_concat(const float* __restrict__ coarse, const float* __restrict__ fine,
                                                    float* __restrict__ cat, unsigned B, unsigned H, unsigned W) {
  const unsigned h = H / 2, w = W / 2, n = B * H * W;
  for (unsigned p = blockIdx.x * 256u + threadIdx.x; p < n; p += gridDim.x * 256u) {
    const unsigned x = p % W, q = p / W, y = q % H, b = q / H;
    const unsigned iy = y >> 1, ix = x >> 1;
    const unsigned y2 = (y & 1) ? min(iy + 1, h - 1) : (iy ? iy - 1 : 0u);
    const unsigned x2 = (x & 1) ? min(ix + 1, w - 1) : (ix ? ix - 1 : 0u);
    const float* cp = coarse + (size_t)b * h * w * C;
    const float *p00 = cp + ((size_t)iy * w + ix) * C, *p01 = cp + ((size_t)iy * w + x2) * C;
    const float *p10 = cp + ((size_t)y2 * w + ix) * C, *p11 = cp + ((size_t)y2 * w + x2) * C;
    const float* ip = fine + (size_t)p * C;
    float* op = cat + (size_t)p * 2 * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      op[c] = 0.75f * (0.75f * p00[c] + 0.25f * p01[c]) + 0.25f * (0.75f * p10[c] + 0.25f * p11[c]);   // as k_lap_merge
      op[C + c] = ip[c];
    }
  }
}
bool launch_lap_concat(const float* coarse, const float* fine, float* cat, int B, int H, int W, int C, hipStream_t s) {
  if (C < 1 || C > 8 || (int64_t)B * H * W >= (1ll << 31)) return false;
  const unsigned gf = (unsigned)grid_for((int64_t)B * H * W);
  switch (C) {
#define MVAE_LC(C_) case C_: hipLaunchKernelGGL((k_lap_concat<C_>), dim3(gf), dim3(256), 0, s, coarse, fine, cat, (unsigned)B, (unsigned)H, (unsigned)W); break;
    MVAE_LC(1) MVAE_LC(2) MVAE_LC(3) MVAE_LC(4) MVAE_LC(5) MVAE_LC(6) MVAE_LC(7) MVAE_LC(8)
#undef MVAE_LC
  }
  return true;
}
bool launch_lap_merge(const float* coarse, const float* fine, float* out, int B, int H, int W, int C, bool final_level,
                      float v0, float v1, hipStream_t s) {
  if (C < 1 || C > 8 || (int64_t)B * H * W >= (1ll << 31)) return false;
  const unsigned gf = (unsigned)grid_for((int64_t)B * H * W);
  switch (C) {
#define MVAE_LM(C_)                                                                                                   \
  case C_:                                                                                                            \
    if (final_level) hipLaunchKernelGGL((k_lap_merge<C_, 1>), dim3(gf), dim3(256), 0, s, coarse, fine, out,           \
                                        (float*)nullptr, (unsigned)B, (unsigned)H, (unsigned)W, v0, v1);              \
    else hipLaunchKernelGGL((k_lap_merge<C_, 0>), dim3(gf), dim3(256), 0, s, coarse, fine, out, (float*)nullptr,      \
                            (unsigned)B, (unsigned)H, (unsigned)W, v0, v1);                                           \
    break;
    MVAE_LM(1) MVAE_LM(2) MVAE_LM(3) MVAE_LM(4) MVAE_LM(5) MVAE_LM(6) MVAE_LM(7) MVAE_LM(8)
#undef MVAE_LM
  }
  return true;
}
bool launch_lap_level(const float* in, float* diff, float* down, int B, int H, int W, int C, const float* gauss9,
                      bool normalise, float v0, float v1, hipStream_t s) {
  if (C < 1 || C > 8 || (int64_t)B * H * W >= (1ll << 31)) return false;
  Gauss9 g;
  for (int k = 0; k < 9; ++k) g.w[k] = gauss9[k];
  const int64_t nd = (int64_t)B * (H / 2) * (W / 2), n = (int64_t)B * H * W;
  const unsigned gd = (unsigned)grid_for(nd), gf = (unsigned)grid_for(n);
  const float k2 = 2.0f / (v1 - v0);
  switch (C) {
#define MVAE_LAP2(C_, N_)                                                                                             \
    hipLaunchKernelGGL((k_lap_down<C_, N_>), dim3(gd), dim3(256), 0, s, in, down, (unsigned)B, (unsigned)H,           \
                       (unsigned)W, g, v0, k2);                                                                       \
    hipLaunchKernelGGL((k_lap_diff<C_, N_>), dim3(gf), dim3(256), 0, s, in, (const float*)down, diff, (unsigned)B,    \
                       (unsigned)H, (unsigned)W, v0, k2);
#define MVAE_LAP(C_)                                                                                                  \
  case C_:                                                                                                            \
    if (normalise) { MVAE_LAP2(C_, true) } else { MVAE_LAP2(C_, false) }                                              \
    break;
    MVAE_LAP(1) MVAE_LAP(2) MVAE_LAP(3) MVAE_LAP(4) MVAE_LAP(5) MVAE_LAP(6) MVAE_LAP(7) MVAE_LAP(8)
#undef MVAE_LAP
#undef MVAE_LAP2
  }
  return true;
}
void launch_denorm_clip(const float* in, float* out, int64_t n, float v0, float v1, hipStream_t s) {
  hipLaunchKernelGGL(k_denorm_clip, dim3(grid_for(n)), dim3(kBlock), 0, s, in, out, n, v0, v1);
}

// ------------------------------------------------------------------------------------------------
// generic convolutions.  One thread per output element, channel fastest (coalesced weights / stores).
// ------------------------------------------------------------------------------------------------
__global__ void k_conv_f(const float* __restrict__ big, const float* __restrict__ w, const float* __restrict__ bias,
                         const float* __restrict__ residual, float* __restrict__ small, ConvGeom g, PreOp pre,
                         int act) {
  int64_t n = (int64_t)g.B * g.OH * g.OW * g.CO;
  GRID_STRIDE(i, n) {
    int co = (int)(i % g.CO);
    int64_t p = i / g.CO;
    int ow = (int)(p % g.OW);
    p /= g.OW;
    int oh = (int)(p % g.OH);
    int64_t b = p / g.OH;
    float acc = bias ? bias[co] : 0.f;
    for (int kh = 0; kh < g.KH; ++kh) {
      int yy = oh * g.SH + kh - g.PT;
      if (yy < 0 || yy >= g.IH) continue;
      for (int kw = 0; kw < g.KW; ++kw) {
        int xx = ow * g.SW + kw - g.PL;
        if (xx < 0 || xx >= g.IW) continue;
        const float* ip = big + ((b * g.IH + yy) * g.IW + xx) * g.CI;
        const float* wp = w + ((int64_t)(kh * g.KW + kw) * g.CI) * g.CO + co;
        for (int ci = 0; ci < g.CI; ++ci) {
          float v = ip[ci];
          if (pre.scale) v = v * pre.scale[ci] + pre.shift[ci];
          if (pre.gate) v *= pre.gate[b * g.CI + ci];
          acc += v * wp[(int64_t)ci * g.CO];
        }
      }
    }
    acc = act_apply(acc, act);
    if (residual) acc += residual[i];
    small[i] = acc;
  }
}
void launch_conv_f_generic(const float* big, const float* w, const float* bias, const float* residual, float* small,
                           ConvGeom g, PreOp pre, int act, hipStream_t s);
void launch_conv_f_any(const float* big, const float* w, const float* bias, const float* residual, float* small,
                       ConvGeom g, int act, hipStream_t s) {
  PreOp none{nullptr, nullptr, nullptr};
  launch_conv_f_generic(big, w, bias, residual, small, g, none, act, s);
}
void launch_conv_f_generic(const float* big, const float* w, const float* bias, const float* residual, float* small,
                           ConvGeom g, PreOp pre, int act, hipStream_t s) {
  int64_t n = (int64_t)g.B * g.OH * g.OW * g.CO;
  hipLaunchKernelGGL(k_conv_f, dim3(grid_for(n)), dim3(kBlock), 0, s, big, w, bias, residual, small, g, pre, act);
}

__global__ void k_conv_t(const float* __restrict__ small, const float* __restrict__ w, const float* __restrict__ bias,
                         const float* __restrict__ residual, float* __restrict__ big, ConvGeom g) {
  int64_t n = (int64_t)g.B * g.IH * g.IW * g.CI;
  GRID_STRIDE(i, n) {
    int ci = (int)(i % g.CI);
    int64_t p = i / g.CI;
    int x = (int)(p % g.IW);
    p /= g.IW;
    int y = (int)(p % g.IH);
    int64_t b = p / g.IH;
    float acc = bias ? bias[ci] : 0.f;
    for (int kh = 0; kh < g.KH; ++kh) {
      int ty = y + g.PT - kh;
      if (ty < 0 || (ty % g.SH) != 0) continue;
      int oh = ty / g.SH;
      if (oh >= g.OH) continue;
      for (int kw = 0; kw < g.KW; ++kw) {
        int tx = x + g.PL - kw;
        if (tx < 0 || (tx % g.SW) != 0) continue;
        int ow = tx / g.SW;
        if (ow >= g.OW) continue;
        const float* sp = small + ((b * g.OH + oh) * g.OW + ow) * g.CO;
        const float* wp = w + ((int64_t)(kh * g.KW + kw) * g.CI + ci) * g.CO;
        for (int co = 0; co < g.CO; ++co) acc += sp[co] * wp[co];
      }
    }
    if (residual) acc += residual[i];
    big[i] = acc;
  }
}
void launch_conv_t_generic(const float* small, const float* w, const float* bias, const float* residual, float* big,
                           ConvGeom g, hipStream_t s) {
  int64_t n = (int64_t)g.B * g.IH * g.IW * g.CI;
  hipLaunchKernelGGL(k_conv_t, dim3(grid_for(n)), dim3(kBlock), 0, s, small, w, bias, residual, big, g);
}

// dW[tap][ci][co] += sum_m pre(big)[m(tap), ci] * small[m, co].  grid = (m-chunks, taps, (ci,co)-tiles of 4096).
// A block stages 16 rows of both operands in LDS; each thread owns up to 16 (ci,co) outputs.
static constexpr int kWgRows = 16;
static constexpr int kWgTile = 4096;
__global__ void __launch_bounds__(256) k_conv_wgrad(const float* __restrict__ big, const float* __restrict__ small,
                                                    float* __restrict__ dW, ConvGeom g, PreOp pre, int64_t M,
                                                    int rows_per_block) {
  extern __shared__ float lds[];
  float* sb = lds;                    // [kWgRows][CI]
  float* ss = lds + kWgRows * g.CI;   // [kWgRows][CO]
  const int tap = blockIdx.y, kh = tap / g.KW, kw = tap % g.KW;
  const int cico = g.CI * g.CO;
  const int tile0 = blockIdx.z * kWgTile;
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  int64_t m0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t m1 = m0 + rows_per_block;
  if (m1 > M) m1 = M;
  for (int64_t mb = m0; mb < m1; mb += kWgRows) {
    __syncthreads();
    for (int t = threadIdx.x; t < kWgRows * g.CI; t += blockDim.x) {
      int r = t / g.CI, ci = t % g.CI;
      int64_t m = mb + r;
      float v = 0.f;
      if (m < m1) {
        int ow = (int)(m % g.OW);
        int64_t p = m / g.OW;
        int oh = (int)(p % g.OH);
        int64_t b = p / g.OH;
        int yy = oh * g.SH + kh - g.PT, xx = ow * g.SW + kw - g.PL;
        if (yy >= 0 && yy < g.IH && xx >= 0 && xx < g.IW) {
          v = big[((b * g.IH + yy) * g.IW + xx) * g.CI + ci];
          if (pre.scale) v = v * pre.scale[ci] + pre.shift[ci];
          if (pre.gate) v *= pre.gate[b * g.CI + ci];
        }
      }
      sb[t] = v;
    }
    for (int t = threadIdx.x; t < kWgRows * g.CO; t += blockDim.x) {
      int r = t / g.CO;
      int64_t m = mb + r;
      ss[t] = (m < m1) ? small[m * g.CO + (t % g.CO)] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      int idx = tile0 + threadIdx.x + k * 256;
      if (idx < cico) {
        int ci = idx / g.CO, co = idx % g.CO;
        float a = acc[k];
#pragma unroll
        for (int r = 0; r < kWgRows; ++r) a += sb[r * g.CI + ci] * ss[r * g.CO + co];
        acc[k] = a;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    int idx = tile0 + threadIdx.x + k * 256;
    if (idx < cico) atomicAdd(&dW[(int64_t)tap * cico + idx], acc[k]);
  }
}
void launch_conv_wgrad_generic(const float* big, const float* small, float* dW, ConvGeom g, PreOp pre,
                               hipStream_t s) {
  int64_t M = (int64_t)g.B * g.OH * g.OW;
  int rows = 2048;
  int64_t chunks = (M + rows - 1) / rows;
  while (chunks > 2048) { rows *= 2; chunks = (M + rows - 1) / rows; }
  int tiles = (g.CI * g.CO + kWgTile - 1) / kWgTile;
  size_t lds = (size_t)kWgRows * (g.CI + g.CO) * sizeof(float);
  hipLaunchKernelGGL(k_conv_wgrad, dim3((unsigned)chunks, g.KH * g.KW, tiles), dim3(256), lds, s, big, small, dW, g,
                     pre, M, rows);
}

__global__ void k_elu_bwd(float* d, const float* y, int64_t n) {
  GRID_STRIDE(i, n) {
    float yy = y[i];
    if (yy <= 0.f) d[i] *= (yy + 1.0f);
  }
}
void launch_elu_bwd(float* d, const float* y, int64_t n, hipStream_t s) {
  ProfScope ps("elu_bwd", (double)(12.0*n), 0.0, s);
  hipLaunchKernelGGL(k_elu_bwd, dim3(grid_for(n)), dim3(kBlock), 0, s, d, y, n);
}

// ------------------------------------------------------------------------------------------------
// depthwise 3x3 (layer_blocks.py:604-614), kernel (3,3,C,1) == [3][3][C]
// ------------------------------------------------------------------------------------------------
__global__ void k_dw_fwd(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                         float* __restrict__ out, int B, int H, int W, int C) {
  int64_t n = (int64_t)B * H * W * C;
  GRID_STRIDE(i, n) {
    int c = (int)(i % C);
    int64_t p = i / C;
    int x = (int)(p % W);
    p /= W;
    int y = (int)(p % H);
    int64_t b = p / H;
    float acc = bias[c];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      int yy = y + a - 1;
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        int xx = x + e - 1;
        if (xx < 0 || xx >= W) continue;
        acc += w[(a * 3 + e) * C + c] * in[((b * H + yy) * W + xx) * C + c];
      }
    }
    out[i] = acc > 0.f ? acc : 0.f;
  }
}
void launch_dw_fwd_generic(const float* in, const float* w, const float* b, float* out, int B, int H, int W, int C,
                           hipStream_t s) {
  int64_t n = (int64_t)B * H * W * C;
  hipLaunchKernelGGL(k_dw_fwd, dim3(grid_for(n)), dim3(kBlock), 0, s, in, w, b, out, B, H, W, C);
}

__global__ void k_dw_bwd_data(const float* __restrict__ dy, const float* __restrict__ w,
                              const float* __restrict__ mask_src, float* __restrict__ dx, int B, int H, int W, int C) {
  int64_t n = (int64_t)B * H * W * C;
  GRID_STRIDE(i, n) {
    float r = 0.f;
    if (mask_src[i] > 0.f) {
      int c = (int)(i % C);
      int64_t p = i / C;
      int x = (int)(p % W);
      p /= W;
      int y = (int)(p % H);
      int64_t b = p / H;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        int yy = y - (a - 1);   // output pixel that read us through tap a
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          int xx = x - (e - 1);
          if (xx < 0 || xx >= W) continue;
          r += w[(a * 3 + e) * C + c] * dy[((b * H + yy) * W + xx) * C + c];
        }
      }
    }
    dx[i] = r;
  }
}
void launch_dw_bwd_data_generic(const float* dy, const float* w, const float* mask_src, float* dx, int B, int H,
                                int W, int C, hipStream_t s) {
  int64_t n = (int64_t)B * H * W * C;
  hipLaunchKernelGGL(k_dw_bwd_data, dim3(grid_for(n)), dim3(kBlock), 0, s, dy, w, mask_src, dx, B, H, W, C);
}

// block = (cpb channels) x (256/cpb pixel lanes); each block walks `ppb` pixels of the flattened [B*H*W] axis.
__global__ void __launch_bounds__(256) k_dw_wgrad(const float* __restrict__ in, const float* __restrict__ dy,
                                                  float* __restrict__ dW, float* __restrict__ db, int B, int H, int W,
                                                  int C, int cpb, int64_t ppb) {
  __shared__ float sh[10 * 256];
  const int tc = threadIdx.x % cpb, tr = threadIdx.x / cpb, nr = blockDim.x / cpb;
  const int c = blockIdx.y * cpb + tc;
  const int64_t P = (int64_t)B * H * W;
  float acc[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = 0.f;
  if (c < C && tr < nr) {
    int64_t p0 = (int64_t)blockIdx.x * ppb, p1 = p0 + ppb;
    if (p1 > P) p1 = P;
    for (int64_t p = p0 + tr; p < p1; p += nr) {
      int x = (int)(p % W);
      int64_t q = p / W;
      int y = (int)(q % H);
      int64_t b = q / H;
      float d = dy[p * C + c];
      acc[9] += d;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        int yy = y + a - 1;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          int xx = x + e - 1;
          if (xx < 0 || xx >= W) continue;
          acc[a * 3 + e] += d * in[((b * H + yy) * W + xx) * C + c];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 10; ++k) sh[k * 256 + threadIdx.x] = acc[k];
  __syncthreads();
  if (tr == 0 && c < C) {
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      float t = 0.f;
      for (int r = 0; r < nr; ++r) t += sh[k * 256 + r * cpb + tc];
      if (k < 9) atomicAdd(&dW[k * C + c], t);
      else atomicAdd(&db[c], t);
    }
  }
}
void launch_dw_wgrad_generic(const float* in, const float* dy, float* dW, float* db, int B, int H, int W, int C,
                             hipStream_t s) {
  int cpb = C < 256 ? C : 256;
  int64_t P = (int64_t)B * H * W;
  int64_t ppb = 1024;
  int64_t chunks = (P + ppb - 1) / ppb;
  while (chunks > 1024) { ppb *= 2; chunks = (P + ppb - 1) / ppb; }
  hipLaunchKernelGGL(k_dw_wgrad, dim3((unsigned)chunks, (C + cpb - 1) / cpb), dim3(256), 0, s, in, dy, dW, db, B, H,
                     W, C, cpb, ppb);
}

// ------------------------------------------------------------------------------------------------
// pixel reductions
// ------------------------------------------------------------------------------------------------
template <int MODE>   // 0: sum a ; 1: sum a*b
__global__ void __launch_bounds__(256) k_spatial(const float* __restrict__ a, const float* __restrict__ bb,
                                                 float* __restrict__ out, int64_t HW, int C, int cpb, float scale,
                                                 int64_t ppb, int use_atomic) {
  __shared__ float sh[256];
  const int tc = threadIdx.x % cpb, tr = threadIdx.x / cpb, nr = blockDim.x / cpb;
  const int c = blockIdx.y * cpb + tc;
  const int64_t b = blockIdx.z;
  float acc = 0.f;
  if (c < C && tr < nr) {
    int64_t p0 = (int64_t)blockIdx.x * ppb, p1 = p0 + ppb;
    if (p1 > HW) p1 = HW;
    const float* ap = a + b * HW * C;
    const float* bp = MODE == 1 ? bb + b * HW * C : nullptr;
    for (int64_t p = p0 + tr; p < p1; p += nr) {
      float v = ap[p * C + c];
      if (MODE == 1) v *= bp[p * C + c];
      acc += v;
    }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (tr == 0 && c < C) {
    float t = 0.f;
    for (int r = 0; r < nr; ++r) t += sh[r * cpb + tc];
    t *= scale;
    if (use_atomic) atomicAdd(&out[b * C + c], t);
    else out[b * C + c] = t;
  }
}
template <int MODE>
static void launch_spatial(const float* a, const float* b, float* out, int B, int64_t HW, int C, float scale,
                           hipStream_t s) {
  int cpb = C < 256 ? C : 256;
  int64_t ppb = det_mode() ? (HW > 4096 ? HW : 4096) : 4096;     // deterministic mode: one block per (image, channel group)
  int64_t chunks = (HW + ppb - 1) / ppb;
  if (chunks > 1) launch_zero(out, (int64_t)B * C, s);
  hipLaunchKernelGGL(k_spatial<MODE>, dim3((unsigned)chunks, (C + cpb - 1) / cpb, B), dim3(256), 0, s, a, b, out, HW,
                     C, cpb, scale, ppb, chunks > 1 ? 1 : 0);
}
void launch_spatial_sum(const float* x, float* out, int B, int64_t HW, int C, float scale, hipStream_t s) {
  ProfScope ps("spatial_reduce", (double)(4.0*B*HW*C), 0.0, s);
  launch_spatial<0>(x, nullptr, out, B, HW, C, scale, s);
}
void launch_spatial_dot(const float* a, const float* b, float* out, int B, int64_t HW, int C, hipStream_t s) {
  ProfScope ps("spatial_reduce", (double)(8.0*B*HW*C), 0.0, s);
  launch_spatial<1>(a, b, out, B, HW, C, 1.0f, s);
}

// column reductions over [M, C] with a per-element functor; out[c] += (atomic)
struct FnSum {
  const float* x;
  __device__ float operator()(int64_t m, int c, int C) const { return x[m * C + c]; }
};
struct FnSqDev {
  const float* x; const float* mean;
  __device__ float operator()(int64_t m, int c, int C) const { float d = x[m * C + c] - mean[c]; return d * d; }
};
template <class F>
__global__ void __launch_bounds__(256) k_colreduce(F f, float* __restrict__ out, int64_t M, int C, int cpb,
                                                   int64_t rpb) {
  __shared__ float sh[256];
  const int tc = threadIdx.x % cpb, tr = threadIdx.x / cpb, nr = blockDim.x / cpb;
  const int c = blockIdx.y * cpb + tc;
  float acc = 0.f;
  if (c < C && tr < nr) {
    int64_t m0 = (int64_t)blockIdx.x * rpb, m1 = m0 + rpb;
    if (m1 > M) m1 = M;
    for (int64_t m = m0 + tr; m < m1; m += nr) acc += f(m, c, C);
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (tr == 0 && c < C) {
    float t = 0.f;
    for (int r = 0; r < nr; ++r) t += sh[r * cpb + tc];
    atomicAdd(&out[c], t);
  }
}
template <class F>
static void launch_colreduce(F f, float* out, int64_t M, int C, hipStream_t s) {
  int cpb = C < 256 ? C : 256;
  int64_t rpb = 1024;
  int64_t chunks = (M + rpb - 1) / rpb;
  while (chunks > 1024) { rpb *= 2; chunks = (M + rpb - 1) / rpb; }
  hipLaunchKernelGGL(k_colreduce<F>, dim3((unsigned)chunks, (C + cpb - 1) / cpb), dim3(256), 0, s, f, out, M, C, cpb,
                     rpb);
}
void launch_colsum(const float* x, float* out, int64_t M, int C, hipStream_t s) {
  ProfScope ps("col_reduce", (double)(4.0*M*C), 0.0, s);
  launch_colreduce(FnSum{x}, out, M, C, s);
}
void launch_colsqdev(const float* x, const float* mean, float* out, int64_t M, int C, hipStream_t s) {
  ProfScope ps("col_reduce", (double)(4.0*M*C), 0.0, s);
  launch_colreduce(FnSqDev{x, mean}, out, M, C, s);
}
struct FnBnD {
  const float* d;
  __device__ float operator()(int64_t m, int c, int C) const { return d[m * C + c]; }
};
struct FnBnDx {
  const float* d; const float* x; const float* mean; const float* invstd;
  __device__ float operator()(int64_t m, int c, int C) const {
    return d[m * C + c] * (x[m * C + c] - mean[c]) * invstd[c];
  }
};
void launch_bn_bwd_reduce(const float* d, const float* x, const float* mean, const float* invstd, float* sum_d,
                          float* sum_dx, int64_t M, int C, hipStream_t s) {
  ProfScope ps("col_reduce", (double)(16.0*M*C), 0.0, s);
  launch_colreduce(FnBnD{d}, sum_d, M, C, s);
  launch_colreduce(FnBnDx{d, x, mean, invstd}, sum_dx, M, C, s);
}

// ------------------------------------------------------------------------------------------------
// MobileNetV3 / squeeze-excite small pieces
// ------------------------------------------------------------------------------------------------
__global__ void k_mn_dt1pre(float* __restrict__ d, const float* __restrict__ t1, const float* __restrict__ g,
                            const float* __restrict__ dgap, int64_t n, int64_t per_img, int C, float inv_hw) {
  GRID_STRIDE(i, n) {
    float r = 0.f;
    if (t1[i] > 0.f) {
      int64_t b = i / per_img;
      int c = (int)(i % C);
      r = d[i] * g[b * C + c] + dgap[b * C + c] * inv_hw;
    }
    d[i] = r;
  }
}
void launch_mn_dt1pre_generic(float* d, const float* t1, const float* g, const float* dgap, int B, int64_t HW, int C,
                              float inv_hw, hipStream_t s) {
  int64_t per = HW * C, n = per * B;
  hipLaunchKernelGGL(k_mn_dt1pre, dim3(grid_for(n)), dim3(kBlock), 0, s, d, t1, g, dgap, n, per, C, inv_hw);
}

__global__ void k_gemm_nn(const float* __restrict__ a, const float* __restrict__ w, const float* __restrict__ bias,
                          float* __restrict__ out, float* __restrict__ out_lin, int B, int K, int N, int act) {
  int64_t n = (int64_t)B * N;
  GRID_STRIDE(i, n) {
    int j = (int)(i % N);
    int64_t b = i / N;
    const float* ap = a + b * K;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int k = 0;
    // 16 independent loads in flight per trip: these small products are latency-, not throughput-bound
#pragma unroll 4
    for (; k + 3 < K; k += 4) {
      acc0 += ap[k] * w[(int64_t)k * N + j];
      acc1 += ap[k + 1] * w[(int64_t)(k + 1) * N + j];
      acc2 += ap[k + 2] * w[(int64_t)(k + 2) * N + j];
      acc3 += ap[k + 3] * w[(int64_t)(k + 3) * N + j];
    }
    for (; k < K; ++k) acc0 += ap[k] * w[(int64_t)k * N + j];
    float acc = (acc0 + acc1) + (acc2 + acc3) + (bias ? bias[j] : 0.f);
    if (out_lin) out_lin[i] = acc;
    out[i] = act_apply(acc, act);
  }
}
void launch_gemm_nn_generic(const float* a, const float* w, const float* bias, float* out, float* out_lin, int B,
                            int K, int N, int act, hipStream_t s) {
  hipLaunchKernelGGL(k_gemm_nn, dim3(grid_for((int64_t)B * N)), dim3(kBlock), 0, s, a, w, bias, out, out_lin, B, K, N,
                     act);
}

__global__ void k_gemm_nt(const float* __restrict__ a, const float* __restrict__ w, float* __restrict__ out, int B,
                          int K, int N, const float* __restrict__ hs_lin, int accumulate) {
  int64_t n = (int64_t)B * K;
  GRID_STRIDE(i, n) {
    int k = (int)(i % K);
    int64_t b = i / K;
    const float* ap = a + b * N;
    const float* wp = w + (int64_t)k * N;
    float acc = 0.f;
    if (hs_lin) {
      const float* hp = hs_lin + b * N;
#pragma unroll 8
      for (int j = 0; j < N; ++j) acc += ap[j] * hsig_grad(hp[j]) * wp[j];
    } else {
#pragma unroll 8
      for (int j = 0; j < N; ++j) acc += ap[j] * wp[j];
    }
    out[i] = accumulate ? out[i] + acc : acc;
  }
}
void launch_gemm_nt_generic(const float* a, const float* w, float* out, int B, int K, int N, const float* hs_lin,
                            int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(k_gemm_nt, dim3(grid_for((int64_t)B * K)), dim3(kBlock), 0, s, a, w, out, B, K, N, hs_lin,
                     accumulate);
}

__global__ void k_gemm_tn(const float* __restrict__ a, const float* __restrict__ g, float* __restrict__ dW,
                          float* __restrict__ db, int B, int K, int N, const float* __restrict__ a_scale,
                          const float* __restrict__ a_shift, const float* __restrict__ hs_lin) {
  int64_t n = (int64_t)K * N;
  GRID_STRIDE(i, n) {
    int j = (int)(i % N);
    int k = (int)(i / N);
    float sc = a_scale ? a_scale[k] : 1.f, sf = a_scale ? a_shift[k] : 0.f;
    float acc = 0.f, accb = 0.f;
    for (int b = 0; b < B; ++b) {
      float gv = g[(int64_t)b * N + j];
      if (hs_lin) gv *= hsig_grad(hs_lin[(int64_t)b * N + j]);
      acc += (a[(int64_t)b * K + k] * sc + sf) * gv;
      accb += gv;
    }
    dW[i] += acc;
    if (k == 0 && db) db[j] += accb;
  }
}
void launch_gemm_tn_generic(const float* a, const float* g, float* dW, float* db, int B, int K, int N,
                            const float* a_scale, const float* a_shift, const float* hs_lin, hipStream_t s) {
  hipLaunchKernelGGL(k_gemm_tn, dim3(grid_for((int64_t)K * N)), dim3(kBlock), 0, s, a, g, dW, db, B, K, N, a_scale,
                     a_shift, hs_lin);
}

// BatchNorm over the batch axis, one block per channel (layer_blocks.py:447-449, keras defaults).
__global__ void __launch_bounds__(256) k_bn1d_fwd(const float* __restrict__ x, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, const float* __restrict__ mov_mean,
                                                  const float* __restrict__ mov_var, float* __restrict__ xhat,
                                                  float* __restrict__ invstd, float* __restrict__ y,
                                                  float* __restrict__ stat_mean, float* __restrict__ stat_var, int B,
                                                  int C, float eps, int training) {
  __shared__ float sh[16];
  __shared__ float s_mean, s_inv;
  const int c = blockIdx.x;
  if (training) {
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) acc += x[(int64_t)b * C + c];
    float t = block_sum(acc, sh);
    if (threadIdx.x == 0) s_mean = t / (float)B;
    __syncthreads();
    float m = s_mean;
    acc = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
      float d = x[(int64_t)b * C + c] - m;
      acc += d * d;
    }
    t = block_sum(acc, sh);
    if (threadIdx.x == 0) {
      float var = t / (float)B;
      s_inv = rsqrtf(var + eps);
      stat_mean[c] = m;
      stat_var[c] = var;
    }
  } else if (threadIdx.x == 0) {
    s_mean = mov_mean[c];
    s_inv = rsqrtf(mov_var[c] + eps);
  }
  __syncthreads();
  float m = s_mean, inv = s_inv, gm = gamma[c], bt = beta[c];
  if (threadIdx.x == 0) invstd[c] = inv;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float xh = (x[(int64_t)b * C + c] - m) * inv;
    xhat[(int64_t)b * C + c] = xh;
    y[(int64_t)b * C + c] = xh * gm + bt;
  }
}
void launch_bn1d_fwd(const float* x, const float* gamma, const float* beta, const float* mov_mean,
                     const float* mov_var, float* xhat, float* invstd, float* y, float* stat_mean, float* stat_var,
                     int B, int C, float eps, int training, hipStream_t s) {
  ProfScope ps("bn1d", (double)(12.0*B*C), 0.0, s);
  hipLaunchKernelGGL(k_bn1d_fwd, dim3(C), dim3(256), 0, s, x, gamma, beta, mov_mean, mov_var, xhat, invstd, y,
                     stat_mean, stat_var, B, C, eps, training);
}

__global__ void __launch_bounds__(256) k_bn1d_bwd(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                  const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                  const float* __restrict__ relu_src, float* __restrict__ dx,
                                                  float* __restrict__ dgamma, float* __restrict__ dbeta, int B,
                                                  int C) {
  __shared__ float sh[16];
  __shared__ float s_a, s_b;
  const int c = blockIdx.x;
  float a0 = 0.f, a1 = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float d = dy[(int64_t)b * C + c];
    a0 += d;
    a1 += d * xhat[(int64_t)b * C + c];
  }
  float t0 = block_sum(a0, sh);
  float t1 = block_sum(a1, sh);
  if (threadIdx.x == 0) {
    s_a = t0; s_b = t1;
    dbeta[c] += t0;
    dgamma[c] += t1;
  }
  __syncthreads();
  float gm = gamma[c], inv = invstd[c], md = s_a / (float)B, mdx = s_b / (float)B;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    int64_t i = (int64_t)b * C + c;
    float v = gm * inv * (dy[i] - md - xhat[i] * mdx);
    dx[i] = relu_src[i] > 0.f ? v : 0.f;
  }
}
void launch_bn1d_bwd(const float* dy, const float* xhat, const float* invstd, const float* gamma,
                     const float* relu_src, float* dx, float* dgamma, float* dbeta, int B, int C, hipStream_t s) {
  ProfScope ps("bn1d", (double)(16.0*B*C), 0.0, s);
  hipLaunchKernelGGL(k_bn1d_bwd, dim3(C), dim3(256), 0, s, dy, xhat, invstd, gamma, relu_src, dx, dgamma, dbeta, B, C);
}

// ------------------------------------------------------------------------------------------------
// decoder BatchNorm helpers (multiscale_vae.py:420-421)
// ------------------------------------------------------------------------------------------------
// sum / sqdev are `nslots` copies C floats apart (slot copies of the column statistics; copy 0 only when nslots = 1)
__global__ void k_bn2d_finalize(const float* sum, const float* sqdev, const float* gamma, const float* beta,
                                const float* mov_mean, const float* mov_var, float* mean, float* invstd, float* scale,
                                float* shift, float* stat_mean, float* stat_var, float inv_m, int C, float eps,
                                int training, int phase, int nslots, const float* pivot) {
  // pivot != nullptr (one-pass statistics, k_colstat4<2>): sum = S1 = sum (x - pivot), sqdev = S2 = sum (x - pivot)^2
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mu = 0.f, s1 = 0.f;
  if (training) {
    for (int k = 0; k < nslots; ++k) s1 += sum[k * C + c];
    mu = s1 * inv_m + (pivot ? pivot[c] : 0.f);
  } else {
    mu = mov_mean[c];
  }
  mean[c] = mu;
  if (phase == 0) return;   // mean only
  float var = 0.f;
  if (training) {
    for (int k = 0; k < nslots; ++k) var += sqdev[k * C + c];
    if (pivot) var = fmaxf(var - s1 * (s1 * inv_m), 0.f);
    var *= inv_m;
  } else {
    var = mov_var[c];
  }
  float inv = rsqrtf(var + eps);
  invstd[c] = inv;
  float sc = gamma[c] * inv;
  scale[c] = sc;
  shift[c] = beta[c] - mu * sc;
  if (training) { stat_mean[c] = mu; stat_var[c] = var; }
}
void launch_bn2d_mean(const float* sum, const float* mov_mean, float* mean, int64_t M, int C, int training,
                      hipStream_t s) {
  ProfScope ps("bn2d_small", (double)(0.0), 0.0, s);
  hipLaunchKernelGGL(k_bn2d_finalize, dim3((C + 63) / 64), dim3(64), 0, s, sum, nullptr, nullptr, nullptr, mov_mean,
                     nullptr, mean, nullptr, nullptr, nullptr, nullptr, nullptr, 1.0f / (float)M, C, 0.f, training, 0, 1, nullptr);
}
void launch_bn2d_finalize(const float* sum, const float* sqdev, const float* gamma, const float* beta,
                          const float* mov_mean, const float* mov_var, float* mean, float* invstd, float* scale,
                          float* shift, float* stat_mean, float* stat_var, int64_t M, int C, float eps, int training,
                          int nslots, hipStream_t s, const float* pivot) {
  ProfScope ps("bn2d_small", (double)(0.0), 0.0, s);
  hipLaunchKernelGGL(k_bn2d_finalize, dim3((C + 63) / 64), dim3(64), 0, s, sum, sqdev, gamma, beta, mov_mean, mov_var,
                     mean, invstd, scale, shift, stat_mean, stat_var, 1.0f / (float)M, C, eps, training, 1, nslots, pivot);
}

__global__ void k_bn2d_bwd_apply(float* __restrict__ d, const float* __restrict__ x, const float* __restrict__ mean,
                                 const float* __restrict__ invstd, const float* __restrict__ gamma,
                                 const float* __restrict__ sum_d, const float* __restrict__ sum_dx, int64_t n, int C,
                                 float inv_m) {
  GRID_STRIDE(i, n) {
    int c = (int)(i % C);
    float xh = (x[i] - mean[c]) * invstd[c];
    d[i] = gamma[c] * invstd[c] * (d[i] - sum_d[c] * inv_m - xh * sum_dx[c] * inv_m);
  }
}
__global__ void k_add_vec2(float* o0, const float* a0, float* o1, const float* a1, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { o0[i] += a0[i]; o1[i] += a1[i]; }
}
void launch_add_vec2(float* o0, const float* a0, float* o1, const float* a1, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_add_vec2, dim3((n + 63) / 64), dim3(64), 0, s, o0, a0, o1, a1, n);
}
void launch_bn2d_bwd_apply(float* d, const float* x, const float* mean, const float* invstd, const float* gamma,
                           const float* sum_d, const float* sum_dx, float* dgamma, float* dbeta, int64_t M, int C,
                           hipStream_t s) {
  ProfScope ps("bn2d_apply", (double)(12.0*M*C), 0.0, s);
  int64_t n = M * C;
  hipLaunchKernelGGL(k_bn2d_bwd_apply, dim3(grid_for(n)), dim3(kBlock), 0, s, d, x, mean, invstd, gamma, sum_d, sum_dx,
                     n, C, 1.0f / (float)M);
  hipLaunchKernelGGL(k_add_vec2, dim3((C + 63) / 64), dim3(64), 0, s, dgamma, sum_dx, dbeta, sum_d, C);
}
__global__ void k_scale_vec(float* v, float a, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] *= a;
}
void launch_scale_vec(float* v, float a, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_scale_vec, dim3((n + 255) / 256), dim3(256), 0, s, v, a, n);
}

// ------------------------------------------------------------------------------------------------
// latent head: sampling (multiscale_vae.py:372-378: mu + exp(log_var) * eps) and KL (:485-488)
// ------------------------------------------------------------------------------------------------
__global__ void k_sample_kl(const float* __restrict__ mu, const float* __restrict__ lv, const float* __restrict__ eps,
                            int eps_stride, int eps_off, float* __restrict__ z, float* __restrict__ kl_out,
                            int kl_stride, int kl_col, int B, int Z) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float kl = 0.f;
  for (int j = 0; j < Z; ++j) {
    float m = mu[(int64_t)b * Z + j], l = lv[(int64_t)b * Z + j];
    float e = expf(l);
    z[(int64_t)b * Z + j] = m + e * eps[(int64_t)b * eps_stride + eps_off + j];
    kl += 1.0f + l - m * m - e;
  }
  kl_out[(int64_t)b * kl_stride + kl_col] = -0.5f * kl;
}
void launch_sample_kl(const float* mu, const float* lv, const float* eps, int eps_stride, int eps_off, float* z,
                      float* kl_out, int kl_stride, int kl_col, int B, int Z, hipStream_t s) {
  ProfScope ps("latent", (double)(16.0*B*Z), 0.0, s);
  hipLaunchKernelGGL(k_sample_kl, dim3((B + 63) / 64), dim3(64), 0, s, mu, lv, eps, eps_stride, eps_off, z, kl_out,
                     kl_stride, kl_col, B, Z);
}
__global__ void k_sample_kl_bwd(const float* __restrict__ dz, const float* __restrict__ mu,
                                const float* __restrict__ lv, const float* __restrict__ eps, int eps_stride,
                                int eps_off, float* __restrict__ dmu, float* __restrict__ dlv,
                                const float* __restrict__ hp, int B, int Z) {
  int64_t n = (int64_t)B * Z;
  const float kfb = hp[HP_KF_OVER_B];
  GRID_STRIDE(i, n) {
    int j = (int)(i % Z);
    int64_t b = i / Z;
    float e = expf(lv[i]);
    float d = dz[i];
    dmu[i] = d + kfb * mu[i];
    dlv[i] = d * eps[b * eps_stride + eps_off + j] * e + kfb * 0.5f * (e - 1.0f);
  }
}
void launch_sample_kl_bwd(const float* dz, const float* mu, const float* lv, const float* eps, int eps_stride,
                          int eps_off, float* dmu, float* dlv, const float* hp, int B, int Z, hipStream_t s) {
  ProfScope ps("latent", (double)(24.0*B*Z), 0.0, s);
  hipLaunchKernelGGL(k_sample_kl_bwd, dim3(grid_for((int64_t)B * Z)), dim3(kBlock), 0, s, dz, mu, lv, eps, eps_stride,
                     eps_off, dmu, dlv, hp, B, Z);
}
__global__ void k_copy_cols(const float* src, int ss, int so, float* dst, int ds, int dof, int B, int n) {
  int64_t t = (int64_t)B * n;
  GRID_STRIDE(i, t) {
    int j = (int)(i % n);
    int64_t b = i / n;
    dst[b * ds + dof + j] = src[b * ss + so + j];
  }
}
void launch_copy_cols(const float* src, int src_stride, int src_off, float* dst, int dst_stride, int dst_off, int B,
                      int n, hipStream_t s) {
  ProfScope ps("latent", (double)(8.0*B*n), 0.0, s);
  hipLaunchKernelGGL(k_copy_cols, dim3(grid_for((int64_t)B * n)), dim3(kBlock), 0, s, src, src_stride, src_off, dst,
                     dst_stride, dst_off, B, n);
}

// ------------------------------------------------------------------------------------------------
// merge (multiscale_vae.py:204-224): UpSampling2D(2, bilinear) = half-pixel centres, edge clamp
// ------------------------------------------------------------------------------------------------
__global__ void k_upsample_add(const float* __restrict__ coarse, const float* __restrict__ fine_in,
                               float* __restrict__ fine_out, float* __restrict__ recon, int B, int H, int W, int C,
                               float v0, float v1) {
  const int h = H / 2, w = W / 2;
  int64_t n = (int64_t)B * H * W * C;
  GRID_STRIDE(i, n) {
    int c = (int)(i % C);
    int64_t p = i / C;
    int x = (int)(p % W);
    p /= W;
    int y = (int)(p % H);
    int64_t b = p / H;
    int iy = y >> 1, ix = x >> 1;
    int y2 = (y & 1) ? min(iy + 1, h - 1) : max(iy - 1, 0);
    int x2 = (x & 1) ? min(ix + 1, w - 1) : max(ix - 1, 0);
    const float* cp = coarse + b * h * w * C + c;
    float a00 = cp[((int64_t)iy * w + ix) * C], a01 = cp[((int64_t)iy * w + x2) * C];
    float a10 = cp[((int64_t)y2 * w + ix) * C], a11 = cp[((int64_t)y2 * w + x2) * C];
    float up = 0.75f * (0.75f * a00 + 0.25f * a01) + 0.25f * (0.75f * a10 + 0.25f * a11);
    float m = up + fine_in[i];
    fine_out[i] = m;
    if (recon) {
      float v = (m + 1.0f) * (v1 - v0) * 0.5f + v0;        // denormalize + K.clip, :86-94
      recon[i] = fminf(fmaxf(v, v0), v1);
    }
  }
}
void launch_upsample_add(const float* coarse, const float* fine_in, float* fine_out, float* recon, int B, int H,
                         int W, int C, float v0, float v1, hipStream_t s) {
  ProfScope ps("merge", (double)(12.0*B*H*W*C), 0.0, s);
  int64_t n = (int64_t)B * H * W * C;
  if (C >= 1 && C <= 4 && (int64_t)B * H * W < (1ll << 31)) {        // one thread per pixel, 32-bit indices (k_lap_merge)
    const unsigned gf = (unsigned)grid_for((int64_t)B * H * W);
    switch (C) {
#define MVAE_UA(C_)                                                                                                   \
  case C_:                                                                                                            \
    if (recon) hipLaunchKernelGGL((k_lap_merge<C_, 2>), dim3(gf), dim3(256), 0, s, coarse, fine_in, fine_out, recon,  \
                                  (unsigned)B, (unsigned)H, (unsigned)W, v0, v1);                                     \
    else hipLaunchKernelGGL((k_lap_merge<C_, 0>), dim3(gf), dim3(256), 0, s, coarse, fine_in, fine_out,               \
                            (float*)nullptr, (unsigned)B, (unsigned)H, (unsigned)W, v0, v1);                          \
    return;
      MVAE_UA(1) MVAE_UA(2) MVAE_UA(3) MVAE_UA(4)
#undef MVAE_UA
    }
  }
  hipLaunchKernelGGL(k_upsample_add, dim3(grid_for(n)), dim3(kBlock), 0, s, coarse, fine_in, fine_out, recon, B, H, W,
                     C, v0, v1);
}
// adjoint of the x2 bilinear upsample: coarse pixel i gathers fine 2i-1..2i+2 with weights .25 .75 .75 .25,
// clamped indices fold onto the border pixel.
template <typename I>
__global__ void k_upsample_bwd(const float* __restrict__ fg, float* __restrict__ cg, int B, int h, int w, int C) {
  const int H = 2 * h, W = 2 * w;
  const float wt[4] = {0.25f, 0.75f, 0.75f, 0.25f};
  int64_t n = (int64_t)B * h * w * C;
  GRID_STRIDE_T(I, i, n) {
    int c = (int)(i % (I)C);
    I p = i / (I)C;
    int x = (int)(p % (I)w);
    p /= (I)w;
    int y = (int)(p % (I)h);
    int64_t b = (int64_t)(p / (I)h);
    const float* fp = fg + b * H * W * C + c;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int fy = min(max(2 * y - 1 + a, 0), H - 1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int fx = min(max(2 * x - 1 + e, 0), W - 1);
        acc += wt[a] * wt[e] * fp[((int64_t)fy * W + fx) * C];
      }
    }
    cg[i] = acc;
  }
}
void launch_upsample_bwd(const float* fine_grad, float* coarse_grad, int B, int h, int w, int C, hipStream_t s) {
  ProfScope ps("merge", (double)(20.0*B*h*w*C), 0.0, s);
  int64_t n = (int64_t)B * h * w * C;
  LAUNCH_IDX(4 * n, k_upsample_bwd, dim3(grid_for(n)), dim3(kBlock), 0, s, fine_grad, coarse_grad, B, h, w, C);
}

// ------------------------------------------------------------------------------------------------
// losses (multiscale_vae.py:453-481).  One block per image.  C <= 8.
// ------------------------------------------------------------------------------------------------
static constexpr int kMaxLossC = 8;
__global__ void __launch_bounds__(256) k_loss_fwd(const float* __restrict__ y, const float* __restrict__ r,
                                                  float* __restrict__ losses, int ls, int nscales,
                                                  float* __restrict__ sgn, int H, int W, int C, int cy0, int cy1,
                                                  int cx0, int cx1) {
  __shared__ float sh[16];
  __shared__ float tot[1 + 2 * kMaxLossC];
  const int64_t b = blockIdx.x;
  const int64_t per = (int64_t)H * W * C;
  const float* yp = y + b * per;
  const float* rp = r + b * per;
  float a_abs = 0.f, a_ch[kMaxLossC], a_cc[kMaxLossC];
#pragma unroll
  for (int c = 0; c < kMaxLossC; ++c) a_ch[c] = a_cc[c] = 0.f;
  for (int64_t p = threadIdx.x; p < (int64_t)H * W; p += blockDim.x) {
    int x = (int)(p % W), yy = (int)(p / W);
    bool in = yy >= cy0 && yy < cy1 && x >= cx0 && x < cx1;
#pragma unroll
    for (int c = 0; c < kMaxLossC; ++c) {
      if (c < C) {
        float d = yp[p * C + c] - rp[p * C + c];
        a_abs += fabsf(d);
        a_ch[c] += d;
        if (in) a_cc[c] += d;
      }
    }
  }
  float t = block_sum(a_abs, sh);
  if (threadIdx.x == 0) tot[0] = t;
#pragma unroll
  for (int c = 0; c < kMaxLossC; ++c) {
    if (c < C) {
      float u = block_sum(a_ch[c], sh);
      float v = block_sum(a_cc[c], sh);
      if (threadIdx.x == 0) { tot[1 + c] = u; tot[1 + kMaxLossC + c] = v; }
    }
  }
  if (threadIdx.x == 0) {
    float hw = (float)H * W, ncrop = (float)(cy1 - cy0) * (cx1 - cx0);
    float rl = tot[0] / (hw * C);
    float ch = 0.f, cc = 0.f;
    for (int c = 0; c < C; ++c) {
      float mch = tot[1 + c] / hw, mcc = ncrop > 0 ? tot[1 + kMaxLossC + c] / ncrop : 0.f;
      ch += fabsf(mch);
      cc += fabsf(mcc);
      sgn[b * 2 * C + c] = (mch > 0.f) - (mch < 0.f);
      sgn[b * 2 * C + C + c] = (mcc > 0.f) - (mcc < 0.f);
    }
    losses[b * ls + 0] = rl;
    losses[b * ls + 1] = rl + 0.5f * (ch / C + cc / C);
    float kl = 0.f;
    for (int s = 0; s < nscales; ++s) kl += losses[b * ls + 3 + s];
    losses[b * ls + 2] = kl;
  }
}
// Large images (one block per image left 64 blocks walking 65536 pixels each at batch 64, 256x256: 218 us for 100 MB):
// S blocks per image leave partial sums part[b][s][1 + 2 kMaxLossC], k_loss_fwd_final adds them in a fixed order and
// finishes as k_loss_fwd does.
__global__ void __launch_bounds__(256) k_loss_fwd_part(const float* __restrict__ y, const float* __restrict__ r,
                                                       float* __restrict__ part, int S, int H, int W, int C, int cy0,
                                                       int cy1, int cx0, int cx1) {
  __shared__ float sh[16];
  const int64_t b = blockIdx.x;
  const int sidx = blockIdx.y;
  const int64_t per = (int64_t)H * W * C, HW = (int64_t)H * W;
  const float* yp = y + b * per;
  const float* rp = r + b * per;
  const int64_t p0 = HW * sidx / S, p1 = HW * (sidx + 1) / S;
  float a_abs = 0.f, a_ch[kMaxLossC], a_cc[kMaxLossC];
#pragma unroll
  for (int c = 0; c < kMaxLossC; ++c) a_ch[c] = a_cc[c] = 0.f;
  for (int64_t p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
    int x = (int)(p % W), yy = (int)(p / W);
    bool in = yy >= cy0 && yy < cy1 && x >= cx0 && x < cx1;
#pragma unroll
    for (int c = 0; c < kMaxLossC; ++c) {
      if (c < C) {
        float d = yp[p * C + c] - rp[p * C + c];
        a_abs += fabsf(d);
        a_ch[c] += d;
        if (in) a_cc[c] += d;
      }
    }
  }
  float* out = part + (b * S + sidx) * (1 + 2 * kMaxLossC);
  float t = block_sum(a_abs, sh);
  if (threadIdx.x == 0) out[0] = t;
#pragma unroll
  for (int c = 0; c < kMaxLossC; ++c) {
    if (c < C) {
      float u = block_sum(a_ch[c], sh);
      float v = block_sum(a_cc[c], sh);
      if (threadIdx.x == 0) { out[1 + c] = u; out[1 + kMaxLossC + c] = v; }
    }
  }
}
__global__ void __launch_bounds__(64) k_loss_fwd_final(const float* __restrict__ part, int S, float* __restrict__ losses,
                                                       int ls, int nscales, float* __restrict__ sgn, int H, int W, int C,
                                                       int cy0, int cy1, int cx0, int cx1) {
  __shared__ float tot[1 + 2 * kMaxLossC];
  const int64_t b = blockIdx.x;
  if (threadIdx.x < 1 + 2 * kMaxLossC) {
    float t = 0.f;
    for (int s2 = 0; s2 < S; ++s2) t += part[(b * S + s2) * (1 + 2 * kMaxLossC) + threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float hw = (float)H * W, ncrop = (float)(cy1 - cy0) * (cx1 - cx0);
    float rl = tot[0] / (hw * C);
    float ch = 0.f, cc = 0.f;
    for (int c = 0; c < C; ++c) {
      float mch = tot[1 + c] / hw, mcc = ncrop > 0 ? tot[1 + kMaxLossC + c] / ncrop : 0.f;
      ch += fabsf(mch);
      cc += fabsf(mcc);
      sgn[b * 2 * C + c] = (mch > 0.f) - (mch < 0.f);
      sgn[b * 2 * C + C + c] = (mcc > 0.f) - (mcc < 0.f);
    }
    losses[b * ls + 0] = rl;
    losses[b * ls + 1] = rl + 0.5f * (ch / C + cc / C);
    float kl = 0.f;
    for (int s2 = 0; s2 < nscales; ++s2) kl += losses[b * ls + 3 + s2];
    losses[b * ls + 2] = kl;
  }
}
static void crop_box(int H, int W, int* cy0, int* cy1, int* cx0, int* cx1) {
  int d0 = H / 2, d1 = W / 2;   // int(H/2), int(d0/2), int(d0*3/2): multiscale_vae.py:459-476
  *cy0 = d0 / 2; *cy1 = (d0 * 3) / 2; *cx0 = d1 / 2; *cx1 = (d1 * 3) / 2;
}
void launch_loss_fwd(const float* y, const float* recon, float* losses, int loss_stride, int nscales, float* sgn,
                     int B, int H, int W, int C, hipStream_t s, float* scratch, int64_t scratch_elems) {
  ProfScope ps("loss", (double)(8.0*B*H*W*C), 0.0, s);
  int cy0, cy1, cx0, cx1;
  crop_box(H, W, &cy0, &cy1, &cx0, &cx1);
  // enough blocks for the chip: S pieces per image (each at least 2048 pixels), partials through `scratch`
  int S = 1;
  while ((int64_t)B * S < 1024 && (int64_t)H * W / (2 * S) >= 2048 && S < 64) S *= 2;
  if (S > 1 && scratch && (int64_t)B * S * (1 + 2 * kMaxLossC) <= scratch_elems) {
    hipLaunchKernelGGL(k_loss_fwd_part, dim3(B, S), dim3(256), 0, s, y, recon, scratch, S, H, W, C, cy0, cy1, cx0, cx1);
    hipLaunchKernelGGL(k_loss_fwd_final, dim3(B), dim3(64), 0, s, (const float*)scratch, S, losses, loss_stride, nscales,
                       sgn, H, W, C, cy0, cy1, cx0, cx1);
    return;
  }
  hipLaunchKernelGGL(k_loss_fwd, dim3(B), dim3(256), 0, s, y, recon, losses, loss_stride, nscales, sgn, H, W, C, cy0,
                     cy1, cx0, cx1);
}
template <typename I>
__global__ void k_loss_bwd(const float* __restrict__ y, const float* __restrict__ r, const float* __restrict__ merged,
                           const float* __restrict__ sgn, float* __restrict__ du, int B, int H, int W, int C, float v0,
                           float v1, const float* __restrict__ hp, int cy0, int cy1, int cx0, int cx1) {
  int64_t n = (int64_t)B * H * W * C;
  const float rfb = hp[HP_RF_OVER_B];
  const float hw = (float)H * W, ncrop = (float)(cy1 - cy0) * (cx1 - cx0);
  const float half_range = (v1 - v0) * 0.5f;
  GRID_STRIDE_T(I, i, n) {
    int c = (int)(i % (I)C);
    I p = i / (I)C;
    int x = (int)(p % (I)W);
    p /= (I)W;
    int yy = (int)(p % (I)H);
    int64_t b = (int64_t)(p / (I)H);
    float v = (merged[i] + 1.0f) * half_range + v0;
    float g = 0.f;
    if (v >= v0 && v <= v1) {
      float d = y[i] - r[i];
      float sg = (float)((d > 0.f) - (d < 0.f));
      g = -sg / (hw * C);
      g -= 0.5f * sgn[b * 2 * C + c] / (C * hw);
      if (yy >= cy0 && yy < cy1 && x >= cx0 && x < cx1) g -= 0.5f * sgn[b * 2 * C + C + c] / (C * ncrop);
      g *= rfb * half_range;
    }
    du[i] = g;
  }
}
void launch_loss_bwd(const float* y, const float* recon, const float* merged, const float* sgn, float* du, int B,
                     int H, int W, int C, float v0, float v1, const float* hp, hipStream_t s) {
  ProfScope ps("loss", (double)(16.0*B*H*W*C), 0.0, s);
  int cy0, cy1, cx0, cx1;
  crop_box(H, W, &cy0, &cy1, &cx0, &cx1);
  int64_t n = (int64_t)B * H * W * C;
  LAUNCH_IDX(n, k_loss_bwd, dim3(grid_for(n)), dim3(kBlock), 0, s, y, recon, merged, sgn, du, B, H, W, C, v0, v1, hp, cy0,
             cy1, cx0, cx1);
}
__global__ void __launch_bounds__(256) k_metrics(const float* __restrict__ losses, int ncol, int B,
                                                 float* __restrict__ metrics) {
  __shared__ float sh[16];
  int j = blockIdx.x;
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) acc += losses[(int64_t)b * ncol + j];
  float t = block_sum(acc, sh);
  if (threadIdx.x == 0) {
    metrics[1 + j] += t;
    if (j == 0) metrics[0] += (float)B;
  }
}
void launch_metrics(const float* losses, int ncol, int B, float* metrics, hipStream_t s) {
  ProfScope ps("loss", (double)(0.0), 0.0, s);
  hipLaunchKernelGGL(k_metrics, dim3(ncol), dim3(256), 0, s, losses, ncol, B, metrics);
}

// ------------------------------------------------------------------------------------------------
// optimiser: Keras regularisers + per-variable clipnorm + Adagrad (multiscale_vae.py:497-499)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_opt_prepare(const float* __restrict__ w, float* __restrict__ g,
                                                     const ChunkDesc* __restrict__ chunks, float* __restrict__ norms,
                                                     const float* __restrict__ hp) {
  __shared__ float sh[16];
  const float grad_scale = hp[HP_GRAD_SCALE];
  ChunkDesc cd = chunks[blockIdx.x];
  float acc = 0.f;
  for (int i = threadIdx.x; i < cd.len; i += blockDim.x) {
    int64_t o = cd.offset + i;
    float gv = g[o] * grad_scale, wv = w[o];
    if (cd.reg == 1) gv += 0.01f * (float)((wv > 0.f) - (wv < 0.f));
    else if (cd.reg == 2) gv += 0.02f * wv;
    g[o] = gv;
    acc += gv * gv;
  }
  float t = block_sum(acc, sh);
  if (threadIdx.x == 0) norms[blockIdx.x] = t;        // per-chunk partial, summed in order by k_opt_apply
}
void launch_opt_prepare(const float* w, float* g, const ChunkDesc* chunks, int nchunks, float* norms,
                        const float* hp, hipStream_t s) {
  ProfScope ps("optimizer", (double)(0.0), 0.0, s);
  hipLaunchKernelGGL(k_opt_prepare, dim3(nchunks), dim3(256), 0, s, w, g, chunks, norms, hp);
}
// ||g||^2 of every tensor: the block of a tensor's FIRST chunk adds up the tensor's per-chunk partials in a fixed
// order (thread t takes partials t, t + 256, ...; block_sum is a fixed tree) -> tnorm[first chunk].  No float atomics:
// the clip factor, and with it the parameter update, is bit-identical on every replica of a data-parallel run.  (Every
// block of k_opt_apply used to walk all `count` partials itself: 4096 dependent loads per block for the 8 M-element
// Dense weights of the 256x256 configuration, 617 us per launch.)
__global__ void __launch_bounds__(256) k_opt_tnorm(const ChunkDesc* __restrict__ chunks, const float* __restrict__ norms,
                                                   float* __restrict__ tnorm) {
  __shared__ float sh[16];
  const ChunkDesc cd = chunks[blockIdx.x];
  if (cd.first != (int)blockIdx.x) return;             // block-uniform
  float acc = 0.f;
  for (int j = threadIdx.x; j < cd.count; j += 256) acc += norms[cd.first + j];
  const float t = block_sum(acc, sh);
  if (threadIdx.x == 0) tnorm[blockIdx.x] = t;
}
__global__ void __launch_bounds__(256) k_opt_apply(float* __restrict__ w, const float* __restrict__ g,
                                                   float* __restrict__ a, const ChunkDesc* __restrict__ chunks,
                                                   const float* __restrict__ tnorm, const float* __restrict__ hp, int clip) {
  ChunkDesc cd = chunks[blockIdx.x];
  const float lr = hp[HP_LR], clip_norm = hp[HP_CLIP];
  float f = 1.0f;
  if (clip) {
    const float nrm = sqrtf(tnorm[cd.first]);
    f = clip_norm / fmaxf(nrm, clip_norm);       // tf.clip_by_norm
  }
  for (int i = threadIdx.x; i < cd.len; i += blockDim.x) {
    int64_t o = cd.offset + i;
    float gv = g[o] * f;
    float av = a[o] + gv * gv;
    a[o] = av;
    w[o] -= lr * gv / (sqrtf(av) + 1e-7f);
  }
}
void launch_opt_apply(float* w, const float* g, float* a, const ChunkDesc* chunks, int nchunks, const float* norms,
                      const float* hp, bool clip, hipStream_t s) {
  ProfScope ps("optimizer", (double)(0.0), 0.0, s);
  float* tnorm = const_cast<float*>(norms) + nchunks;           // second half of the norms buffer (runtime: 2 x nchunks)
  if (clip) hipLaunchKernelGGL(k_opt_tnorm, dim3(nchunks), dim3(256), 0, s, chunks, norms, tnorm);
  hipLaunchKernelGGL(k_opt_apply, dim3(nchunks), dim3(256), 0, s, w, g, a, chunks, (const float*)tnorm, hp, clip ? 1 : 0);
}
__global__ void __launch_bounds__(256) k_reg_loss(const float* __restrict__ w, const ChunkDesc* __restrict__ chunks,
                                                  float* __restrict__ out) {
  __shared__ float sh[16];
  ChunkDesc cd = chunks[blockIdx.x];
  if (cd.reg == 0) return;
  float acc = 0.f;
  for (int i = threadIdx.x; i < cd.len; i += blockDim.x) {
    float wv = w[cd.offset + i];
    acc += cd.reg == 1 ? fabsf(wv) : wv * wv;
  }
  float t = block_sum(acc, sh);
  if (threadIdx.x == 0) atomicAdd(out, 0.01f * t);
}
void launch_reg_loss(const float* w, const ChunkDesc* chunks, int nchunks, float* out, hipStream_t s) {
  ProfScope ps("optimizer", (double)(0.0), 0.0, s);
  hipLaunchKernelGGL(k_reg_loss, dim3(nchunks), dim3(256), 0, s, w, chunks, out);
}
__global__ void k_state_update(float* __restrict__ state, const float* __restrict__ stats,
                               const StateDesc* __restrict__ descs, const float* __restrict__ hp, int B) {
  StateDesc d = descs[blockIdx.x];
  const float stat_scale = hp[HP_GRAD_SCALE];
  float corr = 1.0f;
  if (d.per_image > 0.f) {       // fused 4-D BatchNorm: moving variance takes the Bessel-corrected batch variance
    float n = d.per_image * (float)B;
    if (n > 1.f) corr = n / (n - 1.f);
  }
  for (int i = threadIdx.x; i < d.len; i += blockDim.x) {
    int64_t o = d.offset + i;
    state[o] = state[o] * d.momentum + stats[o] * stat_scale * corr * (1.0f - d.momentum);
  }
}
void launch_state_update(float* state, const float* stats, const StateDesc* descs, int ndesc, const float* hp,
                         int B, hipStream_t s) {
  ProfScope ps("optimizer", (double)(0.0), 0.0, s);
  if (ndesc <= 0) return;
  hipLaunchKernelGGL(k_state_update, dim3(ndesc), dim3(64), 0, s, state, stats, descs, hp, B);
}

// Gradient-slot copies (kernels.h: GradSlots) are laid out like the gradient arena, but only tensors of at most one
// chunk ever receive slot atomics: zero / fold just those chunks (a table of (offset, len)), not n x P floats.
__global__ void __launch_bounds__(256) k_slot_zero(const ChunkDesc* __restrict__ chunks, float* __restrict__ slots,
                                                   int64_t stride) {
  const ChunkDesc cd = chunks[blockIdx.x];
  float* p = slots + (int64_t)blockIdx.y * stride + cd.offset;          // tensor offsets are 256-byte aligned
  const int n4 = cd.len / 4;
  for (int i = threadIdx.x; i < n4; i += 256) reinterpret_cast<float4*>(p)[i] = float4{0.f, 0.f, 0.f, 0.f};
  for (int i = n4 * 4 + threadIdx.x; i < cd.len; i += 256) p[i] = 0.f;
}
__global__ void __launch_bounds__(256) k_slot_sum(const ChunkDesc* __restrict__ chunks, float* __restrict__ g,
                                                  const float* __restrict__ slots, int64_t stride, int n) {
  const ChunkDesc cd = chunks[blockIdx.x];
  float* gp = g + cd.offset;
  const float* sp = slots + cd.offset;
  const int n4 = cd.len / 4;
  for (int i = threadIdx.x; i < n4; i += 256) {
    float4 a = reinterpret_cast<float4*>(gp)[i];
#pragma unroll 8
    for (int k = 0; k < n; ++k) {
      const float4 v = reinterpret_cast<const float4*>(sp + (int64_t)k * stride)[i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    reinterpret_cast<float4*>(gp)[i] = a;
  }
  for (int i = n4 * 4 + threadIdx.x; i < cd.len; i += 256) {
    float a = gp[i];
    for (int k = 0; k < n; ++k) a += sp[(int64_t)k * stride + i];
    gp[i] = a;
  }
}

void launch_slot_zero(const ChunkDesc* chunks, int nchunks, float* slots, int64_t stride, int n, hipStream_t s) {
  ProfScope ps("slot_sum", 0.0, 0.0, s);
  hipLaunchKernelGGL(k_slot_zero, dim3(nchunks, n), dim3(256), 0, s, chunks, slots, stride);
}
void launch_slot_sum(const ChunkDesc* chunks, int nchunks, float* g, const float* slots, int64_t stride, int n,
                     hipStream_t s) {
  ProfScope ps("slot_sum", 0.0, 0.0, s);
  hipLaunchKernelGGL(k_slot_sum, dim3(nchunks), dim3(256), 0, s, chunks, g, slots, stride, n);
}

}  // namespace mvae
