// kernels_fused_fwd.hip -- float32 MobileNetV3 forward across a block boundary in ONE pass: conv2 of block k (squeeze-excite
// gate, bias, residual), conv0 of block k+1 (bias, ReLU) and the depthwise 3x3 + ReLU + global average pool of block k+1
// (layer_blocks.py:625-641, 594-623).  k_conv2_chain (kernels_mfma.hip) already computes the first two from one staged
// tile; what this kernel adds is that t0 of block k+1 -- stored, because the backward pass needs it -- is not read back by
// a depthwise launch: 5 tensor passes (t1, block input in; block output, t0', t1' out) instead of 6.  The float32 step is
// bound by the bytes it moves.
//
// Same skeleton as k_dw_bwd_conv0_s (kernels_fused.hip): C = 64, W = 32 / 16 / 8, one 512-thread block per CU walks whole
// images in 32-pixel tiles, split-bf16 products (split.h), a ring of t0' rows in LDS.  Software pipeline over tile pairs,
// two barriers per pass i:
//   phase A (thread = one float4 of a tile): gated t1 tiles of pair i -> bf16 planes; block output of pair i-1 (float32 in
//            LDS) -> bf16 planes;
//   phase B (matrix cores): waves 0-3: y = (t1 g) W2 + b2 + x for pair i (stored, and kept in LDS for the next phase A);
//            waves 4-7: t0' = relu(y W0' + b0') for pair i-1 (stored, and written into the ring);
//   phase C (thread = one float4): t1' = relu(dw3x3(t0') + bd') for tiles 2i-3, 2i-2 from the ring (stored), GAP sums.
#include "kernels.h"
#include "prof.h"
#include "split.h"
#include <cstdlib>

namespace mvae {

namespace {
constexpr int kFwdRing = 16 * 10 * 16 * 16;      // as kernels_fused.hip: W = 8: 16 slots x 10 px; W = 16: 8 x 18; W = 32: 4 x 34
constexpr int kFwdPlane = 32 * 128;
constexpr int kFwdLds = kFwdRing + 12 * kFwdPlane + 2 * 32 * 64 * 4 + 2 * 24 * 64 * 16 + 4 * 256;
}

// FIRST: the block that opens a chain (its input comes from a k x k convolution or conv_base, no conv2 ahead of it): conv0 +
// depthwise stage only, x -> t0, t1, gap (3 passes instead of 4).  Phase A then splits the INPUT tiles of pair i-1 where
// the chained form splits the block output, and waves 0-3 have nothing to do in phase B.
template <int W_, bool FIRST>
__global__ void __launch_bounds__(512, 1) k_mn_fwd_chain_s(const f32x4* __restrict__ t1, const float* __restrict__ gate,
                                                           const float* __restrict__ x, const float* __restrict__ W2,
                                                           const float* __restrict__ b2, const float* __restrict__ W0n,
                                                           const float* __restrict__ b0n, const f32x4* __restrict__ wdn,
                                                           const f32x4* __restrict__ bdn, float* __restrict__ y,
                                                           float* __restrict__ t0n, f32x4* __restrict__ t1n,
                                                           f32x4* __restrict__ gapn, int H, float inv_hw, int B) {
  constexpr int XSP = W_ + 2, RPT = 32 / W_, NS = W_ == 32 ? 4 : (W_ == 16 ? 8 : 16), TP = kFwdPlane, C = 64;
  static_assert(W_ == 32 || W_ == 16 || W_ == 8, "tile = one, two or four image rows");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  f32x4* ring = reinterpret_cast<f32x4*>(lds);                         // [NS slots][W + 2 px][16 quads]: t0' rows
  char* tA = lds + kFwdRing;                                           // [tile kk][plane][4096]: gated t1
  char* tYp = tA + 6 * TP;                                             // [tile kk][plane][4096]: block output (previous pair)
  float* tY = reinterpret_cast<float*>(tYp + 6 * TP);                  // [tile kk][32 px][64]: block output, float32
  u32x4* wf2 = reinterpret_cast<u32x4*>(tY + 2 * 32 * 64);             // W2 fragments: [nt][kq][plane][lane]
  u32x4* wf0 = wf2 + 24 * 64;                                          // W0' fragments
  float* sv = reinterpret_cast<float*>(wf0 + 24 * 64);                 // b2 [64], b0' [64], (spare)
  const int px = threadIdx.x >> 4, c4 = threadIdx.x & 15;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  {   // fragments of B[k = ci][n = co] = W[ci*64 + co]: wave w prepares (nt = w >> 2, kq = w & 3) of both matrices
    const int nt = wave >> 2, kq = wave & 3;
#pragma unroll
    for (int mtx = 0; mtx < 2; ++mtx) {
      const float* Wm = mtx ? W0n : W2;
      unsigned v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (FIRST && !mtx) ? 0u : __float_as_uint(Wm[(int64_t)(kq * 16 + 8 * h + j) * C + nt * 32 + i]);
      u32x2 a1, a2, a3, b1, b2_, b3;
      split4(u32x4{v[0], v[1], v[2], v[3]}, a1, a2, a3);
      split4(u32x4{v[4], v[5], v[6], v[7]}, b1, b2_, b3);
      u32x4* wf = mtx ? wf0 : wf2;
      wf[((nt * 4 + kq) * 3 + 0) * 64 + lane] = u32x4{a1[0], a1[1], b1[0], b1[1]};
      wf[((nt * 4 + kq) * 3 + 1) * 64 + lane] = u32x4{a2[0], a2[1], b2_[0], b2_[1]};
      wf[((nt * 4 + kq) * 3 + 2) * 64 + lane] = u32x4{a3[0], a3[1], b3[0], b3[1]};
    }
  }
  if (threadIdx.x < 64) { sv[threadIdx.x] = FIRST ? 0.f : b2[threadIdx.x]; sv[64 + threadIdx.x] = b0n[threadIdx.x]; }
  if (threadIdx.x < NS * 32) {                                         // columns 0 and W + 1 of every slot: always zero
    const int slot = threadIdx.x >> 5, side = (threadIdx.x >> 4) & 1;
    ring[(slot * XSP + (side ? W_ + 1 : 0)) * 16 + c4] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 wt[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wt[k] = wdn[k * 16 + c4];
  const f32x4 bdv = bdn[c4];
  const bool gemm1_wave = wave < 4;
  const int gkk = wave & 1, gnt = (wave >> 1) & 1;                     // product role: tile of the pair, output-channel half
  const int ry = px / W_, xc = px % W_;
  f32x4* rbase = ring + xc * 16 + c4;
  const int st_off = dual_off(px, c4 >> 1) + (c4 & 1) * 8;
  const int HW = H * W_, NT = HW / 32, NP = NT / 2;
  const float* gA_b2 = sv + gnt * 32 + i;                              // bias of the wave's output channel

  // the block's images form one stream of fetches (as in k_dw_bwd_conv0_s): tiles NT, NT + 1 of an image are tiles 0, 1 of
  // the block's next one, whose gate vector is requested an image ahead
  f32x4 T1[2];
  float R[16];
  f32x4 g_nx = FIRST ? f32x4{1.f, 1.f, 1.f, 1.f} : reinterpret_cast<const f32x4*>(gate)[(int64_t)blockIdx.x * 16 + c4];
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const bool first = b == (int)blockIdx.x;
    const int bn = b + (int)gridDim.x < B ? b + (int)gridDim.x : b;    // next image of this block (or a harmless refetch)
    const int64_t ioff = (int64_t)b * HW * 16, ioff_n = (int64_t)bn * HW * 16;   // float4 offsets of the two images
    const int64_t poff = (int64_t)b * HW, poff_n = (int64_t)bn * HW;   // pixel offsets
    const f32x4 gq = g_nx;
    if (!FIRST) g_nx = reinterpret_cast<const f32x4*>(gate)[(int64_t)bn * 16 + c4];
    auto fetch_t1 = [&](int t) {                                       // t >= NT: tile t - NT of the next image
      const bool nx = t >= NT;
      return t1[(nx ? ioff_n : ioff) + (int64_t)(min(nx ? t - NT : t, NT - 1) * 32 + px) * 16 + c4];
    };
    // residual x of the wave's y tile, accumulator layout (lane = channel, 16 pixel rows): waves 0-3
    auto fetch_res = [&](int t) {
      const bool nx = t >= NT;
      const float* pr = x + ((nx ? poff_n : poff) + (int64_t)min(nx ? t - NT : t, NT - 1) * 32 + 4 * h) * C + gnt * 32 + i;
#pragma unroll
      for (int r = 0; r < 16; ++r) R[r] = pr[((r & 3) + 8 * (r >> 2)) * C];
    };
    f32x4 gsum = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                  // previous image's LDS reads are done (and the fragments are written)
    if (first) {                                      // later images: fetched by the previous image's last pass
      T1[0] = fetch_t1(0);
      T1[1] = fetch_t1(1);
      if (!FIRST && gemm1_wave) fetch_res(gkk);
    }
    if (ry == 0) {                                    // image row -1: zeros
      rbase[(((-1) & (NS - 1)) * XSP + 1) * 16] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int it = FIRST ? 1 : 0; it <= NP + 1; ++it) {
      // ---- phase A
      if (!FIRST && it < NP) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          u32x2 p1, p2, p3;
          split4(__builtin_bit_cast(u32x4, T1[kk] * gq), p1, p2, p3);
          *reinterpret_cast<u32x2*>(tA + (kk * 3 + 0) * TP + st_off) = p1;
          *reinterpret_cast<u32x2*>(tA + (kk * 3 + 1) * TP + st_off) = p2;
          *reinterpret_cast<u32x2*>(tA + (kk * 3 + 2) * TP + st_off) = p3;
          T1[kk] = fetch_t1(2 * it + 2 + kk);
        }
      }
      if (it >= 1 && it <= NP) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const f32x4 yv = FIRST ? T1[kk] : reinterpret_cast<const f32x4*>(tY)[(kk * 32 + px) * 16 + c4];
          u32x2 p1, p2, p3;
          split4(__builtin_bit_cast(u32x4, yv), p1, p2, p3);
          *reinterpret_cast<u32x2*>(tYp + (kk * 3 + 0) * TP + st_off) = p1;
          *reinterpret_cast<u32x2*>(tYp + (kk * 3 + 1) * TP + st_off) = p2;
          *reinterpret_cast<u32x2*>(tYp + (kk * 3 + 2) * TP + st_off) = p3;
          if (FIRST) T1[kk] = fetch_t1(2 * it + kk);
        }
      }
      __syncthreads();
      // ---- phase B
      if (it == NP + 1 && ry == 0) {                  // image row H: zeros -- only now (phase C of pass NP has read row H - NS)
        rbase[((H & (NS - 1)) * XSP + 1) * 16] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (gemm1_wave) {
        if (!FIRST && it < NP) {
          const int t = 2 * it + gkk;                                  // the wave's tile
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int kq = 0; kq < 4; ++kq) {
            bf16x8 xa[3], wb[3];
            const int off = dual_off(i, 2 * kq + h);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
              xa[p] = as_frag(*reinterpret_cast<const u32x4*>(tA + (gkk * 3 + p) * TP + off));
              wb[p] = as_frag(wf2[((gnt * 4 + kq) * 3 + p) * 64 + lane]);
            }
            MVAE_SPLIT6(acc, xa, wb);
          }
          const float bias = gA_b2[0];
          float* py = y + (poff + (int64_t)t * 32 + 4 * h) * C + gnt * 32 + i;
          float* ps = tY + (gkk * 32 + 4 * h) * C + gnt * 32 + i;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[r] + bias + R[r];
            py[((r & 3) + 8 * (r >> 2)) * C] = v;
            ps[((r & 3) + 8 * (r >> 2)) * C] = v;
          }
          fetch_res(t + 2);
        }
      } else if (it >= 1 && it <= NP) {
        const int t = 2 * it - 2 + gkk;                                // the wave's tile of the previous pair
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
          bf16x8 xa[3], wb[3];
          const int off = dual_off(i, 2 * kq + h);
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            xa[p] = as_frag(*reinterpret_cast<const u32x4*>(tYp + (gkk * 3 + p) * TP + off));
            wb[p] = as_frag(wf0[((gnt * 4 + kq) * 3 + p) * 64 + lane]);
          }
          MVAE_SPLIT6(acc, xa, wb);
        }
        const float bias = gA_b2[64];
        float* pt = t0n + (poff + (int64_t)t * 32 + 4 * h) * C + gnt * 32 + i;
        float* rf = reinterpret_cast<float*>(ring) + gnt * 32 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int pl = (r & 3) + 8 * (r >> 2) + 4 * h;               // pixel inside the tile
          const float v = fmaxf(acc[r] + bias, 0.f);
          pt[((r & 3) + 8 * (r >> 2)) * C] = v;
          const int row = t * RPT + pl / W_, col = pl % W_;
          rf[((row & (NS - 1)) * XSP + col + 1) * 64] = v;
        }
      }
      __syncthreads();
      // ---- phase C: depthwise 3x3 + bias + ReLU of tiles 2 it - 3, 2 it - 2 (their neighbours' rows are in the ring)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int t = 2 * it - 3 + kk;
        if (t >= 0 && t < NT) {
          const int yrow = t * RPT + ry;
          f32x4 acc = bdv;
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const f32x4* rrow = rbase + ((yrow + a - 1) & (NS - 1)) * (XSP * 16);
#pragma unroll
            for (int e = 0; e < 3; ++e) acc += wt[a * 3 + e] * rrow[e * 16];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = fmaxf(acc[q], 0.f);
          t1n[ioff + (int64_t)(t * 32 + px) * 16 + c4] = acc;
          gsum += acc;
        }
      }
    }
    // ---- global average pool of the image: lanes l, l + 16, l + 32, l + 48 share the channel quad, then the 8 waves
    __syncthreads();
    {
      f32x4 v = gsum;
      for (int off = 16; off < 64; off <<= 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += __shfl_xor(v[q], off, 64);
      }
      f32x4* red = reinterpret_cast<f32x4*>(tA);
      if (lane < 16) red[wave * 16 + lane] = v;
      __syncthreads();
      if (threadIdx.x < 16) {
        f32x4 s = red[threadIdx.x];
#pragma unroll
        for (int wv = 1; wv < 8; ++wv) s += red[wv * 16 + threadIdx.x];
        gapn[(int64_t)b * 16 + threadIdx.x] = s * inv_hw;
      }
    }
  }
}

// conv2 (gate, bias, residual) of a C = 64 block, conv0 (bias, ReLU) and depthwise 3x3 + ReLU + GAP of the next one.
// nullptr / false = shape not covered or switched off (the caller then runs k_conv2_chain and the depthwise launch).
const char* mn_fwd_chain_split_kernel(int B, int H, int W, int C) {
  static const int on = [] { const char* e = getenv("MVAE_FUSE_MN_FWD"); return e ? atoi(e) : 1; }();
  if (!on || split_conv_status() != 1 || det_mode()) return nullptr;
  if (C != 64 || (W != 32 && W != 16 && W != 8) || (H * W) % 64 != 0 || H < 4 || B < 1) return nullptr;
  if ((int64_t)B * H * W >= (1LL << 31) / 64) return nullptr;
  static const bool attr =
      hipFuncSetAttribute((const void*)k_mn_fwd_chain_s<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds) == hipSuccess &&
      hipFuncSetAttribute((const void*)k_mn_fwd_chain_s<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds) == hipSuccess &&
      hipFuncSetAttribute((const void*)k_mn_fwd_chain_s<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds) == hipSuccess &&
      hipFuncSetAttribute((const void*)k_mn_fwd_chain_s<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds) == hipSuccess &&
      hipFuncSetAttribute((const void*)k_mn_fwd_chain_s<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds) == hipSuccess &&
      hipFuncSetAttribute((const void*)k_mn_fwd_chain_s<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kFwdLds) == hipSuccess;
  if (!attr) return nullptr;
  return "k_mn_fwd_chain_s";
}
bool launch_mn_fwd_chain_split(const float* t1, const float* gate, const float* x, const float* W2, const float* b2,
                               const float* W0n, const float* b0n, const float* wdn, const float* bdn, float* y, float* t0n,
                               float* t1n, float* gapn, int B, int H, int W, int C, hipStream_t s) {
  if (!mn_fwd_chain_split_kernel(B, H, W, C)) return false;
  static const int cus32 = [] { const char* e = getenv("MVAE_FUSED_CUS"); int n = e ? atoi(e) : 256; return n < 8 ? 8 : (n > 256 ? 256 : n); }();
  static const int cus16 = [] { const char* e = getenv("MVAE_FUSED_CUS16"); int n = e ? atoi(e) : 256; return n < 8 ? 8 : (n > 256 ? 256 : n); }();
  // 8-wide maps on half the CUs: these launches are latency-bound (four images per block cost little) and a block of this
  // kernel has its CU to itself, so the other half of the chip stays open to the other scales' streams (4.98 -> 4.94 ms)
  static const int cus8 = [] { const char* e = getenv("MVAE_FUSED_CUS8"); int n = e ? atoi(e) : 128; return n < 8 ? 8 : (n > 256 ? 256 : n); }();
  const int cus = W == 32 ? cus32 : (W == 16 ? cus16 : cus8);
  const int grid = B < cus ? B : cus;
  fused_launch_note(true, B, grid);
#define MVAE_FF(WW, FI)                                                                                                  \
  hipLaunchKernelGGL((k_mn_fwd_chain_s<WW, FI>), dim3(grid), dim3(512), kFwdLds, s, (const f32x4*)t1, gate, x, W2, b2, W0n, \
                     b0n, (const f32x4*)wdn, (const f32x4*)bdn, y, t0n, (f32x4*)t1n, (f32x4*)gapn, H, 1.0f / (float)(H * W), B)
  if (W == 32) MVAE_FF(32, false); else if (W == 16) MVAE_FF(16, false); else MVAE_FF(8, false);
  return true;
}
bool mn_fwd_first_split_on() {
  static const bool on = [] { const char* e = getenv("MVAE_FUSE_MN_FWD"); return e ? atoi(e) != 2 : true; }();   // 2: chained form only
  return on;
}
// the block that opens a chain: conv0 (bias, ReLU) + depthwise 3x3 + ReLU + GAP from the block input, x -> t0, t1, gap
bool launch_mn_fwd_first_split(const float* x_in, const float* W0, const float* b0, const float* wd, const float* bd, float* t0_out,
                               float* t1_out, float* gap_out, int B, int H, int W, int C, hipStream_t s) {
  if (!mn_fwd_first_split_on() || !mn_fwd_chain_split_kernel(B, H, W, C)) return false;
  static const int cus32 = [] { const char* e = getenv("MVAE_FUSED_CUS"); int n = e ? atoi(e) : 256; return n < 8 ? 8 : (n > 256 ? 256 : n); }();
  static const int cus16 = [] { const char* e = getenv("MVAE_FUSED_CUS16"); int n = e ? atoi(e) : 256; return n < 8 ? 8 : (n > 256 ? 256 : n); }();
  // 8-wide maps on half the CUs: these launches are latency-bound (four images per block cost little) and a block of this
  // kernel has its CU to itself, so the other half of the chip stays open to the other scales' streams (4.98 -> 4.94 ms)
  static const int cus8 = [] { const char* e = getenv("MVAE_FUSED_CUS8"); int n = e ? atoi(e) : 128; return n < 8 ? 8 : (n > 256 ? 256 : n); }();
  const int cus = W == 32 ? cus32 : (W == 16 ? cus16 : cus8);
  const int grid = B < cus ? B : cus;
  fused_launch_note(true, B, grid);
  const float* gate = nullptr; const float* W2 = nullptr; const float* b2 = nullptr; float* y = nullptr;
  const float* x = nullptr;
  const float* W0n = W0; const float* b0n = b0; const float* wdn = wd; const float* bdn = bd;
  float* t0n = t0_out; float* t1n = t1_out; float* gapn = gap_out;
  const float* t1 = x_in;                                              // the kernel's thread-layout input stream
  if (W == 32) MVAE_FF(32, true); else if (W == 16) MVAE_FF(16, true); else MVAE_FF(8, true);
#undef MVAE_FF
  return true;
}

}  // namespace mvae
