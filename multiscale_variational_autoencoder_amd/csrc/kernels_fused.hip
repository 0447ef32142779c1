// kernels_fused.hip -- float32 MobileNetV3 backward: the depthwise backward (k_dw_bwd_ring<true, float>, kernels_dw.hip) and
// conv0's backward pair (k_gemm_dual_s<2>, kernels_split.hip) in ONE pass, so that dt0 -- the depthwise backward's output
// and the pair's input -- never goes to HBM: 5 tensor passes per block element (dt2, t0, block input, dout in; da out)
// instead of 7.  The float32 step is bound by its total HBM traffic (every large kernel moves its
// algorithmic bytes at 4.5-5.2 TB/s, faster kernels did not shorten the step any more), so a pass not made is the lever
// left.  layer_blocks.py:594-623 inverted; the bf16 counterpart is k16_dw_bwd_conv0 (kernels_bf16.hip).
//
// With float32 MFMAs this fusion lost (round 2: 177-182 us against 74 + 100 separately -- 55 us of matrix-core time that
// did not overlap the ring phase); with split-bf16 products (split.h) the two GEMMs of a row pair are 48 bf16 MFMAs per
// wave and the kernel is bound by memory again.
//
// Shape: C = 64, W = 32, 16 or 8 (a 32-pixel MFMA tile is one, two or four whole image rows: no column halo), H W % 64 == 0.  One 512-thread block
// per CU walks whole images top to bottom:
//   thread (px = t >> 4, c4 = t & 15) owns one float4 of every row: d1 rows go HBM -> register FIFO -> 4-row LDS ring,
//   dt0[y] = dwT(d1[y-1 .. y+1]) * (t0[y] > 0) is formed from the ring (3x3 taps), split into three bf16 planes and
//   written to the row's LDS tile, the block-input row likewise; the depthwise weight / bias gradients accumulate in
//   registers.  Every two rows the eight waves run the pair's products on the two tiles:
//     waves 0-3 (row kk, channel tile nt):  da = dt0 . W0^T + dout          24 MFMAs, W0 fragments (pre-split) from LDS
//     waves 4-7 (ci tile a, co tile b):     P[ci][co] += a^T dt0 over 64 px  24 MFMAs, ds_read_b64_tr_b16 fragments
// LDS: ring 34 KB + dt0 planes 24 KB + input planes 24 KB + W0 fragments 24 KB + dout rows 16 KB = 122 KB.
// Rows -1 and H of the ring are stored as zeros, so the tap loops carry no bounds tests: the row loop is straight-line code
// (counted vmcnt waits).
#include "kernels.h"
#include "prof.h"
#include "split.h"
#include <cstdlib>

namespace mvae {

namespace {
constexpr int kFusedRing = 16 * 10 * 16 * 16;    // bytes: W = 8: 16 slots x 10 px; W = 16: 8 x 18 (36864); W = 32: 4 x 34 (34816)
constexpr int kFusedPlane = 32 * 128;            // one bf16 plane of a 32-pixel row
constexpr int kFusedLds = kFusedRing + 12 * kFusedPlane + 24 * 64 * 16 + 2 * 32 * 64 * 4 + 11 * 16 * 16;
}

template <int W_>
__global__ void __launch_bounds__(512, 1) k_dw_bwd_conv0_s(const f32x4* __restrict__ dt2, const f32x4* __restrict__ t0,
                                                           const f32x4* __restrict__ w, const f32x4* __restrict__ gate,
                                                           const f32x4* __restrict__ dgap, const float* __restrict__ W0,
                                                           const f32x4* __restrict__ a_in, const float* __restrict__ dout,
                                                           float* __restrict__ da, float* __restrict__ dW,
                                                           float* __restrict__ db, float* __restrict__ dW0,
                                                           float* __restrict__ db0, int H, float inv_hw, int B, int nslots,
                                                           int64_t slot_stride) {
  // a tile = 32 consecutive pixels = RPT image rows; the ring keeps NS rows (compute(t - 1) may still read rows down to
  // t*RPT - RPT - 1 while the rows of step t are stored: NS >= 2 RPT + 2)
  constexpr int XSP = W_ + 2, RPT = 32 / W_, NS = W_ == 32 ? 4 : (W_ == 16 ? 8 : 16), TP = kFusedPlane, C = 64;
  static_assert(W_ == 32 || W_ == 16 || W_ == 8, "tile = one, two or four image rows");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  f32x4* ring = reinterpret_cast<f32x4*>(lds);                         // [NS slots][W + 2 px][16 quads]: d1, columns 0 / W + 1 zero
  char* tD = lds + kFusedRing;                                         // [row kk][plane][4096]: dt0
  char* tA = tD + 6 * TP;                                              // [row kk][plane][4096]: block input
  u32x4* wfl = reinterpret_cast<u32x4*>(tA + 6 * TP);                  // [nt][kq][plane][lane]: W0^T fragments
  float* tR = reinterpret_cast<float*>(tA + 6 * TP + 24 * 64 * 16);    // [row kk][32 px][64]: dout (the residual of da)
  // depthwise weights [9][16 quads], then the image's gate and dgap / HW: read from LDS at every use -- as registers (44 per
  // thread) they pushed the kernel over 256 VGPRs and hipcc spilled inside the row loop
  f32x4* wl = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(tR) + 2 * 32 * 64 * 4);
  const int px = threadIdx.x >> 4, c4 = threadIdx.x & 15;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
#define RINGF(slot, xs, c4_) ring[((slot) * XSP + (xs)) * 16 + (c4_)]
  {   // wave w prepares fragment (nt = w >> 2, kq = w & 3): Wt[k = co][n = ci] = W0[ci*64 + co]
    const int nt = wave >> 2, kq = wave & 3;
    const f32x4* wp = reinterpret_cast<const f32x4*>(W0 + (int64_t)(nt * 32 + i) * C + kq * 16 + 8 * h);
    u32x2 a1, a2, a3, b1, b2, b3;
    split4(__builtin_bit_cast(u32x4, wp[0]), a1, a2, a3);
    split4(__builtin_bit_cast(u32x4, wp[1]), b1, b2, b3);
    wfl[((nt * 4 + kq) * 3 + 0) * 64 + lane] = u32x4{a1[0], a1[1], b1[0], b1[1]};
    wfl[((nt * 4 + kq) * 3 + 1) * 64 + lane] = u32x4{a2[0], a2[1], b2[0], b2[1]};
    wfl[((nt * 4 + kq) * 3 + 2) * 64 + lane] = u32x4{a3[0], a3[1], b3[0], b3[1]};
  }
  if (threadIdx.x < NS * 32) {                                         // columns 0 and W + 1 of every slot: always zero
    const int slot = threadIdx.x >> 5, side = (threadIdx.x >> 4) & 1;
    RINGF(slot, side ? W_ + 1 : 0, c4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  static_assert(NS * 32 <= 512, "one thread per zero cell");
  f32x4 aw[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) aw[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (threadIdx.x < 9 * 16) wl[threadIdx.x] = w[threadIdx.x];
  const f32x4* wlc = wl + c4;
  f32x4 ab = {0.f, 0.f, 0.f, 0.f}, bs0 = {0.f, 0.f, 0.f, 0.f};
  f32x16 accw;                                                         // waves 4-7: P tile; waves 0-3: the da tile
#pragma unroll
  for (int r = 0; r < 16; ++r) accw[r] = 0.f;
  const bool gemm_wave = wave < 4;
  const int ykk = wave & 1, ynt = (wave >> 1) & 1;                     // data-GEMM role: tile row, output-channel tile
  const int pa = wave & 1, pb = (wave >> 1) & 1;                       // weight-gradient role: ci tile, co tile
  // every LDS access of the row loop is one of these bases plus a compile-time offset
  const int ry = px / W_, xc = px % W_;                                // this thread's pixel inside a tile: image row ry, column xc
  f32x4* rbase = ring + xc * 16 + c4;                                  // ring[slot][xc + xs][c4] = rbase[(slot*XSP + xs)*16]
  char* stD = tD + dual_off(px, c4 >> 1) + (c4 & 1) * 8;               // this thread's 8 bytes inside a plane of tD (tA = +6 planes)
  f32x4* stR = reinterpret_cast<f32x4*>(tR) + px * 16 + c4;
  const char* gA[4];                                                   // data GEMM: A fragments of the wave's dt0 row, k-step kq
#pragma unroll
  for (int kq = 0; kq < 4; ++kq) gA[kq] = tD + ykk * 3 * TP + dual_off(i, 2 * kq + h);
  const u32x4* gW = wfl + ynt * 12 * 64 + lane;
  const float* gR = tR + (ykk * 32 + 4 * h) * C + ynt * 32 + i;
  // weight gradient: the two 8-byte pieces of a transposed fragment (rows q and q + 4 of the 16-lane group's 8 rows)
  const char *pA0, *pA1, *pD0, *pD1;
  {
    const int g16 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, rb = 8 * (g16 >> 1);
    const int ca = pa * 32 + 16 * (g16 & 1) + 4 * pp, cd = pb * 32 + 16 * (g16 & 1) + 4 * pp;
    pA0 = tA + dual_off(rb + q, ca >> 3) + (ca & 7) * 2;
    pA1 = tA + dual_off(rb + 4 + q, ca >> 3) + (ca & 7) * 2;
    pD0 = tD + dual_off(rb + q, cd >> 3) + (cd & 7) * 2;
    pD1 = tD + dual_off(rb + 4 + q, cd >> 3) + (cd & 7) * 2;
  }
  auto tr_frag = [](const char* o0, const char* o1) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)o0);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)o1);
    s16x8 f = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, f);
  };

  // The block's images form ONE stream of fetches: what the last two steps of an image would fetch past its end (row sets
  // NT, NT + 1 and t0 tiles NT, NT + 1) are the next image's first sets / tiles, and its gate / dgap values are requested an
  // image ahead -- an image then starts with its data in registers instead of a 2-3 us round trip (at batch 512 a block
  // walks two images: 16 x 16 maps ran at 3.85, 8 x 8 maps at 1.95 TB/s against 4.65 on 32 x 32 ones).
  const int HW = H * W_, NT = HW / 32;
  const float* gate_f = reinterpret_cast<const float*>(gate);
  const float* dgap_f = reinterpret_cast<const float*>(dgap);
  float g_nx = gate_f[(int64_t)blockIdx.x * 64 + lane], d_nx = dgap_f[(int64_t)blockIdx.x * 64 + lane];
  f32x4 Fd[2], T[2];
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const bool first = b == (int)blockIdx.x;
    const int bn = b + (int)gridDim.x < B ? b + (int)gridDim.x : b;    // next image of this block (or a harmless refetch)
    const int64_t ioff = (int64_t)b * HW * 16, ioff_n = (int64_t)bn * HW * 16;   // float4 offsets of the two images
    const f32x4* dout4 = reinterpret_cast<const f32x4*>(dout);
    // row set j = the RPT image rows j*RPT + 1 .. j*RPT + RPT (one element per thread): pixel j*32 + W + px of the image.
    // Step t stores set t into the ring (rows outside the image as zeros) and computes tile t from rows t*RPT - 1 .. t*RPT + RPT.
    auto fetch_set = [&](int j) {                                      // j >= NT: set j - NT - 1 of the next image
      const bool nx = j >= NT;
      const int p = (nx ? j - NT - 1 : j) * 32 + W_ + px;
      return dt2[(nx ? ioff_n : ioff) + (int64_t)min(max(p, 0), HW - 1) * 16 + c4];
    };
    auto store_set = [&](int j, const f32x4 rd) {
      const int p = j * 32 + W_ + px, row = j * RPT + 1 + ry;
      const bool inside = p >= 0 && p < HW;
      const f32x4 gg_c = wlc[9 * 16], dg_c = wlc[10 * 16];
      f32x4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned bits = __float_as_uint(rd[q]);
        v[q] = (inside && (bits & 1u)) ? __uint_as_float(bits & ~1u) * gg_c[q] + dg_c[q] : 0.f;   // ReLU mask t1 > 0 in the LSB
      }
      rbase[((row & (NS - 1)) * XSP + 1) * 16] = v;
    };
    auto fetch_t0 = [&](int t) {                                       // t >= NT: tile t - NT of the next image
      const bool nx = t >= NT;
      return t0[(nx ? ioff_n : ioff) + (int64_t)(min(nx ? t - NT : t, NT - 1) * 32 + px) * 16 + c4];
    };
    __syncthreads();                                  // previous image's ring / tile reads are done (and wfl is written)
    if (threadIdx.x < 64) {
      reinterpret_cast<float*>(wl + 9 * 16)[lane] = g_nx;
      reinterpret_cast<float*>(wl + 10 * 16)[lane] = d_nx * inv_hw;
    }
    g_nx = gate_f[(int64_t)bn * 64 + lane];
    d_nx = dgap_f[(int64_t)bn * 64 + lane];
    if (first) {                                      // later images: fetched by the previous image's last two steps
      Fd[1] = fetch_set(-1);
      Fd[0] = fetch_set(0);
      T[0] = fetch_t0(0);
      T[1] = fetch_t0(1);
    }
    __syncthreads();                                  // gate / dgap of this image are in LDS
    store_set(-2, Fd[1]);                             // rows < 0: zeros whatever the data
    store_set(-1, Fd[1]);
    Fd[1] = Fd[0];
    Fd[0] = fetch_set(1);
    // (unrolling this loop over four steps, to make the ring slots compile-time constants, doubled hipcc's register demand and
    // spilled 100 registers inside the loop: the slot offsets are run-time values, three pointer adds per step)
    {
#pragma unroll 1
      for (int t2 = 0; t2 < NT; t2 += 2) {
        const int64_t prow = (int64_t)b * HW + t2 * 32;                // pixel index of the first tile
        const f32x4 la0 = a_in[(prow + px) * 16 + c4];
        const f32x4 la1 = a_in[(prow + 32 + px) * 16 + c4];
        const f32x4 lr0 = dout4[(prow + px) * 16 + c4];
        const f32x4 lr1 = dout4[(prow + 32 + px) * 16 + c4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int t = t2 + kk, y = t * RPT + ry;                     // this thread's output pixel: (y, xc)
          store_set(t, Fd[(kk + 1) & 1]);
          Fd[(kk + 1) & 1] = fetch_set(t + 2);
          __syncthreads();
          const f32x4 tvk = T[kk];
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const f32x4* rrow = rbase + ((y - a + 1) & (NS - 1)) * (XSP * 16);
#pragma unroll
            for (int e = 0; e < 3; ++e) {
              const f32x4 sv = rrow[(2 - e) * 16];
              acc += wlc[(a * 3 + e) * 16] * sv;
              aw[a * 3 + e] += tvk * sv;
            }
          }
          ab += rbase[((y & (NS - 1)) * XSP + 1) * 16];
          f32x4 rv;
#pragma unroll
          for (int q = 0; q < 4; ++q) rv[q] = tvk[q] > 0.f ? acc[q] : 0.f;
          bs0 += rv;
          u32x2 p1, p2, p3;
          split4(__builtin_bit_cast(u32x4, rv), p1, p2, p3);
          *reinterpret_cast<u32x2*>(stD + (kk * 3 + 0) * TP) = p1;
          *reinterpret_cast<u32x2*>(stD + (kk * 3 + 1) * TP) = p2;
          *reinterpret_cast<u32x2*>(stD + (kk * 3 + 2) * TP) = p3;
          T[kk] = fetch_t0(t + 2);
        }
        {
          u32x2 p1, p2, p3;
          split4(__builtin_bit_cast(u32x4, la0), p1, p2, p3);
          *reinterpret_cast<u32x2*>(stD + 6 * TP) = p1;                // tA = tD + 6 planes
          *reinterpret_cast<u32x2*>(stD + 7 * TP) = p2;
          *reinterpret_cast<u32x2*>(stD + 8 * TP) = p3;
          split4(__builtin_bit_cast(u32x4, la1), p1, p2, p3);
          *reinterpret_cast<u32x2*>(stD + 9 * TP) = p1;
          *reinterpret_cast<u32x2*>(stD + 10 * TP) = p2;
          *reinterpret_cast<u32x2*>(stD + 11 * TP) = p3;
          stR[0] = lr0;
          stR[32 * 16] = lr1;
        }
        __syncthreads();                              // all tiles of the row pair complete
        if (gemm_wave) {
          // ---- da tile: tile t2 + ykk, output channels 32 ynt ..   (A = dt0 pixels, B = W0^T fragments)
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int kq = 0; kq < 4; ++kq) {
            bf16x8 xa[3], wb[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
              xa[p] = as_frag(*reinterpret_cast<const u32x4*>(gA[kq] + p * TP));
              wb[p] = as_frag(gW[(kq * 3 + p) * 64]);
            }
            MVAE_SPLIT6(acc, xa, wb);
          }
          float* py = da + (prow + ykk * 32 + 4 * h) * C + ynt * 32 + i;
#pragma unroll
          for (int r = 0; r < 16; ++r) py[((r & 3) + 8 * (r >> 2)) * C] = acc[r] + gR[((r & 3) + 8 * (r >> 2)) * C];
        } else {
          // ---- P[ci][co] += a^T dt0 over the 64 pixels of the pair
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              bf16x8 fa[3], fb[3];
#pragma unroll
              for (int p = 0; p < 3; ++p) {
                fa[p] = tr_frag(pA0 + (kk * 3 + p) * TP + half * 2048, pA1 + (kk * 3 + p) * TP + half * 2048);
                fb[p] = tr_frag(pD0 + (kk * 3 + p) * TP + half * 2048, pD1 + (kk * 3 + p) * TP + half * 2048);
              }
              MVAE_SPLIT6(accw, fa, fb);
            }
        }
      }
    }
  }
  // ---- depthwise weight / bias gradients and db0: lanes l, l + 16, l + 32, l + 48 share the channel quad
  __syncthreads();
  f32x4* red = reinterpret_cast<f32x4*>(tD);                           // [8 waves][11][16 quads] = 22.5 KB <= tD + tA
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    f32x4 v = k < 9 ? aw[k < 9 ? k : 0] : (k == 9 ? ab : bs0);
    for (int off = 16; off < 64; off <<= 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] += __shfl_xor(v[q], off, 64);
    }
    if (lane < 16) red[(wave * 11 + k) * 16 + lane] = v;
  }
  __syncthreads();
  const int64_t slot = (int64_t)(blockIdx.x % nslots) * slot_stride;
  for (int idx = threadIdx.x; idx < 11 * 64; idx += 512) {
    const int q = idx & 3, cc = (idx >> 2) & 15, k = idx >> 6;
    float t = 0.f;
#pragma unroll
    for (int wv = 0; wv < 8; ++wv) t += red[(wv * 11 + k) * 16 + cc][q];
    float* dst = k < 9 ? dW + slot + (int64_t)k * 64 : (k == 9 ? db + slot : (db0 ? db0 + slot : nullptr));
    if (dst) atomicAdd(dst + cc * 4 + q, t);
  }
  // ---- conv0 weight gradient: the four P tiles through LDS (dW0[ci][co] row-major), one coalesced atomic set per block
  __syncthreads();
  float* redw = reinterpret_cast<float*>(lds);                         // 16 KB of the ring
  if (!gemm_wave) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = pa * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, co = pb * 32 + i;
      redw[ci * 64 + co] = accw[r];
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 64 * 64; idx += 512) atomicAdd(&dW0[slot + idx], redw[idx]);
#undef RINGF
}

// dt2 (float32, ReLU mask of t1 in the mantissa LSB) -> da, dW_dw, db_dw, dW0, db0.  false = shape not covered or switched
// off (the caller then runs launch_dw_bwd_fused and launch_gemm_dual_mfma).
const char* dw_bwd_conv0_split_kernel(int B, int H, int W, int C) {
  static const int on = [] { const char* e = getenv("MVAE_FUSE_DW_CONV0_F32"); return e ? atoi(e) : 1; }();   // 2: 32-wide maps only, 3: 32 and 16
  if (!on || split_conv_status() != 1 || (on == 2 && W != 32) || (on == 3 && W < 16)) return nullptr;
  if (C != 64 || (W != 32 && W != 16 && W != 8) || (H * W) % 64 != 0 || H < 4 || B < 1) return nullptr;
  if ((int64_t)B * H * W >= (1LL << 31) / 64) return nullptr;
  return "k_dw_bwd_conv0_s";
}
bool launch_dw_bwd_conv0_split(const float* dt2, const float* t0, const float* w, const float* gate, const float* dgap,
                               const float* W0, const float* a_in, const float* dout, float* da, float* dW, float* db,
                               float* dW0, float* db0, GradSlots sl, int B, int H, int W, int C, hipStream_t s) {
  if (!dw_bwd_conv0_split_kernel(B, H, W, C)) return false;
  static const bool attr =
      hipFuncSetAttribute((const void*)k_dw_bwd_conv0_s<32>, hipFuncAttributeMaxDynamicSharedMemorySize, kFusedLds) == hipSuccess &&
      hipFuncSetAttribute((const void*)k_dw_bwd_conv0_s<16>, hipFuncAttributeMaxDynamicSharedMemorySize, kFusedLds) == hipSuccess &&
      hipFuncSetAttribute((const void*)k_dw_bwd_conv0_s<8>, hipFuncAttributeMaxDynamicSharedMemorySize, kFusedLds) == hipSuccess;
  if (!attr) return false;
  static const int cus32 = [] { const char* e = getenv("MVAE_FUSED_CUS"); int n = e ? atoi(e) : 256; return n < 8 ? 8 : (n > 256 ? 256 : n); }();
  static const int cus16 = [] { const char* e = getenv("MVAE_FUSED_CUS16"); int n = e ? atoi(e) : 256; return n < 8 ? 8 : (n > 256 ? 256 : n); }();
  // 8-wide maps on half the CUs: these launches are latency-bound (four images per block cost little) and a block of this
  // kernel has its CU to itself, so the other half of the chip stays open to the other scales' streams (4.98 -> 4.94 ms)
  static const int cus8 = [] { const char* e = getenv("MVAE_FUSED_CUS8"); int n = e ? atoi(e) : 128; return n < 8 ? 8 : (n > 256 ? 256 : n); }();
  const int cus = W == 32 ? cus32 : (W == 16 ? cus16 : cus8);
  int grid = B < cus ? B : cus;
  if (det_mode() && grid > kDetSlots) grid = kDetSlots;
  fused_launch_note(false, B, grid);
#define MVAE_FB(WW)                                                                                                      \
  hipLaunchKernelGGL(k_dw_bwd_conv0_s<WW>, dim3(grid), dim3(512), kFusedLds, s, (const f32x4*)dt2, (const f32x4*)t0,    \
                     (const f32x4*)w, (const f32x4*)gate, (const f32x4*)dgap, W0, (const f32x4*)a_in, dout, da, sl.at(dW), \
                     sl.at(db), sl.at(dW0), sl.at(db0), H, 1.0f / (float)(H * W), B, sl.count(), sl.stride)
  if (W == 32) MVAE_FB(32); else if (W == 16) MVAE_FB(16); else MVAE_FB(8);
#undef MVAE_FB
  return true;
}

// =================================================================================================
// The same pass one step further: dt2 = dout . W2^T is RECOMPUTED here, row set by row set, on the matrix cores -- dout is
// read anyway (the residual of da) -- so the conv2 pair (k_gemm_dual_s<3>) stores no dt2 and this kernel reads none: the
// block's backward moves 2 + 4 tensor passes (dout, t1 | dout, t0, block input, da) instead of 3 + 5.  What the conv2 pair
// leaves behind instead is the ReLU mask of t1 as two 32-bit words per pixel (1/32 of a pass).
// Per step (one row set = 32 pixels): the threads split their dout float4 into planes and keep the raw values in a
// 96-pixel LDS ring (the residual of the da tile one or two steps later); all eight waves then run the set's 32 x 64 x 64
// product as 16 x 16 tiles (wave = pixel half x channel quarter, 12 v_mfma_f32_16x16x32_bf16) and write
// d1 = mask ? dt2 * gate + dgap / HW : 0 straight from the accumulators into the ring; from there on the kernel is
// k_dw_bwd_conv0_s.  LDS: ring 36 KB + dt0 planes 24 + input planes 24 (the dout planes of a set live in their first half
// until the set's product is done) + W0 and W2 fragments 48 + raw dout 24 + small = 159 KB; W = 32 or 16.
// =================================================================================================
#define MVAE_SPLIT6_16(ACC, A, B)                                                      \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2], B[0], ACC, 0, 0, 0);             \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], B[1], ACC, 0, 0, 0);             \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[2], ACC, 0, 0, 0);             \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], B[0], ACC, 0, 0, 0);             \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[1], ACC, 0, 0, 0);             \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[0], ACC, 0, 0, 0)
namespace {
constexpr int kMnRing = 8 * 18 * 16 * 16;
constexpr int kMnLds = kMnRing + 4 * 6 * kFusedPlane + 97 * 256 + 11 * 256 + 256;   // raw dout: 96 pixels + one dummy
}

template <int W_>
__global__ void __launch_bounds__(512, 1) k_mn_bwd_s(const f32x4* __restrict__ dout4, const unsigned* __restrict__ mask,
                                                     const f32x4* __restrict__ t0, const f32x4* __restrict__ w,
                                                     const f32x4* __restrict__ gate, const f32x4* __restrict__ dgap,
                                                     const float* __restrict__ W2, const float* __restrict__ W0,
                                                     const f32x4* __restrict__ a_in, float* __restrict__ da,
                                                     float* __restrict__ dW, float* __restrict__ db, float* __restrict__ dW0,
                                                     float* __restrict__ db0, int H, float inv_hw, int B, int nslots,
                                                     int64_t slot_stride) {
  constexpr int XSP = W_ + 2, RPT = 32 / W_, NS = W_ == 32 ? 4 : 8, TP = kFusedPlane, C = 64;
  static_assert(W_ == 32 || W_ == 16, "tile = one or two image rows");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  f32x4* ring = reinterpret_cast<f32x4*>(lds);                         // [NS slots][W + 2 px][16 quads]: d1
  char* tD = lds + kMnRing;                                            // [tile kk][plane][4096]: dt0
  char* tA = tD + 6 * TP;                                              // [tile kk][plane][4096]: block input; [0..2]: dout planes of a set
  u32x4* wfl = reinterpret_cast<u32x4*>(tA + 6 * TP);                  // W0^T fragments (32x32x16): [nt][kq][plane][lane]
  u32x4* wf2 = wfl + 24 * 64;                                          // W2^T fragments (16x16x32): [q][ks][plane][lane]
  f32x4* tR = reinterpret_cast<f32x4*>(wf2 + 24 * 64);                 // raw dout, ring of 96 pixels
  f32x4* wl = tR + 97 * 16;                                            // depthwise weights [9][16], gate [16], dgap / HW [16]
  unsigned* mk = reinterpret_cast<unsigned*>(wl + 11 * 16);            // mask words of the current set: [32 px][2]
  const int px = threadIdx.x >> 4, c4 = threadIdx.x & 15;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int l16 = lane & 15, g4 = lane >> 4, ph = wave & 1, q4 = wave >> 1;      // set product: pixel half, channel quarter
  const int ci1 = q4 * 16 + l16;                                                 // its output channel (a conv2 INPUT channel)
  {   // W0^T fragments: wave w prepares (nt = w >> 2, kq = w & 3): Wt[k = co][n = ci] = W0[ci*64 + co]
    const int nt = wave >> 2, kq = wave & 3;
    const f32x4* wp = reinterpret_cast<const f32x4*>(W0 + (int64_t)(nt * 32 + i) * C + kq * 16 + 8 * h);
    u32x2 a1, a2, a3, b1, b2, b3;
    split4(__builtin_bit_cast(u32x4, wp[0]), a1, a2, a3);
    split4(__builtin_bit_cast(u32x4, wp[1]), b1, b2, b3);
    wfl[((nt * 4 + kq) * 3 + 0) * 64 + lane] = u32x4{a1[0], a1[1], b1[0], b1[1]};
    wfl[((nt * 4 + kq) * 3 + 1) * 64 + lane] = u32x4{a2[0], a2[1], b2[0], b2[1]};
    wfl[((nt * 4 + kq) * 3 + 2) * 64 + lane] = u32x4{a3[0], a3[1], b3[0], b3[1]};
  }
  {   // W2^T fragments: wave w prepares (q = w >> 1, ks = w & 1): B[k = co][n = ci] = W2[ci*64 + co], k = ks*32 + 8 g4 + j
    const int qq = wave >> 1, ks = wave & 1;
    const f32x4* wp = reinterpret_cast<const f32x4*>(W2 + (int64_t)(qq * 16 + l16) * C + ks * 32 + 8 * g4);
    u32x2 a1, a2, a3, b1, b2, b3;
    split4(__builtin_bit_cast(u32x4, wp[0]), a1, a2, a3);
    split4(__builtin_bit_cast(u32x4, wp[1]), b1, b2, b3);
    wf2[((qq * 2 + ks) * 3 + 0) * 64 + lane] = u32x4{a1[0], a1[1], b1[0], b1[1]};
    wf2[((qq * 2 + ks) * 3 + 1) * 64 + lane] = u32x4{a2[0], a2[1], b2[0], b2[1]};
    wf2[((qq * 2 + ks) * 3 + 2) * 64 + lane] = u32x4{a3[0], a3[1], b3[0], b3[1]};
  }
  if (threadIdx.x < NS * 32) {                                         // columns 0 and W + 1 of every slot: always zero
    const int slot = threadIdx.x >> 5, side = (threadIdx.x >> 4) & 1;
    ring[(slot * XSP + (side ? W_ + 1 : 0)) * 16 + c4] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 aw[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) aw[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (threadIdx.x < 9 * 16) wl[threadIdx.x] = w[threadIdx.x];
  const f32x4* wlc = wl + c4;
  f32x4 ab = {0.f, 0.f, 0.f, 0.f}, bs0 = {0.f, 0.f, 0.f, 0.f};
  f32x16 accw;                                                         // waves 4-7: P tile of conv0's weight gradient
#pragma unroll
  for (int r = 0; r < 16; ++r) accw[r] = 0.f;
  const bool gemm_wave = wave < 4;
  const int ykk = wave & 1, ynt = (wave >> 1) & 1;
  const int pa = wave & 1, pb = (wave >> 1) & 1;
  const int ry = px / W_, xc = px % W_;
  f32x4* rbase = ring + xc * 16 + c4;
  const int st_off = dual_off(px, c4 >> 1) + (c4 & 1) * 8;
  char* stD = tD + st_off;
  auto tr_frag = [](const char* o0, const char* o1) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)o0);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)o1);
    s16x8 f = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, f);
  };

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const int HW = H * W_;
    const int64_t ioff = (int64_t)b * HW * 16;                         // float4 offset of the image
    const f32x4 gg_in = gate[(int64_t)b * 16 + c4];
    const f32x4 dg_in = dgap[(int64_t)b * 16 + c4] * inv_hw;
    // row set j = pixels j*32 + W .. + 31 of the image (image rows j*RPT + 1 .. j*RPT + RPT)
    auto fetch_set = [&](int j) {
      const int p = j * 32 + W_ + px;
      return dout4[ioff + (int64_t)min(max(p, 0), HW - 1) * 16 + c4];
    };
    // every wave fetches and stores the set's 64 mask words (pixel lane >> 1, half lane & 1): no branch around the load
    auto fetch_mask = [&](int j) {
      const int p = j * 32 + W_ + (lane >> 1);
      return mask[((int64_t)b * HW + min(max(p, 0), HW - 1)) * 2 + (lane & 1)];
    };
    // set j, first half: planes of dout for the set's product, raw dout for the residual, the set's mask words
    auto produce_a = [&](int j, const f32x4 rd, unsigned mword) {
      const int p = j * 32 + W_ + px;
      u32x2 p1, p2, p3;
      split4(__builtin_bit_cast(u32x4, rd), p1, p2, p3);
      *reinterpret_cast<u32x2*>(tA + st_off) = p1;
      *reinterpret_cast<u32x2*>(tA + TP + st_off) = p2;
      *reinterpret_cast<u32x2*>(tA + 2 * TP + st_off) = p3;
      tR[((p >= 0 && p < HW) ? p % 96 : 96) * 16 + c4] = rd;           // outside the image: the dummy pixel
      mk[lane] = mword;
    };
    // second half: dt2 = dout . W2^T on 16 x 16 tiles, d1 = mask ? dt2 * gate + dgap / HW : 0 into the ring
    auto produce_c = [&](int j) {
      typedef float f32x4_ __attribute__((ext_vector_type(4)));
      f32x4_ acc = {0.f, 0.f, 0.f, 0.f};
      int lane_c = lane;
      asm volatile("" : "+v"(lane_c));
      const int l16 = lane_c & 15, g4 = lane_c >> 4, ci1 = q4 * 16 + l16;
      const u32x4* g1W = wf2 + q4 * 6 * 64 + lane_c;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 xa[3], wb[3];
        const char* g1A = tA + dual_off(ph * 16 + l16, ks * 4 + g4);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          xa[p] = as_frag(*reinterpret_cast<const u32x4*>(g1A + p * TP));
          wb[p] = as_frag(g1W[(ks * 3 + p) * 64]);
        }
        MVAE_SPLIT6_16(acc, xa, wb);
      }
      const float gci = reinterpret_cast<const float*>(wl + 9 * 16)[ci1], dci = reinterpret_cast<const float*>(wl + 10 * 16)[ci1];
      float* rf = reinterpret_cast<float*>(ring);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int pl = ph * 16 + 4 * g4 + r, p = j * 32 + W_ + pl;     // pixel inside the set / inside the image
        const int row = j * RPT + 1 + pl / W_, col = pl % W_;
        const unsigned word = mk[pl * 2 + (ci1 >> 5)];
        const unsigned on = (unsigned)(p >= 0) & (unsigned)(p < HW) & (word >> (ci1 & 31));
        rf[((row & (NS - 1)) * XSP + col + 1) * 64 + ci1] = (on & 1u) ? acc[r] * gci + dci : 0.f;
      }
    };
    auto fetch_t0 = [&](int t) { return t0[ioff + (int64_t)(min(t, HW / 32 - 1) * 32 + px) * 16 + c4]; };
    f32x4 Fd[2], T[2];
    unsigned Mk[2];
    __syncthreads();                                  // previous image's LDS reads are done (and the fragments are written)
    if (threadIdx.x < 16) { wl[9 * 16 + c4] = gg_in; wl[10 * 16 + c4] = dg_in; }
    Fd[1] = fetch_set(-1);
    Mk[1] = fetch_mask(-1);
    T[0] = fetch_t0(0);
    T[1] = fetch_t0(1);
    rbase[(((-2 * RPT + 1 + ry) & (NS - 1)) * XSP + 1) * 16] = f32x4{0.f, 0.f, 0.f, 0.f};    // rows of set -2: zeros
    produce_a(-1, Fd[1], Mk[1]);
    Fd[1] = fetch_set(0);
    Mk[1] = fetch_mask(0);
    Fd[0] = fetch_set(1);
    Mk[0] = fetch_mask(1);
    __syncthreads();
    produce_c(-1);
#pragma unroll 1
    for (int t2 = 0; t2 < HW / 32; t2 += 2) {
      const int64_t prow = (int64_t)b * HW + t2 * 32;                  // pixel index of the first tile
      const f32x4 la0 = a_in[(prow + px) * 16 + c4];
      const f32x4 la1 = a_in[(prow + 32 + px) * 16 + c4];
      __syncthreads();                                // the previous pair's products (tA, tD, tR readers) and set are done
      // fragment addresses of the products: recomputed per pair from an opaque copy of the lane id -- as loop invariants
      // they cost ~25 registers that hipcc spilled and reloaded inside this loop (scratch loads share vmcnt with the prefetches)
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const int i = lane_o & 31, h = lane_o >> 5, g4 = lane_o >> 4;
      const char* gA[4];
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) gA[kq] = tD + ykk * 3 * TP + dual_off(i, 2 * kq + h);
      const u32x4* gW = wfl + ynt * 12 * 64 + lane_o;
      const char *pA0, *pA1, *pD0, *pD1;
      {
        const int g16 = g4, q = (lane_o >> 2) & 3, pp = lane_o & 3, rb = 8 * (g16 >> 1);
        const int ca = pa * 32 + 16 * (g16 & 1) + 4 * pp, cd = pb * 32 + 16 * (g16 & 1) + 4 * pp;
        pA0 = tA + dual_off(rb + q, ca >> 3) + (ca & 7) * 2;
        pA1 = tA + dual_off(rb + 4 + q, ca >> 3) + (ca & 7) * 2;
        pD0 = tD + dual_off(rb + q, cd >> 3) + (cd & 7) * 2;
        pD1 = tD + dual_off(rb + 4 + q, cd >> 3) + (cd & 7) * 2;
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int t = t2 + kk, y = t * RPT + ry;                       // this thread's output pixel: (y, xc)
        produce_a(t, Fd[(kk + 1) & 1], Mk[(kk + 1) & 1]);
        Fd[(kk + 1) & 1] = fetch_set(t + 2);
        Mk[(kk + 1) & 1] = fetch_mask(t + 2);
        __syncthreads();
        produce_c(t);
        __syncthreads();
        const f32x4 tvk = T[kk];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const f32x4* rrow = rbase + ((y - a + 1) & (NS - 1)) * (XSP * 16);
#pragma unroll
          for (int e = 0; e < 3; ++e) {
            const f32x4 sv = rrow[(2 - e) * 16];
            acc += wlc[(a * 3 + e) * 16] * sv;
            aw[a * 3 + e] += tvk * sv;
          }
        }
        ab += rbase[((y & (NS - 1)) * XSP + 1) * 16];
        f32x4 rv;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) rv[qq] = tvk[qq] > 0.f ? acc[qq] : 0.f;
        bs0 += rv;
        u32x2 p1, p2, p3;
        split4(__builtin_bit_cast(u32x4, rv), p1, p2, p3);
        *reinterpret_cast<u32x2*>(stD + (kk * 3 + 0) * TP) = p1;
        *reinterpret_cast<u32x2*>(stD + (kk * 3 + 1) * TP) = p2;
        *reinterpret_cast<u32x2*>(stD + (kk * 3 + 2) * TP) = p3;
        T[kk] = fetch_t0(t + 2);
      }
      {
        u32x2 p1, p2, p3;
        split4(__builtin_bit_cast(u32x4, la0), p1, p2, p3);
        *reinterpret_cast<u32x2*>(stD + 6 * TP) = p1;                  // tA = tD + 6 planes
        *reinterpret_cast<u32x2*>(stD + 7 * TP) = p2;
        *reinterpret_cast<u32x2*>(stD + 8 * TP) = p3;
        split4(__builtin_bit_cast(u32x4, la1), p1, p2, p3);
        *reinterpret_cast<u32x2*>(stD + 9 * TP) = p1;
        *reinterpret_cast<u32x2*>(stD + 10 * TP) = p2;
        *reinterpret_cast<u32x2*>(stD + 11 * TP) = p3;
      }
      __syncthreads();                                // all tiles of the pair complete
      if (gemm_wave) {
        // ---- da tile: tile t2 + ykk, output channels 32 ynt ..   (A = dt0 pixels, B = W0^T fragments)
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
          bf16x8 xa[3], wb[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            xa[p] = as_frag(*reinterpret_cast<const u32x4*>(gA[kq] + p * TP));
            wb[p] = as_frag(gW[(kq * 3 + p) * 64]);
          }
          MVAE_SPLIT6(acc, xa, wb);
        }
        float* py = da + (prow + ykk * 32 + 4 * h) * C + ynt * 32 + i;
        const float* gR = reinterpret_cast<const float*>(tR) + (((t2 + ykk) % 3) * 32 + 4 * h) * C + ynt * 32 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) py[((r & 3) + 8 * (r >> 2)) * C] = acc[r] + gR[((r & 3) + 8 * (r >> 2)) * C];
      } else {
        // ---- P[ci][co] += a^T dt0 over the 64 pixels of the pair
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            bf16x8 fa[3], fb[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
              fa[p] = tr_frag(pA0 + (kk * 3 + p) * TP + half * 2048, pA1 + (kk * 3 + p) * TP + half * 2048);
              fb[p] = tr_frag(pD0 + (kk * 3 + p) * TP + half * 2048, pD1 + (kk * 3 + p) * TP + half * 2048);
            }
            MVAE_SPLIT6(accw, fa, fb);
          }
      }
    }
  }
  // ---- depthwise weight / bias gradients and db0: lanes l, l + 16, l + 32, l + 48 share the channel quad
  __syncthreads();
  f32x4* red = reinterpret_cast<f32x4*>(tD);
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    f32x4 v = k < 9 ? aw[k < 9 ? k : 0] : (k == 9 ? ab : bs0);
    for (int off = 16; off < 64; off <<= 1) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) v[qq] += __shfl_xor(v[qq], off, 64);
    }
    if (lane < 16) red[(wave * 11 + k) * 16 + lane] = v;
  }
  __syncthreads();
  const int64_t slot = (int64_t)(blockIdx.x % nslots) * slot_stride;
  for (int idx = threadIdx.x; idx < 11 * 64; idx += 512) {
    const int qq = idx & 3, cc = (idx >> 2) & 15, k = idx >> 6;
    float t = 0.f;
#pragma unroll
    for (int wv = 0; wv < 8; ++wv) t += red[(wv * 11 + k) * 16 + cc][qq];
    float* dst = k < 9 ? dW + slot + (int64_t)k * 64 : (k == 9 ? db + slot : (db0 ? db0 + slot : nullptr));
    if (dst) atomicAdd(dst + cc * 4 + qq, t);
  }
  __syncthreads();
  float* redw = reinterpret_cast<float*>(lds);
  if (!gemm_wave) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = pa * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, co = pb * 32 + i;
      redw[ci * 64 + co] = accw[r];
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 64 * 64; idx += 512) atomicAdd(&dW0[slot + idx], redw[idx]);
}

static bool mn_bwd_attr() {
  static const bool attr =
      hipFuncSetAttribute((const void*)k_mn_bwd_s<32>, hipFuncAttributeMaxDynamicSharedMemorySize, kMnLds) == hipSuccess &&
      hipFuncSetAttribute((const void*)k_mn_bwd_s<16>, hipFuncAttributeMaxDynamicSharedMemorySize, kMnLds) == hipSuccess;
  return attr;
}
// dout + mask words (launch_gemm_dual_stats) -> da, dW_dw, db_dw, dW0, db0; false = shape not covered or switched off
const char* mn_bwd_split_kernel(int B, int H, int W, int C) {
  // Off unless MVAE_FUSE_MN_BWD=1: parity-green and 1.7 GB per headline step lighter on HBM, but SLOWER -- the recomputed
  // product adds two barriers and a latency-bound MFMA phase to every step of a block that has the CU to itself, and 26
  // registers spill: 96 us per launch against 77 for k_dw_bwd_conv0_s on the same launches, step 5.27 against 5.08 ms.
  static const int on = [] { const char* e = getenv("MVAE_FUSE_MN_BWD"); return e ? atoi(e) : 0; }();
  if (!on || det_mode() || !dw_bwd_conv0_split_kernel(B, H, W, C)) return nullptr;
  if (W != 32 && W != 16) return nullptr;
  if (!mn_bwd_attr()) return nullptr;
  return "k_mn_bwd_s";
}
bool launch_mn_bwd_split(const float* dout, const unsigned* mask, const float* t0, const float* w, const float* gate,
                         const float* dgap, const float* W2, const float* W0, const float* a_in, float* da, float* dW, float* db,
                         float* dW0, float* db0, GradSlots sl, int B, int H, int W, int C, hipStream_t s) {
  if (!mn_bwd_split_kernel(B, H, W, C)) return false;
  static const int cus32 = [] { const char* e = getenv("MVAE_FUSED_CUS"); int n = e ? atoi(e) : 256; return n < 8 ? 8 : (n > 256 ? 256 : n); }();
  static const int cus16 = [] { const char* e = getenv("MVAE_FUSED_CUS16"); int n = e ? atoi(e) : 256; return n < 8 ? 8 : (n > 256 ? 256 : n); }();
  // 8-wide maps on half the CUs: these launches are latency-bound (four images per block cost little) and a block of this
  // kernel has its CU to itself, so the other half of the chip stays open to the other scales' streams (4.98 -> 4.94 ms)
  static const int cus8 = [] { const char* e = getenv("MVAE_FUSED_CUS8"); int n = e ? atoi(e) : 128; return n < 8 ? 8 : (n > 256 ? 256 : n); }();
  const int cus = W == 32 ? cus32 : (W == 16 ? cus16 : cus8);
  const int grid = B < cus ? B : cus;
#define MVAE_MB(WW)                                                                                                    \
  hipLaunchKernelGGL(k_mn_bwd_s<WW>, dim3(grid), dim3(512), kMnLds, s, (const f32x4*)dout, mask, (const f32x4*)t0,     \
                     (const f32x4*)w, (const f32x4*)gate, (const f32x4*)dgap, W2, W0, (const f32x4*)a_in, da, sl.at(dW), \
                     sl.at(db), sl.at(dW0), sl.at(db0), H, 1.0f / (float)(H * W), B, sl.count(), sl.stride)
  if (W == 32) MVAE_MB(32); else MVAE_MB(16);
#undef MVAE_MB
  return true;
}

}  // namespace mvae
