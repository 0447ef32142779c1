// kernels_split.hip -- float32 k x k convolutions on the bf16 matrix cores: "split" products.
//
// gfx950 has no fast float32 MFMA: v_mfma_f32_32x32x2_f32 runs at the vector rate (157 TFLOP/s), 1/16 of
// v_mfma_f32_32x32x16_bf16.  The 5x5 stride-2 Conv2D / Conv2DTranspose layers of the encoder / decoder
// (layer_blocks.py:946-951; K = 800, 13.4 GFLOP per launch at batch 512) are the only kernels of the float32 step that
// are bound by that rate (k_conv_taps / k_wgrad_taprow, kernels_mfma.hip: 90-107 TFLOP/s).  Here every float32 operand
// is written as the EXACT sum of three bfloat16 numbers,
//        x = x1 + x2 + x3,   x1 = top 8 significand bits of x,  x2 = top 8 bits of x - x1,  x3 = x - x1 - x2
// (a float32 has 24 significand bits, a bfloat16 8 and the same exponent range: the three truncations are exact), and a
// product a.b is accumulated in float32 from the six partial products a1b1, a1b2, a2b1, a1b3, a2b2, a3b1.  Every bf16 x
// bf16 product is exact in float32 (16 significand bits); the three terms left out (a2b3, a3b2, a3b3) are below 2^-24 of
// |a b| -- the size of the rounding error the float32 MFMA itself commits.  Six bf16 MFMAs cost 6/16 of one float32
// MFMA: 2.7x the matrix throughput at float32 accuracy (tests/test_split_conv_gpu.py: against float64 the split kernels are
// as close as the float32-MFMA kernels they replace).
//
// The weights are split once per step into bf16 planes laid out exactly as the kernels stage them in LDS
// (k_split_weights); activations are split by the wave that gathers them, on their way into its LDS tile.
#include "kernels.h"
#include "prof.h"
#include "split.h"
#include <cstdio>
#include <cstdlib>

namespace mvae {

// ---- weights -> bf16 planes in the kernels' LDS layout ------------------------------------------------------------
// W: [taps][CI][CO] float32 (F-form coordinates: a Conv2D kernel HWIO, or a Conv2DTranspose kernel (kh,kw,out,in)).
//   outF: [tap][plane][n = co][k = ci]   (F-form launches: KC = CI, NC = CO)
//   outT: [tap][plane][n = ci][k = co]   (T-form launches: KC = CO, NC = CI)
// rows are wrow_off-swizzled, a plane is CI*CO*2 bytes, a tap 3 planes.
__global__ void __launch_bounds__(256) k_split_weights(const float* __restrict__ W, char* __restrict__ outF,
                                                       char* __restrict__ outT, int taps, int CI, int CO) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= taps * CI * CO) return;
  const int tap = idx / (CI * CO), rem = idx % (CI * CO), ci = rem / CO, co = rem % CO;
  const unsigned x = __float_as_uint(W[idx]);
  const float a = __uint_as_float(x) - __uint_as_float(x & 0xFFFF0000u);
  const unsigned r1 = __float_as_uint(a);
  const unsigned r2 = __float_as_uint(a - __uint_as_float(r1 & 0xFFFF0000u));
  const uint16_t pl[3] = {(uint16_t)(x >> 16), (uint16_t)(r1 >> 16), (uint16_t)(r2 >> 16)};
  const int plane = CI * CO * 2;
  const int offF = CI == 32 ? wrow_off<32>(co, ci >> 3) : wrow_off<64>(co, ci >> 3);     // KC = CI
  const int offT = CO == 32 ? wrow_off<32>(ci, co >> 3) : wrow_off<64>(ci, co >> 3);     // KC = CO
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    *reinterpret_cast<uint16_t*>(outF + (size_t)(tap * 3 + p) * plane + offF + (ci & 7) * 2) = pl[p];
    *reinterpret_cast<uint16_t*>(outT + (size_t)(tap * 3 + p) * plane + offT + (co & 7) * 2) = pl[p];
  }
}

// =================================================================================================
// k x k strided SAME convolution (F-form) / transposed convolution (T-form), float32 in and out, split-bf16 products.
// Same coordinates, tap list, padding-by-buffer-range and epilogue as k_conv_taps (kernels_mfma.hip); what differs is the
// arithmetic of a tap: the wave's 32 gathered pixels go to its LDS tile as three bf16 planes, the block's weight slice
// arrives already split, and a tap is KC/16 k-steps of six v_mfma_f32_32x32x16_bf16 per 32 output channels.
// A = activations (row = pixel), B = weights (column = output channel): the accumulator layout, and with it the
// coalesced 128-byte row-segment stores of the epilogue, are those of the float32 kernel.
//
// A tap is ~3x shorter than in the float32 kernel (~0.4 us), shorter than the latency of the gather that feeds the next one:
// with the float32 kernel's one tap of prefetch this kernel ran at the SAME speed as the float32 one (latency-bound: 121 /
// 161 / 155 / 121 us against 136 / 170 / 143 / 156).  So the tap loop is FULLY UNROLLED (template on the tap counts of the
// launch: 5 x 5 for F-form, {3,2} x {3,2} for the sub-pixel phases of a stride-2 T-form) with PF taps of register prefetch:
// in straight-line code hipcc counts the outstanding loads exactly (s_waitcnt vmcnt(N > 0)), so a tap's data is waited
// for with the PF - 1 younger fetches still in flight.
// =================================================================================================
template <int KC, int NC, bool TFORM, int NKH, int NKW, int PF, int NW>
struct ConvTapsS {
  static constexpr int NT = NC / 32, KK = KC / 16, Q = KC / 8;       // Q = 16-byte float32 chunks a lane fetches per tap
  static constexpr int CPP = KC / 4, PPI = 64 / CPP;                 // lanes per pixel, pixels per load instruction
  static constexpr int PLANE_W = NC * KC * 2, TAPB = 3 * PLANE_W;    // bytes
  static constexpr int PLANE_A = 32 * KC * 2;
  static constexpr int NTHR = 64 * NW;                               // NW waves share one staged weight slice
  static constexpr int WLD = (TAPB + NTHR * 16 - 1) / (NTHR * 16);   // 16-byte pieces of a weight slice per thread
  static constexpr int NTAPS = NKH * NKW;
  static constexpr bool WRAG = TAPB % (NTHR * 16) != 0;              // the last piece exists for part of the threads only

  // kernel column of tap-list position tw: F-form with stride 2 walks a kernel row as kw = 0, 2, 4, 1, 3 (consecutive taps
  // touch the same cache lines, see k_conv_taps)
  static __device__ __forceinline__ constexpr int col_of(int tw) {
    return TFORM ? tw : (tw < (NKW + 1) / 2 ? 2 * tw : 2 * (tw - (NKW + 1) / 2) + 1);
  }

  static __device__ __forceinline__ void run(const float* __restrict__ in, const char* __restrict__ Wp,
                                             const float* __restrict__ bias, float* __restrict__ out, const ConvGeom& g,
                                             unsigned in_bytes, unsigned out_bytes, char (*sW)[TAPB], char* myA, int py, int px,
                                             int kh0, int kw0, int CH, int CW, unsigned p0, unsigned Mc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int khs = TFORM ? g.SH : 1, kws = TFORM ? g.SW : 1;
    const int SHh = TFORM ? g.OH : g.IH, SWw = TFORM ? g.OW : g.IW;      // the tensor the taps read
    const int lp = lane / CPP, ch = lane % CPP;
    unsigned base[Q], inv[Q];
    const bool pow2 = (CW & (CW - 1)) == 0 && (CH & (CH - 1)) == 0;
    const int lgw = 31 - __builtin_clz((unsigned)CW), lgh = 31 - __builtin_clz((unsigned)CH);
    auto split = [&](unsigned p, int& cx, int& cy, int& b) {
      if (pow2) { cx = (int)(p & (unsigned)(CW - 1)); cy = (int)((p >> lgw) & (unsigned)(CH - 1)); b = (int)(p >> (lgw + lgh)); }
      else { cx = (int)(p % (unsigned)CW); const unsigned q = p / (unsigned)CW; cy = (int)(q % (unsigned)CH); b = (int)(q / (unsigned)CH); }
    };
    // per tap-list row / column: the offset of the pixel it reads relative to the window origin
    int dyv[NKH], dxv[NKW];
#pragma unroll
    for (int t = 0; t < NKH; ++t) { const int kh = kh0 + t * khs; dyv[t] = TFORM ? (py + g.PT - kh) / g.SH : kh - g.PT; }
#pragma unroll
    for (int t = 0; t < NKW; ++t) { const int kw = kw0 + col_of(t) * kws; dxv[t] = TFORM ? (px + g.PL - kw) / g.SW : kw - g.PL; }
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      const unsigned p = p0 + wave * 32 + lp + PPI * j;
      int cx, cy, b;
      split(p < Mc ? p : 0u, cx, cy, b);
      const int y0 = TFORM ? cy : cy * g.SH, x0 = TFORM ? cx : cx * g.SW;
      base[j] = (unsigned)(((b * SHh + y0) * SWw + x0) * KC + ch * 4) * 4u;
      unsigned m = p < Mc ? 0u : 0xFFFFu;                    // bit t: tap row t leaves the image, bit 8 + t: tap column t does
#pragma unroll
      for (int t = 0; t < NKH; ++t)
        if ((unsigned)(y0 + dyv[t]) >= (unsigned)SHh) m |= 1u << t;
#pragma unroll
      for (int t = 0; t < NKW; ++t)
        if ((unsigned)(x0 + dxv[t]) >= (unsigned)SWw) m |= 0x100u << t;
      inv[j] = m;
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)in_bytes, 0x00020000);

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    u32x4 wreg[PF][WLD], areg[PF][Q];
    auto fetch = [&](int t, int slot) {                     // t, slot: compile-time after unrolling
      const int th = t / NKW, tw = t % NKW;
      const int kh = kh0 + th * khs, kw = kw0 + col_of(tw) * kws;
      const u32x4* wt = reinterpret_cast<const u32x4*>(Wp + (size_t)(kh * g.KW + kw) * TAPB);
#pragma unroll
      for (int u = 0; u < WLD; ++u) {
        const int idx = (int)threadIdx.x + u * NTHR;
        wreg[slot][u] = wt[(WRAG && idx * 16 >= TAPB) ? (int)threadIdx.x : idx];        // (clamped: loaded, not stored)
      }
      const unsigned delta = (unsigned)((dyv[th] * SWw + dxv[tw]) * KC * 4);
      const unsigned sel = (1u << th) | (0x100u << tw);
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        const unsigned off = (inv[j] & sel) ? 0x80000000u : base[j] + delta;
        areg[slot][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
      }
    };
    // The split of the next tap's pixels (VALU) is spread between this tap's MFMA groups, its LDS stores follow the last
    // fragment read: a wave then never issues more than six MFMAs back to back.
    u32x2 sp[Q][3];
    auto split_slot = [&](int slot, int j) { split4(areg[slot][j], sp[j][0], sp[j][1], sp[j][2]); };
    auto store = [&](int slot, int buf) {                   // weights as they are; pixels: the three planes of every slot
#pragma unroll
      for (int u = 0; u < WLD; ++u) {
        const int idx = (int)threadIdx.x + u * NTHR;
        if (!WRAG || idx * 16 < TAPB) reinterpret_cast<u32x4*>(sW[buf])[idx] = wreg[slot][u];
      }
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        const int pl = lp + PPI * j;
        const int o = tile_off<KC>(pl, ch >> 1) + (ch & 1) * 8;
        *reinterpret_cast<u32x2*>(myA + o) = sp[j][0];
        *reinterpret_cast<u32x2*>(myA + PLANE_A + o) = sp[j][1];
        *reinterpret_cast<u32x2*>(myA + 2 * PLANE_A + o) = sp[j][2];
      }
    };
#pragma unroll
    for (int t = 0; t < PF; ++t)
      if (t < NTAPS) fetch(t, t);
#pragma unroll
    for (int j = 0; j < Q; ++j) split_slot(0, j);
    store(0, 0);
    constexpr int GROUPS = KK * NT;                       // MFMA groups of six per tap
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
      __syncthreads();       // sW[t & 1] and the wave's A tile are complete, sW[(t + 1) & 1] is free
      if (t + PF < NTAPS) fetch(t + PF, t % PF);            // into the registers tap t has just left
      __builtin_amdgcn_sched_barrier(0);
      const char* w = sW[t & 1];
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        bf16x8 xa[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
          xa[p] = as_frag(*reinterpret_cast<const u32x4*>(myA + p * PLANE_A + tile_off<KC>(i, 2 * kk + h)));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          bf16x8 wb[3];
          const int wo = wrow_off<KC>(nt * 32 + i, 2 * kk + h);
#pragma unroll
          for (int p = 0; p < 3; ++p) wb[p] = as_frag(*reinterpret_cast<const u32x4*>(w + p * PLANE_W + wo));
          // smallest terms first
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[2], wb[0], acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[1], wb[1], acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0], wb[2], acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[1], wb[0], acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0], wb[1], acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0], wb[0], acc[nt], 0, 0, 0);
          // a share of the next tap's split behind every group of six
          if (t + 1 < NTAPS) {
            constexpr int PER = (Q + GROUPS - 1) / GROUPS;
            const int gi = kk * NT + nt;
#pragma unroll
            for (int j = gi * PER; j < (gi + 1) * PER && j < Q; ++j) split_slot((t + 1) % PF, j);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // the A tile is wave-private: this wave's fragment reads of it are done (they fed the MFMAs above)
      if (t + 1 < NTAPS) store((t + 1) % PF, (t + 1) & 1);
    }
    // ---- epilogue (k_conv_taps): unconditional bias load, row offsets through the idle A tile, buffer stores
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = (bias ? bias : in)[nt * 32 + i];
    unsigned* sOff = reinterpret_cast<unsigned*>(myA);
    __builtin_amdgcn_wave_barrier();
    if (h == 0) {
      const unsigned p = p0 + wave * 32 + i;
      unsigned off = 0x80000000u;
      if (p < Mc) {
        int cx, cy, b;
        split(p, cx, cy, b);
        off = (TFORM ? (unsigned)(((b * g.IH + (cy * g.SH + py)) * g.IW + (cx * g.SW + px)) * NC) : p * NC) * 4u;
      }
      sOff[i] = off;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    unsigned offs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) offs[r] = sOff[(r & 3) + 8 * (r >> 2) + 4 * h];
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)out_bytes, 0x00020000);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const unsigned n4 = (unsigned)(nt * 32 + i) * 4u;
      const float b = bias ? bv[nt] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[nt][r] + b), orsrc, offs[r] + n4, 0, 0);
    }
  }
};

// F-form: KH x KW = 5 x 5 (any stride / padding).  T-form: stride 2 x 2, 5 x 5: a sub-pixel phase (blockIdx.y) has 3 or 2 tap
// rows and columns.  Other geometries take the float32 kernels (launcher).
template <int KC, int NC, bool TFORM, int PF, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 1 : ((KC == 32 && PF == 2) ? 3 : 2))
k_conv_taps_s(const float* __restrict__ in, const char* __restrict__ Wp, const float* __restrict__ bias,
              float* __restrict__ out, ConvGeom g, unsigned in_bytes, unsigned out_bytes) {
  constexpr int TAPB = 3 * NC * KC * 2, PLANE_A = 32 * KC * 2;
  __shared__ __attribute__((aligned(16))) char sW[2][TAPB];
  __shared__ __attribute__((aligned(16))) char sA[NW][3 * PLANE_A];
  const int wave = threadIdx.x >> 6;
  int py = 0, px = 0, CH = g.OH, CW = g.OW;
  if (TFORM) {
    py = blockIdx.y / g.SW; px = blockIdx.y % g.SW;
    CH = (g.IH - py + g.SH - 1) / g.SH; CW = (g.IW - px + g.SW - 1) / g.SW;
  }
  const unsigned Mc = (unsigned)(g.B * CH * CW);
  const unsigned p0 = blockIdx.x * (32u * NW);
  if (p0 >= Mc) return;                         // block-uniform
  if constexpr (!TFORM) {
    ConvTapsS<KC, NC, false, 5, 5, PF, NW>::run(in, Wp, bias, out, g, in_bytes, out_bytes, sW, sA[wave], 0, 0, 0, 0, CH, CW, p0, Mc);
  } else {
    const int kh0 = (py + g.PT) % g.SH, kw0 = (px + g.PL) % g.SW;     // 0 -> taps 0, 2, 4;  1 -> taps 1, 3
#define MVAE_PH(A, B_) ConvTapsS<KC, NC, true, A, B_, PF, NW>::run(in, Wp, bias, out, g, in_bytes, out_bytes, sW, sA[wave], py, px, kh0, kw0, CH, CW, p0, Mc)
    if (kh0 == 0) { if (kw0 == 0) MVAE_PH(3, 3); else MVAE_PH(3, 2); }
    else { if (kw0 == 0) MVAE_PH(2, 3); else MVAE_PH(2, 2); }
#undef MVAE_PH
  }
}

// =================================================================================================
// T-form with the four sub-pixel phases of a stride-2 5 x 5 transposed convolution MERGED in one block (round 4).
// Output pixel (2 cy + py, 2 cx + px) sums the taps kh = py + 1 - 2 dy, kw = px + 1 - 2 dx over the input pixels
// (cy + dy, cx + dx), dy, dx in {-1, 0, 1}: the 25 taps of the four phases read only NINE distinct input offsets
// (2 x 2, 2, 2 or 1 taps per offset).  k_conv_taps_s<.., TFORM = true> runs one phase per block and gathers its 32 pixels
// once per TAP -- 25 gathers and 25 three-plane splits per 4 output pixels, in 4096 short blocks (4 - 9 taps each: the
// pipeline fill of a block is as long as its work).  Here a block owns 128 input-grid positions and all four phases of
// them: 9 gathers / splits, 25 taps into acc[phase], 1024 blocks of 25 taps.  Requires IH = 2 OH, IW = 2 OW, PT = PL = 1.
// Tap order: the offsets with four taps first, the single-tap corner last; the next offset's pixels are fetched at the
// first tap of the current offset and split between the MFMA groups of its last tap.
// =================================================================================================
template <int KC, int NC, int NW>
struct ConvTMergedS {
  static constexpr int NT = NC / 32, KK = KC / 16, Q = KC / 8;
  static constexpr int CPP = KC / 4, PPI = 64 / CPP;
  static constexpr int PLANE_W = NC * KC * 2, TAPB = 3 * PLANE_W;
  static constexpr int PLANE_A = 32 * KC * 2;
  static constexpr int NTHR = 64 * NW;
  static constexpr int WLD = (TAPB + NTHR * 16 - 1) / (NTHR * 16);
  static constexpr bool WRAG = TAPB % (NTHR * 16) != 0;
  static constexpr int PF = 2;                                       // weight slices of register prefetch
  // offset list (dy, dx): 4-tap offsets, 2-tap offsets, the 1-tap corner
  static __device__ __forceinline__ constexpr int o_dy(int oi) { return oi == 0 ? 0 : oi == 1 ? 0 : oi == 2 ? -1 : oi == 3 ? -1 : oi == 4 ? 1 : oi == 5 ? 1 : oi == 6 ? 0 : oi == 7 ? -1 : 1; }
  static __device__ __forceinline__ constexpr int o_dx(int oi) { return oi == 0 ? 0 : oi == 1 ? -1 : oi == 2 ? 0 : oi == 3 ? -1 : oi == 4 ? 0 : oi == 5 ? -1 : oi == 6 ? 1 : oi == 7 ? 1 : 1; }
  static __device__ __forceinline__ constexpr int o_start(int oi) { return oi < 4 ? 4 * oi : (oi < 8 ? 16 + 2 * (oi - 4) : 24); }
  static __device__ __forceinline__ constexpr int t_oi(int t) { return t < 16 ? t / 4 : (t < 24 ? 4 + (t - 16) / 2 : 8); }
  static __device__ __forceinline__ constexpr int t_py(int t) {
    const int oi = t_oi(t), j = t - o_start(oi), nx = o_dx(oi) == 1 ? 1 : 2;
    return o_dy(oi) == 1 ? 1 : j / nx;
  }
  static __device__ __forceinline__ constexpr int t_px(int t) {
    const int oi = t_oi(t), j = t - o_start(oi), nx = o_dx(oi) == 1 ? 1 : 2;
    return o_dx(oi) == 1 ? 1 : j % nx;
  }
  static __device__ __forceinline__ constexpr int t_kh(int t) { return t_py(t) + 1 - 2 * o_dy(t_oi(t)); }
  static __device__ __forceinline__ constexpr int t_kw(int t) { return t_px(t) + 1 - 2 * o_dx(t_oi(t)); }
  static __device__ __forceinline__ constexpr bool t_first(int t) { return t == o_start(t_oi(t)); }
  static __device__ __forceinline__ constexpr bool t_last(int t) { return t == 24 || t + 1 == o_start(t_oi(t) + 1); }

  static __device__ __forceinline__ void run(const float* __restrict__ in, const char* __restrict__ Wp,
                                             const float* __restrict__ bias, float* __restrict__ out, const ConvGeom& g,
                                             unsigned in_bytes, unsigned out_bytes, char (*sW)[TAPB], char* myA, unsigned p0,
                                             unsigned Mc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int CH = g.OH, CW = g.OW;                                  // the input (small) grid = the phase grid
    const int lp = lane / CPP, ch = lane % CPP;
    unsigned base[Q], inv[Q];
    const bool pow2 = (CW & (CW - 1)) == 0 && (CH & (CH - 1)) == 0;
    const int lgw = 31 - __builtin_clz((unsigned)CW), lgh = 31 - __builtin_clz((unsigned)CH);
    auto split = [&](unsigned p, int& cx, int& cy, int& b) {
      if (pow2) { cx = (int)(p & (unsigned)(CW - 1)); cy = (int)((p >> lgw) & (unsigned)(CH - 1)); b = (int)(p >> (lgw + lgh)); }
      else { cx = (int)(p % (unsigned)CW); const unsigned q = p / (unsigned)CW; cy = (int)(q % (unsigned)CH); b = (int)(q / (unsigned)CH); }
    };
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      const unsigned p = p0 + wave * 32 + lp + PPI * j;
      int cx, cy, b;
      split(p < Mc ? p : 0u, cx, cy, b);
      base[j] = (unsigned)(((b * CH + cy) * CW + cx) * KC + ch * 4) * 4u;
      unsigned m = p < Mc ? 0u : 0x77u;                      // bit dy + 1: row cy + dy leaves the image; bit 4 + dx + 1: column
      if (cy == 0) m |= 1u;
      if (cy == CH - 1) m |= 4u;
      if (cx == 0) m |= 0x10u;
      if (cx == CW - 1) m |= 0x40u;
      inv[j] = m;
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)in_bytes, 0x00020000);

    f32x16 acc[4][NT];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ph][nt][r] = 0.f;

    u32x4 wreg[PF][WLD], areg[Q];
    auto fetch_w = [&](int t, int slot) {
      const u32x4* wt = reinterpret_cast<const u32x4*>(Wp + (size_t)(t_kh(t) * 5 + t_kw(t)) * TAPB);
#pragma unroll
      for (int u = 0; u < WLD; ++u) {
        const int idx = (int)threadIdx.x + u * NTHR;
        wreg[slot][u] = wt[(WRAG && idx * 16 >= TAPB) ? (int)threadIdx.x : idx];
      }
    };
    auto fetch_a = [&](int oi) {
      const int dy = o_dy(oi), dx = o_dx(oi);
      const unsigned delta = (unsigned)((dy * CW + dx) * KC * 4);
      const unsigned sel = (1u << (dy + 1)) | (0x10u << (dx + 1));
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        const unsigned off = (inv[j] & sel) ? 0x80000000u : base[j] + delta;
        areg[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
      }
    };
    u32x2 sp[Q][3];
    auto split_slot = [&](int j) { split4(areg[j], sp[j][0], sp[j][1], sp[j][2]); };
    auto store_w = [&](int slot, int buf) {
#pragma unroll
      for (int u = 0; u < WLD; ++u) {
        const int idx = (int)threadIdx.x + u * NTHR;
        if (!WRAG || idx * 16 < TAPB) reinterpret_cast<u32x4*>(sW[buf])[idx] = wreg[slot][u];
      }
    };
    auto store_a = [&]() {
#pragma unroll
      for (int j = 0; j < Q; ++j) {
        const int pl = lp + PPI * j;
        const int o = tile_off<KC>(pl, ch >> 1) + (ch & 1) * 8;
        *reinterpret_cast<u32x2*>(myA + o) = sp[j][0];
        *reinterpret_cast<u32x2*>(myA + PLANE_A + o) = sp[j][1];
        *reinterpret_cast<u32x2*>(myA + 2 * PLANE_A + o) = sp[j][2];
      }
    };
    fetch_a(0);
#pragma unroll
    for (int t = 0; t < PF; ++t) fetch_w(t, t);
#pragma unroll
    for (int j = 0; j < Q; ++j) split_slot(j);
    store_a();
    store_w(0, 0);
    constexpr int GROUPS = KK * NT;
#pragma unroll
    for (int t = 0; t < 25; ++t) {
      __syncthreads();       // sW[t & 1] and the wave's A tile are complete, sW[(t + 1) & 1] is free
      if (t_first(t) && t_oi(t) + 1 < 9) fetch_a(t_oi(t) + 1);       // the registers the LDS tile was split from are free
      if (t + PF < 25) fetch_w(t + PF, t % PF);
      __builtin_amdgcn_sched_barrier(0);
      const char* w = sW[t & 1];
      const int ph = t_py(t) * 2 + t_px(t);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        bf16x8 xa[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
          xa[p] = as_frag(*reinterpret_cast<const u32x4*>(myA + p * PLANE_A + tile_off<KC>(i, 2 * kk + h)));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          bf16x8 wb[3];
          const int wo = wrow_off<KC>(nt * 32 + i, 2 * kk + h);
#pragma unroll
          for (int p = 0; p < 3; ++p) wb[p] = as_frag(*reinterpret_cast<const u32x4*>(w + p * PLANE_W + wo));
          acc[ph][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[2], wb[0], acc[ph][nt], 0, 0, 0);
          acc[ph][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[1], wb[1], acc[ph][nt], 0, 0, 0);
          acc[ph][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0], wb[2], acc[ph][nt], 0, 0, 0);
          acc[ph][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[1], wb[0], acc[ph][nt], 0, 0, 0);
          acc[ph][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0], wb[1], acc[ph][nt], 0, 0, 0);
          acc[ph][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[0], wb[0], acc[ph][nt], 0, 0, 0);
          if (t_last(t) && t < 24) {                        // the next offset's split behind the groups of this offset's last tap
            constexpr int PER = (Q + GROUPS - 1) / GROUPS;
            const int gi = kk * NT + nt;
#pragma unroll
            for (int j = gi * PER; j < (gi + 1) * PER && j < Q; ++j) split_slot(j);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (t + 1 < 25) store_w((t + 1) % PF, (t + 1) & 1);
      if (t_last(t) && t < 24) store_a();                  // wave-private tile: this wave's fragment reads of it are done
    }
    // ---- epilogue: per phase the pixel (2 cy + py, 2 cx + px); row offsets through the idle A tile
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = (bias ? bias : in)[nt * 32 + i];
    unsigned* sOff = reinterpret_cast<unsigned*>(myA);
    __builtin_amdgcn_wave_barrier();
    if (h == 0) {
      const unsigned p = p0 + wave * 32 + i;
      unsigned off = 0x80000000u;
      if (p < Mc) {
        int cx, cy, b;
        split(p, cx, cy, b);
        off = (unsigned)(((b * g.IH + cy * 2) * g.IW + cx * 2) * NC) * 4u;
      }
      sOff[i] = off;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    unsigned offs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) offs[r] = sOff[(r & 3) + 8 * (r >> 2) + 4 * h];
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)out_bytes, 0x00020000);
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      const unsigned pho = (unsigned)(((ph >> 1) * g.IW + (ph & 1)) * NC) * 4u;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const unsigned n4 = (unsigned)(nt * 32 + i) * 4u + pho;
        const float b = bias ? bv[nt] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)      // (an out-of-range row offset 0x80000000 stays out of range with the phase added)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[ph][nt][r] + b), orsrc, offs[r] + n4, 0, 0);
      }
    }
  }
};

template <int KC, int NC, int NW>
__global__ void __launch_bounds__(64 * NW, KC == 32 ? 2 : 2)
k_convt_merged_s(const float* __restrict__ in, const char* __restrict__ Wp, const float* __restrict__ bias,
                 float* __restrict__ out, ConvGeom g, unsigned in_bytes, unsigned out_bytes) {
  constexpr int TAPB = 3 * NC * KC * 2, PLANE_A = 32 * KC * 2;
  __shared__ __attribute__((aligned(16))) char sW[2][TAPB];
  __shared__ __attribute__((aligned(16))) char sA[NW][3 * PLANE_A];
  const int wave = threadIdx.x >> 6;
  const unsigned Mc = (unsigned)(g.B * g.OH * g.OW);
  const unsigned p0 = blockIdx.x * (32u * NW);
  if (p0 >= Mc) return;                         // block-uniform
  ConvTMergedS<KC, NC, NW>::run(in, Wp, bias, out, g, in_bytes, out_bytes, sW, sA[wave], p0, Mc);
}

// =================================================================================================
// F-form (5 x 5, stride 2, SAME) with the input rows of a KERNEL ROW staged in LDS once (round 4).
// k_conv_taps_s<.., TFORM = false> gathers a wave's 32 input pixels once per TAP: 25 gathers of 32 pixels and 25 three-plane
// splits per tile, and every input pixel crosses the fabric ~6 times because the 25 taps' footprint of the blocks resident on
// an XCD does not fit its L2 (PMC, round 3: 352 MB of fabric reads for a 151 MB launch; the kernel ran at the ~30 GB/s per CU
// of gathers served behind L2, MI355X_MICROARCH.md "Indexed rows").  Here a wave's tile is still 32 consecutive output pixels
// = R = 32 / COLS output rows of COLS = min(32, OW) columns, but per kernel row kh it loads the R input row segments those
// pixels read through the five kw taps -- 2 COLS + 3 pixels each, e.g. 2 x 35 instead of 5 x 32 pixels at OW = 16 -- splits
// them ONCE into the three bf16 planes and keeps them in its LDS region; the five taps of the row then read their A
// fragments from LDS at pixel 2 c + kw.  Even and odd segment pixels are stored apart (row index = ((q & 1) HALF +
// (q >> 1)) R + j), so the 32 lanes of a fragment read walk rows at a fixed stride whatever kw is.
// Gathered bytes per tile: 5 x 70 instead of 25 x 32 pixels (2.3x fewer), split work likewise.
// =================================================================================================
template <int KC, int NC, int COLS, int NW>
struct ConvFRowsS {
  static constexpr int NT = NC / 32, KK = KC / 16;
  static constexpr int CPP = KC / 4, PPI = 64 / CPP;                 // lanes per pixel, pixels per load instruction
  static constexpr int R = 32 / COLS, SEG = 2 * COLS + 3, HALF = COLS + 2, NPX = R * SEG;
  static constexpr int NL = (NPX + PPI - 1) / PPI;                   // load slots per lane and kernel row
  static constexpr int PLANE_W = NC * KC * 2, TAPB = 3 * PLANE_W;
  static constexpr int PLANE_A = ((NPX + 7) / 8 * 8) * KC * 2;       // one bf16 plane of the staged rows
  static constexpr int NTHR = 64 * NW;
  static constexpr int WLD = (TAPB + NTHR * 16 - 1) / (NTHR * 16);
  static constexpr bool WRAG = TAPB % (NTHR * 16) != 0;
  static constexpr int PF = 2;

  static __device__ __forceinline__ void run(const float* __restrict__ in, const char* __restrict__ Wp,
                                             const float* __restrict__ bias, float* __restrict__ out, const ConvGeom& g,
                                             unsigned in_bytes, unsigned out_bytes, char (*sW)[TAPB], char* myA, unsigned p0,
                                             unsigned Mc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int lp = lane / CPP, ch = lane % CPP;
    const int CW = g.OW, CH = g.OH;
    const unsigned ptile = p0 + wave * 32;
    // ---- staging slots: pixel s = u PPI + lp of the R x SEG staged pixels -> (output row j, segment column q)
    unsigned base[NL];                 // byte offset of (input row of kernel row 0, segment column q, channel chunk)
    unsigned rmask[NL];                // bit kh: that kernel row's input row is outside the image (or the slot is unused)
    int lds_o[NL];                     // byte offset of the slot's 8-byte piece inside a plane
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int sidx = u * PPI + lp;
      const int j = sidx / SEG, q = sidx - j * SEG;
      const unsigned pj = ptile + (unsigned)(j * COLS);
      const bool live = sidx < NPX && pj < Mc;
      const unsigned pp = live ? pj : 0u;
      const int ox0 = (int)(pp % (unsigned)CW);
      const unsigned t2 = pp / (unsigned)CW;
      const int oy = (int)(t2 % (unsigned)CH), b = (int)(t2 / (unsigned)CH);
      const int x = ox0 * 2 - g.PL + q, y0 = oy * 2 - g.PT;
      unsigned m = (live && (unsigned)x < (unsigned)g.IW) ? 0u : 0x1Fu;
#pragma unroll
      for (int kh = 0; kh < 5; ++kh)
        if ((unsigned)(y0 + kh) >= (unsigned)g.IH) m |= 1u << kh;
      rmask[u] = m;
      base[u] = (unsigned)(((b * g.IH + y0) * g.IW + x) * KC + ch * 4) * 4u;      // (wraps for y0 < 0: only used unmasked)
      const int rho = ((q & 1) * HALF + (q >> 1)) * R + j;      // output row j minor: see the fragment rows below
      lds_o[u] = tile_off<KC>(rho, ch >> 1) + (ch & 1) * 8;
    }
    const unsigned row_bytes = (unsigned)(g.IW * KC * 4);
    // ---- fragment rows: output pixel i = (j, c) reads LDS row ((kw & 1) HALF + c + (kw >> 1)) R + j.  With the output row as
    // the MINOR index the 16 lanes of a ds_read_b128 group ({0-3, 12-15, 20-27}, ...) hit 16 rows that are distinct mod 16 for
    // R = 1, 2, 4, 8: no bank conflicts (with j major, rows 39.. of the second output row collided with rows 12.. of the first)
    const int rho0 = (i % COLS) * R + (i / COLS);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)in_bytes, 0x00020000);

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    u32x4 wreg[PF][WLD], areg[NL];
    auto fetch_w = [&](int t, int slot) {
      const u32x4* wt = reinterpret_cast<const u32x4*>(Wp + (size_t)t * TAPB);          // t = kh * 5 + kw
#pragma unroll
      for (int u = 0; u < WLD; ++u) {
        const int idx = (int)threadIdx.x + u * NTHR;
        wreg[slot][u] = wt[(WRAG && idx * 16 >= TAPB) ? (int)threadIdx.x : idx];
      }
    };
    auto fetch_a = [&](int kh) {
#pragma unroll
      for (int u = 0; u < NL; ++u) {
        const unsigned off = (rmask[u] >> kh) & 1u ? 0x80000000u : base[u] + (unsigned)kh * row_bytes;
        areg[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
      }
    };
    auto stage_a = [&]() {               // split + store slot by slot (the region's readers -- this wave -- are done)
#pragma unroll
      for (int u = 0; u < NL; ++u) {
        u32x2 p1, p2, p3;
        split4(areg[u], p1, p2, p3);
        if (u * PPI + lp < NPX) {
          *reinterpret_cast<u32x2*>(myA + lds_o[u]) = p1;
          *reinterpret_cast<u32x2*>(myA + PLANE_A + lds_o[u]) = p2;
          *reinterpret_cast<u32x2*>(myA + 2 * PLANE_A + lds_o[u]) = p3;
        }
      }
    };
    auto store_w = [&](int slot, int buf) {
#pragma unroll
      for (int u = 0; u < WLD; ++u) {
        const int idx = (int)threadIdx.x + u * NTHR;
        if (!WRAG || idx * 16 < TAPB) reinterpret_cast<u32x4*>(sW[buf])[idx] = wreg[slot][u];
      }
    };
    fetch_a(0);
#pragma unroll
    for (int t = 0; t < PF; ++t) fetch_w(t, t);
    stage_a();
    store_w(0, 0);
#pragma unroll
    for (int kh = 0; kh < 5; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        const int t = kh * 5 + kw;
        __syncthreads();       // sW[t & 1] and the wave's staged rows are complete, sW[(t + 1) & 1] is free
        if (kw == 0 && kh + 1 < 5) fetch_a(kh + 1);           // the registers the rows were split from are free
        if (t + PF < 25) fetch_w(t + PF, t % PF);
        __builtin_amdgcn_sched_barrier(0);
        const char* w = sW[t & 1];
        const int rho = rho0 + ((kw & 1) * HALF + (kw >> 1)) * R;
        // fragment reads run ONE MFMA group ahead of their MFMAs (two register sets): with the reads of a group issued right
        // in front of its own MFMAs every group opened with an exposed LDS round trip (tools/isa_flow.py: R128x3 wait M ...)
        constexpr int G = KK * NT;
        bf16x8 xa[2][3], wb[2][3];
        auto load_x = [&](int kk, bf16x8 (&d)[3]) {
          const int ao = tile_off<KC>(rho, 2 * kk + h);
#pragma unroll
          for (int p = 0; p < 3; ++p) d[p] = as_frag(*reinterpret_cast<const u32x4*>(myA + p * PLANE_A + ao));
        };
        auto load_w = [&](int kk, int nt, bf16x8 (&d)[3]) {
          const int wo = wrow_off<KC>(nt * 32 + i, 2 * kk + h);
#pragma unroll
          for (int p = 0; p < 3; ++p) d[p] = as_frag(*reinterpret_cast<const u32x4*>(w + p * PLANE_W + wo));
        };
        load_x(0, xa[0]);
        load_w(0, 0, wb[0]);
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          const int kk = gi / NT, nt = gi % NT;
          if (gi + 1 < G) {
            const int kk2 = (gi + 1) / NT, nt2 = (gi + 1) % NT;
            if (kk2 != kk) load_x(kk2, xa[kk2 & 1]);
            load_w(kk2, nt2, wb[(gi + 1) & 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
          MVAE_SPLIT6(acc[nt], xa[kk & 1], wb[gi & 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 1 < 25) store_w((t + 1) % PF, (t + 1) & 1);
        if (kw == 4 && kh + 1 < 5) stage_a();                 // wave-private region: this wave's reads of it are done
      }
    }
    // ---- epilogue (as k_conv_taps_s): bias, row offsets through the idle staging region, buffer stores
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = (bias ? bias : in)[nt * 32 + i];
    unsigned* sOff = reinterpret_cast<unsigned*>(myA);
    __builtin_amdgcn_wave_barrier();
    if (h == 0) sOff[i] = (ptile + i < Mc) ? (ptile + i) * (unsigned)NC * 4u : 0x80000000u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    unsigned offs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) offs[r] = sOff[(r & 3) + 8 * (r >> 2) + 4 * h];
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)out_bytes, 0x00020000);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const unsigned n4 = (unsigned)(nt * 32 + i) * 4u;
      const float b = bias ? bv[nt] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[nt][r] + b), orsrc, offs[r] + n4, 0, 0);
    }
  }
};

template <int KC, int NC, int COLS, int NW>
__global__ void __launch_bounds__(64 * NW, KC == 32 ? 2 : 1)
k_convf_rows_s(const float* __restrict__ in, const char* __restrict__ Wp, const float* __restrict__ bias,
               float* __restrict__ out, ConvGeom g, unsigned in_bytes, unsigned out_bytes) {
  using K = ConvFRowsS<KC, NC, COLS, NW>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char (*sW)[K::TAPB] = reinterpret_cast<char (*)[K::TAPB]>(smem);
  char* sA = smem + 2 * K::TAPB;
  const int wave = threadIdx.x >> 6;
  const unsigned Mc = (unsigned)(g.B * g.OH * g.OW);
  const unsigned p0 = blockIdx.x * (32u * NW);
  if (p0 >= Mc) return;                         // block-uniform
  K::run(in, Wp, bias, out, g, in_bytes, out_bytes, sW, sA + wave * 3 * K::PLANE_A, p0, Mc);
}
template <int KC, int NC, int COLS>
static bool launch_convf_rows(const float* in, const char* pF, const float* bias, float* out, const ConvGeom& g, unsigned in_bytes,
                              unsigned out_bytes, hipStream_t s) {
  using K = ConvFRowsS<KC, NC, COLS, 4>;
  constexpr int lds = 2 * K::TAPB + 4 * 3 * K::PLANE_A;
  static_assert(lds <= 160 * 1024, "staged rows do not fit the LDS");
  static const bool attr = hipFuncSetAttribute((const void*)k_convf_rows_s<KC, NC, COLS, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               lds) == hipSuccess;
  if (!attr) return false;
  const int64_t Mc = (int64_t)g.B * g.OH * g.OW;
  hipLaunchKernelGGL((k_convf_rows_s<KC, NC, COLS, 4>), dim3((unsigned)((Mc + 127) / 128)), dim3(256), lds, s, in, pF, bias, out, g,
                     in_bytes, out_bytes);
  return true;
}

int64_t split_planes_bytes(const ConvGeom& g);
bool launch_conv_taps_mfma(bool transposed, const float* in, const float* w, const float* bias, float* out, const ConvGeom& g,
                           hipStream_t s);           // kernels_mfma.hip (the self-test's control experiment)
// read at plan time (mvae_create), not cached: a test builds one engine with and one without the split kernels
static bool split_enabled() {
  const char* e = getenv("MVAE_SPLIT_CONV");
  return e ? atoi(e) != 0 : true;
}

// =================================================================================================
// Hardware self-test.  While a wave interleaves float32 VALU work with v_mfma_f32_32x32x16_bf16 --
// which is what every split kernel does -- a v_pk_fma_f32 of ANOTHER wave on the same SIMD now and then returns a wrong
// value in one 16-lane group (found as 16 wrong outputs of the decoder's Dense layer in a few percent of its launches,
// tools/split_debug2.py; integer VALU and plain v_fma_f32 victims are not affected, the float32-MFMA convolution as the
// neighbour does not trigger it, the kernels share no memory).  The library is therefore built without packed float32
// instructions (_build.py).  The first bind of a process whose plan uses split kernels checks exactly that: the split
// kernels on one stream, a self-checking VALU kernel as this library compiles it on another -- a single wrong value and
// the process keeps the float32-MFMA kernels (mvae_split_conv_status) -- and the same check kernel compiled WITH packed
// float32, whose count of wrong values is reported (mvae_split_conv_erratum; MVAE_SPLIT_SELFTEST=0 skips both).
// =================================================================================================
template <int VV>
__global__ void __launch_bounds__(256) __attribute__((target("packed-fp32-ops"))) k_selftest_victim_v(const float* __restrict__ z, const f32x4* __restrict__ W,
                                                           const f32x4* __restrict__ bias, f32x4* __restrict__ out, int B, int Z, int N4) {
  // debug variants of the check kernel: 1 = the same sums with single v_fma_f32 (no packed float32 instructions),
  // 2 = integer multiply-adds
  __shared__ float sz[4][32];
  const int b0 = blockIdx.y * 4;
  for (int t = threadIdx.x; t < 4 * Z; t += 256) {
    const int r = t / Z, j = t % Z;
    sz[r][j] = (b0 + r < B) ? z[(b0 + r) * Z + j] : 0.f;
  }
  __syncthreads();
  const int n4 = blockIdx.x * 256 + threadIdx.x;
  if (n4 >= N4) return;
  const f32x4 bv = bias[n4];
  float acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = bv[c];
  for (int k = 0; k < Z; ++k) {
    const f32x4 w = W[(int64_t)k * N4 + n4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (VV == 3) continue;
        else if (VV == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[r][c]) : "v"(w[c]), "v"(sz[r][k]));
        else acc[r][c] = __uint_as_float(__float_as_uint(acc[r][c]) + __float_as_uint(w[c]) * (__float_as_uint(sz[r][k]) | 1u));
      }
    if (VV == 3) {                               // the packed instruction itself, whatever the build's code generation does
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x2 s2 = {sz[r][k], sz[r][k]};
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
          f32x2 a2 = {acc[r][c], acc[r][c + 1]};
          const f32x2 w2 = {w[c], w[c + 1]};
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(w2), "v"(s2));
          acc[r][c] = a2[0]; acc[r][c + 1] = a2[1];
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (b0 + r < B) out[(int64_t)(b0 + r) * N4 + n4] = f32x4{acc[r][0], acc[r][1], acc[r][2], acc[r][3]};
}
#define MVAE_SELFTEST_VICTIM_BODY                                                                     \
  __shared__ float sz[4][32];                                                                         \
  const int b0 = blockIdx.y * 4;                                                                      \
  for (int t = threadIdx.x; t < 4 * Z; t += 256) {                                                    \
    const int r = t / Z, j = t % Z;                                                                   \
    sz[r][j] = (b0 + r < B) ? z[(b0 + r) * Z + j] : 0.f;                                              \
  }                                                                                                   \
  __syncthreads();                                                                                    \
  const int n4 = blockIdx.x * 256 + threadIdx.x;                                                      \
  if (n4 >= N4) return;                                                                               \
  const f32x4 bv = bias[n4];                                                                          \
  f32x4 acc[4] = {bv, bv, bv, bv};                                                                    \
  for (int k = 0; k < Z; ++k) {                                                                       \
    const f32x4 w = W[(int64_t)k * N4 + n4];                                                          \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[r] += w * sz[r][k];                             \
  }                                                                                                   \
  _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                       \
    if (b0 + r < B) out[(int64_t)(b0 + r) * N4 + n4] = acc[r];
// the same source with packed-float32 code generation switched back on (what every kernel looked like before the build flag)
__global__ void __launch_bounds__(256) __attribute__((target("packed-fp32-ops")))
k_selftest_victim_packed(const float* __restrict__ z, const f32x4* __restrict__ W, const f32x4* __restrict__ bias,
                         f32x4* __restrict__ out, int B, int Z, int N4) {
  MVAE_SELFTEST_VICTIM_BODY
}
__global__ void __launch_bounds__(256) k_selftest_victim(const float* __restrict__ z, const f32x4* __restrict__ W,
                                                         const f32x4* __restrict__ bias, f32x4* __restrict__ out, int B, int Z, int N4) {
  __shared__ float sz[4][32];                       // the decoder Dense layer's kernel (kernels_dense.hip: k_dense_expand)
  const int b0 = blockIdx.y * 4;
  for (int t = threadIdx.x; t < 4 * Z; t += 256) {
    const int r = t / Z, j = t % Z;
    sz[r][j] = (b0 + r < B) ? z[(b0 + r) * Z + j] : 0.f;
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
    if (b0 + r < B) out[(int64_t)(b0 + r) * N4 + n4] = acc[r];
}
__global__ void k_selftest_fill(float* p, int64_t n, unsigned seed, float scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 2654435761u ^ seed;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
  p[i] = ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) * scale;
}
__global__ void k_selftest_compare(const unsigned* __restrict__ got, const unsigned* __restrict__ ref, int64_t n_ref, int64_t n,
                                   unsigned* bad) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n && got[i] != ref[i % n_ref]) atomicAdd(bad, 1u);
}

static int g_split_state = -1;                 // -1 untested, 1 usable, 2 disabled by the self-test
static int g_split_erratum = -1;               // wrong values the PACKED check kernel returned (-1 = not measured)
static int g_k16_erratum = -1;                 // the same beside the bfloat16-storage kernels (k16_taps) as the neighbour
int split_conv_status() { return !split_enabled() ? 0 : (g_split_state == 2 ? 2 : 1); }
int split_conv_erratum_count() { return g_split_erratum; }
int k16_erratum_count() { return g_k16_erratum; }
// kernels_bf16.hip: the bf16-storage 5x5 kernels (float32 VALU for bias / conversion between v_mfma_f32_32x32x16_bf16)
bool launch16_taps(bool transposed, const void* in, const float* w, const float* bias, void* out, const ConvGeom& g,
                   hipStream_t s, const float* w2, const float* bias2, void* out2, bool* chained);

// One measurement: the split kernels on one stream, check kernel `vv` back to back on another; number of wrong values,
// or -1 if it could not run.  vv: 0 = as this library is compiled, 1 = v_fma_f32, 2 = integer, 3 = v_pk_fma_f32 (asm), 4 = compiled with packed float32;
// conv: 0 = split kernels, 1 = float32-MFMA kernels (control), 2 = the bfloat16-storage kernels (k16_taps: the tensors are then
// read as bf16 -- same byte footprint or less, values irrelevant to what is measured)
static long selftest_run(int vv, int conv, int VB) {
  const int nb = 256, Z = 16, N = 32768, NV = 96 * 2 / VB > 8 ? 96 * 2 / VB : 8, ROUNDS = 6;
  ConvGeom g{nb, 32, 32, 64, 16, 16, 32, 5, 5, 2, 2, 1, 1};        // decoder layer: T-form 32 -> 64 and F-form 64 -> 32
  const int64_t nbig = (int64_t)nb * 32 * 32 * 64, nsm = (int64_t)nb * 16 * 16 * 32, nvic = (int64_t)VB * N;
  float *big = nullptr, *small = nullptr, *w = nullptr, *vz = nullptr, *vw = nullptr, *vb = nullptr, *vout = nullptr, *vref = nullptr;
  void* planes = nullptr;
  unsigned* bad = nullptr;
  hipStream_t sa = nullptr, sb = nullptr;
  bool ok = true;
  unsigned hbad = 0;
  auto chk = [&](hipError_t err) { if (err != hipSuccess) ok = false; return err == hipSuccess; };
  if (chk(hipMalloc(&big, nbig * 4)) && chk(hipMalloc(&small, nsm * 4)) && chk(hipMalloc(&w, 25 * 64 * 32 * 4)) &&
      chk(hipMalloc(&planes, split_planes_bytes(g))) && chk(hipMalloc(&vz, VB * Z * 4)) && chk(hipMalloc(&vw, (int64_t)Z * N * 4)) &&
      chk(hipMalloc(&vb, N * 4)) && chk(hipMalloc(&vout, nvic * NV * 4)) && chk(hipMalloc(&vref, nvic * 4)) &&
      chk(hipMalloc(&bad, 4)) && chk(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)) &&
      chk(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking))) {
    auto fill = [&](float* p, int64_t n, unsigned seed, float sc) {
      hipLaunchKernelGGL(k_selftest_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sa, p, n, seed, sc);
    };
    fill(small, nsm, 1u, 1.f); fill(big, nbig, 2u, 1.f); fill(w, 25 * 64 * 32, 3u, 0.05f);
    fill(vz, VB * Z, 4u, 0.5f); fill(vw, (int64_t)Z * N, 5u, 0.03f); fill(vb, N, 6u, 0.2f);
    (void)hipMemsetAsync(bad, 0, 4, sa);
    launch_split_weights(w, planes, g, sa);
    const dim3 vgrid((N / 4 + 255) / 256, (VB + 3) / 4);
    auto victim = [&](hipStream_t st, float* dst) {
#define MVAE_VIC(K) hipLaunchKernelGGL(K, vgrid, dim3(256), 0, st, vz, (const f32x4*)vw, (const f32x4*)vb, (f32x4*)dst, VB, Z, N / 4)
      if (vv == 1) MVAE_VIC(k_selftest_victim_v<1>); else if (vv == 2) MVAE_VIC(k_selftest_victim_v<2>);
      else if (vv == 3) MVAE_VIC(k_selftest_victim_v<3>); else if (vv == 4) MVAE_VIC(k_selftest_victim_packed); else MVAE_VIC(k_selftest_victim);
#undef MVAE_VIC
    };
    victim(sa, vref);
    chk(hipStreamSynchronize(sa));                 // reference = the same kernel on an idle GPU
    const int saved = g_split_state;
    g_split_state = 1;                             // (launch_conv_taps_split checks the state)
    for (int r = 0; r < ROUNDS && ok; ++r) {
      for (int k = 0; k < 4; ++k) {                // ~1 ms of the convolution kernels on stream a ...
        if (conv == 1) {
          launch_conv_taps_mfma(true, small, w, nullptr, big, g, sa);
          launch_conv_taps_mfma(false, big, w, nullptr, small, g, sa);
        } else if (conv == 2) {
          if (!launch16_taps(true, small, w, nullptr, big, g, sa, nullptr, nullptr, nullptr, nullptr) ||
              !launch16_taps(false, big, w, nullptr, small, g, sa, nullptr, nullptr, nullptr, nullptr)) ok = false;
        } else {
          launch_conv_taps_split(true, small, planes, nullptr, big, g, sa);
          launch_conv_taps_split(false, big, planes, nullptr, small, g, sa);
        }
      }
      for (int k = 0; k < NV; ++k) victim(sb, vout + (int64_t)k * nvic);     // ... the checked kernel back to back on stream b
      hipLaunchKernelGGL(k_selftest_compare, dim3((unsigned)((nvic * NV + 255) / 256)), dim3(256), 0, sb, (const unsigned*)vout,
                         (const unsigned*)vref, nvic, nvic * NV, bad);
      chk(hipStreamSynchronize(sb));
      chk(hipStreamSynchronize(sa));
      fill(small, nsm, 1u, 1.f);                   // (the F-form launches overwrote the small tensor)
    }
    g_split_state = saved;
    chk(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
  }
  for (void* p : {(void*)big, (void*)small, (void*)w, planes, (void*)vz, (void*)vw, (void*)vb, (void*)vout, (void*)vref, (void*)bad})
    if (p) (void)hipFree(p);
  if (sa) (void)hipStreamDestroy(sa);
  if (sb) (void)hipStreamDestroy(sb);
  (void)hipGetLastError();
  return ok ? (long)hbad : -1;
}

// first bind of a process whose plan has split convolutions.  false = the split kernels stay off for this process.
bool split_selftest() {
  if (g_split_state >= 0) return g_split_state == 1;
  const char* e = getenv("MVAE_SPLIT_SELFTEST");
  const int mode = e ? atoi(e) : 1;
  if (mode == 0) { g_split_state = 1; return true; }
  if (mode >= 3) {                                 // diagnostics: MVAE_SELFTEST_VICTIM / _CONV / _VB pick one measurement
    const int vv = getenv("MVAE_SELFTEST_VICTIM") ? atoi(getenv("MVAE_SELFTEST_VICTIM")) : 0;
    const int cv = getenv("MVAE_SELFTEST_CONV") ? atoi(getenv("MVAE_SELFTEST_CONV")) : 0;
    const int VB = getenv("MVAE_SELFTEST_VB") ? atoi(getenv("MVAE_SELFTEST_VB")) : 2;
    fprintf(stderr, "mvae: self-test diagnostics: check kernel %d, conv %d, batch %d: %ld wrong values\n", vv, cv, VB, selftest_run(vv, cv, VB));
    g_split_state = 1;
    return true;
  }
  const long shipped = selftest_run(0, 0, 2);      // the instructions this library is built from
  const long packed = selftest_run(4, 0, 2);       // the same source compiled WITH packed float32: reports the erratum itself
  g_split_erratum = (int)(packed < 0 ? -1 : (packed > 2000000000L ? 2000000000L : packed));
  const long packed16 = selftest_run(4, 2, 2);     // ... and beside the bfloat16-storage kernels (are THEY such a neighbour?)
  g_k16_erratum = (int)(packed16 < 0 ? -1 : (packed16 > 2000000000L ? 2000000000L : packed16));
  g_split_state = shipped > 0 ? 2 : 1;
  if (shipped > 0 || mode >= 2)
    fprintf(stderr, "mvae: split-bf16 convolution self-test: %ld wrong values in the check kernel as compiled, %ld in its "
                    "v_pk_fma_f32 form%s\n", shipped, packed, shipped > 0 ? " -- this board keeps the float32-MFMA 5x5 kernels" : "");
  return shipped <= 0;
}

// bytes of split weight planes one k x k layer needs (both forms)
int64_t split_planes_bytes(const ConvGeom& g) { return (int64_t)2 * g.KH * g.KW * 3 * g.CI * g.CO * 2; }

// does the split path cover this layer: 5 x 5 kernels at stride 2 x 2 between 32 and 64 channels (the tap loops are unrolled
// for exactly these tap counts); anything else keeps the float32-MFMA / generic kernels
bool split_conv_covers(const ConvGeom& g) {
  if (!split_enabled()) return false;
  if (g.KH != 5 || g.KW != 5 || g.SH != 2 || g.SW != 2) return false;
  return (g.CI == 32 && g.CO == 64) || (g.CI == 64 && g.CO == 32);
}

// W [KH*KW][CI][CO] -> planes (F-form block first, T-form block behind it); one launch per layer and step
void launch_split_weights(const float* W, void* planes, const ConvGeom& g, hipStream_t s) {
  const int taps = g.KH * g.KW, n = taps * g.CI * g.CO;
  char* pF = static_cast<char*>(planes);
  char* pT = pF + (int64_t)taps * 3 * g.CI * g.CO * 2;
  hipLaunchKernelGGL(k_split_weights, dim3((n + 255) / 256), dim3(256), 0, s, W, pF, pT, taps, g.CI, g.CO);
}

// =================================================================================================
// The backward pair of a 1x1 convolution 64 -> 64 of the MobileNetV3 block (k_gemm_dual<64, 1 / 2>, kernels_mfma.hip) with
// split products.  The float32 kernel spends 55 us of matrix-core time per launch at M = 2^19 next to 77 us of memory time
// (AI 21 FLOP/B needs 70 % of the float32 MFMA peak at 5.2 TB/s): the two do not overlap that well and it runs at 3.9 TB/s.
// Here a tile is 48 bf16 MFMAs per wave (1536 cycles instead of 4096) plus ~230 VALU instructions per thread for the split,
// which each thread does ONCE on the 32 values it stages (the same element is an MFMA operand of two waves).
//   block tile 64 rows, waves = 2 row groups x 2 column groups as in k_gemm_dual; LDS: three bf16 planes of X and of
//   (gated) aux, [plane][row][64 channels], 16-byte chunks XOR-swizzled with a pattern that is conflict-free both for the
//   row-major ds_read_b128 fragments (data GEMM, A operand) and for the ds_read_b64_tr_b16 fragments (weight gradient,
//   both operands pixel-major); MODE 1 keeps the raw float32 aux tile as well (ReLU mask + gate-gradient dot in the epilogue).
//   Wt planes live in registers (48).  Epilogue, residual prefetch, slots: as k_gemm_dual.  db = column sums of the values
//   a thread stages (its four float4 share the channel quad), folded once at the end.
// M % 64 == 0, rows_per_image a power of two >= 64 (a tile lies in one image: one gate vector, one dot atomic per wave).
// =================================================================================================

#ifndef MVAE_DUAL_VARIANT
#define MVAE_DUAL_VARIANT 0        /* tools/dual_variants.sh: 1 no weight gradient, 2 no data GEMM, 3 no split arithmetic, 4 plain epilogue, 5 no tile loads */
#endif
template <int MODE>
__global__ void __launch_bounds__(256, 2) k_gemm_dual_s(const float* __restrict__ X, const float* __restrict__ W,
                                                        const float* __restrict__ aux, const float* __restrict__ gate,
                                                        const float* __restrict__ residual, float* __restrict__ Y,
                                                        float* __restrict__ dW, float* __restrict__ db,
                                                        float* __restrict__ dot_out, int64_t M, int rpi_shift, int nslots,
                                                        int64_t slot_stride, int dslots, int64_t dstride) {
  // MODE 3 = MODE 1 for the fully fused backward (k_mn_bwd_s, kernels_fused.hip, recomputes Y = dt2 itself): nothing is
  // stored at Y's 256 bytes per pixel; instead the ReLU mask of aux (t1 > 0) goes out as two 32-bit words per pixel,
  // mask[pixel*2 + (channel >> 5)] bit (channel & 31), through Y's pointer.
  constexpr bool GATED = MODE == 1 || MODE == 3;
  constexpr int C = 64, C4 = 16, TR = 64, LD = 4, PL = TR * C * 2;          // PL = bytes of one plane
  __shared__ __attribute__((aligned(16))) char lds[6 * PL + (GATED ? TR * C * 4 : 0)];
  char* pX = lds;
  char* pA = lds + 3 * PL;
  float* sA = reinterpret_cast<float*>(lds + 6 * PL);                      // MODE 1: raw aux (t1), float32, SWZ layout
  if (GATED) dot_out += (int64_t)(blockIdx.x % dslots) * dstride;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = wave & 1, rw = wave >> 1, n0 = nw * 32;
  const int i = lane & 31, h = lane >> 5;
  const int sr = threadIdx.x >> 4, sc4 = threadIdx.x & 15;                 // staging: rows sr + 16 j, channel quad sc4
  // raw aux tile: plain row-major (one address register for the 16 epilogue reads; the two lane halves read rows 4 apart,
  // i.e. the same banks: 2 cycles per ds_read_b32 instead of 1, against ~15 registers of swizzled addresses)
#define SWZ4(r, c4) ((r) * C4 + (c4))
#define SWZ1(r, c) ((r) * C + (c))
  // Wt[k][n = n0 + i] = W[n * 64 + k], k = kk*16 + 8h + j: three planes per k-step, resident
  bf16x8 wreg[4][3];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const f32x4* wp = reinterpret_cast<const f32x4*>(W + (int64_t)(n0 + i) * C + kk * 16 + 8 * h);
    u32x2 a1, a2, a3, b1, b2, b3;
    split4(__builtin_bit_cast(u32x4, wp[0]), a1, a2, a3);
    split4(__builtin_bit_cast(u32x4, wp[1]), b1, b2, b3);
    wreg[kk][0] = as_frag(u32x4{a1[0], a1[1], b1[0], b1[1]});
    wreg[kk][1] = as_frag(u32x4{a2[0], a2[1], b2[0], b2[1]});
    wreg[kk][2] = as_frag(u32x4{a3[0], a3[1], b3[0], b3[1]});
  }
  const int64_t ntiles = M / TR;
  const f32x4* X4 = reinterpret_cast<const f32x4*>(X);
  const f32x4* A4 = reinterpret_cast<const f32x4*>(aux);
  f32x16 accw[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[kt][r] = 0.f;
  f32x4 bs4 = {0.f, 0.f, 0.f, 0.f};
  struct Stage { f32x4 x[LD], a[LD], g; };                 // g: the tile's image gate (a 64-row tile lies in one image)
  Stage S;
  auto load_tile = [&](int64_t tile) {
    const f32x4* px = X4 + tile * (TR * C4) + threadIdx.x;
    const f32x4* pa = A4 + tile * (TR * C4) + threadIdx.x;
#pragma unroll
    for (int j = 0; j < LD; ++j) { S.x[j] = px[j * 256]; S.a[j] = pa[j * 256]; }
    if constexpr (GATED)
      S.g = reinterpret_cast<const f32x4*>(gate)[(int64_t)((uint32_t)(tile * TR) >> rpi_shift) * C4 + sc4];
  };
  float res[16];
  auto load_res = [&](int64_t tile) {
    const float* pr = residual + (tile * TR + rw * 32 + 4 * h) * C + n0 + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = pr[((r & 3) + 8 * (r >> 2)) * C];
  };
  int64_t tile = blockIdx.x;
  const int64_t g1 = gridDim.x;
  if (tile < ntiles) {
    load_tile(tile);
    if constexpr (MODE == 2) load_res(tile);
  }
  for (; tile < ntiles; tile += g1) {
    const int64_t row0 = tile * TR + rw * 32;
    const int64_t next = tile + g1 < ntiles ? tile + g1 : tile;
    __syncthreads();                                       // previous tile fully consumed by all waves
#pragma unroll
    for (int j = 0; j < LD; ++j) {
      const int r = sr + 16 * j;
      const int off = dual_off(r, sc4 >> 1) + (sc4 & 1) * 8;
      u32x2 p1, p2, p3;
#if MVAE_DUAL_VARIANT == 3
#define split4(V, A, B_, C_) do { const u32x4 v_ = (V); A = u32x2{v_[0], v_[1]}; B_ = u32x2{v_[2], v_[3]}; C_ = u32x2{v_[1], v_[2]}; } while (0)
#endif
      split4(__builtin_bit_cast(u32x4, S.x[j]), p1, p2, p3);
      *reinterpret_cast<u32x2*>(pX + off) = p1;
      *reinterpret_cast<u32x2*>(pX + PL + off) = p2;
      *reinterpret_cast<u32x2*>(pX + 2 * PL + off) = p3;
      bs4 += S.x[j];
      f32x4 a = S.a[j];
      if constexpr (GATED) {
        reinterpret_cast<f32x4*>(sA)[SWZ4(r, sc4)] = a;
        a = a * S.g;
      }
      split4(__builtin_bit_cast(u32x4, a), p1, p2, p3);
      *reinterpret_cast<u32x2*>(pA + off) = p1;
      *reinterpret_cast<u32x2*>(pA + PL + off) = p2;
      *reinterpret_cast<u32x2*>(pA + 2 * PL + off) = p3;
#if MVAE_DUAL_VARIANT == 3
#undef split4
#endif
    }
    __syncthreads();
#if MVAE_DUAL_VARIANT != 5
    load_tile(next);                                       // prefetch under the MFMAs (past the end: refetch, unused)
#endif
    const int64_t ebase = (row0 + 4 * h) * C + n0 + i;
    // ---- Y tile = X . Wt (32 rows x 32 columns per wave), then dW[:, n0..n0+31] += (aux * gate)^T X over the wave's 32 rows.
    // The LDS fragments of a phase are requested one phase ahead of its MFMAs (sched_barrier keeps hipcc from sinking the
    // reads next to their first use, where every group of MFMAs waited for its own ds_read).
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr bool kData = MVAE_DUAL_VARIANT != 2, kWgrad = MVAE_DUAL_VARIANT != 1;
    bf16x8 xa[4][3], fa[2][2][3], fb[2][3];
    auto read_x = [&](int kk) {
      const int off = dual_off(rw * 32 + i, 2 * kk + h);
#pragma unroll
      for (int p = 0; p < 3; ++p) xa[kk][p] = as_frag(*reinterpret_cast<const u32x4*>(pX + p * PL + off));
    };
    auto read_b = [&](int sidx) {
#pragma unroll
      for (int p = 0; p < 3; ++p) fb[sidx][p] = dual_frag_cols(pX + p * PL, lane, rw * 32, nw, sidx);
    };
    auto read_a = [&](int sidx, int kt) {
#pragma unroll
      for (int p = 0; p < 3; ++p) fa[sidx][kt][p] = dual_frag_cols(pA + p * PL, lane, rw * 32, kt, sidx);
    };
#define MVAE_SB() __builtin_amdgcn_sched_barrier(0)
    // at most 12 fragments (48 registers) requested or waiting at any time
    if (kData) { read_x(0); read_x(1); }
    MVAE_SB();
    if (kData) { read_x(2); read_x(3); MVAE_SPLIT6(acc, xa[0], wreg[0]); MVAE_SPLIT6(acc, xa[1], wreg[1]); }
    MVAE_SB();
    if (kWgrad) { read_b(0); read_a(0, 0); }
    if (kData) { MVAE_SPLIT6(acc, xa[2], wreg[2]); MVAE_SPLIT6(acc, xa[3], wreg[3]); }
    MVAE_SB();
    if (kWgrad) { read_a(0, 1); read_b(1); MVAE_SPLIT6(accw[0], fa[0][0], fb[0]); }
    MVAE_SB();
    if (kWgrad) { read_a(1, 0); MVAE_SPLIT6(accw[1], fa[0][1], fb[0]); }
    MVAE_SB();
    float av[16];                                          // MODE 1: raw aux in the accumulator layout (mask + dot)
    if (kWgrad) { read_a(1, 1); MVAE_SPLIT6(accw[0], fa[1][0], fb[1]); }
    MVAE_SB();
    if (kWgrad) { MVAE_SPLIT6(accw[1], fa[1][1], fb[1]); }
    if constexpr (GATED && MVAE_DUAL_VARIANT != 4) {
#pragma unroll
      for (int r = 0; r < 16; ++r) av[r] = sA[SWZ1(rw * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, n0 + i)];
    }
    MVAE_SB();
#undef MVAE_SB
    // ---- epilogue straight from the accumulator layout
    if constexpr (MODE == 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += res[r];
      load_res(next);                                      // ahead of the stores (k_gemm_dual's header comment)
    }
    float* py = Y + ebase;
    if constexpr (GATED && MVAE_DUAL_VARIANT != 4) {
      float dsum = 0.f;                                    // the wave's 32 rows lie in one image
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dsum += acc[r] * av[r];
        if constexpr (MODE == 1) {
          py[((r & 3) + 8 * (r >> 2)) * C] = __uint_as_float((__float_as_uint(acc[r]) & ~1u) | (av[r] > 0.f ? 1u : 0u));
        } else {
          // lanes 0-31: pixel row0 + rc, lanes 32-63: pixel row0 + rc + 4, channels n0 .. n0 + 31 in lane order
          const unsigned long long bal = __ballot(av[r] > 0.f);
          if (i == 0) {
            unsigned* mw = reinterpret_cast<unsigned*>(Y) + (row0 + (r & 3) + 8 * (r >> 2) + 4 * h) * 2 + nw;
            *mw = h ? (unsigned)(bal >> 32) : (unsigned)bal;
          }
        }
      }
      dsum += __shfl_xor(dsum, 32, 64);
      if (h == 0) atomicAdd(dot_out + (int64_t)((uint32_t)row0 >> rpi_shift) * C + n0 + i, dsum);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) py[((r & 3) + 8 * (r >> 2)) * C] = acc[r];
    }
  }
  // ---- reduce the row groups' dW slabs through LDS, then one coalesced float-atomic set per block
  float* red = reinterpret_cast<float*>(lds);              // C*C floats = 16 KB <= the X planes
  for (int step = 0; step < 2; ++step) {
    __syncthreads();
    if (rw == step) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int idx = ci * C + n0 + i;
          red[idx] = (step == 0 ? 0.f : red[idx]) + accw[kt][r];
        }
    }
  }
  f32x4* redb = reinterpret_cast<f32x4*>(pA);              // [16 staging rows][16 channel quads]
  redb[sr * C4 + sc4] = bs4;
  __syncthreads();
  const int64_t slot = (int64_t)(blockIdx.x % nslots) * slot_stride;
  for (int idx = threadIdx.x; idx < C * C; idx += 256) atomicAdd(&dW[slot + idx], red[idx]);
  if (db != nullptr && threadIdx.x < C) {
    const float* rb = reinterpret_cast<const float*>(pA);
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += rb[q * C + threadIdx.x];
    atomicAdd(&db[slot + threadIdx.x], t);
  }
#undef SWZ4
#undef SWZ1
}

static bool dual_split_enabled() {
  static const bool on = [] { const char* e = getenv("MVAE_SPLIT_DUAL"); return e ? atoi(e) != 0 : true; }();
  return on && split_enabled();
}
// which kernel launch_gemm_dual_split would run for this shape (profile tags); nullptr = not covered
const char* gemm_dual_split_kernel(bool gated, int64_t M, int64_t rows_per_image, int C) {
  if (!dual_split_enabled() || g_split_state == 2) return nullptr;
  if (C != 64 || M % 64 != 0 || M >= (1LL << 31)) return nullptr;
  if (rows_per_image < 64 || (rows_per_image & (rows_per_image - 1)) != 0) return nullptr;      // a 64-row tile lies in one image
  return gated ? "k_gemm_dual_s<1>" : "k_gemm_dual_s<2>";
}
// conv2 pair (gate + dot_out, no residual) or conv0 pair (residual, no gate / dot) of a 64 -> 64 1x1 convolution
bool launch_gemm_dual_split(const float* X, const float* W, const float* aux, const float* gate, const float* residual,
                            float* Y, float* dW, float* db, float* dot_out, int64_t M, int64_t rows_per_image, int C,
                            GradSlots sl, int dslots, int64_t dstride, int cap, hipStream_t s) {
  const bool m1 = gate && dot_out && !residual, m2 = residual && !gate && !dot_out;
  if (!(m1 || m2) || !gemm_dual_split_kernel(m1, M, rows_per_image, C)) return false;
  const int64_t ntiles = M / 64;
  const int grid = (int)(ntiles < cap ? ntiles : cap);
  const int sh = 63 - __builtin_clzll((unsigned long long)rows_per_image);
#define MVAE_DS(MODE)                                                                                                  \
  hipLaunchKernelGGL((k_gemm_dual_s<MODE>), dim3(grid), dim3(256), 0, s, X, W, aux, gate, residual, Y, sl.at(dW), sl.at(db), \
                     dot_out, M, sh, sl.count(), sl.stride, dslots < 1 ? 1 : dslots, dstride)
  if (m1) MVAE_DS(1); else MVAE_DS(2);
#undef MVAE_DS
  return true;
}

// conv2 pair WITHOUT dt2: dW2, db2, the gate gradient, and the ReLU mask of t1 as bit words (mask[pixel*2 + (c >> 5)]) for
// k_mn_bwd_s, which recomputes dt2 from dout.  Same coverage as the gated split pair.
bool launch_gemm_dual_stats(const float* X, const float* W, const float* aux, const float* gate, unsigned* mask, float* dW,
                            float* db, float* dot_out, int64_t M, int64_t rows_per_image, int C, GradSlots sl, int dslots,
                            int64_t dstride, int cap, hipStream_t s) {
  if (!gate || !dot_out || !mask || !gemm_dual_split_kernel(true, M, rows_per_image, C)) return false;
  const int64_t ntiles = M / 64;
  static const int cus = [] { const char* e = getenv("MVAE_BIG_CUS"); int n = e ? atoi(e) : 224; return n < 8 ? 8 : (n > 256 ? 256 : n); }();
  if (cap <= 0) cap = 2 * cus;                             // two resident blocks per CU (run_gemm_dual, kernels_mfma.hip)
  const int grid = (int)(ntiles < cap ? ntiles : cap);
  const int sh = 63 - __builtin_clzll((unsigned long long)rows_per_image);
  hipLaunchKernelGGL((k_gemm_dual_s<3>), dim3(grid), dim3(256), 0, s, X, W, aux, gate, (const float*)nullptr,
                     reinterpret_cast<float*>(mask), sl.at(dW), sl.at(db), dot_out, M, sh, sl.count(), sl.stride,
                     dslots < 1 ? 1 : dslots, dstride);
  return true;
}

// F-form (in = big) / T-form (in = small) k x k convolution from pre-split weight planes.  false = shape not covered.
bool launch_conv_taps_split(bool transposed, const float* in, const void* planes, const float* bias, float* out,
                            const ConvGeom& g, hipStream_t s) {
  if (g_split_state == 2) return false;
  if (!planes || g.KH != 5 || g.KW != 5 || g.SH != 2 || g.SW != 2 || !((g.CI == 32 && g.CO == 64) || (g.CI == 64 && g.CO == 32))) return false;
  if ((int64_t)g.B * g.IH * g.IW * g.CI * 4 >= (1LL << 31) || (int64_t)g.B * g.OH * g.OW * g.CO * 4 >= (1LL << 31)) return false;
  const int KC = transposed ? g.CO : g.CI;
  int64_t Mc;
  int classes = 1;
  if (transposed) { classes = g.SH * g.SW; Mc = (int64_t)g.B * ((g.IH + g.SH - 1) / g.SH) * ((g.IW + g.SW - 1) / g.SW); }
  else Mc = (int64_t)g.B * g.OH * g.OW;
  const unsigned in_bytes = (unsigned)((int64_t)g.B * (transposed ? g.OH * g.OW : g.IH * g.IW) * KC * 4);
  const unsigned out_bytes = (unsigned)((int64_t)g.B * (transposed ? g.IH * g.IW * g.CI : g.OH * g.OW * g.CO) * 4);
  // 4 waves per block (128 output pixels share a staged weight slice); 8-wave blocks measured 5-10 % slower on all four forms
  const dim3 grid((unsigned)((Mc + 127) / 128), classes);
  const int taps = g.KH * g.KW;
  const char* pF = static_cast<const char*>(planes);
  const char* pT = pF + (int64_t)taps * 3 * g.CI * g.CO * 2;
  // transposed, stride 2, even sizes, SAME padding of a 5 x 5 kernel: the four sub-pixel phases merged in one block
  static const bool merged = [] { const char* e = getenv("MVAE_CONVT_MERGED"); return e ? atoi(e) != 0 : true; }();
  if (transposed && merged && g.IH == 2 * g.OH && g.IW == 2 * g.OW && g.PT == 1 && g.PL == 1) {
    const char* pTm = static_cast<const char*>(planes) + (int64_t)g.KH * g.KW * 3 * g.CI * g.CO * 2;
    const int64_t Mm = (int64_t)g.B * g.OH * g.OW;
    const dim3 gm((unsigned)((Mm + 127) / 128));
    if (g.CO == 32) hipLaunchKernelGGL((k_convt_merged_s<32, 64, 4>), gm, dim3(256), 0, s, in, pTm, bias, out, g, in_bytes, out_bytes);
    else hipLaunchKernelGGL((k_convt_merged_s<64, 32, 4>), gm, dim3(256), 0, s, in, pTm, bias, out, g, in_bytes, out_bytes);
    return true;
  }
  // F-form with the input rows of a kernel row staged in LDS: output width a power of two up to 32, or a multiple of 32
  // Default 2 = the 32 -> 64 layers only: there the staged rows fit twice per CU (79 KB per block) and the kernel is 10 % faster
  // (97 -> 87 us at batch 512); with 64 input channels a block needs 132 KB, one block per CU, and runs as fast as the
  // per-tap gathers (138 -> 134 us; MVAE_CONVF_ROWS=1 selects it: 2.3x fewer gathered bytes, same time).
  static const int rows = [] { const char* e = getenv("MVAE_CONVF_ROWS"); return e ? atoi(e) : 2; }();
  if (!transposed && (rows == 1 || (rows == 2 && g.CI == 32)) && (g.OW % 32 == 0 || g.OW == 16 || g.OW == 8 || g.OW == 4)) {
    const char* pFr = static_cast<const char*>(planes);
    const int cols = g.OW >= 32 ? 32 : g.OW;
    bool ok = false;
#define MVAE_CR(A, B_) (cols == 32 ? launch_convf_rows<A, B_, 32>(in, pFr, bias, out, g, in_bytes, out_bytes, s)          \
                        : cols == 16 ? launch_convf_rows<A, B_, 16>(in, pFr, bias, out, g, in_bytes, out_bytes, s)        \
                        : cols == 8 ? launch_convf_rows<A, B_, 8>(in, pFr, bias, out, g, in_bytes, out_bytes, s)          \
                                    : launch_convf_rows<A, B_, 4>(in, pFr, bias, out, g, in_bytes, out_bytes, s))
    if (g.CI == 32) ok = MVAE_CR(32, 64); else ok = MVAE_CR(64, 32);
#undef MVAE_CR
    if (ok) return true;
  }
  // PF taps of register prefetch: 3 where a tap's pixels are 16 registers (KC = 32), 2 where they are 32 (KC = 64)
#define MVAE_CS(A, B_, TF, PF_, P) hipLaunchKernelGGL((k_conv_taps_s<A, B_, TF, PF_, 4>), grid, dim3(256), 0, s, in, P, bias, out, g, in_bytes, out_bytes)
  if (!transposed) { if (g.CI == 32) MVAE_CS(32, 64, false, 3, pF); else MVAE_CS(64, 32, false, 2, pF); }
  else { if (g.CO == 32) MVAE_CS(32, 64, true, 3, pT); else MVAE_CS(64, 32, true, 2, pT); }
#undef MVAE_CS
  return true;
}

}  // namespace mvae
