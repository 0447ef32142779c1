// kernels_split_wgrad.hip -- float32 5 x 5 weight gradients as split-bf16 products on the bf16 matrix cores (round 4).
//
//   dW[kh][kw][ci][co] += sum_m big[gather(m, kh, kw)][ci] * small[m][co] ;  db[co] += sum_m small[m][co]
//
// k_wgrad_taprow<*, *, 5> (kernels_mfma.hip) was the last kernel of the float32 step bound by the float32 MFMA rate
// (95 - 101 of 157 TFLOP/s: v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate).  Same decomposition here -- one block
// per (row chunk, kernel row kh), the five kw taps share the staged `small` tile, 1-D grid laid out so that the KH blocks
// of a chunk share an XCD's L2 -- with the arithmetic of kernels_split.hip: every float32 operand is the exact sum of three
// bf16 values (split.h) and a product is six bf16 MFMAs accumulated in float32 (terms below 2^-24 dropped), 6 / 16 of the
// float32 MFMA time.  The contraction runs over pixels, so BOTH operands are needed pixel-major per channel: the tiles are
// staged row-major [16 pixels][C] as three bf16 planes (split once per element, by the lane that loads it) and read back
// with ds_read_b64_tr_b16 (two per fragment), as k16_wgrad does for bf16 storage.
#include "kernels.h"
#include "prof.h"
#include "split.h"

namespace mvae {

namespace {
typedef short s16x4w __attribute__((ext_vector_type(4)));
typedef short s16x8w __attribute__((ext_vector_type(8)));

#define WAVE_LDS_SYNCW()                                     \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)

// row-major plane of 16 rows x C bf16, 16-byte chunks XOR-swizzled (the layout of kernels_bf16.hip's tiles)
template <int C>
__device__ __forceinline__ int wtile_off(int row, int chunk) {
  if constexpr (C == 64) return row * 128 + ((chunk ^ (row & 7)) << 4);
  else return (row >> 1) * 128 + ((((row & 1) * 4 + chunk) ^ ((row >> 1) & 7)) << 4);
}
// transposed fragment of one plane: channel ct*32 + (lane & 31), rows 8 (lane >> 5) .. + 7 of the 16-row tile
template <int C>
__device__ __forceinline__ bf16x8 wfrag_cols(const char* plane, int lane, int ct) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int c0 = ct * 32 + 16 * (g & 1) + 4 * p;
  const int rb = 8 * (g >> 1);
  typedef __attribute__((address_space(3))) s16x4w lds_s16x4;
  const int o0 = wtile_off<C>(rb + q, c0 >> 3) + (c0 & 7) * 2;
  const int o1 = wtile_off<C>(rb + 4 + q, c0 >> 3) + (c0 & 7) * 2;
  const s16x4w a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(plane + o0));
  const s16x4w b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(plane + o1));
  s16x8w f = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, f);
}
}  // namespace

template <int CI, int CO, int TG>
__global__ void __launch_bounds__(256, 2) k_wgrad_taprow_s(const float* __restrict__ big, const float* __restrict__ small,
                                                           float* __restrict__ dW, float* __restrict__ db, ConvGeom g, int64_t M,
                                                           int64_t rows_per_block) {
  constexpr int KT = CI / 32, NT = CO / 32, R = 16;
  constexpr int PB = R * CI * 2, PS = R * CO * 2;              // bytes of one plane of the big / small tile
  constexpr int TILE = 3 * (PB + PS);
  __shared__ __attribute__((aligned(16))) char lds[(4 * TILE > CI * CO * 4) ? 4 * TILE : CI * CO * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  char* tb = lds + wave * TILE;                   // big planes   [3][16][CI]
  char* ts = tb + 3 * PB;                         // small planes [3][16][CO]
  const int xcd = blockIdx.x & 7, kh = (blockIdx.x >> 3) % g.KH;
  const uint32_t chunk = ((blockIdx.x >> 3) / g.KH) * 8u + xcd;
  if ((int64_t)chunk * rows_per_block >= M) return;            // block-uniform, before any barrier
  f32x16 acc[TG][KT][NT];
#pragma unroll
  for (int t = 0; t < TG; ++t)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][kt][nt][e] = 0.f;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};              // this lane's four columns of `small` (column group lane % (CO / 4))
  uint32_t m_begin = chunk * (uint32_t)rows_per_block;
  uint32_t m_end = m_begin + (uint32_t)rows_per_block;
  if (m_end > (uint32_t)M) m_end = (uint32_t)M;
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(big), 0,
      (int)((unsigned)g.B * g.IH * g.IW * CI * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(small), 0,
      (int)((unsigned)M * CO * 4u), 0x00020000);
  constexpr int LX = R * CI / 4 / 64, LG = (R * CO / 4 + 63) / 64;      // 16-byte float4 pieces per lane per tile
  constexpr int CPX = CI / 4, CPG = CO / 4;                            // float4 pieces per pixel
  const int HWo = g.OH * g.OW;
  const bool pow2 = (g.OW & (g.OW - 1)) == 0 && (g.OH & (g.OH - 1)) == 0;
  const int lgw = 31 - __builtin_clz((unsigned)g.OW), lgh = 31 - __builtin_clz((unsigned)g.OH);
  unsigned pbase[LX];
  int px0[LX];
  auto decode = [&](uint32_t row0) {
#pragma unroll
    for (int j = 0; j < LX; ++j) {
      const int c = j * 64 + lane, pr = c / CPX, ch = c % CPX;
      const uint32_t m = row0 + pr;
      int bi, oh, ow;
      if (pow2) { ow = (int)(m & (uint32_t)(g.OW - 1)); oh = (int)((m >> lgw) & (uint32_t)(g.OH - 1)); bi = (int)(m >> (lgw + lgh)); }
      else { const uint32_t b = m / (uint32_t)HWo, rem = m - b * (uint32_t)HWo; oh = (int)(rem / (uint32_t)g.OW); ow = (int)(rem - (uint32_t)oh * (uint32_t)g.OW); bi = (int)b; }
      const int yy = oh * g.SH + kh - g.PT, x0 = ow * g.SW - g.PL;
      const bool ok = m < m_end && (unsigned)yy < (unsigned)g.IH;
      pbase[j] = ok ? (unsigned)(((bi * g.IH + yy) * g.IW + x0) * CI + ch * 4) * 4u : 0xC0000000u;
      px0[j] = x0;
    }
  };
  u32x4 xq[LX], gq[LG];
  auto load_big = [&](int kw) {
#pragma unroll
    for (int j = 0; j < LX; ++j) {
      const unsigned off = (unsigned)(px0[j] + kw) < (unsigned)g.IW ? pbase[j] + (unsigned)(kw * CI * 4) : 0x80000000u;
      xq[j] = __builtin_amdgcn_raw_buffer_load_b128(brs, off, 0, 0);
    }
  };
  auto load_small = [&](uint32_t row0) {
#pragma unroll
    for (int j = 0; j < LG; ++j) {
      const int c = j * 64 + lane, pr = c / CPG, ch = c % CPG;
      const uint32_t mm = row0 + pr;
      const unsigned off = (pr < R && mm < m_end) ? (mm * CO + ch * 4) * 4u : 0x80000000u;
      gq[j] = __builtin_amdgcn_raw_buffer_load_b128(srs, off, 0, 0);
    }
  };
  uint32_t row0 = m_begin + wave * R;
  if (row0 < m_end) { decode(row0); load_small(row0); load_big(0); }
  for (; row0 < m_end; row0 += 4 * R) {
    WAVE_LDS_SYNCW();                               // the previous tile's fragment reads are done
#pragma unroll
    for (int j = 0; j < LG; ++j) {
      const int c = j * 64 + lane, pr = c / CPG, ch = c % CPG;
      u32x2 p1, p2, p3;
      split4(gq[j], p1, p2, p3);
      if (pr < R) {
        const int o = wtile_off<CO>(pr, ch >> 1) + (ch & 1) * 8;
        *reinterpret_cast<u32x2*>(ts + o) = p1;
        *reinterpret_cast<u32x2*>(ts + PS + o) = p2;
        *reinterpret_cast<u32x2*>(ts + 2 * PS + o) = p3;
#pragma unroll
        for (int e = 0; e < 4; ++e) bsum[e] += __uint_as_float(gq[j][e]);        // (rows past the end were loaded as zeros)
      }
    }
    const bool more = row0 + 4 * R < m_end;
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      if (t > 0) WAVE_LDS_SYNCW();                  // tap t-1's fragments are in registers
#pragma unroll
      for (int j = 0; j < LX; ++j) {
        const int c = j * 64 + lane, pr = c / CPX, ch = c % CPX;
        u32x2 p1, p2, p3;
        split4(xq[j], p1, p2, p3);
        const int o = wtile_off<CI>(pr, ch >> 1) + (ch & 1) * 8;
        *reinterpret_cast<u32x2*>(tb + o) = p1;
        *reinterpret_cast<u32x2*>(tb + PB + o) = p2;
        *reinterpret_cast<u32x2*>(tb + 2 * PB + o) = p3;
      }
      WAVE_LDS_SYNCW();
      if (t + 1 < TG) load_big(t + 1);
      else if (more) { decode(row0 + 4 * R); load_small(row0 + 4 * R); load_big(0); }
      // (the small tile's fragments are re-read per tap: held in registers across the five taps they spilled the accumulators)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        bf16x8 fs[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) fs[p] = wfrag_cols<CO>(ts + p * PS, lane, nt);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          bf16x8 fb[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) fb[p] = wfrag_cols<CI>(tb + p * PB, lane, kt);
          MVAE_SPLIT6(acc[t][kt][nt], fb, fs);
        }
      }
    }
  }
  // ---- per tap: reduce the 4 waves through LDS, then one coalesced float-atomic set per block
  float* red = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int t = 0; t < TG; ++t) {
    for (int wv = 0; wv < 4; ++wv) {
      __syncthreads();
      if (wave == wv) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
              const int ci = kt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
              const int idx = ci * CO + nt * 32 + r;
              red[idx] = (wv == 0 ? 0.f : red[idx]) + acc[t][kt][nt][reg];
            }
      }
    }
    __syncthreads();
    float* dWt = dW + (int64_t)(kh * g.KW + t) * CI * CO;
    for (int idx = threadIdx.x; idx < CI * CO; idx += 256) atomicAdd(&dWt[idx], red[idx]);
  }
  if (db != nullptr && kh == 0) {
    __syncthreads();
    // lanes with equal lane % CPG hold partial sums of the same four columns
#pragma unroll
    for (int o = 32; o >= CPG; o >>= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) bsum[e] += __shfl_xor(bsum[e], o, 64);
    if (lane < CPG) *reinterpret_cast<f32x4*>(red + wave * CO + lane * 4) = bsum;
    __syncthreads();
    if (threadIdx.x < CO)
      atomicAdd(&db[threadIdx.x], red[threadIdx.x] + red[CO + threadIdx.x] + red[2 * CO + threadIdx.x] + red[3 * CO + threadIdx.x]);
  }
}

// false = geometry not covered (the caller then runs k_wgrad_taprow)
template <int CI, int CO>
static bool run_wgrad_taprow_split(const float* big, const float* small, float* dW, float* db, const ConvGeom& g, hipStream_t s) {
  const int64_t M = (int64_t)g.B * g.OH * g.OW;
  if (g.KW != 5 || M * CO * 4 >= (1ll << 31) || (int64_t)g.B * g.IH * g.IW * CI * 4 >= (1ll << 31)) return false;
  // as run_wgrad_taprow: at most 64 / KH row chunks per XCD (two resident 4-wave blocks per CU)
  int64_t chunks = 8 * (64 / g.KH);
  if (chunks < 8) chunks = 8;
  int64_t rpb = (M + chunks - 1) / chunks;
  rpb = (rpb + 63) / 64 * 64;
  if (rpb < 64) rpb = 64;
  chunks = (M + rpb - 1) / rpb;
  const unsigned groups = (unsigned)((chunks + 7) / 8);
  hipLaunchKernelGGL((k_wgrad_taprow_s<CI, CO, 5>), dim3(groups * 8u * g.KH), dim3(256), 0, s, big, small, dW, db, g, M, rpb);
  return true;
}

bool launch_conv_wgrad_split(const float* big, const float* small, float* dW, float* db, const ConvGeom& g, hipStream_t s) {
  static const bool on = [] { const char* e = getenv("MVAE_SPLIT_WGRAD"); return e ? atoi(e) != 0 : true; }();
  if (!on || split_conv_status() != 1) return false;
  if (g.CI == 64 && g.CO == 32) return run_wgrad_taprow_split<64, 32>(big, small, dW, db, g, s);
  if (g.CI == 32 && g.CO == 64) return run_wgrad_taprow_split<32, 64>(big, small, dW, db, g, s);
  return false;
}

}  // namespace mvae
