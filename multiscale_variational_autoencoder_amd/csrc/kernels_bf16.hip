// kernels_bf16.hip -- the MFMA kernels of the MVAE_ACT_BF16 path (BASELINE configs 4-5: 256x256x3, bf16): activations,
// saved tensors and activation gradients are bfloat16 in HBM, weights / gradients / accumulators float32, products on
// v_mfma_f32_32x32x16_bf16 (16x the f32-MFMA rate: with half the bytes and f32 MFMAs the 1x1 convolutions of the
// MobileNetV3 block would be MFMA-bound at today's speed; with bf16 MFMAs every kernel here is bound by memory).
//
// Orientation.  All data GEMMs are computed TRANSPOSED: D[row = output channel][col = pixel] = W^T . X^T, i.e. the
// weights are the A operand (register / LDS resident) and the activation tile the B operand.  Lane l = (r = l & 31,
// h = l >> 5) then holds, for pixel r, output channels (reg & 3) + 8 (reg >> 2) + 4 h: four CONSECUTIVE channels per
// register quad -> 8 bytes of the NHWC row, and one v_permlane32_swap per dword turns two quads into 16 contiguous
// bytes per lane (cdna_hip_programming.md T21): the result leaves as 16-byte stores with no LDS transposition.
// The B operand of pixel r, k-step kk is channels kk*16 + 8h .. +7 of that pixel = one 16-byte chunk of its row.
//
// Weight gradients contract over pixels: both operands are needed pixel-major per channel, which is what
// ds_read_b64_tr_b16 delivers from a row-major LDS tile (T10): lane (c, h) receives rows 8h .. 8h+7 of channel c.
#include "kernels.h"
#include "act16.h"
#include "prof.h"

namespace mvae {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define WAVE_LDS_SYNC16()                                    \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)

__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
// 8 floats -> one fragment
__device__ __forceinline__ bf16x8 frag_of(const float (&v)[8]) {
  u32x4 u = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
  return as_frag(u);
}
__device__ __forceinline__ float act16(float v, int act) {
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == ACT_ELU) return v > 0.f ? v : expm1f(v);
  return v;
}

// ---- row-major LDS tile of 32 rows x C bf16 (wave private), 16-byte chunks XOR-swizzled -------------------------------
// C = 64: 128-byte rows, chunk' = chunk ^ (row & 7).  C = 32: two rows share a 128-byte line, chunk8 = (row & 1) * 4 +
// chunk, chunk8' = chunk8 ^ ((row >> 1) & 7).  Returns the BYTE offset of (row, 16-byte chunk) inside the tile.
template <int C>
__device__ __forceinline__ int tile_off(int row, int chunk) {
  if constexpr (C == 64) return row * 128 + ((chunk ^ (row & 7)) << 4);
  else return (row >> 1) * 128 + ((((row & 1) * 4 + chunk) ^ ((row >> 1) & 7)) << 4);
}
// the wave's 32 x C tile is ONE contiguous run of 32*C*2 bytes in HBM: lane l takes 16-byte chunks l, l + 64, ...
template <int C>
struct TileRegs { u32x4 v[C / 16]; };
template <int C>
__device__ __forceinline__ void tile_load(const bf16_t* __restrict__ base, int64_t row0, int lane, TileRegs<C>& t) {
  const u32x4* p = reinterpret_cast<const u32x4*>(base + row0 * C) + lane;
#pragma unroll
  for (int j = 0; j < C / 16; ++j) t.v[j] = p[j * 64];
}
template <int C>
__device__ __forceinline__ void tile_store_lds(char* tile, int lane, const TileRegs<C>& t) {
#pragma unroll
  for (int j = 0; j < C / 16; ++j) {
    const int c = j * 64 + lane, row = c / (C / 8), ch = c % (C / 8);
    *reinterpret_cast<u32x4*>(tile + tile_off<C>(row, ch)) = t.v[j];
  }
}
// B operand of the data GEMM / A operand of a row-major product: pixel (row) r, channels kk*16 + 8h .. +7
template <int C>
__device__ __forceinline__ bf16x8 frag_rows(const char* tile, int r, int h, int kk) {
  return as_frag(*reinterpret_cast<const u32x4*>(tile + tile_off<C>(r, 2 * kk + h)));
}
// transposed fragment: channel ct*32 + r, rows 16 s + 8 h .. + 7   (two ds_read_b64_tr_b16)
template <int C>
__device__ __forceinline__ bf16x8 frag_cols(const char* tile, int lane, int ct, int s) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;       // 16-lane group g, its lane 4q + p
  const int c0 = ct * 32 + 16 * (g & 1) + 4 * p;                    // the 4 channels this lane ADDRESSES
  const int rb = 16 * s + 8 * (g >> 1);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int o0 = tile_off<C>(rb + q, c0 >> 3) + (c0 & 7) * 2;
  const int o1 = tile_off<C>(rb + 4 + q, c0 >> 3) + (c0 & 7) * 2;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + o0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + o1));
  s16x8 f = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, f);
}
// acc + sum of the 8 elements.  (v_dot2c_f32_bf16 against (1, 1) would do a pair per instruction, but it returned wrong
// sums in tools/bf16_unit.hip on gfx950 / ROCm 7.2; two shifts / masks and two adds per dword are cheap next to the
// tile's memory time.)
__device__ __forceinline__ float frag_sum(bf16x8 f, float acc) {
  const u32x4 u = __builtin_bit_cast(u32x4, f);
#pragma unroll
  for (int e = 0; e < 4; ++e) acc += bf16_lo(u[e]) + bf16_hi(u[e]);
  return acc;
}

// weights of a 1x1 convolution as A-operand fragments, resident in registers for the whole kernel:
// wf[nt][kk] = Wm[k = kk*16 + 8h + j][n = nt*32 + r], Wm[k][n] = WT ? W[n*K + k] : W[k*N + n]
template <int K, int N, bool WT>
__device__ __forceinline__ void load_wfrags(const float* __restrict__ W, int r, int h, bf16x8 (&wf)[N / 32][K / 16]) {
#pragma unroll
  for (int nt = 0; nt < N / 32; ++nt)
#pragma unroll
    for (int kk = 0; kk < K / 16; ++kk) {
      float v[8];
      const int n = nt * 32 + r, k0 = kk * 16 + 8 * h;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = WT ? W[(int64_t)n * K + k0 + j] : W[(int64_t)(k0 + j) * N + n];
      wf[nt][kk] = frag_of(v);
    }
}

// epilogue of a transposed-orientation tile: acc[nt] holds channels nt*32 + 8q + 4h + e (e = reg & 3, q = reg >> 2) of
// pixel row0 + r.  v = act(acc + bias) + residual, packed to bf16, two quads -> one 16-byte store.  The bias sits in
// registers (a per-tile bias load inside the tile loop came with its own s_waitcnt vmcnt(0), i.e. every tile waited for
// the NEXT tile's prefetch: the first version ran at 2.8 TB/s, latency-bound).  The residual tile
// was staged in LDS with the same coalesced 16-byte loads as the input (one tile ahead); here it is read back 8 bytes
// per lane in the accumulator's layout.
// float32 result of a transposed-orientation tile: the register quad of lane (r, h) is 16 contiguous bytes of pixel row0+r.
// Used where the consumer is a BatchNorm: the decoder's last block output keeps float32 storage, because normalisation
// subtracts the batch mean and turns bf16's relative rounding error of the VALUE into (mean / std) times as much of the
// normalised signal (measured: reconstruction RMS error 0.9 % of the range with a bf16 BatchNorm input).
// STATS: the lane also accumulates sum (v - pivot) and sum (v - pivot)^2 of what it stores (its 4 channels of quad q), for the
// consumer BatchNorm's batch statistics (one pass about a pivot: k_colstat4<2>'s arithmetic without its read pass)
template <int N, bool RES, bool BIAS, bool STATS = false>
__device__ __forceinline__ void store_tile_t32(const f32x16 (&acc)[N / 32], const f32x4 (&bz)[N / 32][4],
                                               const char* __restrict__ res_tile, float* __restrict__ Y, int64_t row0, int r,
                                               int h, const f32x4 (*piv)[4] = nullptr, f32x4 (*a1)[4] = nullptr,
                                               f32x4 (*a2)[4] = nullptr) {
  const int64_t rowoff = (row0 + r) * N;
#pragma unroll
  for (int nt = 0; nt < N / 32; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c0 = nt * 32 + 8 * q + 4 * h;
      f32x4 v = {acc[nt][4 * q], acc[nt][4 * q + 1], acc[nt][4 * q + 2], acc[nt][4 * q + 3]};
      if constexpr (BIAS) v += bz[nt][q];
      if constexpr (RES) v += unpack4(*reinterpret_cast<const uint2*>(res_tile + tile_off<N>(r, c0 >> 3) + (c0 & 7) * 2));
      *reinterpret_cast<f32x4*>(Y + rowoff + c0) = v;
      if constexpr (STATS) {
        const f32x4 d = v - piv[nt][q];
        a1[nt][q] += d;
        a2[nt][q] += d * d;
      }
    }
}

// two floats -> bf16 pair rounded to ONE bit less (nearest even at 2^17), the freed mantissa LSB of each carrying a mask
// bit: the depthwise backward takes the ReLU mask (t1 > 0) from there instead of reading t1 (a quarter of its bytes)
__device__ __forceinline__ unsigned pack_bf16_mask(float lo, float hi, bool mlo, bool mhi) {
  const unsigned a = __float_as_uint(lo), b = __float_as_uint(hi);
  const unsigned ra = ((a + 0xFFFFu + ((a >> 17) & 1u)) & 0xFFFE0000u) | (mlo ? 0x10000u : 0u);
  const unsigned rb = ((b + 0xFFFFu + ((b >> 17) & 1u)) & 0xFFFE0000u) | (mhi ? 0x10000u : 0u);
  return (ra >> 16) | (rb & 0xFFFF0000u);
}
// MASK: res_tile is not added but supplies the mask source (aux = t1): the stored value carries (aux > 0) in its LSB
template <int N, bool RES, int ACT, bool BIAS, bool MASK = false>
__device__ __forceinline__ void store_tile_t(const f32x16 (&acc)[N / 32], const f32x4 (&bz)[N / 32][4],
                                             const char* __restrict__ res_tile, bf16_t* __restrict__ Y, int64_t row0, int r,
                                             int h) {
  const int64_t rowoff = (row0 + r) * N;
#pragma unroll
  for (int nt = 0; nt < N / 32; ++nt) {
    uint2 pk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c0 = nt * 32 + 8 * q + 4 * h;
      f32x4 v = {acc[nt][4 * q], acc[nt][4 * q + 1], acc[nt][4 * q + 2], acc[nt][4 * q + 3]};
      if constexpr (BIAS) v += bz[nt][q];
      if constexpr (ACT != ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act16(v[e], ACT);
      }
      if constexpr (RES) v += unpack4(*reinterpret_cast<const uint2*>(res_tile + tile_off<N>(r, c0 >> 3) + (c0 & 7) * 2));
      if constexpr (MASK) {
        const uint2 m = *reinterpret_cast<const uint2*>(res_tile + tile_off<N>(r, c0 >> 3) + (c0 & 7) * 2);
        // aux = t1 >= 0 (a ReLU output): > 0 <=> any bit below the sign set
        pk[q] = uint2{pack_bf16_mask(v[0], v[1], (m.x & 0x7FFFu) != 0u, (m.x & 0x7FFF0000u) != 0u),
                      pack_bf16_mask(v[2], v[3], (m.y & 0x7FFFu) != 0u, (m.y & 0x7FFF0000u) != 0u)};
      } else {
        pk[q] = pack4(v);
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      uint2 a = pk[2 * p], b = pk[2 * p + 1];
      auto s0 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
      auto s1 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
      const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
      *reinterpret_cast<u32x4*>(Y + rowoff + nt * 32 + 16 * p + 8 * h) = o;
    }
  }
}

// =================================================================================================
// Y[M,N] = act( (X * gate[image]) . Wm + bias ) + residual     1x1 convolution (and its transpose), bf16 storage
// (layer_blocks.py:594-602 conv0, :625-641 conv2, :946-951 the 1x1 Conv2D / Conv2DTranspose of basic_block)
// block = 4 waves, wave = 32 rows (wave-private LDS tile, no block barrier); M % 32 == 0; gate: rows_per_image % 32 == 0
// =================================================================================================
template <int K, int N, bool WT, bool GATE, bool RES, int ACT, bool OUT32 = false>
__global__ void __launch_bounds__(256) k16_pw(const bf16_t* __restrict__ X, const float* __restrict__ W,
                                              const float* __restrict__ bias, const float* __restrict__ gate,
                                              const bf16_t* __restrict__ res, void* __restrict__ Yv, int64_t ntiles,
                                              int64_t rows_per_image, const float* __restrict__ pivot = nullptr,
                                              float* __restrict__ st1 = nullptr, float* __restrict__ st2 = nullptr, int nslots = 1,
                                              int64_t slot_stride = 0) {
  constexpr int TB = 32 * K * 2, RB = RES ? 32 * N * 2 : 0;
  __shared__ __attribute__((aligned(16))) char lds[4 * (TB + RB)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  char* tile = lds + wave * (TB + RB);
  char* rtile = tile + TB;
  bf16x8 wf[N / 32][K / 16];
  load_wfrags<K, N, WT>(W, r, h, wf);
  f32x4 bz[N / 32][4];
#pragma unroll
  for (int nt = 0; nt < N / 32; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bz[nt][q] = bias ? *reinterpret_cast<const f32x4*>(bias + nt * 32 + 8 * q + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
  // OUT32 with st1 != nullptr: per-channel sums of (y - pivot) and (y - pivot)^2 over everything this wave stores (the decoder
  // BatchNorm's batch statistics without a pass over y): per-lane partials, folded over the 32 pixel lanes at the end
  f32x4 piv[OUT32 ? N / 32 : 1][4], a1[OUT32 ? N / 32 : 1][4], a2[OUT32 ? N / 32 : 1][4];
  if constexpr (OUT32) {
#pragma unroll
    for (int nt = 0; nt < N / 32; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        piv[nt][q] = pivot ? *reinterpret_cast<const f32x4*>(pivot + nt * 32 + 8 * q + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
        a1[nt][q] = a2[nt][q] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t t = (int64_t)blockIdx.x * 4 + wave;
  TileRegs<K> cur, nxt;
  TileRegs<RES ? N : 16> rcur, rnxt;
  // squeeze-excite gate of a tile's image: the lane's chunks all have channel chunk lane % (K/8), i.e. 8 gate values per
  // tile.  They travel WITH the tile's prefetch (loaded one tile ahead): a gate load issued inside the tile would sit
  // behind the next tile's prefetch in the in-order vmcnt queue, and waiting for it would drain that prefetch too.
  f32x4 gc0 = {1.f, 1.f, 1.f, 1.f}, gc1 = gc0, gn0 = gc0, gn1 = gc0;
  auto gate_load = [&](int64_t tile, f32x4& a, f32x4& b) {
    const f32x4* gp = reinterpret_cast<const f32x4*>(gate + (tile * 32 / rows_per_image) * K + (lane % (K / 8)) * 8);
    a = gp[0]; b = gp[1];
  };
  if (t < ntiles) {
    tile_load<K>(X, t * 32, lane, cur);
    if constexpr (RES) tile_load<N>(res, t * 32, lane, rcur);
    if constexpr (GATE) gate_load(t, gc0, gc1);
  }
  for (; t < ntiles; t += stride) {
    const int64_t row0 = t * 32;
    const int64_t tn = t + stride < ntiles ? t + stride : t;
    tile_load<K>(X, tn * 32, lane, nxt);                       // next tile in flight under this tile's work
    if constexpr (RES) tile_load<N>(res, tn * 32, lane, rnxt);
    if constexpr (GATE) {
      gate_load(tn, gn0, gn1);
#pragma unroll
      for (int j = 0; j < K / 16; ++j) {
        const u32x4 u = cur.v[j];
        float v[8] = {bf16_lo(u[0]) * gc0[0], bf16_hi(u[0]) * gc0[1], bf16_lo(u[1]) * gc0[2], bf16_hi(u[1]) * gc0[3],
                      bf16_lo(u[2]) * gc1[0], bf16_hi(u[2]) * gc1[1], bf16_lo(u[3]) * gc1[2], bf16_hi(u[3]) * gc1[3]};
        cur.v[j] = __builtin_bit_cast(u32x4, frag_of(v));
      }
    }
    WAVE_LDS_SYNC16();                                         // the previous tile's LDS reads are done
    tile_store_lds<K>(tile, lane, cur);
    if constexpr (RES) tile_store_lds<N>(rtile, lane, rcur);
    WAVE_LDS_SYNC16();
    f32x16 acc[N / 32];
#pragma unroll
    for (int nt = 0; nt < N / 32; ++nt) acc[nt] = zero16();
#pragma unroll
    for (int kk = 0; kk < K / 16; ++kk) {
      const bf16x8 xb = frag_rows<K>(tile, r, h, kk);
#pragma unroll
      for (int nt = 0; nt < N / 32; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt][kk], xb, acc[nt], 0, 0, 0);
    }
    if constexpr (OUT32) store_tile_t32<N, RES, true, true>(acc, bz, rtile, static_cast<float*>(Yv), row0, r, h, piv, a1, a2);
    else store_tile_t<N, RES, ACT, true>(acc, bz, rtile, static_cast<bf16_t*>(Yv), row0, r, h);
    cur = nxt;
    if constexpr (RES) rcur = rnxt;
    if constexpr (GATE) { gc0 = gn0; gc1 = gn1; }
  }
  if constexpr (OUT32) {
    if (st1 != nullptr) {                                      // (kernel-uniform)
      float* o1 = st1 + (int64_t)((blockIdx.x * 4 + wave) % (unsigned)nslots) * slot_stride;
      float* o2 = st2 + (int64_t)((blockIdx.x * 4 + wave) % (unsigned)nslots) * slot_stride;
#pragma unroll
      for (int nt = 0; nt < N / 32; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float u = a1[nt][q][e], v = a2[nt][q][e];
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) { u += __shfl_xor(u, off, 64); v += __shfl_xor(v, off, 64); }
            if (r == 0) { atomicAdd(o1 + nt * 32 + 8 * q + 4 * h + e, u); atomicAdd(o2 + nt * 32 + 8 * q + 4 * h + e, v); }
          }
    }
  }
}

// =================================================================================================
// conv2 of one MobileNetV3 block and conv0 of the NEXT block in one launch (64 -> 64 both, layer_blocks.py:625-641 then
// :594-602 of the following block):   Y = (X * gate) . W + bias + residual  (stored: it is the next block's residual and a
// saved tensor)  and  Y2 = relu(Y . W2 + bias2)  from the SAME registers -- the 16-byte chunks lane (r, h) has just
// formed for the store of pixel r (channels 32 nt + 16 p + 8 h ..) are exactly the B-operand fragments kk = 2 nt + p of the
// second product, so Y is not read back (one tensor pass less per block).  The second product sees the bf16-rounded Y,
// as the separate launch would.  W2 fragments and bias2 sit in LDS (registers: 208 + 16).
// =================================================================================================
// MID: between the two blocks sits a 1x1 convolution 64 -> 32 (basic_block's Conv2D / Conv2DTranspose of the last
// level, layer_blocks.py:946-951; WTM = the transposed form) and the next block is 32 wide: three products in a row,
//   Y = conv2(X) (stored), Ym = Y . Wm + bm (stored: the next block's residual / conv output), Y2 = relu(Ym . W2 + b2),
// each from the chunks the previous one has just stored (Ym's two chunks per lane are the K = 32 fragments of the third).
template <bool MID, bool WTM>
__global__ void __launch_bounds__(256) k16_pw_chain(const bf16_t* __restrict__ X, const float* __restrict__ W,
                                                    const float* __restrict__ bias, const float* __restrict__ gate,
                                                    const bf16_t* __restrict__ res, bf16_t* __restrict__ Y,
                                                    const float* __restrict__ Wm, const float* __restrict__ biasm,
                                                    bf16_t* __restrict__ Ym,
                                                    const float* __restrict__ W2, const float* __restrict__ bias2,
                                                    bf16_t* __restrict__ Y2, int64_t ntiles, int64_t rows_per_image) {
  constexpr int K = 64, N = 64, TB = 32 * K * 2, RB = 32 * N * 2;
  constexpr int N2 = MID ? 32 : 64;                                     // width of the next block
  __shared__ __attribute__((aligned(16))) char lds[4 * (TB + RB) + 8 * 64 * 16 + 64 * 4 + (MID ? 4 * 64 * 16 + 32 * 4 : 0)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  char* tile = lds + wave * (TB + RB);
  char* rtile = tile + TB;
  u32x4* wfl2 = reinterpret_cast<u32x4*>(lds + 4 * (TB + RB));
  float* b2s = reinterpret_cast<float*>(lds + 4 * (TB + RB) + 8 * 64 * 16);
  bf16x8 wf[2][4];
  load_wfrags<K, N, false>(W, r, h, wf);
  u32x4* wflm = reinterpret_cast<u32x4*>(lds + 4 * (TB + RB) + 8 * 64 * 16 + 64 * 4);
  float* bms = reinterpret_cast<float*>(lds + 4 * (TB + RB) + 8 * 64 * 16 + 64 * 4 + 4 * 64 * 16);
  if (wave == 0) {
    if constexpr (MID) {
      bf16x8 w2[1][2];
      load_wfrags<32, 32, false>(W2, r, h, w2);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) wfl2[kk * 64 + lane] = __builtin_bit_cast(u32x4, w2[0][kk]);
      if (lane < 32) b2s[lane] = bias2 ? bias2[lane] : 0.f;
    } else {
      bf16x8 w2[2][4];
      load_wfrags<64, 64, false>(W2, r, h, w2);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) wfl2[(nt * 4 + kk) * 64 + lane] = __builtin_bit_cast(u32x4, w2[nt][kk]);
      b2s[lane] = bias2 ? bias2[lane] : 0.f;
    }
  }
  if constexpr (MID) {
    if (wave == 1) {
      bf16x8 wm[1][4];
      load_wfrags<64, 32, WTM>(Wm, r, h, wm);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wflm[kk * 64 + lane] = __builtin_bit_cast(u32x4, wm[0][kk]);
      if (lane < 32) bms[lane] = biasm ? biasm[lane] : 0.f;
    }
  }
  f32x4 bz[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bz[nt][q] = bias ? *reinterpret_cast<const f32x4*>(bias + nt * 32 + 8 * q + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t t = (int64_t)blockIdx.x * 4 + wave;
  TileRegs<K> cur, nxt;
  TileRegs<N> rcur, rnxt;
  f32x4 gc0 = {1.f, 1.f, 1.f, 1.f}, gc1 = gc0, gn0 = gc0, gn1 = gc0;
  auto gate_load = [&](int64_t tl, f32x4& a, f32x4& b) {
    const f32x4* gp = reinterpret_cast<const f32x4*>(gate + (tl * 32 / rows_per_image) * K + (lane % (K / 8)) * 8);
    a = gp[0]; b = gp[1];
  };
  if (t < ntiles) {
    tile_load<K>(X, t * 32, lane, cur);
    tile_load<N>(res, t * 32, lane, rcur);
    gate_load(t, gc0, gc1);
  }
  for (; t < ntiles; t += stride) {
    const int64_t row0 = t * 32;
    const int64_t tn = t + stride < ntiles ? t + stride : t;
    tile_load<K>(X, tn * 32, lane, nxt);
    tile_load<N>(res, tn * 32, lane, rnxt);
    gate_load(tn, gn0, gn1);
#pragma unroll
    for (int j = 0; j < K / 16; ++j) {
      const u32x4 u = cur.v[j];
      float v[8] = {bf16_lo(u[0]) * gc0[0], bf16_hi(u[0]) * gc0[1], bf16_lo(u[1]) * gc0[2], bf16_hi(u[1]) * gc0[3],
                    bf16_lo(u[2]) * gc1[0], bf16_hi(u[2]) * gc1[1], bf16_lo(u[3]) * gc1[2], bf16_hi(u[3]) * gc1[3]};
      cur.v[j] = __builtin_bit_cast(u32x4, frag_of(v));
    }
    WAVE_LDS_SYNC16();
    tile_store_lds<K>(tile, lane, cur);
    tile_store_lds<N>(rtile, lane, rcur);
    WAVE_LDS_SYNC16();
    f32x16 acc[2] = {zero16(), zero16()};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bf16x8 xb = frag_rows<K>(tile, r, h, kk);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt][kk], xb, acc[nt], 0, 0, 0);
    }
    // ---- Y = acc + bias + residual -> bf16; o[2 nt + p] = channels 32 nt + 16 p + 8 h .. + 7 of pixel row0 + r
    const int64_t rowoff = (row0 + r) * N;
    u32x4 o[4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      uint2 pk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = nt * 32 + 8 * q + 4 * h;
        f32x4 v = {acc[nt][4 * q], acc[nt][4 * q + 1], acc[nt][4 * q + 2], acc[nt][4 * q + 3]};
        v += bz[nt][q];
        v += unpack4(*reinterpret_cast<const uint2*>(rtile + tile_off<N>(r, c0 >> 3) + (c0 & 7) * 2));
        pk[q] = pack4(v);
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        uint2 a = pk[2 * p], b = pk[2 * p + 1];
        auto s0 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
        o[2 * nt + p] = u32x4{s0[0], s1[0], s0[1], s1[1]};
        *reinterpret_cast<u32x4*>(Y + rowoff + nt * 32 + 16 * p + 8 * h) = o[2 * nt + p];
      }
    }
    if constexpr (MID) {
      // ---- Ym = Y . Wm + bm (32 channels), then Y2 = relu(Ym . W2 + b2) from Ym's two chunks
      acc[0] = zero16();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(wflm[kk * 64 + lane]), as_frag(o[kk]), acc[0], 0, 0, 0);
      const int64_t rowm = (row0 + r) * 32;
      u32x4 om[2];
      {
        uint2 pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v = {acc[0][4 * q], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]};
          v += *reinterpret_cast<const f32x4*>(bms + 8 * q + 4 * h);
          pk[q] = pack4(v);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          uint2 a = pk[2 * p], b = pk[2 * p + 1];
          auto s0 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
          auto s1 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
          om[p] = u32x4{s0[0], s1[0], s0[1], s1[1]};
          *reinterpret_cast<u32x4*>(Ym + rowm + 16 * p + 8 * h) = om[p];
        }
      }
      acc[0] = zero16();
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(wfl2[kk * 64 + lane]), as_frag(om[kk]), acc[0], 0, 0, 0);
      {
        uint2 pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v = {acc[0][4 * q], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]};
          v += *reinterpret_cast<const f32x4*>(b2s + 8 * q + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
          pk[q] = pack4(v);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          uint2 a = pk[2 * p], b = pk[2 * p + 1];
          auto s0 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
          auto s1 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
          *reinterpret_cast<u32x4*>(Y2 + rowm + 16 * p + 8 * h) = u32x4{s0[0], s1[0], s0[1], s1[1]};
        }
      }
    } else {
    // ---- Y2 = relu(Y . W2 + bias2): the chunks above are the B fragments
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[nt] = zero16();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(wfl2[(nt * 4 + kk) * 64 + lane]), as_frag(o[kk]), acc[nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      uint2 pk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = nt * 32 + 8 * q + 4 * h;
        f32x4 v = {acc[nt][4 * q], acc[nt][4 * q + 1], acc[nt][4 * q + 2], acc[nt][4 * q + 3]};
        v += *reinterpret_cast<const f32x4*>(b2s + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        pk[q] = pack4(v);
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        uint2 a = pk[2 * p], b = pk[2 * p + 1];
        auto s0 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
        *reinterpret_cast<u32x4*>(Y2 + rowoff + nt * 32 + 16 * p + 8 * h) = u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
    }
    }   // !MID
    cur = nxt; rcur = rnxt; gc0 = gn0; gc1 = gn1;
  }
}

// =================================================================================================
// Backward pair of a 1x1 C -> C convolution of the MobileNetV3 block in ONE pass over (X, aux) (layer_blocks.py:594-641
// inverted; the f32 counterpart is k_gemm_dual):
//     Y[M,C]  = X . W^T (+ residual)                       W = Conv2D kernel [ci][co]
//     dW[ci][co] += gate[b,ci] * sum_m aux[m,ci] X[m,co] ;  db[co] += sum_m X[m,co]
//     dot_out[b,ci] += sum_co W[ci][co] * P_b[ci][co],  P_b = sum_{m in image b} aux[m,ci] X[m,co]
//         ( = sum_m Y[m,ci] aux[m,ci], the squeeze-excite gate gradient, from the weight-gradient product itself )
// MODE 1 = conv2 pair (gate + dot, no residual): aux = t1, X = dout.   MODE 2 = conv0 pair (residual): aux = block input,
// X = dt0.  A wave takes a CONTIGUOUS run of 32-row tiles inside one image, so the ungated P accumulates in registers
// over the run and the gate scaling / gate gradient cost one pass over the accumulators per wave, not per tile.
// =================================================================================================
template <int C, int MODE>
__global__ void __launch_bounds__(256) k16_dual(const bf16_t* __restrict__ X, const float* __restrict__ W,
                                                const bf16_t* __restrict__ aux, const float* __restrict__ gate,
                                                const bf16_t* __restrict__ res, bf16_t* __restrict__ Y,
                                                float* __restrict__ dW, float* __restrict__ db,
                                                float* __restrict__ dot_out, int64_t ntiles, int64_t tiles_per_wave,
                                                int64_t tiles_per_image, int nslots, int64_t slot_stride, int embed_mask) {
  constexpr int CT = C / 32, TB = 32 * C * 2, NTL = MODE == 2 ? 3 : 2;    // tiles per wave in LDS: X, aux (, residual)
  constexpr int TILES = (4 * NTL * TB > C * C * 4) ? 4 * NTL * TB : C * C * 4;
  // the data GEMM's weight fragments live in LDS, one 16-byte slot per (fragment, lane): 32 registers less per lane
  // (C = 64), which is what keeps the residual form under 256 VGPRs = two blocks per CU
  __shared__ __attribute__((aligned(16))) char lds[TILES + CT * (C / 16) * 64 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  char* tx = lds + wave * (NTL * TB);
  char* ta = tx + TB;
  char* tr = ta + TB;
  u32x4* wfl = reinterpret_cast<u32x4*>(lds + TILES);
  if (wave == 0) {
    bf16x8 wf[CT][C / 16];
    load_wfrags<C, C, true>(W, r, h, wf);                      // Wm[k = co][n = ci] = W[ci*C + co]
#pragma unroll
    for (int nt = 0; nt < CT; ++nt)
#pragma unroll
      for (int kk = 0; kk < C / 16; ++kk) wfl[(nt * (C / 16) + kk) * 64 + lane] = __builtin_bit_cast(u32x4, wf[nt][kk]);
  }
  __syncthreads();
  f32x16 accw[CT][CT];                                         // [nt = co tile][kt = ci tile]: D'[row = co][col = ci]
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int b = 0; b < CT; ++b) accw[a][b] = zero16();
  float bsum[CT];
#pragma unroll
  for (int a = 0; a < CT; ++a) bsum[a] = 0.f;
  const f32x4 bz0[CT][4] = {};                                 // (no bias in the backward pair)
  // a wave takes a contiguous run of tiles that lies inside ONE image (launcher: tiles_per_wave divides
  // tiles_per_image), so the ungated product P = aux^T X accumulates in accw over the whole run; the gate scaling and
  // the gate gradient are applied once, after the loop
  const int64_t w_id = (int64_t)blockIdx.x * 4 + wave;
  const int64_t t0 = w_id * tiles_per_wave;
  int64_t t1 = t0 + tiles_per_wave;
  if (t1 > ntiles) t1 = ntiles;
  TileRegs<C> cx, ca, nx, na;
  TileRegs<MODE == 2 ? C : 16> cr, nr;
  if (t0 < t1) {
    tile_load<C>(X, t0 * 32, lane, cx); tile_load<C>(aux, t0 * 32, lane, ca);
    if constexpr (MODE == 2) tile_load<C>(res, t0 * 32, lane, cr);
  }
  for (int64_t t = t0; t < t1; ++t) {
    const int64_t row0 = t * 32;
    const int64_t tn = t + 1 < t1 ? t + 1 : t;
    tile_load<C>(X, tn * 32, lane, nx);
    tile_load<C>(aux, tn * 32, lane, na);
    if constexpr (MODE == 2) tile_load<C>(res, tn * 32, lane, nr);
    WAVE_LDS_SYNC16();
    tile_store_lds<C>(tx, lane, cx);
    tile_store_lds<C>(ta, lane, ca);
    if constexpr (MODE == 2) tile_store_lds<C>(tr, lane, cr);
    WAVE_LDS_SYNC16();
    // ---- Y tile = X . W^T (+ residual)
    f32x16 acc[CT];
#pragma unroll
    for (int nt = 0; nt < CT; ++nt) acc[nt] = zero16();
#pragma unroll
    for (int kk = 0; kk < C / 16; ++kk) {
      const bf16x8 xb = frag_rows<C>(tx, r, h, kk);
#pragma unroll
      for (int nt = 0; nt < CT; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(wfl[(nt * (C / 16) + kk) * 64 + lane]), xb, acc[nt], 0, 0, 0);
    }
    if constexpr (MODE == 1) {
      if (embed_mask) store_tile_t<C, false, ACT_NONE, false, true>(acc, bz0, ta, Y, row0, r, h);   // dt2 + ReLU mask of t1
      else store_tile_t<C, false, ACT_NONE, false>(acc, bz0, ta, Y, row0, r, h);
    } else {
      store_tile_t<C, true, ACT_NONE, false>(acc, bz0, tr, Y, row0, r, h);
    }
    // ---- P[co][ci] += X^T aux over the tile's 32 rows
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fa[CT], fb[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) { fa[ct] = frag_cols<C>(tx, lane, ct, s); fb[ct] = frag_cols<C>(ta, lane, ct, s); }
#pragma unroll
      for (int nt = 0; nt < CT; ++nt) {
        bsum[nt] = frag_sum(fa[nt], bsum[nt]);
#pragma unroll
        for (int kt = 0; kt < CT; ++kt)
          accw[nt][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[nt], fb[kt], accw[nt][kt], 0, 0, 0);
      }
    }
    cx = nx; ca = na;
    if constexpr (MODE == 2) cr = nr;
  }
  if constexpr (MODE == 1) {
    if (t0 < t1) {                                             // wave-uniform
      const int64_t img = t0 / tiles_per_image;
#pragma unroll
      for (int kt = 0; kt < CT; ++kt) {
        const int ci = kt * 32 + r;
        const float gl = gate[img * C + ci];
        float ds = 0.f;
#pragma unroll
        for (int nt = 0; nt < CT; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(W + (int64_t)ci * C + nt * 32 + 8 * q + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              ds += w4[e] * accw[nt][kt][4 * q + e];
              accw[nt][kt][4 * q + e] *= gl;
            }
          }
        ds += __shfl_xor(ds, 32, 64);
        if (h == 0) atomicAdd(dot_out + img * C + ci, ds);
      }
    }
  }
  // ---- reduce the 4 waves' dW through LDS (tile memory is free now), one coalesced atomic set per block
  float* red = reinterpret_cast<float*>(lds);
  for (int wv = 0; wv < 4; ++wv) {
    __syncthreads();
    if (wave == wv) {
#pragma unroll
      for (int nt = 0; nt < CT; ++nt)
#pragma unroll
        for (int kt = 0; kt < CT; ++kt)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int co = nt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h, ci = kt * 32 + r;
            const int idx = ci * C + co;
            red[idx] = (wv == 0 ? 0.f : red[idx]) + accw[nt][kt][reg];
          }
    }
  }
  __syncthreads();
  const int64_t slot = (int64_t)(blockIdx.x % nslots) * slot_stride;
  for (int idx = threadIdx.x; idx < C * C; idx += 256) atomicAdd(&dW[slot + idx], red[idx]);
  if (db != nullptr) {
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < CT; ++nt) {
      float b = bsum[nt] + __shfl_xor(bsum[nt], 32, 64);
      if (h == 0) red[wave * C + nt * 32 + r] = b;
    }
    __syncthreads();
    if (threadIdx.x < C)
      atomicAdd(&db[slot + threadIdx.x], red[threadIdx.x] + red[C + threadIdx.x] + red[2 * C + threadIdx.x] + red[3 * C + threadIdx.x]);
  }
}

// =================================================================================================
// k x k strided SAME convolution / transposed convolution (5x5 stride 2 in the notebook blocks), bf16 storage.
// Same F-form / T-form coordinates as k_conv_taps (kernels_mfma.hip): F: out = small[B,OH,OW,NC=CO], in = big (KC = CI);
// T: out = big[B,IH,IW,NC=CI], in = small (KC = CO), one sub-pixel phase per blockIdx.y.
// A 512-thread block keeps the weight slices of ALL taps in LDS as bf16 [tap][n][k] (25 x 4 KB for 32 <-> 64): the tap
// loop has no block barrier.  A wave owns 32 output pixels x NC channels; per tap it gathers the 32 input pixels' rows
// with coalesced 16-byte loads (SAME padding = an out-of-range buffer offset, which returns zeros), one tap ahead of the
// MFMAs, and stages them in a wave-private LDS tile from which the B fragments (lane = pixel) are read.
// =================================================================================================
// CHAIN (T-form, NC = 64): the MobileNetV3 block that follows the transposed convolution starts with conv0 (64 -> 64,
// bias, ReLU): out2 = relu(out . W2 + bias2) is computed from the chunks just stored (see k16_pw_chain), so that block's
// conv0 does not read `out` back.
template <int KC, int NC, bool TFORM, int MAXTAPS, int NTHR, bool CHAIN = false>
__global__ void __launch_bounds__(NTHR) k16_taps(const bf16_t* __restrict__ in, const float* __restrict__ W,
                                                const float* __restrict__ bias, bf16_t* __restrict__ out, ConvGeom g,
                                                unsigned in_bytes, int tiles_per_wave, const float* __restrict__ W2,
                                                const float* __restrict__ bias2, bf16_t* __restrict__ out2) {
  constexpr int NT = NC / 32, KK = KC / 16;
  static_assert(!CHAIN || (TFORM && NC == 64), "chained conv0: T-form, 64 output channels");
  __shared__ __attribute__((aligned(16))) char sW2[CHAIN ? 8 * 64 * 16 + 64 * 4 : 16];
  u32x4* wfl2 = reinterpret_cast<u32x4*>(sW2);
  float* b2s = reinterpret_cast<float*>(sW2 + 8 * 64 * 16);
  if constexpr (CHAIN) {
    if (threadIdx.x < 64) {
      const int l = threadIdx.x;
      bf16x8 w2[2][4];
      load_wfrags<64, 64, false>(W2, l & 31, l >> 5, w2);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) wfl2[(nt * 4 + kk) * 64 + l] = __builtin_bit_cast(u32x4, w2[nt][kk]);
      b2s[l] = bias2 ? bias2[l] : 0.f;
    }
  }
  constexpr int CPP = KC / 8, LX = KC / 16;          // 16-byte chunks per pixel; gather slots per lane (32 px * CPP / 64)
  __shared__ __attribute__((aligned(16))) char sW[MAXTAPS * NC * KC * 2];
  __shared__ __attribute__((aligned(16))) char sA[(NTHR / 64) * 32 * KC * 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  char* myA = sA + wave * (32 * KC * 2);
  int py = 0, px = 0, CH = g.OH, CW = g.OW;
  if (TFORM) {
    py = blockIdx.y / g.SW; px = blockIdx.y % g.SW;
    CH = (g.IH - py + g.SH - 1) / g.SH; CW = (g.IW - px + g.SW - 1) / g.SW;
  }
  const unsigned Mc = (unsigned)(g.B * CH * CW);
  const int SHh = TFORM ? g.OH : g.IH, SWw = TFORM ? g.OW : g.IW;      // the tensor the taps read
  int kh0 = 0, kw0 = 0, khs = 1, kws = 1;
  if (TFORM) { kh0 = (py + g.PT) % g.SH; kw0 = (px + g.PL) % g.SW; khs = g.SH; kws = g.SW; }
  const int nkh = kh0 < g.KH ? (g.KH - kh0 + khs - 1) / khs : 0;
  const int nkw = kw0 < g.KW ? (g.KW - kw0 + kws - 1) / kws : 0;
  const int ntaps = nkh * nkw;
  // ---- this phase's weight slices -> LDS, bf16, [tap][n][k] with the 16-byte chunks of a row XOR-swizzled
  //      F: W[tap][k][n] (Keras HWIO);  T: W[tap][n][k] (the same array read with big = output channels)
  for (int idx = threadIdx.x; idx < ntaps * NC * KC; idx += NTHR) {
    const int tl = idx / (NC * KC), rem = idx % (NC * KC);
    const int th = tl / nkw, tw = tl % nkw;
    const int tap = (kh0 + th * khs) * g.KW + kw0 + tw * kws;
    int n, k;
    if (TFORM) { n = rem / KC; k = rem % KC; } else { k = rem / NC; n = rem % NC; }
    const float v = W[(int64_t)tap * NC * KC + rem];
    const int off = tl * NC * KC * 2 + n * KC * 2 + (((k >> 3) ^ ((KC == 32 ? n >> 1 : n) & (KC / 8 - 1))) << 4) + (k & 7) * 2;
    *reinterpret_cast<uint16_t*>(sW + off) = (uint16_t)(pack_bf16(v, 0.f) & 0xFFFFu);
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(in), 0, (int)in_bytes, 0x00020000);
  const bool pow2 = (CW & (CW - 1)) == 0 && (CH & (CH - 1)) == 0;
  const int lgw = 31 - __builtin_clz((unsigned)CW), lgh = 31 - __builtin_clz((unsigned)CH);
  auto split = [&](unsigned p, int& cx, int& cy, int& b) {
    if (pow2) { cx = (int)(p & (unsigned)(CW - 1)); cy = (int)((p >> lgw) & (unsigned)(CH - 1)); b = (int)(p >> (lgw + lgh)); }
    else { cx = (int)(p % (unsigned)CW); const unsigned q = p / (unsigned)CW; cy = (int)(q % (unsigned)CH); b = (int)(q / (unsigned)CH); }
  };
  float bv[NT][4][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[nt][q][e] = bias ? bias[nt * 32 + 8 * q + 4 * h + e] : 0.f;

  const unsigned wtile0 = ((unsigned)blockIdx.x * (unsigned)(NTHR / 64) + wave) * (unsigned)tiles_per_wave;
  for (int ti = 0; ti < tiles_per_wave; ++ti) {
    const unsigned p0 = (wtile0 + ti) * 32u;
    if (p0 >= Mc) break;                                       // wave-uniform
    // The lane FETCHES 16-byte chunks of whole pixel rows (CPP lanes cover one pixel's KC channels: every load
    // instruction reads full 128-byte lines; a lane-per-pixel gather of 32-byte pieces ran the texture path at an
    // eighth of its rate) and the wave turns them into B fragments (lane = pixel) through its private LDS tile.
    // slot j = pixel j*(64/CPP) + lane/CPP of the wave, chunk lane % CPP
    unsigned base[LX], inv[LX];
#pragma unroll
    for (int j = 0; j < LX; ++j) {
      const unsigned p = p0 + (unsigned)(j * (64 / CPP) + lane / CPP);
      int cx, cy, b;
      split(p < Mc ? p : 0u, cx, cy, b);
      const int y0 = TFORM ? cy : cy * g.SH, x0 = TFORM ? cx : cx * g.SW;
      base[j] = (unsigned)(((b * SHh + y0) * SWw + x0) * KC + (lane % CPP) * 8) * 2u;
      unsigned m = p < Mc ? 0u : 0xFFFFu;
      for (int t = 0; t < nkh; ++t) {
        const int kh = kh0 + t * khs;
        const int dy = TFORM ? (py + g.PT - kh) / g.SH : kh - g.PT;
        if ((unsigned)(y0 + dy) >= (unsigned)SHh) m |= 1u << t;
      }
      for (int t = 0; t < nkw; ++t) {
        const int kw = kw0 + t * kws;
        const int dx = TFORM ? (px + g.PL - kw) / g.SW : kw - g.PL;
        if ((unsigned)(x0 + dx) >= (unsigned)SWw) m |= 0x100u << t;
      }
      inv[j] = m;
    }
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = zero16();
    u32x4 xc[LX], xn[LX];
    auto fetch = [&](int lin, u32x4 (&dst)[LX]) {               // tap number lin of this phase's list
      const int th = lin / nkw, tw = lin - th * nkw;
      const int kh = kh0 + th * khs, kw = kw0 + tw * kws;
      const int dy = TFORM ? (py + g.PT - kh) / g.SH : kh - g.PT;
      const int dx = TFORM ? (px + g.PL - kw) / g.SW : kw - g.PL;
      const unsigned delta = (unsigned)((dy * SWw + dx) * KC * 2);
      const unsigned sel = (1u << th) | (0x100u << tw);
#pragma unroll
      for (int j = 0; j < LX; ++j)
        dst[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (inv[j] & sel) ? 0x80000000u : base[j] + delta, 0, 0);
    };
    if (ntaps > 0) fetch(0, xc);
    for (int it = 0; it < ntaps; ++it) {
      if (it + 1 < ntaps) fetch(it + 1, xn);                    // next tap's rows in flight under this tap's work
      WAVE_LDS_SYNC16();                                         // the previous tap's fragment reads are done
#pragma unroll
      for (int j = 0; j < LX; ++j) {
        const int c = j * 64 + lane;
        *reinterpret_cast<u32x4*>(myA + tile_off<KC>(c / CPP, c % CPP)) = xc[j];
      }
      WAVE_LDS_SYNC16();
      const char* wt = sW + it * NC * KC * 2;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const bf16x8 xb = frag_rows<KC>(myA, r, h, kk);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int n = nt * 32 + r;
          const bf16x8 wa = as_frag(*reinterpret_cast<const u32x4*>(wt + n * KC * 2 + (((2 * kk + h) ^ ((KC == 32 ? n >> 1 : n) & (KC / 8 - 1))) << 4)));
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, xb, acc[nt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < LX; ++j) xc[j] = xn[j];
    }
    // ---- epilogue: bias, pack, 16-byte stores (pixels past the end are dropped; the lane swaps need every lane, so only
    //      the store itself is masked).  Lane r's OUTPUT pixel:
    const unsigned p = p0 + r;
    int cx, cy, b;
    split(p < Mc ? p : 0u, cx, cy, b);
    const int64_t opix = TFORM ? ((int64_t)(b * g.IH + (cy * g.SH + py)) * g.IW + (cx * g.SW + px)) : (int64_t)(p < Mc ? p : 0u);
    u32x4 och[CHAIN ? 4 : 1];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      uint2 pk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[nt][4 * q] + bv[nt][q][0], acc[nt][4 * q + 1] + bv[nt][q][1], acc[nt][4 * q + 2] + bv[nt][q][2],
                   acc[nt][4 * q + 3] + bv[nt][q][3]};
        pk[q] = pack4(v);
      }
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        uint2 a = pk[2 * pp], bb = pk[2 * pp + 1];
        auto s0 = __builtin_amdgcn_permlane32_swap(a.x, bb.x, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(a.y, bb.y, false, false);
        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
        if constexpr (CHAIN) och[2 * nt + pp] = o;
        if (p < Mc) *reinterpret_cast<u32x4*>(out + opix * NC + nt * 32 + 16 * pp + 8 * h) = o;
      }
    }
    if constexpr (CHAIN) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[nt] = zero16();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(wfl2[(nt * 4 + kk) * 64 + lane]), as_frag(och[kk]), acc[nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        uint2 pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v = {acc[nt][4 * q], acc[nt][4 * q + 1], acc[nt][4 * q + 2], acc[nt][4 * q + 3]};
          v += *reinterpret_cast<const f32x4*>(b2s + nt * 32 + 8 * q + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
          pk[q] = pack4(v);
        }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          uint2 a = pk[2 * pp], bb = pk[2 * pp + 1];
          auto s0 = __builtin_amdgcn_permlane32_swap(a.x, bb.x, false, false);
          auto s1 = __builtin_amdgcn_permlane32_swap(a.y, bb.y, false, false);
          if (p < Mc) *reinterpret_cast<u32x4*>(out2 + opix * 64 + nt * 32 + 16 * pp + 8 * h) = u32x4{s0[0], s1[0], s0[1], s1[1]};
        }
      }
    }
  }
}

// =================================================================================================
// T-form with the four sub-pixel phases of a stride-2 5 x 5 transposed convolution MERGED (round 4; the bf16 counterpart of
// k_convt_merged_s, kernels_split.hip).  Output pixel (2 cy + py, 2 cx + px) sums the taps kh = py + 1 - 2 dy,
// kw = px + 1 - 2 dx over the input pixels (cy + dy, cx + dx), dy, dx in {-1, 0, 1}: the 25 taps of the four phases read
// only NINE distinct input offsets.  k16_taps<.., TFORM> runs one phase per blockIdx.y and gathers its 32 pixels once per
// tap (25 gathers, 25 LDS tile writes and 25 x KK fragment reads per 4 x 32 output pixels); here a wave's tile is 32
// input-grid positions and all four phases of them: 9 gathers, 9 tile writes, 9 x KK fragment reads, 25 x KK x NT MFMAs
// into acc[phase].  All 25 weight slices stay in LDS as bf16 ([tap][n][k], 100 KB), like the F-form kernel.
// Requires KH = KW = 5, SH = SW = 2, PT = PL = 1, IH = 2 OH, IW = 2 OW.
// =================================================================================================
namespace tm16 {
__device__ __forceinline__ constexpr int o_dy(int oi) { return oi == 0 ? 0 : oi == 1 ? 0 : oi == 2 ? -1 : oi == 3 ? -1 : oi == 4 ? 1 : oi == 5 ? 1 : oi == 6 ? 0 : oi == 7 ? -1 : 1; }
__device__ __forceinline__ constexpr int o_dx(int oi) { return oi == 0 ? 0 : oi == 1 ? -1 : oi == 2 ? 0 : oi == 3 ? -1 : oi == 4 ? 0 : oi == 5 ? -1 : oi == 6 ? 1 : oi == 7 ? 1 : 1; }
__device__ __forceinline__ constexpr int o_ny(int oi) { return o_dy(oi) == 1 ? 1 : 2; }
__device__ __forceinline__ constexpr int o_nx(int oi) { return o_dx(oi) == 1 ? 1 : 2; }
}  // namespace tm16

template <int KC, int NC, int NTHR, bool CHAIN = false, bool ROWS = false>
__global__ void __launch_bounds__(NTHR) k16_taps_tm(const bf16_t* __restrict__ in, const float* __restrict__ W,
                                                   const float* __restrict__ bias, bf16_t* __restrict__ out, ConvGeom g,
                                                   unsigned in_bytes, int tiles_per_wave, const float* __restrict__ W2,
                                                   const float* __restrict__ bias2, bf16_t* __restrict__ out2, int dbg) {
  constexpr int NT = NC / 32, KK = KC / 16;
  static_assert(!CHAIN || NC == 64, "chained conv0: 64 output channels");
  __shared__ __attribute__((aligned(16))) char sW2[CHAIN ? 8 * 64 * 16 + 64 * 4 : 16];
  u32x4* wfl2 = reinterpret_cast<u32x4*>(sW2);
  float* b2s = reinterpret_cast<float*>(sW2 + 8 * 64 * 16);
  if constexpr (CHAIN) {
    if (threadIdx.x < 64) {
      const int l = threadIdx.x;
      bf16x8 w2[2][4];
      load_wfrags<64, 64, false>(W2, l & 31, l >> 5, w2);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) wfl2[(nt * 4 + kk) * 64 + l] = __builtin_bit_cast(u32x4, w2[nt][kk]);
      b2s[l] = bias2 ? bias2[l] : 0.f;
    }
  }
  constexpr int CPP = KC / 8, LX = KC / 16;
  __shared__ __attribute__((aligned(16))) char sW[25 * NC * KC * 2];
  // per wave: the gathered 32 x KC tile of an offset; in the epilogue the same bytes stage one phase's (NC = 64: 32 pixels x
  // 128 B) or one output row's (NC = 32: 64 pixels x 64 B) packed results, row pitch + 16 B, for the line-contiguous stores
  constexpr int OROW = NC == 64 ? 128 : 64, OSLOTS = NC == 64 ? 32 : 64, OPITCH = OROW + 16;
  // ROWS (launched when CW % 32 == 0, KC = 32): the tile's 3 x 34 input pixels (rows cy - 1 .. cy + 1, columns cx0 - 1 .. cx0 + 32)
  // are requested ONE TILE AHEAD (7 coalesced 16-byte loads per lane), written to the wave's LDS tile once, and the nine offsets
  // read their fragments from it at block dy + 1, row r + dx + 1: one wait and 6.5 KB of LDS writes per tile instead of nine
  // gathers of 2 KB with a wait each (the gathers' latency, one offset of prefetch deep, was what the waves waited for).
  constexpr int RPX = 34, RCH = RPX * (KC / 8), RTOT = 3 * RCH, NLR = (RTOT + 63) / 64;
  constexpr int GBYTES = ROWS ? 3 * RPX * KC * 2 : 32 * KC * 2;
  constexpr int ABYTES = (GBYTES > OSLOTS * OPITCH) ? GBYTES : OSLOTS * OPITCH;
  __shared__ __attribute__((aligned(16))) char sA[(NTHR / 64) * ABYTES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  char* myA = sA + wave * ABYTES;
  const int CH = g.OH, CW = g.OW;
  const unsigned Mc = (unsigned)(g.B * CH * CW);
  // ---- all 25 weight slices -> LDS, bf16, T layout [tap][n][k] (W[tap][n = ci][k = co]), chunks XOR-swizzled as k16_taps
  for (int idx = threadIdx.x; idx < 25 * NC * KC; idx += NTHR) {
    const int tap = idx / (NC * KC), rem = idx % (NC * KC);
    const int n = rem / KC, k = rem % KC;
    const float v = W[idx];
    const int off = tap * NC * KC * 2 + n * KC * 2 + (((k >> 3) ^ ((KC == 32 ? n >> 1 : n) & (KC / 8 - 1))) << 4) + (k & 7) * 2;
    *reinterpret_cast<uint16_t*>(sW + off) = (uint16_t)(pack_bf16(v, 0.f) & 0xFFFFu);
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(in), 0, (int)in_bytes, 0x00020000);
  const bool pow2 = (CW & (CW - 1)) == 0 && (CH & (CH - 1)) == 0;
  const int lgw = 31 - __builtin_clz((unsigned)CW), lgh = 31 - __builtin_clz((unsigned)CH);
  auto split = [&](unsigned p, int& cx, int& cy, int& b) {
    if (pow2) { cx = (int)(p & (unsigned)(CW - 1)); cy = (int)((p >> lgw) & (unsigned)(CH - 1)); b = (int)(p >> (lgw + lgh)); }
    else { cx = (int)(p % (unsigned)CW); const unsigned q = p / (unsigned)CW; cy = (int)(q % (unsigned)CH); b = (int)(q / (unsigned)CH); }
  };
  // tile order.  CW % 32 == 0: a tile is 32 positions of ONE row; the block walks down a 32-wide column strip with its waves on
  // vertically adjacent rows (tile t of the block's run -> row cy = t % CH of strip (t / CH) % strips), so the rows a wave
  // gathers for dy = -1, 0, 1 are the rows its neighbours gather too and come from the CU's L1 instead of nine times from L2.
  const bool strips_on = (CW & 31) == 0 && !(dbg & 8);       // (dbg bit 3: MVAE_TD_DBG=8, plain row-major tile order)
  const unsigned nstrip = (unsigned)(CW >> 5);
  auto tile_p0 = [&](int ti) -> unsigned {
    if (ti >= tiles_per_wave) return Mc;
    if (!strips_on) return (((unsigned)blockIdx.x * (unsigned)(NTHR / 64) + wave) * (unsigned)tiles_per_wave + (unsigned)ti) * 32u;
    const unsigned t = (unsigned)blockIdx.x * (unsigned)(NTHR / 64) * (unsigned)tiles_per_wave + (unsigned)ti * (unsigned)(NTHR / 64) + wave;
    const unsigned cy = t % (unsigned)CH, q = t / (unsigned)CH;
    const unsigned sx = q % nstrip, b = q / nstrip;
    return b >= (unsigned)g.B ? Mc : ((b * (unsigned)CH + cy) * (unsigned)CW + sx * 32u);
  };
  u32x4 xr[ROWS ? NLR : 1];
  auto request_rows = [&](unsigned q0) {                         // the 3 x 34 pixels around the tile that starts at position q0
    const bool valid = q0 < Mc;
    int cx0, cy0, b0;
    split(valid ? q0 : 0u, cx0, cy0, b0);
#pragma unroll
    for (int u = 0; u < (ROWS ? NLR : 1); ++u) {
      const int i = u * 64 + lane, rowi = i / RCH, rem = i - rowi * RCH;
      const int y = cy0 + rowi - 1, x = cx0 - 1 + rem / CPP;
      const bool ok = valid && i < RTOT && (unsigned)y < (unsigned)CH && (unsigned)x < (unsigned)CW;
      xr[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? (unsigned)(((b0 * CH + y) * CW + x) * KC + (rem % CPP) * 8) * 2u : 0x80000000u, 0, 0);
    }
  };
  if constexpr (ROWS) request_rows(tile_p0(0));
  for (int ti = 0; ti < tiles_per_wave; ++ti) {
    const unsigned p0 = tile_p0(ti);
    if (p0 >= Mc) break;                                       // wave-uniform
    unsigned base[LX], inv[LX];
#pragma unroll
    for (int j = 0; j < LX; ++j) {
      const unsigned p = p0 + (unsigned)(j * (64 / CPP) + lane / CPP);
      int cx, cy, b;
      split(p < Mc ? p : 0u, cx, cy, b);
      base[j] = (unsigned)(((b * CH + cy) * CW + cx) * KC + (lane % CPP) * 8) * 2u;
      unsigned m = p < Mc ? 0u : 0x77u;                        // bit dy + 1 / bit 4 + dx + 1: the offset leaves the image
      if (cy == 0) m |= 1u;
      if (cy == CH - 1) m |= 4u;
      if (cx == 0) m |= 0x10u;
      if (cx == CW - 1) m |= 0x40u;
      inv[j] = m;
    }
    f32x16 acc[4][NT];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[ph][nt] = zero16();
    u32x4 xc[LX], xn[LX];
    auto fetch = [&](int oi, u32x4 (&dst)[LX]) {
      const int dy = tm16::o_dy(oi), dx = tm16::o_dx(oi);
      const unsigned delta = (unsigned)((dy * CW + dx) * KC * 2);
      const unsigned sel = (1u << (dy + 1)) | (0x10u << (dx + 1));
#pragma unroll
      for (int j = 0; j < LX; ++j)
        dst[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (inv[j] & sel) ? 0x80000000u : base[j] + delta, 0, 0);
    };
    if constexpr (ROWS) {
      WAVE_LDS_SYNC16();                                         // the previous tile's staged results have left
#pragma unroll
      for (int u = 0; u < NLR; ++u) {
        const int i = u * 64 + lane, rowi = i / RCH, rem = i - rowi * RCH;
        if (i < RTOT) *reinterpret_cast<u32x4*>(myA + rowi * (RPX * KC * 2) + tile_off<KC>(rem / CPP, rem % CPP)) = xr[u];
      }
      request_rows(tile_p0(ti + 1));                             // the next tile's rows: in flight under this tile's 100 MFMAs
      WAVE_LDS_SYNC16();
    } else {
      fetch(0, xc);
    }
#pragma unroll
    for (int oi = 0; oi < 9; ++oi) {
      bf16x8 xb[KK];
      if constexpr (ROWS) {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)       // (image row dy + 1 is its own swizzled block: six distinct lane addresses in all)
          xb[kk] = frag_rows<KC>(myA + (tm16::o_dy(oi) + 1) * (RPX * KC * 2), r + tm16::o_dx(oi) + 1, h, kk);
      } else {
        if (oi + 1 < 9) fetch(oi + 1, xn);                      // next offset's rows in flight under this offset's taps
        WAVE_LDS_SYNC16();                                       // the previous offset's fragment reads are done
#pragma unroll
        for (int j = 0; j < LX; ++j) {
          const int c = j * 64 + lane;
          *reinterpret_cast<u32x4*>(myA + tile_off<KC>(c / CPP, c % CPP)) = xc[j];
        }
        WAVE_LDS_SYNC16();
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) xb[kk] = frag_rows<KC>(myA, r, h, kk);
      }
#pragma unroll
      for (int jy = 0; jy < tm16::o_ny(oi); ++jy)
#pragma unroll
        for (int jx = 0; jx < tm16::o_nx(oi); ++jx) {
          const int py = tm16::o_dy(oi) == 1 ? 1 : jy, px = tm16::o_dx(oi) == 1 ? 1 : jx;
          const int kh = py + 1 - 2 * tm16::o_dy(oi), kw = px + 1 - 2 * tm16::o_dx(oi);
          const char* wt = sW + (kh * 5 + kw) * NC * KC * 2;
#pragma unroll
          for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const int n = nt * 32 + r;
              const bf16x8 wa = as_frag(*reinterpret_cast<const u32x4*>(wt + n * KC * 2 + (((2 * kk + h) ^ ((KC == 32 ? n >> 1 : n) & (KC / 8 - 1))) << 4)));
              acc[py * 2 + px][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, xb[kk], acc[py * 2 + px][nt], 0, 0, 0);
            }
        }
      if constexpr (!ROWS) {
#pragma unroll
        for (int j = 0; j < LX; ++j) xc[j] = xn[j];
      }
    }
    if constexpr (ROWS) WAVE_LDS_SYNC16();                       // the fragment reads are done: the tile's bytes now stage the results
    // ---- epilogue per phase: bias, pack; stores.  Per-lane stores (lane (r, h) -> 16 bytes of output pixel 2 (cx0 + r) + px) put
    // 64 pieces of 16 / 32 bytes, 256 bytes apart, into every store instruction: 1024 partial-line requests per tile, and the
    // memory pipeline's request rate -- not its bandwidth -- set the kernel's time (skipping the stores: -65 .. -140 us of ~300 at
    // 256 x 256).  With the tile in one image row (strips_on) the packed results go through the wave's LDS tile and leave as
    // whole 128-byte lines: NC = 64: per phase 32 pixels x 128 B (8 lanes per pixel); NC = 32: per output row the 64 pixels
    // 2 cx0 .. 2 cx0 + 63 x 64 B = 4 KB contiguous.
    const unsigned p = p0 + r;
    int cx, cy, b;
    split(p < Mc ? p : 0u, cx, cy, b);
    const int64_t opix00 = (int64_t)(b * g.IH + cy * 2) * g.IW + cx * 2;
    int cx0, cy0, b0;
    split(p0, cx0, cy0, b0);
    auto stage_put = [&](int slot, int chunk, const u32x4& v) { *reinterpret_cast<u32x4*>(myA + slot * OPITCH + chunk * 16) = v; };
    auto flush64 = [&](bf16_t* dst, int px, int py) {            // NC = 64: staged 32 pixels x 128 B of phase (py, px)
      WAVE_LDS_SYNC16();
      char* row = reinterpret_cast<char*>(dst) + ((int64_t)(b0 * g.IH + cy0 * 2 + py) * g.IW + cx0 * 2 + px) * 128;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = j * 64 + lane, slot = q >> 3, c = q & 7;
        const u32x4 v = *reinterpret_cast<const u32x4*>(myA + slot * OPITCH + c * 16);
        *reinterpret_cast<u32x4*>(row + (int64_t)slot * 256 + c * 16) = v;
      }
      WAVE_LDS_SYNC16();
    };
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      const int64_t opix = opix00 + (ph >> 1) * g.IW + (ph & 1);
      u32x4 och[CHAIN ? 4 : 1];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        uint2 pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 bq = {0.f, 0.f, 0.f, 0.f};
          if (bias) bq = *reinterpret_cast<const f32x4*>(bias + nt * 32 + 8 * q + 4 * h);
          f32x4 v = {acc[ph][nt][4 * q] + bq[0], acc[ph][nt][4 * q + 1] + bq[1], acc[ph][nt][4 * q + 2] + bq[2],
                     acc[ph][nt][4 * q + 3] + bq[3]};
          pk[q] = pack4(v);
        }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          uint2 a = pk[2 * pp], bb = pk[2 * pp + 1];
          auto s0 = __builtin_amdgcn_permlane32_swap(a.x, bb.x, false, false);
          auto s1 = __builtin_amdgcn_permlane32_swap(a.y, bb.y, false, false);
          const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
          if constexpr (CHAIN) och[2 * nt + pp] = o;
          if (strips_on) {
            if constexpr (NC == 64) stage_put(r, nt * 4 + pp * 2 + h, o);
            else stage_put(2 * r + (ph & 1), pp * 2 + h, o);
          } else if (p < Mc) {
            *reinterpret_cast<u32x4*>(out + opix * NC + nt * 32 + 16 * pp + 8 * h) = o;
          }
        }
      }
      if (strips_on) {
        if constexpr (NC == 64) {
          flush64(out, ph & 1, ph >> 1);
        } else if (ph & 1) {                                     // NC = 32: both px phases of output row py are staged
          WAVE_LDS_SYNC16();
          char* row = reinterpret_cast<char*>(out) + ((int64_t)(b0 * g.IH + cy0 * 2 + (ph >> 1)) * g.IW + cx0 * 2) * 64;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int q = j * 64 + lane, slot = q >> 2, c = q & 3;
            *reinterpret_cast<u32x4*>(row + (int64_t)q * 16) = *reinterpret_cast<const u32x4*>(myA + slot * OPITCH + c * 16);
          }
          WAVE_LDS_SYNC16();
        }
      }
      if constexpr (CHAIN) {
        f32x16 a2[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) a2[nt] = zero16();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            a2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(wfl2[(nt * 4 + kk) * 64 + lane]), as_frag(och[kk]), a2[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          uint2 pk[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = {a2[nt][4 * q], a2[nt][4 * q + 1], a2[nt][4 * q + 2], a2[nt][4 * q + 3]};
            v += *reinterpret_cast<const f32x4*>(b2s + nt * 32 + 8 * q + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            pk[q] = pack4(v);
          }
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) {
            uint2 a = pk[2 * pp], bb = pk[2 * pp + 1];
            auto s0 = __builtin_amdgcn_permlane32_swap(a.x, bb.x, false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(a.y, bb.y, false, false);
            const u32x4 o2 = {s0[0], s1[0], s0[1], s1[1]};
            if (strips_on) stage_put(r, nt * 4 + pp * 2 + h, o2);
            else if (p < Mc) *reinterpret_cast<u32x4*>(out2 + opix * 64 + nt * 32 + 16 * pp + 8 * h) = o2;
          }
        }
        if (strips_on) flush64(out2, ph & 1, ph >> 1);
      }
    }
  }
}

// =================================================================================================
// F-form (5 x 5, stride 2, SAME) in bf16 with the input rows of a KERNEL ROW staged in LDS once (round 4; the bf16
// counterpart of k_convf_rows_s, kernels_split.hip).  k16_taps<.., TFORM = false> gathers a wave's 32 input pixels once per
// tap: 25 gather -> LDS -> fragment phases per tile, each a dependent chain, and every input pixel crosses the fabric 2 - 4
// times (PMC, round 2/3).  Here a wave loads, per kernel row kh, the R = 32 / COLS input row segments its 32 output pixels
// read through the five kw taps (2 COLS + 3 pixels each), once; the five taps of the row read their B fragments from that
// LDS region at pixel 2 c + kw (even and odd segment pixels stored apart, output row minor: row index ((q & 1) HALF +
// (q >> 1)) R + j): 5 load / store phases per tile instead of 25, 2.3x fewer gathered bytes.  All 25 weight slices stay in
// LDS (100 KB), which leaves room for NW = 6 (KC = 64) or 12 (KC = 32) waves of staged rows.
// =================================================================================================
static int cus16();
template <int C>
__device__ __forceinline__ int rtile_off(int row, int chunk) {         // split.h's row_off: conflict-free for rows at stride R
  if constexpr (C == 64) return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
  else return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}
template <int KC, int NC, int COLS, int NW>
__global__ void __launch_bounds__(64 * NW) k16_taps_fr(const bf16_t* __restrict__ in, const float* __restrict__ W,
                                                     const float* __restrict__ bias, bf16_t* __restrict__ out, ConvGeom g,
                                                     unsigned in_bytes, int tiles_per_wave) {
  constexpr int NT = NC / 32, KK = KC / 16;
  constexpr int CPP = KC / 8, PPI = 64 / CPP;                        // 16-byte chunks per pixel, pixels per load instruction
  constexpr int R = 32 / COLS, SEG = 2 * COLS + 3, HALF = COLS + 2, NPX = R * SEG;
  constexpr int NL = (NPX + PPI - 1) / PPI;
  constexpr int ABYTES = ((NPX + 7) / 8 * 8) * KC * 2;
  extern __shared__ __attribute__((aligned(16))) char fr_lds[];
  char* sW = fr_lds;                                                 // [25][NC][KC] bf16
  char* sA = fr_lds + 25 * NC * KC * 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  char* myA = sA + wave * ABYTES;
  const int CH = g.OH, CW = g.OW;
  const unsigned Mc = (unsigned)(g.B * CH * CW);
  // all 25 weight slices -> LDS, bf16, [tap][n][k] from the Keras HWIO array W[tap][k][n]
  for (int idx = threadIdx.x; idx < 25 * NC * KC; idx += 64 * NW) {
    const int tap = idx / (NC * KC), rem = idx % (NC * KC);
    const int k = rem / NC, n = rem % NC;
    const float v = W[idx];
    const int off = tap * NC * KC * 2 + n * KC * 2 + (((k >> 3) ^ ((KC == 32 ? n >> 1 : n) & (KC / 8 - 1))) << 4) + (k & 7) * 2;
    *reinterpret_cast<uint16_t*>(sW + off) = (uint16_t)(pack_bf16(v, 0.f) & 0xFFFFu);
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(in), 0, (int)in_bytes, 0x00020000);
  const int lp = lane / CPP, ch = lane % CPP;
  const unsigned row_bytes = (unsigned)(g.IW * KC * 2);
  const int rho0 = (r % COLS) * R + (r / COLS);
  const unsigned wtile0 = ((unsigned)blockIdx.x * (unsigned)NW + wave) * (unsigned)tiles_per_wave;
  for (int ti = 0; ti < tiles_per_wave; ++ti) {
    const unsigned p0 = (wtile0 + ti) * 32u;
    if (p0 >= Mc) break;                                             // wave-uniform
    unsigned base[NL], rmask[NL];
    int lds_o[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int sidx = u * PPI + lp;
      const int j = sidx / SEG, q = sidx - j * SEG;
      const unsigned pj = p0 + (unsigned)(j * COLS);
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
      base[u] = (unsigned)(((b * g.IH + y0) * g.IW + x) * KC + ch * 8) * 2u;
      lds_o[u] = rtile_off<KC>(((q & 1) * HALF + (q >> 1)) * R + j, ch);
    }
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = zero16();
    u32x4 xc[NL], xn[NL];
    auto fetch = [&](int kh, u32x4 (&dst)[NL]) {
#pragma unroll
      for (int u = 0; u < NL; ++u)
        dst[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (rmask[u] >> kh) & 1u ? 0x80000000u : base[u] + (unsigned)kh * row_bytes, 0, 0);
    };
    fetch(0, xc);
#pragma unroll
    for (int kh = 0; kh < 5; ++kh) {
      if (kh + 1 < 5) fetch(kh + 1, xn);                             // next kernel row's segments in flight under this row's taps
      WAVE_LDS_SYNC16();                                             // the previous row's fragment reads are done
#pragma unroll
      for (int u = 0; u < NL; ++u)
        if (u * PPI + lp < NPX) *reinterpret_cast<u32x4*>(myA + lds_o[u]) = xc[u];
      WAVE_LDS_SYNC16();
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        const int rho = rho0 + ((kw & 1) * HALF + (kw >> 1)) * R;
        const char* wt = sW + (kh * 5 + kw) * NC * KC * 2;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
          const bf16x8 xb = as_frag(*reinterpret_cast<const u32x4*>(myA + rtile_off<KC>(rho, 2 * kk + h)));
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 32 + r;
            const bf16x8 wa = as_frag(*reinterpret_cast<const u32x4*>(wt + n * KC * 2 + (((2 * kk + h) ^ ((KC == 32 ? n >> 1 : n) & (KC / 8 - 1))) << 4)));
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, xb, acc[nt], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < NL; ++u) xc[u] = xn[u];
    }
    // ---- epilogue (as k16_taps, F-form): bias, pack, 16-byte stores
    const unsigned p = p0 + r;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      uint2 pk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 bq = {0.f, 0.f, 0.f, 0.f};
        if (bias) bq = *reinterpret_cast<const f32x4*>(bias + nt * 32 + 8 * q + 4 * h);
        f32x4 v = {acc[nt][4 * q] + bq[0], acc[nt][4 * q + 1] + bq[1], acc[nt][4 * q + 2] + bq[2], acc[nt][4 * q + 3] + bq[3]};
        pk[q] = pack4(v);
      }
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        uint2 a = pk[2 * pp], bb = pk[2 * pp + 1];
        auto s0 = __builtin_amdgcn_permlane32_swap(a.x, bb.x, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(a.y, bb.y, false, false);
        if (p < Mc) *reinterpret_cast<u32x4*>(out + (int64_t)p * NC + nt * 32 + 16 * pp + 8 * h) = u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
    }
  }
}
template <int KC, int NC, int COLS, int NW>
static bool launch16_taps_fr(const void* in, const float* w, const float* bias, void* out, const ConvGeom& g, unsigned in_bytes,
                             int64_t tiles, hipStream_t s) {
  constexpr int R = 32 / COLS, NPX = R * (2 * COLS + 3), ABYTES = ((NPX + 7) / 8 * 8) * KC * 2;
  constexpr int lds = 25 * NC * KC * 2 + NW * ABYTES;
  static_assert(lds <= 160 * 1024, "staged rows + weight slices do not fit the LDS");
  static const bool attr = hipFuncSetAttribute((const void*)k16_taps_fr<KC, NC, COLS, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               lds) == hipSuccess;
  if (!attr) return false;
  int64_t waves = (int64_t)NW * cus16();
  if (waves > tiles) waves = tiles;
  const int tpw = (int)((tiles + waves - 1) / waves);
  const unsigned gx = (unsigned)(((tiles + tpw - 1) / tpw + NW - 1) / NW);
  hipLaunchKernelGGL((k16_taps_fr<KC, NC, COLS, NW>), dim3(gx), dim3(64 * NW), lds, s, (const bf16_t*)in, w, bias, (bf16_t*)out, g,
                     in_bytes, tpw);
  return true;
}

// =================================================================================================
// Weight gradient of a strided SAME convolution (F-form coordinates), bf16 storage:
//   dW[tap][ci][co] += sum_m big[gather(m, tap)][ci] * small[m][co] ;  db[co] += sum_m small[m][co]
// One block per (row chunk, kernel row kh) as k_wgrad_taprow: the TG = KW taps of the row share the staged `small` tile
// (its transposed fragments stay in registers), the gathered `big` tile changes per tap.  1x1 convolutions: TG = 1.
// D[row = ci][col = co]: A = big^T, B = small, both read with ds_read_b64_tr_b16 from row-major wave-private tiles.
// =================================================================================================
template <int CI, int CO, int TG>
__global__ void __launch_bounds__(256, 2) k16_wgrad(const bf16_t* __restrict__ big, const bf16_t* __restrict__ small,
                                                 float* __restrict__ dW, float* __restrict__ db, ConvGeom g, int64_t M,
                                                 int64_t rows_per_block, int nslots, int64_t slot_stride) {
  constexpr int KT = CI / 32, NT = CO / 32;
  constexpr int TILE = 32 * (CI + CO) * 2;
  __shared__ __attribute__((aligned(16))) char lds[(4 * TILE > CI * CO * 4) ? 4 * TILE : CI * CO * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  char* tb = lds + wave * TILE;                   // gathered big tile [32][CI]
  char* ts = tb + 32 * CI * 2;                    // small tile [32][CO]
  const int xcd = blockIdx.x & 7, kh = (blockIdx.x >> 3) % g.KH;
  const uint32_t chunk = ((blockIdx.x >> 3) / g.KH) * 8u + xcd;
  if ((int64_t)chunk * rows_per_block >= M) return;            // block-uniform, before any barrier
  f32x16 acc[TG][KT][NT];
#pragma unroll
  for (int t = 0; t < TG; ++t)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[t][kt][nt] = zero16();
  float bsum[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bsum[nt] = 0.f;
  uint32_t m_begin = chunk * (uint32_t)rows_per_block;
  uint32_t m_end = m_begin + (uint32_t)rows_per_block;
  if (m_end > (uint32_t)M) m_end = (uint32_t)M;
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(big), 0,
      (int)((unsigned)g.B * g.IH * g.IW * CI * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(small), 0,
      (int)((unsigned)M * CO * 2u), 0x00020000);
  constexpr int LX = CI / 16, LG = CO / 16;       // 16-byte chunks per lane per tile
  constexpr int CPX = CI / 8, CPG = CO / 8;       // chunks per pixel
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
      pbase[j] = ok ? (unsigned)(((bi * g.IH + yy) * g.IW + x0) * CI + ch * 8) * 2u : 0xC0000000u;
      px0[j] = x0;
    }
  };
  u32x4 xq[LX], gq[LG];
  auto load_big = [&](int kw) {
#pragma unroll
    for (int j = 0; j < LX; ++j) {
      const unsigned off = (unsigned)(px0[j] + kw) < (unsigned)g.IW ? pbase[j] + (unsigned)(kw * CI * 2) : 0x80000000u;
      xq[j] = __builtin_amdgcn_raw_buffer_load_b128(brs, off, 0, 0);
    }
  };
  auto load_small = [&](uint32_t row0) {
#pragma unroll
    for (int j = 0; j < LG; ++j) {
      const int c = j * 64 + lane, pr = c / CPG, ch = c % CPG;
      const uint32_t mm = row0 + pr;
      const unsigned off = mm < m_end ? (mm * CO + ch * 8) * 2u : 0x80000000u;
      gq[j] = __builtin_amdgcn_raw_buffer_load_b128(srs, off, 0, 0);
    }
  };
  uint32_t row0 = m_begin + wave * 32;
  if (row0 < m_end) { decode(row0); load_small(row0); load_big(0); }
  for (; row0 < m_end; row0 += 4 * 32) {
    WAVE_LDS_SYNC16();
#pragma unroll
    for (int j = 0; j < LG; ++j) {
      const int c = j * 64 + lane;
      *reinterpret_cast<u32x4*>(ts + tile_off<CO>(c / CPG, c % CPG)) = gq[j];
    }
    bf16x8 fs[NT][2];
    const bool more = row0 + 4 * 32 < m_end;
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      if (t > 0) WAVE_LDS_SYNC16();
#pragma unroll
      for (int j = 0; j < LX; ++j) {
        const int c = j * 64 + lane;
        *reinterpret_cast<u32x4*>(tb + tile_off<CI>(c / CPX, c % CPX)) = xq[j];
      }
      WAVE_LDS_SYNC16();
      if (t + 1 < TG) load_big(t + 1);
      else if (more) { decode(row0 + 4 * 32); load_small(row0 + 4 * 32); load_big(0); }
      if (t == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int s = 0; s < 2; ++s) { fs[nt][s] = frag_cols<CO>(ts, lane, nt, s); bsum[nt] = frag_sum(fs[nt][s], bsum[nt]); }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const bf16x8 fb = frag_cols<CI>(tb, lane, kt, s);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[t][kt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fs[nt][s], acc[t][kt][nt], 0, 0, 0);
        }
    }
  }
  float* red = reinterpret_cast<float*>(lds);
  const int64_t gslot = (int64_t)(blockIdx.x % nslots) * slot_stride;
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
    float* dWt = dW + gslot + (int64_t)(kh * g.KW + t) * CI * CO;
    for (int idx = threadIdx.x; idx < CI * CO; idx += 256) atomicAdd(&dWt[idx], red[idx]);
  }
  if (db != nullptr && kh == 0) {
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float b = bsum[nt] + __shfl_xor(bsum[nt], 32, 64);
      if (h == 0) red[wave * CO + nt * 32 + r] = b;
    }
    __syncthreads();
    if (threadIdx.x < CO)
      atomicAdd(&db[gslot + threadIdx.x], red[threadIdx.x] + red[CO + threadIdx.x] + red[2 * CO + threadIdx.x] + red[3 * CO + threadIdx.x]);
  }
}

// =================================================================================================
// 5 x 5 stride-2 weight gradient in bf16 with the `big` operand's row segment staged in LDS once per tile (round 4).
// k16_wgrad<*, *, 5> gathers the 32 input pixels of a tile once per tap: five gather -> LDS -> transposed-fragment phases per
// tile.  With OW a multiple of 32 a tile is 32 consecutive pixels of ONE output row, and the five kw taps of the block's
// kernel row read the same input row at columns 2 c + kw: the 67-pixel segment is loaded once, even and odd pixels stored
// apart (row = (q & 1) 34 + (q >> 1)), and tap kw reads its transposed fragments at row offset (kw & 1) 34 + (kw >> 1).
// 2.4x fewer gathered bytes and LDS writes; same block decomposition (row chunk x kernel row, XCD-aware grid) and epilogue.
// =================================================================================================
template <int C>
__device__ __forceinline__ bf16x8 frag_cols_at(const char* tile, int lane, int ct, int s, int roff) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int c0 = ct * 32 + 16 * (g & 1) + 4 * p;
  const int rb = roff + 16 * s + 8 * (g >> 1);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int o0 = tile_off<C>(rb + q, c0 >> 3) + (c0 & 7) * 2;
  const int o1 = tile_off<C>(rb + 4 + q, c0 >> 3) + (c0 & 7) * 2;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + o0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + o1));
  s16x8 f = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, f);
}
template <int CI, int CO>
__global__ void __launch_bounds__(256, 2) k16_wgrad_seg(const bf16_t* __restrict__ big, const bf16_t* __restrict__ small,
                                                        float* __restrict__ dW, float* __restrict__ db, ConvGeom g, int64_t M,
                                                        int64_t rows_per_block, float* __restrict__ dbig, int nslots,
                                                        int64_t slot_stride) {
  constexpr int KT = CI / 32, NT = CO / 32, TG = 5;
  constexpr int SEG = 67, HALF = 34, SROWS = 72;                     // staged pixels, even-pixel rows, rows reserved
  constexpr int TB = SROWS * CI * 2, TS = 32 * CO * 2, TILE = TB + TS;
  __shared__ __attribute__((aligned(16))) char lds[(4 * TILE > CI * CO * 4) ? 4 * TILE : CI * CO * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  char* tb = lds + wave * TILE;
  char* ts = tb + TB;
  const int xcd = blockIdx.x & 7, kh = (blockIdx.x >> 3) % g.KH;
  const uint32_t chunk = ((blockIdx.x >> 3) / g.KH) * 8u + xcd;
  if ((int64_t)chunk * rows_per_block >= M) return;
  f32x16 acc[TG][KT][NT];
#pragma unroll
  for (int t = 0; t < TG; ++t)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[t][kt][nt] = zero16();
  float bsum[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bsum[nt] = 0.f;
  uint32_t m_begin = chunk * (uint32_t)rows_per_block;
  uint32_t m_end = m_begin + (uint32_t)rows_per_block;
  if (m_end > (uint32_t)M) m_end = (uint32_t)M;
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(big), 0,
      (int)((unsigned)g.B * g.IH * g.IW * CI * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(small), 0,
      (int)((unsigned)M * CO * 2u), 0x00020000);
  constexpr int CPX = CI / 8, CPG = CO / 8;                          // 16-byte chunks per pixel
  constexpr int PPI = 64 / CPX, NLB = (SEG + PPI - 1) / PPI, LG = CO / 16;
  const int lpx = lane / CPX, chx = lane % CPX;
  u32x4 xq[NLB], gq[LG];
  // dbig != nullptr (a Conv2DTranspose's bias gradient = the column sums of `big`, PT = PL = 1, IH = 2 OH, IW = 2 OW): the blocks
  // of kernel rows 1 and 2 meet every row of `big` exactly once (y = 2 oh + kh - 1), and the pixels q = 1 .. 64 of a tile's
  // segment are the columns 2 ow0 .. 2 ow0 + 63 no other tile owns: their sum rides on the segment's way to LDS instead of a
  // separate read pass over the tensor (k_colstat4).
  const bool own_rows = dbig != nullptr && (kh == 1 || kh == 2);
  float bacc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bacc[e] = 0.f;
  auto load_seg = [&](uint32_t row0) {                               // the tile's input row segment: 67 pixels from column 2 ow0 - PL
    const uint32_t ow0 = row0 % (uint32_t)g.OW, t2 = row0 / (uint32_t)g.OW;
    const int oh = (int)(t2 % (uint32_t)g.OH), bi = (int)(t2 / (uint32_t)g.OH);
    const int yy = oh * g.SH + kh - g.PT, x0 = (int)ow0 * g.SW - g.PL;
    const bool rowok = row0 < m_end && (unsigned)yy < (unsigned)g.IH;
    const unsigned rbase = (unsigned)(((bi * g.IH + yy) * g.IW + x0) * CI) * 2u;
#pragma unroll
    for (int u = 0; u < NLB; ++u) {
      const int q = u * PPI + lpx;
      const bool ok = rowok && q < SEG && (unsigned)(x0 + q) < (unsigned)g.IW;
      xq[u] = __builtin_amdgcn_raw_buffer_load_b128(brs, ok ? rbase + (unsigned)((q * CI + chx * 8) * 2) : 0x80000000u, 0, 0);
    }
  };
  auto load_small = [&](uint32_t row0) {
#pragma unroll
    for (int j = 0; j < LG; ++j) {
      const int c = j * 64 + lane, pr = c / CPG, ch = c % CPG;
      const uint32_t mm = row0 + pr;
      gq[j] = __builtin_amdgcn_raw_buffer_load_b128(srs, mm < m_end ? (mm * CO + ch * 8) * 2u : 0x80000000u, 0, 0);
    }
  };
  uint32_t row0 = m_begin + wave * 32;
  if (row0 < m_end) { load_small(row0); load_seg(row0); }
  for (; row0 < m_end; row0 += 4 * 32) {
    WAVE_LDS_SYNC16();                              // the previous tile's fragment reads are done
#pragma unroll
    for (int j = 0; j < LG; ++j) {
      const int c = j * 64 + lane;
      *reinterpret_cast<u32x4*>(ts + tile_off<CO>(c / CPG, c % CPG)) = gq[j];
    }
#pragma unroll
    for (int u = 0; u < NLB; ++u) {
      const int q = u * PPI + lpx;
      if (q < SEG) *reinterpret_cast<u32x4*>(tb + tile_off<CI>((q & 1) * HALF + (q >> 1), chx)) = xq[u];
      if (own_rows && q >= 1 && q <= 64) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { bacc[2 * e] += bf16_lo(xq[u][e]); bacc[2 * e + 1] += bf16_hi(xq[u][e]); }
      }
    }
    WAVE_LDS_SYNC16();
    if (row0 + 4 * 32 < m_end) { load_small(row0 + 4 * 32); load_seg(row0 + 4 * 32); }     // next tile in flight under the MFMAs
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fs[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) { fs[nt] = frag_cols<CO>(ts, lane, nt, s); bsum[nt] = frag_sum(fs[nt], bsum[nt]); }
#pragma unroll
      for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const bf16x8 fb = frag_cols_at<CI>(tb, lane, kt, s, (t & 1) * HALF + (t >> 1));
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[t][kt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fs[nt], acc[t][kt][nt], 0, 0, 0);
        }
    }
  }
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
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float b = bsum[nt] + __shfl_xor(bsum[nt], 32, 64);
      if (h == 0) red[wave * CO + nt * 32 + r] = b;
    }
    __syncthreads();
    if (threadIdx.x < CO)
      atomicAdd(&db[threadIdx.x], red[threadIdx.x] + red[CO + threadIdx.x] + red[2 * CO + threadIdx.x] + red[3 * CO + threadIdx.x]);
  }
  if (own_rows) {                                          // lanes with the same channel chunk (lane % CPX) -> one sum; 4 waves -> LDS
#pragma unroll
    for (int e = 0; e < 8; ++e)
      for (int off = CPX; off < 64; off <<= 1) bacc[e] += __shfl_xor(bacc[e], off, 64);
    __syncthreads();
    if (lane < CPX) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wave * CI + lane * 8 + e] = bacc[e];
    }
    __syncthreads();
    if (threadIdx.x < CI)
      atomicAdd(&dbig[(int64_t)(blockIdx.x % (unsigned)nslots) * slot_stride + threadIdx.x],
                red[threadIdx.x] + red[CI + threadIdx.x] + red[2 * CI + threadIdx.x] + red[3 * CI + threadIdx.x]);
  }
}

// ---- launchers ------------------------------------------------------------------------------------------------------
static int cus16() {
  static const int v = [] { const char* e = getenv("MVAE_BIG_CUS16"); int n = e ? atoi(e) : 256; return n < 8 ? 8 : (n > 256 ? 256 : n); }();
  return v;
}

// 1x1 convolution forward / transposed.  false = shape not covered.
bool launch16_pw(bool transposed, const void* in, const float* w, const float* bias, const float* gate, const void* residual,
                 void* out, int64_t M, int64_t rows_per_image, int K, int N, int act, hipStream_t s, bool out_f32,
                 const float* pivot, float* st1, float* st2, int nslots, int64_t slot_stride) {
  if (M % 32 != 0 || M <= 0) return false;
  if (gate && (rows_per_image % 32 != 0)) return false;
  const int64_t ntiles = M / 32;
  const int64_t nblk = (ntiles + 3) / 4;
  const int grid = (int)(nblk < 4 * cus16() ? nblk : 4 * cus16());
  const bf16_t* X = (const bf16_t*)in;
  const bf16_t* R = (const bf16_t*)residual;
  void* Y = out;
  if (out_f32) {                                 // the conv2 of a decoder's last block: gate + residual, float32 result
    if (!gate || !residual || transposed || act != ACT_NONE) return false;
    ProfScope ps("k16_pw", 2.0 * M * (K + N) + 4.0 * M * N, 2.0 * M * K * N, s);
    if (K == 32 && N == 32) hipLaunchKernelGGL((k16_pw<32, 32, false, true, true, ACT_NONE, true>), dim3(grid), dim3(256), 0, s, X, w, bias, gate, R, Y, ntiles, rows_per_image, pivot, st1, st2, nslots < 1 ? 1 : nslots, slot_stride);
    else if (K == 64 && N == 64) hipLaunchKernelGGL((k16_pw<64, 64, false, true, true, ACT_NONE, true>), dim3(grid), dim3(256), 0, s, X, w, bias, gate, R, Y, ntiles, rows_per_image, pivot, st1, st2, nslots < 1 ? 1 : nslots, slot_stride);
    else return false;
    return true;
  }
  ProfScope ps("k16_pw", 2.0 * M * (K + N * (residual ? 2 : 1)), 2.0 * M * K * N, s);
#define MVAE_PW1(KK, NN, WT_, G_, R_, A_)                                                                            \
  hipLaunchKernelGGL((k16_pw<KK, NN, WT_, G_, R_, A_>), dim3(grid), dim3(256), 0, s, X, w, bias, gate, R, Y, ntiles, rows_per_image, \
                     nullptr, nullptr, nullptr, 1, (int64_t)0)
#define MVAE_PW(KK, NN, WT_)                                                                                          \
  if (K == KK && N == NN && transposed == WT_) {                                                                      \
    if (act != ACT_NONE && (act != ACT_RELU || gate || residual)) return false;                                       \
    if (gate && residual) MVAE_PW1(KK, NN, WT_, true, true, ACT_NONE);                                                \
    else if (gate) MVAE_PW1(KK, NN, WT_, true, false, ACT_NONE);                                                      \
    else if (residual) MVAE_PW1(KK, NN, WT_, false, true, ACT_NONE);                                                  \
    else if (act == ACT_RELU) MVAE_PW1(KK, NN, WT_, false, false, ACT_RELU);                                          \
    else MVAE_PW1(KK, NN, WT_, false, false, ACT_NONE);                                                               \
    return true;                                                                                                      \
  }
  MVAE_PW(64, 64, false) MVAE_PW(32, 32, false) MVAE_PW(64, 32, false) MVAE_PW(32, 64, false)
  MVAE_PW(64, 64, true) MVAE_PW(32, 32, true) MVAE_PW(64, 32, true) MVAE_PW(32, 64, true)
#undef MVAE_PW1
#undef MVAE_PW
  return false;
}

// conv2 of a block chained with conv0 of the next one (k16_pw_chain; both 64 -> 64); with wm: a 1x1 convolution 64 -> 32
// (mid_transposed: the Conv2DTranspose form) sits between them and the next block is 32 wide.  false = not covered.
bool launch16_pw_chain(const void* in, const float* w, const float* bias, const float* gate, const void* residual, void* out,
                       const float* wm, const float* biasm, void* outm, bool mid_transposed,
                       const float* w2, const float* bias2, void* out2, int64_t M, int64_t rows_per_image, int C,
                       hipStream_t s) {
  static const bool on = [] { const char* e = getenv("MVAE_FUSE_PW_CHAIN"); return e ? atoi(e) != 0 : true; }();
  if (!on || C != 64 || M % 32 != 0 || M <= 0 || rows_per_image % 32 != 0 || !gate || !residual) return false;
  const int64_t ntiles = M / 32;
  const int64_t nblk = (ntiles + 3) / 4;
  const int grid = (int)(nblk < 4 * cus16() ? nblk : 4 * cus16());
  ProfScope ps(wm ? "k16_pw_chain3" : "k16_pw_chain", 2.0 * M * (wm ? 3 * C + 64 : 4 * C), (wm ? 2.0 * C * C + 2.0 * C * 32 + 2.0 * 32 * 32 : 4.0 * C * C) * M, s);
#define MVAE_CH(MID_, WT_)                                                                                             \
  hipLaunchKernelGGL((k16_pw_chain<MID_, WT_>), dim3(grid), dim3(256), 0, s, (const bf16_t*)in, w, bias, gate,         \
                     (const bf16_t*)residual, (bf16_t*)out, wm, biasm, (bf16_t*)outm, w2, bias2, (bf16_t*)out2, ntiles, \
                     rows_per_image)
  if (!wm) MVAE_CH(false, false);
  else if (mid_transposed) MVAE_CH(true, true);
  else MVAE_CH(true, false);
#undef MVAE_CH
  return true;
}

// MobileNetV3 backward pair (see k16_dual).  false = shape not covered.
bool launch16_dual(const void* X, const float* W, const void* aux, const float* gate, const void* residual, void* Y,
                   float* dW, float* db, float* dot_out, int64_t M, int64_t rows_per_image, int C, GradSlots sl,
                   hipStream_t s, bool embed_mask) {
  if (M % 32 != 0 || rows_per_image % 32 != 0) return false;
  const int mode = (gate && dot_out && !residual) ? 1 : ((residual && !gate && !dot_out) ? 2 : 0);
  if (!mode || (C != 32 && C != 64)) return false;
  const int64_t ntiles = M / 32, tpi = rows_per_image / 32;
  // a wave takes a contiguous run of tiles inside ONE image: the largest divisor-by-halving of tiles_per_image that still
  // leaves >= 8 waves per CU's worth of runs (2 register-limited blocks per CU, several rounds)
  int64_t tpw = tpi;
  while (tpw % 2 == 0 && ntiles / tpw < 16 * cus16()) tpw /= 2;
  while (tpw % 2 == 0 && tpw > 64) tpw /= 2;
  const int grid = (int)((ntiles / tpw + 3) / 4);
  ProfScope ps(mode == 1 ? (C == 64 ? "k16_dual<64,1>" : "k16_dual<32,1>") : (C == 64 ? "k16_dual<64,2>" : "k16_dual<32,2>"),
               2.0 * M * C * (mode == 1 ? 3 : 4), 4.0 * M * C * C, s);
#define MVAE_D16(CC, MM)                                                                                              \
  hipLaunchKernelGGL((k16_dual<CC, MM>), dim3(grid), dim3(256), 0, s, (const bf16_t*)X, W, (const bf16_t*)aux, gate,  \
                     (const bf16_t*)residual, (bf16_t*)Y, sl.at(dW), sl.at(db), dot_out, ntiles, tpw, tpi, sl.count(), sl.stride, \
                     embed_mask ? 1 : 0)
  if (C == 64) { if (mode == 1) MVAE_D16(64, 1); else MVAE_D16(64, 2); }
  else { if (mode == 1) MVAE_D16(32, 1); else MVAE_D16(32, 2); }
#undef MVAE_D16
  return true;
}

// =================================================================================================
// Depthwise backward (k_dw_bwd_ring<true, bf16_t>, kernels_dw.hip) and conv0's backward pair (k16_dual MODE 2) in ONE
// pass: dt0, the depthwise backward's output, never goes to HBM (5 tensor passes per block element instead of 7, and
// the ring phase is VALU-bound in bf16, so the two extra streams ride under it).  C = 64, W % 32 == 0, H even, the ReLU
// mask of t1 in the bf16 LSB of dt2.  A strip is 32 columns: after the ring step of image row y the strip's dt0 row is a
// 32 x 64 bf16 tile in LDS; every two rows the four waves run, on tiles (y, y + 1) and the block-input / dout tiles
// fetched beside them,
//     da^T[co tile][32 px] = W0^T . dt0^T (+ dout)     wave (kk = w & 1, nt = w >> 1): 4 MFMAs, 16-byte stores
//     P[co][ci] += dt0^T a over the 64 px               wave (nt = w & 1, kt = w >> 1): 4 MFMAs from ds_read_b64_tr_b16
// The finished da tile is stored one step late, behind the next pair's tile loads (vmcnt is one in-order queue).
// =================================================================================================
template <bool HALO>
__global__ void __launch_bounds__(256, 2) k16_dw_bwd_conv0(const uint2* __restrict__ dt2, const uint2* __restrict__ t0,
                                                           const f32x4* __restrict__ w, const f32x4* __restrict__ gate,
                                                           const f32x4* __restrict__ dgap, const float* __restrict__ W0,
                                                           const bf16_t* __restrict__ a_in, const bf16_t* __restrict__ dout,
                                                           bf16_t* __restrict__ da, float* __restrict__ dW,
                                                           float* __restrict__ db, float* __restrict__ dW0,
                                                           float* __restrict__ db0, int H, int W, int strips, float inv_hw,
                                                           int B, int RS, int nseg, int nslots, int64_t slot_stride) {
  constexpr int XSP = 34, RINGB = 4 * XSP * 16 * 16, TB = 32 * 64 * 2;
  extern __shared__ __attribute__((aligned(16))) char lds16[];
  f32x4* ring = reinterpret_cast<f32x4*>(lds16);                       // 4 x 34 x 16 float4 of d1
  char* tD = lds16 + RINGB;                                            // dt0 tiles of rows y2, y2 + 1
  char* tA = tD + 2 * TB;                                              // block input tiles
  char* tR = tA + 2 * TB;                                              // dout tiles
  u32x4* wfl = reinterpret_cast<u32x4*>(tR + 2 * TB);                  // W0 fragments, one 16-byte slot per (fragment, lane)
  const int x0 = blockIdx.x * 32;
  const int c4 = threadIdx.x & 15;
  const int xl0 = threadIdx.x >> 4, xl1 = xl0 + 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  if (wave == 0) {
    bf16x8 wf[2][4];
    load_wfrags<64, 64, true>(W0, r, h, wf);                           // Wm[k = co][n = ci] = W0[ci*64 + co]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wfl[(nt * 4 + kk) * 64 + lane] = __builtin_bit_cast(u32x4, wf[nt][kk]);
  }
#define RING16(slot, xs, c4_) ring[((slot) * XSP + (xs)) * 16 + (c4_)]
  if (!HALO && threadIdx.x < 128) {                                    // columns 0 and 33 of the four slots: always zero
    const int slot = threadIdx.x >> 5, side = (threadIdx.x >> 4) & 1;
    RING16(slot, side ? 33 : 0, c4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 wt[9], aw[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) { wt[k] = w[k * 16 + c4]; aw[k] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  f32x4 ab = {0.f, 0.f, 0.f, 0.f};
  f32x16 accw = zero16();                                              // P tile: co = 32 (wave & 1) + .., ci = 32 (wave >> 1) + r
  float bsum = 0.f;
  const int pnt = wave & 1, pkt = wave >> 1;                           // weight-gradient role
  const int ykk = wave & 1, ynt = wave >> 1;                           // data-GEMM role: tile row, output-channel tile

  for (int item = blockIdx.y; item < B * nseg; item += gridDim.y) {
    const int b = item / nseg, seg = item % nseg;
    const int ya = seg * RS, yb = min(H, ya + RS);
    const int64_t ioff = (int64_t)b * H * W * 16;                      // 4-element offset of the image
    constexpr int NU = HALO ? 3 : 2;
    uint2 Fd[2][NU], T[2][2];
    const f32x4 gg_c = gate[(int64_t)b * 16 + c4];
    const f32x4 dg_c = dgap[(int64_t)b * 16 + c4] * inv_hw;
    const int ybc = min(yb, H - 1);
    auto fetch_row = [&](int y, uint2 (&rd)[NU]) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int t = threadIdx.x + 256 * u;
        const int x = HALO ? x0 - 1 + (t >> 4) : (u == 0 ? xl0 : xl1);
        const bool ok = HALO ? (t < XSP * 16 && x >= 0 && x < W) : true;
        rd[u] = dt2[ok ? ioff + ((int64_t)y * W + x) * 16 + c4 : 0];
      }
    };
    auto store_row = [&](int y, const uint2 (&rdr)[NU]) {              // d1 row y (+ halo) -> ring slot y & 3
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int t = HALO ? threadIdx.x + 256 * u : ((u == 0 ? xl0 : xl1) + 1) * 16 + c4;
        const int x = x0 - 1 + (t >> 4);
        const bool ok = HALO ? (x >= 0 && x < W) : true;
        if (!HALO || t < XSP * 16) {
          const f32x4 rd = unpack4(rdr[u]);
          f32x4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned bits = __float_as_uint(rd[q]);
            const bool on = (bits & 0x10000u) != 0u;                   // ReLU mask t1 > 0 in the bf16 LSB, cleared for the value
            v[q] = (ok && on) ? __uint_as_float(bits & ~0x10000u) * gg_c[q] + dg_c[q] : 0.f;
          }
          RING16(y & 3, t >> 4, c4) = v;
        }
      }
    };
    auto fetch_t0 = [&](int y, uint2 (&tv)[2]) {
      tv[0] = t0[ioff + ((int64_t)y * W + x0 + xl0) * 16 + c4];
      tv[1] = t0[ioff + ((int64_t)y * W + x0 + xl1) * 16 + c4];
    };
    __syncthreads();                                  // previous item's ring / tile reads are done (and wfl is written)
    fetch_row(max(ya - 1, 0), Fd[0]);
    fetch_row(ya, Fd[1]);
    fetch_t0(ya, T[0]);
    fetch_t0(min(ya + 1, yb - 1), T[1]);
    store_row(ya - 1, Fd[0]);
    store_row(ya, Fd[1]);
    fetch_row(min(ya + 1, ybc), Fd[1]);
    fetch_row(min(ya + 2, ybc), Fd[0]);
    u32x4 out0 = {0u, 0u, 0u, 0u}, out1 = out0;       // the finished da tile of the previous pair (this lane's 2 x 16 bytes)
    int64_t out_off = 0;
    bool pending = false;
    for (int y2 = ya; y2 < yb; y2 += 2) {
      // block input and dout of rows (y2, y2 + 1), columns x0 .. x0 + 31: four 4 KB runs, thread t takes 16-byte chunk t of
      // each (chunk = pixel * 8 + channel chunk)
      const int64_t px0 = (int64_t)b * H * W + (int64_t)y2 * W + x0;   // pixel index of (y2, x0)
      const u32x4 la0 = reinterpret_cast<const u32x4*>(a_in + px0 * 64)[threadIdx.x];
      const u32x4 la1 = reinterpret_cast<const u32x4*>(a_in + (px0 + W) * 64)[threadIdx.x];
      const u32x4 lr0 = reinterpret_cast<const u32x4*>(dout + px0 * 64)[threadIdx.x];
      const u32x4 lr1 = reinterpret_cast<const u32x4*>(dout + (px0 + W) * 64)[threadIdx.x];
      if (pending) {
        *reinterpret_cast<u32x4*>(da + out_off) = out0;
        *reinterpret_cast<u32x4*>(da + out_off + 16) = out1;
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int y = y2 + kk;
        store_row(y + 1, Fd[(kk + 1) & 1]);
        fetch_row(min(y + 3, ybc), Fd[(kk + 1) & 1]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int xl = k == 0 ? xl0 : xl1;
          const f32x4 tvk = unpack4(T[kk][k]);
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const int yy = y - a + 1;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int e = 0; e < 3; ++e) {
              const f32x4 sv = RING16(yy & 3, xl + 2 - e, c4);
              acc += wt[a * 3 + e] * sv;
              aw[a * 3 + e] += tvk * sv;
            }
          }
          ab += RING16(y & 3, xl + 1, c4);
          f32x4 rv;
#pragma unroll
          for (int q = 0; q < 4; ++q) rv[q] = tvk[q] > 0.f ? acc[q] : 0.f;
          *reinterpret_cast<uint2*>(tD + kk * TB + tile_off<64>(xl, c4 >> 1) + (c4 & 1) * 8) = pack4(rv);
        }
        fetch_t0(min(y + 2, yb - 1), T[kk]);
      }
      {
        const int row = threadIdx.x >> 3, ch = threadIdx.x & 7;
        *reinterpret_cast<u32x4*>(tA + tile_off<64>(row, ch)) = la0;
        *reinterpret_cast<u32x4*>(tA + TB + tile_off<64>(row, ch)) = la1;
        *reinterpret_cast<u32x4*>(tR + tile_off<64>(row, ch)) = lr0;
        *reinterpret_cast<u32x4*>(tR + TB + tile_off<64>(row, ch)) = lr1;
      }
      __syncthreads();                                // all six tiles complete
      // ---- P[co][ci] += dt0^T a over both rows
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
          const bf16x8 fa = frag_cols<64>(tD + kk * TB, lane, pnt, sidx);
          const bf16x8 fb = frag_cols<64>(tA + kk * TB, lane, pkt, sidx);
          if (pkt == 0) bsum = frag_sum(fa, bsum);
          accw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, accw, 0, 0, 0);
        }
      // ---- da tile: row y2 + ykk, output channels 32 ynt ..
      f32x16 acc = zero16();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(wfl[(ynt * 4 + kk) * 64 + lane]),
                                                      frag_rows<64>(tD + ykk * TB, r, h, kk), acc, 0, 0, 0);
      uint2 pk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = ynt * 32 + 8 * q + 4 * h;
        f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        v += unpack4(*reinterpret_cast<const uint2*>(tR + ykk * TB + tile_off<64>(r, c0 >> 3) + (c0 & 7) * 2));
        pk[q] = pack4(v);
      }
      {
        auto s0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
        out0 = u32x4{s0[0], s1[0], s0[1], s1[1]};
        auto s2 = __builtin_amdgcn_permlane32_swap(pk[2].x, pk[3].x, false, false);
        auto s3 = __builtin_amdgcn_permlane32_swap(pk[2].y, pk[3].y, false, false);
        out1 = u32x4{s2[0], s3[0], s2[1], s3[1]};
      }
      out_off = (px0 + (int64_t)ykk * W + r) * 64 + ynt * 32 + 8 * h;
      pending = true;
    }
    if (pending) {
      *reinterpret_cast<u32x4*>(da + out_off) = out0;
      *reinterpret_cast<u32x4*>(da + out_off + 16) = out1;
    }
  }
  // ---- depthwise weight / bias gradients: block reduction as in k_dw_bwd_ring
  __syncthreads();
  f32x4* red = ring;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    f32x4 v = k < 9 ? aw[k < 9 ? k : 0] : ab;
    for (int off = 16; off < 64; off <<= 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] += __shfl_xor(v[q], off, 64);
    }
    if (lane < 16) red[(wave * 10 + k) * 16 + lane] = v;
  }
  __syncthreads();
  const int64_t slot = (int64_t)((blockIdx.y * gridDim.x + blockIdx.x) % nslots) * slot_stride;
  for (int idx = threadIdx.x; idx < 10 * 16 * 4; idx += 256) {
    const int q = idx & 3, cc = (idx >> 2) % 16, k = (idx >> 2) / 16;
    float t = 0.f;
    for (int wv = 0; wv < 4; ++wv) t += red[(wv * 10 + k) * 16 + cc][q];
    atomicAdd((k < 9 ? dW + slot + (int64_t)k * 64 : db + slot) + cc * 4 + q, t);
  }
  // ---- conv0 weight gradient: the four P tiles through LDS (dW0[ci][co] row-major), one coalesced atomic set per block
  __syncthreads();
  float* redw = reinterpret_cast<float*>(lds16);
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int co = pnt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h, ci = pkt * 32 + r;
    redw[ci * 64 + co] = accw[reg];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) atomicAdd(&dW0[slot + idx], redw[idx]);
  bsum += __shfl_xor(bsum, 32, 64);
  if (pkt == 0 && h == 0 && db0 != nullptr) atomicAdd(&db0[slot + pnt * 32 + r], bsum);
#undef RING16
}

// dt2 (bf16, ReLU mask in the LSB) -> da, dW_dw, db_dw, dW0, db0.  false = shape not covered (the caller then runs
// launch_dw_bwd_fused and launch16_dual).
bool launch16_dw_bwd_conv0(const void* dt2, const void* t0, const float* w, const float* gate, const float* dgap,
                           const float* W0, const void* a_in, const void* dout, void* da, float* dW, float* db, float* dW0,
                           float* db0, GradSlots sl, int B, int H, int W, int C, hipStream_t s) {
  static const bool on = [] { const char* e = getenv("MVAE_FUSE_DW_CONV0"); return e ? atoi(e) != 0 : true; }();
  if (!on || C != 64 || W % 32 != 0 || H % 2 != 0 || H < 4) return false;
  const int strips = W / 32;
  int nseg = 1;
  while ((int64_t)B * strips * nseg < 512 && (H / (nseg * 2)) % 2 == 0 && H / (nseg * 2) >= 4) nseg *= 2;
  const int RS = H / nseg;
  const int64_t work = (int64_t)B * nseg;
  const int gy = (int)(work < kDwMaxBlocks / strips ? work : kDwMaxBlocks / strips);
  if (gy < 1) return false;
  const size_t lds = (size_t)4 * 34 * 16 * 16 + 6 * 4096 + 8 * 64 * 16;
  static const bool attr = [] {
    return hipFuncSetAttribute((const void*)k16_dw_bwd_conv0<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess &&
           hipFuncSetAttribute((const void*)k16_dw_bwd_conv0<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
  }();
  if (!attr) return false;
  const double M = (double)B * H * W;
  ProfScope ps("k16_dw_bwd_conv0", 2.0 * M * C * 5, (40.0 + 4.0 * C) * M * C, s);
#define MVAE_DWC16(HALO)                                                                                               \
  hipLaunchKernelGGL(k16_dw_bwd_conv0<HALO>, dim3(strips, gy), dim3(256), lds, s, (const uint2*)dt2, (const uint2*)t0, \
                     (const f32x4*)w, (const f32x4*)gate, (const f32x4*)dgap, W0, (const bf16_t*)a_in,                 \
                     (const bf16_t*)dout, (bf16_t*)da, sl.at(dW), sl.at(db), sl.at(dW0), sl.at(db0), H, W, strips,     \
                     1.0f / (float)(H * W), B, RS, nseg, sl.count(), sl.stride)
  if (strips > 1) MVAE_DWC16(true); else MVAE_DWC16(false);
#undef MVAE_DWC16
  return true;
}

// k x k convolution (F-form: in = big) / transposed convolution (T-form: in = small).  false = shape not covered.
bool launch16_taps(bool transposed, const void* in, const float* w, const float* bias, void* out, const ConvGeom& g,
                   hipStream_t s, const float* w2, const float* bias2, void* out2, bool* chained) {
  if (chained) *chained = false;
  const int KC = transposed ? g.CO : g.CI, NC = transposed ? g.CI : g.CO;
  if ((int64_t)g.B * g.IH * g.IW * g.CI * 2 >= (1LL << 31) || (int64_t)g.B * g.OH * g.OW * g.CO * 2 >= (1LL << 31)) return false;
  if (g.KH > 8 || g.KW > 8 || g.KH * g.KW > 25) return false;
  if (!((KC == 32 && NC == 64) || (KC == 64 && NC == 32))) return false;
  int64_t Mc;
  int classes = 1;
  if (transposed) { classes = g.SH * g.SW; Mc = (int64_t)g.B * ((g.IH + g.SH - 1) / g.SH) * ((g.IW + g.SW - 1) / g.SW); }
  else Mc = (int64_t)g.B * g.OH * g.OW;
  const unsigned in_bytes = (unsigned)((int64_t)g.B * (transposed ? g.OH * g.OW : g.IH * g.IW) * KC * 2);
  const int64_t tiles = (Mc + 31) / 32;
  // The tap loop is a chain of (gather -> LDS -> MFMA) steps fed from L2 with one tap of prefetch: what hides the ~1 us
  // of each gather is the number of waves per CU.  T-form phases stage at most 9 tap slices (36 KB): two 512-thread
  // blocks per CU; F-form stages all 25 (100 KB): one block of 16 waves (KC = 32) or 12 (KC = 64: 4 KB tiles).
  const int kWaves = transposed ? 8 : (KC == 32 ? 16 : 12);
  const int per_cu = transposed ? 2 : 1;
  int64_t waves = (int64_t)kWaves * per_cu * cus16() / classes;
  if (waves < kWaves) waves = kWaves;
  if (waves > tiles) waves = tiles;
  const int tpw = (int)((tiles + waves - 1) / waves);
  const unsigned gx = (unsigned)(((tiles + tpw - 1) / tpw + kWaves - 1) / kWaves);
  ProfScope ps(transposed ? "k16_taps<T>" : "k16_taps<F>", 2.0 * ((double)g.B * g.IH * g.IW * g.CI + (double)g.B * g.OH * g.OW * g.CO),
               2.0 * g.B * g.OH * g.OW * g.CO * g.KH * g.KW * g.CI, s);
#define MVAE_T16(A, B_, TF, MT, NW)                                                                                   \
  hipLaunchKernelGGL((k16_taps<A, B_, TF, MT, 64 * NW>), dim3(gx, classes), dim3(64 * NW), 0, s,                      \
                     (const bf16_t*)in, w, bias, (bf16_t*)out, g, in_bytes, tpw, nullptr, nullptr, nullptr)
  if (transposed && (((g.KH + g.SH - 1) / g.SH) * ((g.KW + g.SW - 1) / g.SW) > 9)) return false;   // taps per phase
  // the four sub-pixel phases merged in one block (k16_taps_tm): 5 x 5, stride 2, even sizes, SAME padding
  static const bool merged_on = [] { const char* e = getenv("MVAE_CONVT_MERGED16"); return e ? atoi(e) != 0 : true; }();
  if (transposed && merged_on && g.KH == 5 && g.KW == 5 && g.SH == 2 && g.SW == 2 && g.PT == 1 && g.PL == 1 && g.IH == 2 * g.OH &&
      g.IW == 2 * g.OW) {
    static const bool chain_m = [] { const char* e = getenv("MVAE_FUSE_PW_CHAIN"); return e ? atoi(e) != 0 : true; }();
    const int64_t Mm = (int64_t)g.B * g.OH * g.OW, tiles_m = (Mm + 31) / 32;
    int64_t wv = (int64_t)8 * cus16();                                 // one 8-wave block per CU (100 KB of weight slices)
    if (wv > tiles_m) wv = tiles_m;
    const int tpw_m = (int)((tiles_m + wv - 1) / wv);
    const unsigned gxm = (unsigned)(((tiles_m + tpw_m - 1) / tpw_m + 7) / 8);
    // MVAE_TM_FLAGS16 (diagnostic bit mask): 8 = plain row-major tile order and per-lane 16-byte stores (the round-4 first form)
    static const int dbg = [] { const char* e = getenv("MVAE_TM_FLAGS16"); return e ? atoi(e) : 0; }();
    const bool chain_here = KC == 32 && NC == 64 && w2 && out2 && chain_m;
    // KC = 32 and whole 32-position tiles per row: the tile's 3 x 34 input pixels staged once, one tile ahead (ROWS)
    static const bool rows_t = [] { const char* e = getenv("MVAE_TM_ROWS16"); return e ? atoi(e) != 0 : true; }();
    const bool rows_here = rows_t && KC == 32 && g.OW % 32 == 0 && !(dbg & 8);
#define MVAE_TM(A, B_, CH_, RW_)                                                                                       \
  hipLaunchKernelGGL((k16_taps_tm<A, B_, 512, CH_, RW_>), dim3(gxm), dim3(512), 0, s, (const bf16_t*)in, w, bias, (bf16_t*)out, g, \
                     in_bytes, tpw_m, CH_ ? w2 : nullptr, CH_ ? bias2 : nullptr, CH_ ? (bf16_t*)out2 : nullptr, dbg)
    if (chain_here) { if (rows_here) MVAE_TM(32, 64, true, true); else MVAE_TM(32, 64, true, false); }
    else if (KC == 32 && NC == 64) { if (rows_here) MVAE_TM(32, 64, false, true); else MVAE_TM(32, 64, false, false); }
    else MVAE_TM(64, 32, false, false);
#undef MVAE_TM
    if (chained) *chained = chain_here;
    return true;
  }
  // F-form with the input rows of a kernel row staged in LDS (k16_taps_fr): 5 x 5, stride 2, output width 4 / 8 / 16 / 32k
  static const bool rows_on = [] { const char* e = getenv("MVAE_CONVF_ROWS16"); return e ? atoi(e) != 0 : true; }();
  if (!transposed && rows_on && g.KH == 5 && g.KW == 5 && g.SH == 2 && g.SW == 2 &&
      (g.OW % 32 == 0 || g.OW == 16 || g.OW == 8 || g.OW == 4)) {
    const int cols = g.OW >= 32 ? 32 : g.OW;
    bool ok = false;
#define MVAE_FR(A, B_, NWV) (cols == 32 ? launch16_taps_fr<A, B_, 32, NWV>(in, w, bias, out, g, in_bytes, tiles, s)        \
                             : cols == 16 ? launch16_taps_fr<A, B_, 16, NWV>(in, w, bias, out, g, in_bytes, tiles, s)      \
                             : cols == 8 ? launch16_taps_fr<A, B_, 8, NWV>(in, w, bias, out, g, in_bytes, tiles, s)        \
                                         : launch16_taps_fr<A, B_, 4, NWV>(in, w, bias, out, g, in_bytes, tiles, s))
    if (KC == 32) ok = MVAE_FR(32, 64, 10); else ok = MVAE_FR(64, 32, 5);
#undef MVAE_FR
    if (ok) return true;
  }
  static const bool chain_on = [] { const char* e = getenv("MVAE_FUSE_PW_CHAIN"); return e ? atoi(e) != 0 : true; }();
  if (KC == 32 && NC == 64 && transposed && w2 && out2 && chain_on) {
    hipLaunchKernelGGL((k16_taps<32, 64, true, 9, 512, true>), dim3(gx, classes), dim3(512), 0, s, (const bf16_t*)in, w, bias,
                       (bf16_t*)out, g, in_bytes, tpw, w2, bias2, (bf16_t*)out2);
    if (chained) *chained = true;
    return true;
  }
  if (KC == 32 && NC == 64) { if (transposed) MVAE_T16(32, 64, true, 9, 8); else MVAE_T16(32, 64, false, 25, 16); }
  else { if (transposed) MVAE_T16(64, 32, true, 9, 8); else MVAE_T16(64, 32, false, 25, 12); }
#undef MVAE_T16
  return true;
}

// convolution weight (+ bias) gradient.  false = shape not covered.
bool launch16_wgrad(const void* big, const void* small, float* dW, float* db, const ConvGeom& g, GradSlots sl, hipStream_t s,
                    float* db_big, bool* db_big_done) {
  if (db_big_done) *db_big_done = false;
  const GradSlots sl_in = sl;
  const int64_t M = (int64_t)g.B * g.OH * g.OW;
  if (M * g.CO * 2 >= (1ll << 31) || (int64_t)g.B * g.IH * g.IW * g.CI * 2 >= (1ll << 31)) return false;
  if (!((g.CI == 32 && g.CO == 64) || (g.CI == 64 && g.CO == 32))) return false;
  const bool pointwise = g.KH == 1 && g.KW == 1 && g.SH == 1 && g.SW == 1;
  if (!pointwise && g.KW != 5) return false;
  int64_t chunks = 8 * (64 / g.KH);
  int64_t rpb = (M + chunks - 1) / chunks;
  rpb = (rpb + 127) / 128 * 128;
  if (rpb < 128) rpb = 128;
  chunks = (M + rpb - 1) / rpb;
  const unsigned groups = (unsigned)((chunks + 7) / 8);
  if (!pointwise) sl = GradSlots();                            // only small gradients go through the slots
  ProfScope ps(pointwise ? "k16_wgrad<1x1>" : "k16_wgrad<5x5>", 2.0 * ((double)g.B * g.IH * g.IW * g.CI + (double)M * g.CO),
               2.0 * M * g.CO * g.KH * g.KW * g.CI, s);
#define MVAE_W16(A, B_, TG)                                                                                           \
  hipLaunchKernelGGL((k16_wgrad<A, B_, TG>), dim3(groups * 8u * g.KH), dim3(256), 0, s, (const bf16_t*)big,           \
                     (const bf16_t*)small, sl.at(dW), sl.at(db), g, M, rpb, sl.count(), sl.stride)
  // 5 x 5, stride 2, SAME, output rows of whole 32-pixel tiles: the input row segment staged once per tile (k16_wgrad_seg)
  static const bool seg_on = [] { const char* e = getenv("MVAE_WGRAD_SEG16"); return e ? atoi(e) != 0 : true; }();
  if (!pointwise && seg_on && g.KH == 5 && g.SH == 2 && g.SW == 2 && g.OW % 32 == 0 && rpb % 32 == 0) {
    // the bias gradient of a Conv2DTranspose (column sums of `big`) rides along: MVAE_WGRAD_DBIG16=0 leaves it to k_colstat4
    static const bool dbig_on = [] { const char* e = getenv("MVAE_WGRAD_DBIG16"); return e ? atoi(e) != 0 : true; }();
    float* dbg = (dbig_on && db_big && db_big_done && !det_mode() && g.PT == 1 && g.PL == 1 && g.IH == 2 * g.OH && g.IW == 2 * g.OW)
                     ? sl_in.at(db_big) : nullptr;
    if (g.CI == 32) hipLaunchKernelGGL((k16_wgrad_seg<32, 64>), dim3(groups * 8u * g.KH), dim3(256), 0, s, (const bf16_t*)big,
                                       (const bf16_t*)small, dW, db, g, M, rpb, dbg, sl_in.count(), sl_in.stride);
    else hipLaunchKernelGGL((k16_wgrad_seg<64, 32>), dim3(groups * 8u * g.KH), dim3(256), 0, s, (const bf16_t*)big,
                            (const bf16_t*)small, dW, db, g, M, rpb, dbg, sl_in.count(), sl_in.stride);
    if (dbg) *db_big_done = true;
    return true;
  }
  if (g.CI == 32) { if (pointwise) MVAE_W16(32, 64, 1); else MVAE_W16(32, 64, 5); }
  else { if (pointwise) MVAE_W16(64, 32, 1); else MVAE_W16(64, 32, 5); }
#undef MVAE_W16
  return true;
}

}  // namespace mvae
