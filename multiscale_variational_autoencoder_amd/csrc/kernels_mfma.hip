// kernels_mfma.hip -- LDS-tiled, f32-MFMA kernels for the wide-channel convolutions of the multiscale VAE
// (gfx950 / CDNA4: wave = 64 lanes, v_mfma_f32_32x32x2_f32 = exact fp32 at the vector rate, 160 KiB LDS / CU).
//
// Every conv on the hot path is a tall-skinny GEMM: M = B*H*W rows (10^5..10^6), N and K are 32 or 64 channels
// (x taps).  One wave owns 32 rows x all N columns, the 32-row activation tile is staged through LDS with fully
// coalesced 16-byte loads, the (tiny) weight matrix lives in LDS for the whole kernel, and everything that
// the reference materialises as separate Keras layers around the matmul (bias, ReLU/ELU, squeeze-excite gate,
// folded BatchNorm, residual add, bias gradient) is fused into the load / store of the tile.
//
// MFMA operand maps (cdna_hip_programming.md section 3):  v_mfma_f32_32x32x2_f32, lane l:
//   A[i = l & 31][k = l >> 5],  B[k = l >> 5][j = l & 31],  D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]
// The k index is a dummy summation index, so lane-half h = l >> 5 may take ANY half of the k range as long as the
// A and B operands agree: here half h owns k in [h*K/2, (h+1)*K/2), which makes each lane's A fragment a
// contiguous run of floats (16-byte LDS reads).
#include "kernels.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace mvae {

// Blocks of the persistent (grid-stride) kernels: two register-limited workgroups per CU on `big_grid_cus()` CUs.
// Fewer than all 256 CUs leaves slots where the short kernels of the other pyramid scales (own HIP streams) can start
// at once instead of waiting for a persistent block to retire.
// 224 of the 256 CUs by default: with the pyramid scales on their own streams the headline step measured 5.55 ms at 256,
// 5.48 at 224, 5.51 at 208, 5.59 at 192 (same box, back to back; MVAE_BIG_CUS overrides).
static int big_grid_cus() {
  static const int v = [] { const char* e = getenv("MVAE_BIG_CUS"); int n = e ? atoi(e) : 224; return n < 8 ? 8 : (n > 256 ? 256 : n); }();
  return v;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// image index of a row: rows_per_image is a power of two for every feature map of the reference's configs -> a shift.
// (As `row / rows_per_image` on int64 the tile loops carried a ~100-instruction software division per use -- VALU work
// that fp32 MFMAs do not overlap with, see k_conv_taps.)   Rows < 2^31 (launchers).
struct ImgOf {
  uint32_t rpi;
  int sh;
  __device__ explicit ImgOf(int64_t rows_per_image)
      : rpi((uint32_t)rows_per_image),
        sh((rows_per_image & (rows_per_image - 1)) == 0 ? 63 - __builtin_clzll((unsigned long long)rows_per_image) : -1) {}
  __device__ uint32_t operator()(int64_t row) const { return sh >= 0 ? (uint32_t)row >> sh : (uint32_t)row / rpi; }
};

// A wave's LDS instructions execute in issue order, so data one lane wrote is visible to a later ds_read of any
// lane of the SAME wave; only the compiler has to be kept from reordering across the hand-off.
#define WAVE_LDS_SYNC()                                      \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)

__device__ __forceinline__ float act_apply_m(float v, int act) {
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == ACT_ELU) return v > 0.f ? v : expm1f(v);
  return v;
}

// =================================================================================================
// Y[M,N] = act( pre(X)[M,K] . Wm[K,N] + bias ) + residual        (1x1 convolution, stride 1)
//   WT = false: Wm[k][n] = W[k*N + n]   (Conv2D forward, kernel [1,1,CI=K,CO=N])
//   WT = true : Wm[k][n] = W[n*K + k]   (its backward-data / Conv2DTranspose forward: kernel [1,1,CI=N,CO=K])
// block = 4 waves; wave w owns rows [tile*128 + 32w, +32).  Persistent over tiles.
// =================================================================================================
// FAST = (M % 128 == 0 and no fused dot): the tile loop is then straight-line code -- no row guards, every load
// unconditional (clamped / dummy sources + selects), no conditional VMEM operation -- so hipcc can COUNT the outstanding
// memory operations and waits for the prefetched tile with s_waitcnt vmcnt(N > 0).  With any branch around a load,
// store or atomic it falls back to vmcnt(0) at the top of the loop, i.e. every tile also waited for the previous
// tile's STORES to be acknowledged (stores share the in-order vmcnt counter on gfx9).
// D2 = FAST and no residual: two tiles of prefetch (a residual launch already keeps two streams in flight, and its
// registers do not leave room for a second tile buffer)
template <int K, int N, bool WT, bool FAST, bool D2>
__global__ void __launch_bounds__(256, 2) k_gemm_rows(const float* __restrict__ X, const float* __restrict__ W,
                                                   const float* __restrict__ bias, const float* __restrict__ residual,
                                                   float* __restrict__ Y, int64_t M, int64_t rows_per_image, PreOp pre,
                                                   int act, const float* __restrict__ dot_src,
                                                   float* __restrict__ dot_out) {
  // optional fused reduction (squeeze-excite gate gradient): dot_out[b, n] += sum_{rows of image b} Y[m,n]*dot_src[m,n]
  constexpr int KH = K / 2;                    // k's per lane half
  constexpr int TS = (K > N ? K : N) + 4;      // padded LDS row stride (floats) of the A tile and of the C tile
  constexpr int NT = N / 32;
  constexpr int C4 = K / 4, N4 = N / 4;        // float4 chunks per input / output row
  constexpr int LD = K / 8, ST = N / 8;        // float4 loads / stores per lane per 32-row tile
  constexpr int RPL = 64 / C4, RPS = 64 / N4;  // rows covered by one wave-wide float4 load / store
  __shared__ __attribute__((aligned(16))) float lds[4 * 32 * TS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* sT = lds + wave * 32 * TS;            // this wave's tile: A operand first, C result afterwards
  const int i = lane & 31, h = lane >> 5;
  // The B operand of every MFMA is a WEIGHT: Wm[k = h*KH + t][n = nt*32 + i] is the same for every tile, so each lane
  // keeps its KH x NT weights in registers for the whole kernel (64 VGPRs at 64x64) -- no LDS traffic, no waits in
  // the MFMA phase.
  float breg[KH][NT];
#pragma unroll
  for (int t = 0; t < KH; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int k = h * KH + t, n = nt * 32 + i;
      breg[t][nt] = WT ? W[(int64_t)n * K + k] : W[(int64_t)k * N + n];
    }
  const int lc4 = lane % C4, lr = lane / C4;   // load mapping: row = j*RPL + lr, chunk lc4
  const int sc4 = lane % N4, sr = lane / N4;   // store mapping
  const int64_t ntiles = (M + 127) / 128;
  const f32x4* X4 = reinterpret_cast<const f32x4*>(X);
  const f32x4* R4 = reinterpret_cast<const f32x4*>(residual);
  const f32x4* D4 = reinterpret_cast<const f32x4*>(dot_src);
  f32x4* Y4 = reinterpret_cast<f32x4*>(Y);
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (bias) bias4 = reinterpret_cast<const f32x4*>(bias)[sc4];
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  if (pre.scale) { psc = reinterpret_cast<const f32x4*>(pre.scale)[lc4]; psh = reinterpret_cast<const f32x4*>(pre.shift)[lc4]; }

  // Prefetch = RAW loads only, all issued back to back (pre-ops are applied when the tile is written to LDS, a whole
  // MFMA phase later): anything computed between the loads makes hipcc wait for each load before issuing the next.
  // The squeeze-excite gate is constant over each 16-row half of a tile (launcher: rows_per_image % 16 == 0, so 4x4
  // feature maps qualify): two 16-byte loads per tile.
  struct Stage { f32x4 st[LD]; f32x4 g0, g1; };
  // FAST runs TWO tiles of prefetch ahead (buffers A and B alternate): with one tile (8 KB per wave, 64 KB per CU) the
  // plain conv0 launch (one input stream) sat at 3.4 TB/s while conv2 (input + residual in flight) reached 5 TB/s --
  // the kernel is limited by bytes in flight, not by MFMA or LDS work.
  Stage A, Bq;
  A.g0 = A.g1 = Bq.g0 = Bq.g1 = f32x4{1.f, 1.f, 1.f, 1.f};
  const ImgOf img_of(rows_per_image);
  const f32x4* G4 = reinterpret_cast<const f32x4*>(pre.gate ? pre.gate : X);   // dummy source keeps the load unconditional
  const f32x4* RS4 = residual ? R4 : reinterpret_cast<const f32x4*>(Y);        // likewise ([M,N] like the residual)
  auto load_tile = [&](int64_t tile, Stage& S) {
    const int64_t row0 = tile * 128 + wave * 32;
    if constexpr (FAST) {
#pragma unroll
      for (int j = 0; j < LD; ++j) S.st[j] = X4[(row0 + j * RPL + lr) * C4 + lc4];
      const uint32_t b0 = pre.gate ? img_of(row0) : 0u, b1 = pre.gate ? img_of(row0 + 16) : 0u;
      S.g0 = G4[(int64_t)b0 * C4 + lc4];
      S.g1 = G4[(int64_t)b1 * C4 + lc4];
    } else {
#pragma unroll
      for (int j = 0; j < LD; ++j) {
        int64_t row = row0 + j * RPL + lr;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        S.st[j] = row < M ? X4[row * C4 + lc4] : z;
      }
      if (pre.gate && row0 < M) {
        S.g0 = reinterpret_cast<const f32x4*>(pre.gate)[(int64_t)img_of(row0) * C4 + lc4];
        const uint32_t r1 = (uint32_t)(row0 + 16 < M ? row0 + 16 : row0);
        S.g1 = reinterpret_cast<const f32x4*>(pre.gate)[(int64_t)img_of(r1) * C4 + lc4];
      }
    }
  };

  int64_t tile = blockIdx.x;
  const int64_t g1 = gridDim.x, g2 = 2 * (int64_t)gridDim.x;
  if (tile < ntiles) {
    load_tile(tile, A);
    if constexpr (D2) load_tile(tile + g1 < ntiles ? tile + g1 : tile, Bq);
  }
  // FAST with a residual: the residual of tile t+1 is fetched at the END of tile t's epilogue, into the registers tile t
  // has just finished with -- a whole tile ahead of its use at no register cost (fetched at the start of its own tile
  // it had one MFMA phase, less than the loaded HBM latency, to arrive).
  f32x4 resq[ST];
  auto load_res = [&](int64_t tile) {
    const int64_t row0 = tile * 128 + wave * 32;
#pragma unroll
    for (int j = 0; j < ST; ++j) resq[j] = RS4[(row0 + j * RPS + sr) * N4 + sc4];
  };
  if constexpr (FAST && !D2) {
    if (tile < ntiles) load_res(tile);
  }
  // S holds this tile (fetched two iterations ago in FAST mode); it is refilled for tile + ahead once it is in LDS
  auto body = [&](int64_t tile, Stage& S, int64_t ahead) {    // everything below is wave-private: no block barrier
    const int64_t row0 = tile * 128 + wave * 32;
    WAVE_LDS_SYNC();     // the previous tile's C read-back is done
#pragma unroll
    for (int j = 0; j < LD; ++j) {
      f32x4 v = S.st[j];
      if (pre.scale) v = v * psc + psh;
      if (pre.gate) v = v * (j * RPL >= 16 ? S.g1 : S.g0);             // RPL divides 16: rows j*RPL+lr < 16 <=> j*RPL < 16
      if (!FAST && row0 + j * RPL + lr >= M) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(&sT[(j * RPL + lr) * TS + lc4 * 4]) = v;
    }
    WAVE_LDS_SYNC();
    f32x4 res[ST];
    if constexpr (FAST) {
      // prefetch under the MFMAs; past the last tile the current one is fetched again (never used)
      load_tile(tile + ahead < ntiles ? tile + ahead : tile, S);
    } else {
      if (tile + ahead < ntiles) load_tile(tile + ahead, S);
      if (residual) {
#pragma unroll
        for (int j = 0; j < ST; ++j) {
          int64_t row = row0 + j * RPS + sr;
          f32x4 z = {0.f, 0.f, 0.f, 0.f};
          res[j] = row < M ? R4[row * N4 + sc4] : z;
        }
      }
    }
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    // fragments in two batches (KH/8 float4 each): all reads of a batch in flight ahead of its MFMAs (progressive
    // lgkmcnt), half the registers of a single batch
    constexpr int QH = KH / 8 > 0 ? KH / 8 : 1, NB = (KH / 4) / QH;
#pragma unroll
    for (int bq = 0; bq < NB; ++bq) {
      f32x4 afr[QH];
#pragma unroll
      for (int q = 0; q < QH; ++q)
        afr[q] = *reinterpret_cast<const f32x4*>(&sT[i * TS + h * KH + (bq * QH + q) * 4]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < QH; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[q][e], breg[(bq * QH + q) * 4 + e][nt], acc[nt], 0, 0, 0);
    }
    WAVE_LDS_SYNC();     // every lane's A fragments are consumed: the tile buffer becomes the C tile
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) sT[((r & 3) + 8 * (r >> 2) + 4 * h) * TS + nt * 32 + i] = acc[nt][r];
    WAVE_LDS_SYNC();
    f32x4 dsum = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FAST) {
      // all outputs are formed first; the next tile's residual is then requested BEFORE the stores (vmcnt is one
      // in-order queue: a wait for a load issued behind the stores would also wait for their acknowledgement)
      f32x4 vout[ST];
#pragma unroll
      for (int j = 0; j < ST; ++j) {
        f32x4 v = *reinterpret_cast<const f32x4*>(&sT[(j * RPS + sr) * TS + sc4 * 4]) + bias4;
        if (act != ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = act_apply_m(v[e], act);
        }
        if constexpr (!D2) {
          if (residual) v = v + resq[j];
        }
        vout[j] = v;
      }
      if constexpr (!D2) load_res(tile + ahead < ntiles ? tile + ahead : tile);
#pragma unroll
      for (int j = 0; j < ST; ++j) Y4[(row0 + j * RPS + sr) * N4 + sc4] = vout[j];
    } else {
#pragma unroll
      for (int j = 0; j < ST; ++j) {
        const int r = j * RPS + sr;
        int64_t row = row0 + r;
        f32x4 v = *reinterpret_cast<const f32x4*>(&sT[r * TS + sc4 * 4]) + bias4;
        if (act != ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = act_apply_m(v[e], act);
        }
        if (residual) v = v + res[j];
        if (row < M) {
          Y4[row * N4 + sc4] = v;
          if (dot_src) dsum += v * D4[row * N4 + sc4];
        }
      }
    }
    if (!FAST && dot_src) {
      // lanes that share the channel chunk sc4 are N4 apart: fold them, then one atomic per channel per tile
      // (the launcher guarantees rows_per_image % 32 == 0, so a 32-row tile never straddles two images)
#pragma unroll
      for (int off = N4; off < 64; off <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) dsum[e] += __shfl_xor(dsum[e], off, 64);
      if (lane < N4 && row0 < M) {
        float* dst = dot_out + (int64_t)img_of(row0) * N + sc4 * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dst + e, dsum[e]);
      }
    }
  };
  // The first tile is peeled: at the loop header the queue of outstanding memory operations then looks the same on
  // entry and on the back edge (prefetch loads, residual loads, stores -- in that order), and the wait for the
  // prefetched tile can leave the younger residual loads and stores in flight.
  // (the loop must be reachable ONLY through the peeled iteration: any other path into its header carries the
  // prologue's queue state and forces the conservative wait again)
  if constexpr (D2) {
    if (tile < ntiles) {
      body(tile, A, g2);
      tile += g1;
      if (tile < ntiles) {
        body(tile, Bq, g2);
        tile += g1;
        while (tile < ntiles) {
          body(tile, A, g2);
          tile += g1;
          if (tile >= ntiles) break;
          body(tile, Bq, g2);
          tile += g1;
        }
      }
    }
  } else if constexpr (FAST) {
    if (tile < ntiles) {
      body(tile, A, g1);
      for (tile += g1; tile < ntiles; tile += g1) body(tile, A, g1);
    }
  } else {
    for (; tile < ntiles; tile += g1) body(tile, A, g1);
  }
}

template <int K, int N, bool WT>
static void run_gemm_rows(const float* X, const float* W, const float* bias, const float* residual, float* Y,
                          int64_t M, int64_t rows_per_image, PreOp pre, int act, const float* dot_src, float* dot_out,
                          hipStream_t s) {
  int64_t ntiles = (M + 127) / 128;
  static const int rows_cap = [] { const char* e = getenv("MVAE_ROWS_GRID"); return e ? atoi(e) : 0; }();
  const int cap = rows_cap > 0 ? rows_cap : 2 * big_grid_cus();   // 2 resident workgroups per CU (register-limited)
  int grid = (int)(ntiles < cap ? ntiles : cap);
  if (M % 128 == 0 && !dot_src && !residual)
    hipLaunchKernelGGL((k_gemm_rows<K, N, WT, true, true>), dim3(grid), dim3(256), 0, s, X, W, bias, residual, Y, M,
                       rows_per_image, pre, act, dot_src, dot_out);
  else if (M % 128 == 0 && !dot_src)
    hipLaunchKernelGGL((k_gemm_rows<K, N, WT, true, false>), dim3(grid), dim3(256), 0, s, X, W, bias, residual, Y, M,
                       rows_per_image, pre, act, dot_src, dot_out);
  else
    hipLaunchKernelGGL((k_gemm_rows<K, N, WT, false, false>), dim3(grid), dim3(256), 0, s, X, W, bias, residual, Y, M,
                       rows_per_image, pre, act, dot_src, dot_out);
}

// returns false when the shape is not covered (caller falls back to the generic kernel)
bool launch_conv1x1_mfma(bool transposed, const float* in, const float* w, const float* bias, const float* residual,
                         float* out, const ConvGeom& g, PreOp pre, int act, const float* dot_src, float* dot_out,
                         hipStream_t s) {
  if (g.KH != 1 || g.KW != 1 || g.SH != 1 || g.SW != 1) return false;
  if (dot_src && (((int64_t)g.IH * g.IW) % 32) != 0) return false;
  if (pre.gate && (((int64_t)g.IH * g.IW) % 16) != 0) return false;
  const int K = transposed ? g.CO : g.CI, N = transposed ? g.CI : g.CO;
  const int64_t M = (int64_t)g.B * g.IH * g.IW, rpi = (int64_t)g.IH * g.IW;
#define MVAE_GR(KK, NN)                                                                                   \
  if (K == KK && N == NN) {                                                                               \
    if (transposed) run_gemm_rows<KK, NN, true>(in, w, bias, residual, out, M, rpi, pre, act, dot_src, dot_out, s);         \
    else run_gemm_rows<KK, NN, false>(in, w, bias, residual, out, M, rpi, pre, act, dot_src, dot_out, s);                   \
    return true;                                                                                          \
  }
  MVAE_GR(64, 64) MVAE_GR(32, 32) MVAE_GR(64, 32) MVAE_GR(32, 64)
#undef MVAE_GR
  return false;
}

// =================================================================================================
// dW[tap][ci][co] += sum_m pre(big)[gather(m, tap)][ci] * small[m][co]   ;   db[co] += sum_m small[m][co]
// (weight gradient of a strided SAME conv in F-form coordinates; 1x1 is the tap-less special case.)
// grid = (row chunks, taps).  Wave w of a block takes every 4th 32-row tile of the chunk, accumulates a
// CI x CO tile set in MFMA accumulators (D[i = ci][j = co]), the 4 waves are reduced through LDS and the block
// issues ONE set of coalesced float atomics (256-byte segments).
// =================================================================================================
template <int CI, int CO>
__global__ void __launch_bounds__(256, 3) k_wgrad_rows(const float* __restrict__ big, const float* __restrict__ small,
                                                       float* __restrict__ dW, float* __restrict__ db, ConvGeom g,
                                                       PreOp pre, int64_t M, int64_t rows_per_block, int nslots,
                                                       int64_t slot_stride) {
  constexpr int KT = CI / 32, NT = CO / 32;
  constexpr int R = 16;                          // rows per wave tile
  constexpr int CI4 = CI / 4, CO4 = CO / 4;
  constexpr int LX = R * CI4 / 64, LG = R * CO4 / 64;       // float4 loads per lane per tile
  constexpr int RPX = 64 / CI4, RPG = 64 / CO4;             // rows covered by one wave-wide float4 load
  constexpr int TILE = R * (CI + CO);
  __shared__ __attribute__((aligned(16))) float lds[(4 * TILE > CI * CO) ? 4 * TILE : CI * CO];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* sX = lds + wave * TILE;
  float* sG = sX + R * CI;
  const int i = lane & 31, h = lane >> 5;
  const int tap = blockIdx.y, kh = tap / g.KW, kw = tap % g.KW;
  const bool pointwise = (g.KH == 1 && g.KW == 1 && g.SH == 1 && g.SW == 1);
  const int64_t HWo = (int64_t)g.OH * g.OW;
  const f32x4* big4 = reinterpret_cast<const f32x4*>(big);
  const f32x4* small4 = reinterpret_cast<const f32x4*>(small);
  const int xc4 = lane % CI4, xr = lane / CI4, gc4 = lane % CO4, gr = lane / CO4;
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  if (pre.scale) { psc = reinterpret_cast<const f32x4*>(pre.scale)[xc4]; psh = reinterpret_cast<const f32x4*>(pre.shift)[xc4]; }

  f32x16 acc[KT][NT];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kt][nt][r] = 0.f;
  float bsum[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bsum[nt] = 0.f;

  const int64_t m_begin = (int64_t)blockIdx.x * rows_per_block;
  int64_t m_end = m_begin + rows_per_block;
  if (m_end > M) m_end = M;

  // Prefetch registers hold RAW loads: every address is clamped to a valid one and the validity mask, the folded
  // BatchNorm and the squeeze-excite gate are applied when the tile is written to LDS.  With the loads unconditional the
  // compiler issues them back to back (exec-masked `if (ok) load` bodies each came with their own s_waitcnt vmcnt(0)).
  f32x4 px[LX], pq[LX], pg[LG];
  float pm[LX];
  const f32x4* gate4 = reinterpret_cast<const f32x4*>(pre.gate ? pre.gate : big);   // never null: loads stay uniform
  const float gate_on = pre.gate ? 1.f : 0.f;
  auto load_tile = [&](int64_t row0) {
    // decode this lane's first row once, then step RPX rows per load (branch-free; the launcher guarantees 2*OW >= RPX)
    int64_t m = row0 + xr;
    int64_t b = m / HWo;
    int rem = (int)(m - b * HWo);
    int oh = rem / g.OW, ow = rem - oh * g.OW;
#pragma unroll
    for (int j = 0; j < LX; ++j) {
      bool ok = m < m_end;
      int64_t src = m;
      if (!pointwise) {
        const int yy = oh * g.SH + kh - g.PT, xx = ow * g.SW + kw - g.PL;
        ok = ok && yy >= 0 && yy < g.IH && xx >= 0 && xx < g.IW;
        src = (b * g.IH + yy) * g.IW + xx;
      }
      src = ok ? src : 0;
      const int64_t bb = ok ? b : 0;
      px[j] = big4[src * CI4 + xc4];
      pq[j] = gate4[bb * CI4 + xc4];
      pm[j] = ok ? 1.f : 0.f;
      m += RPX;
      ow += RPX;
#pragma unroll
      for (int wr = 0; wr < 2; ++wr) {
        const bool c = ow >= g.OW;
        ow -= c ? g.OW : 0;
        oh += c ? 1 : 0;
        const bool c2 = oh >= g.OH;
        oh = c2 ? 0 : oh;
        b += c2 ? 1 : 0;
      }
    }
#pragma unroll
    for (int j = 0; j < LG; ++j) {
      const int64_t mm = row0 + j * RPG + gr;
      pg[j] = small4[(mm < m_end ? mm : m_begin) * CO4 + gc4];
    }
  };

  int64_t row0 = m_begin + wave * R;
  if (row0 < m_end) load_tile(row0);
  for (; row0 < m_end; row0 += 4 * R) {
    WAVE_LDS_SYNC();       // the previous tile's fragment reads are done
#pragma unroll
    for (int j = 0; j < LX; ++j) {
      f32x4 v = px[j] * psc + psh;                                     // identity when there is no folded BatchNorm
      const f32x4 q = pq[j] * gate_on + (1.f - gate_on);               // gate, or 1
      *reinterpret_cast<f32x4*>(&sX[(j * RPX + xr) * CI + xc4 * 4]) = v * q * pm[j];
    }
#pragma unroll
    for (int j = 0; j < LG; ++j) {
      const int64_t mm = row0 + j * RPG + gr;
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(&sG[(j * RPG + gr) * CO + gc4 * 4]) = mm < m_end ? pg[j] : z;
    }
    WAVE_LDS_SYNC();
    if (row0 + 4 * R < m_end) load_tile(row0 + 4 * R);       // prefetch under the MFMAs
    // all operands of the tile are read from LDS up front (R/2 * (KT+NT) registers): the MFMAs then issue back to
    // back instead of each waiting for its own ds_read
    float av[R / 2][KT], bv[R / 2][NT];
#pragma unroll
    for (int tt = 0; tt < R / 2; ++tt) {
      const int r = h * (R / 2) + tt;       // lane half h sums rows [8h, 8h+8) of the tile
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) av[tt][kt] = sX[r * CI + kt * 32 + i];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bv[tt][nt] = sG[r * CO + nt * 32 + i];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tt = 0; tt < R / 2; ++tt) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bsum[nt] += bv[tt][nt];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[kt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tt][kt], bv[tt][nt], acc[kt][nt], 0, 0, 0);
    }
  }
  // ---- reduce the 4 waves through LDS, then ONE set of coalesced float atomics per block
  __syncthreads();
  float* red = lds;
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int ci = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            int idx = ci * CO + nt * 32 + i;
            red[idx] = (wv == 0 ? 0.f : red[idx]) + acc[kt][nt][r];
          }
    }
    __syncthreads();
  }
  const int64_t gslot = (int64_t)(blockIdx.x % nslots) * slot_stride;    // gradient slot (1x1 convs; kernels.h)
  float* dWt = dW + gslot + (int64_t)tap * CI * CO;
  for (int idx = threadIdx.x; idx < CI * CO; idx += 256) atomicAdd(&dWt[idx], red[idx]);
  if (db != nullptr && tap == 0) {
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) red[(wave * 2 + h) * CO + nt * 32 + i] = bsum[nt];
    __syncthreads();
    if (threadIdx.x < CO) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += red[q * CO + threadIdx.x];
      atomicAdd(&db[gslot + threadIdx.x], t);
    }
  }
}

// k x k weight gradient with one block per KERNEL ROW kh: the TG = KW taps of the row share the `small` tile (staged
// once per 16-row tile, its fragments kept in registers across the taps) and only the gathered `big` tile changes per
// tap, prefetched one tap ahead.  Against one-tap-per-block (k_wgrad_rows with grid.y = taps) this cuts the L2 -> CU
// traffic of the 5x5 convolutions from 25 x (big + small) to 25 x big + 5 x small, which is what bounded them.
// The KH blocks that walk the same rows read the same `big` pixels: the 1-D grid is laid out so that they are dispatched
// back to back AND land on the same XCD (workgroup L runs on XCD L % 8), i.e. share an L2: L = xcd + 8 * (kh + KH * group),
// row chunk = 8 * group + xcd.  With (chunk, kh) as (blockIdx.x, blockIdx.y) the kh passes were a whole tensor sweep
// apart and `big` came across the fabric ~2.5 times (PMC: 374 MB per launch for a 134 MB tensor).
template <int CI, int CO, int TG>
__global__ void __launch_bounds__(256, 2) k_wgrad_taprow(const float* __restrict__ big, const float* __restrict__ small,
                                                         float* __restrict__ dW, float* __restrict__ db, ConvGeom g,
                                                         int64_t M, int64_t rows_per_block) {
  constexpr int KT = CI / 32, NT = CO / 32;
  constexpr int R = 16;
  constexpr int CI4 = CI / 4, CO4 = CO / 4;
  constexpr int LX = R * CI4 / 64, LG = R * CO4 / 64;
  constexpr int RPX = 64 / CI4, RPG = 64 / CO4;
  constexpr int TILE = R * (CI + CO);
  __shared__ __attribute__((aligned(16))) float lds[(4 * TILE > CI * CO) ? 4 * TILE : CI * CO];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* sX = lds + wave * TILE;
  float* sG = sX + R * CI;
  const int i = lane & 31, h = lane >> 5;
  const int xcd = blockIdx.x & 7, kh = (blockIdx.x >> 3) % g.KH;
  const uint32_t chunk = ((blockIdx.x >> 3) / g.KH) * 8u + xcd;      // may be past the last chunk
  if ((int64_t)chunk * rows_per_block >= M) return;                  // block-uniform, before any barrier
  const int HWo = g.OH * g.OW;
  const f32x4* big4 = reinterpret_cast<const f32x4*>(big);
  const f32x4* small4 = reinterpret_cast<const f32x4*>(small);
  const int xc4 = lane % CI4, xr = lane / CI4, gc4 = lane % CO4, gr = lane / CO4;

  f32x16 acc[TG][KT][NT];
#pragma unroll
  for (int t = 0; t < TG; ++t)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][kt][nt][r] = 0.f;
  float bsum[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bsum[nt] = 0.f;

  // M < 2^31 (launcher): rows are 32-bit
  uint32_t m_begin = chunk * (uint32_t)rows_per_block;
  if (m_begin > (uint32_t)M) m_begin = (uint32_t)M;
  uint32_t m_end = m_begin + (uint32_t)rows_per_block;
  if (m_end > (uint32_t)M) m_end = (uint32_t)M;

  // Both operands come through raw buffer resources with 32-bit byte offsets: a row past the block's range or a tap
  // in the SAME padding gets an offset >= 2^31, out of range for the buffer, which returns zeros -- no masks or
  // multiplies on the data, ~4 VALU instructions per 16-byte load (fp32 MFMAs share their issue slots with VALU work,
  // see k_conv_taps: the first version's ~110 VALU instructions per 16 MFMAs held it at 70 - 83 TFLOP/s).
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(big), 0,
      (int)((unsigned)g.B * g.IH * g.IW * CI * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(small), 0,
      (int)((unsigned)M * CO * 4u), 0x00020000);
  u32x4 px[LX], pg[LG];
  unsigned pbase[LX];                  // byte offset of (row kh of the window, column -PL, chunk) per slot; >= 2^31: no row
  int px0[LX];                         // input column of tap kw = 0
  const bool pow2 = (g.OW & (g.OW - 1)) == 0 && (g.OH & (g.OH - 1)) == 0;      // block-uniform
  const int lgw = 31 - __builtin_clz((unsigned)g.OW), lgh = 31 - __builtin_clz((unsigned)g.OH);
  auto decode_tile = [&](uint32_t row0) {
    const uint32_t m = row0 + xr;
    int bi, oh, ow;
    if (pow2) { ow = (int)(m & (uint32_t)(g.OW - 1)); oh = (int)((m >> lgw) & (uint32_t)(g.OH - 1)); bi = (int)(m >> (lgw + lgh)); }
    else {
      const uint32_t b = m / (uint32_t)HWo, rem = m - b * (uint32_t)HWo;
      oh = (int)(rem / (uint32_t)g.OW); ow = (int)(rem - (uint32_t)oh * (uint32_t)g.OW); bi = (int)b;
    }
#pragma unroll
    for (int j = 0; j < LX; ++j) {
      const int yy = oh * g.SH + kh - g.PT, x0 = ow * g.SW - g.PL;
      const bool ok = (row0 + xr + j * RPX < m_end) && (unsigned)yy < (unsigned)g.IH;
      pbase[j] = ok ? (unsigned)(((bi * g.IH + yy) * g.IW + x0) * CI + xc4 * 4) * 4u : 0xC0000000u;
      px0[j] = x0;
      ow += RPX;
#pragma unroll
      for (int wr = 0; wr < 2; ++wr) {
        const bool c = ow >= g.OW;
        ow -= c ? g.OW : 0;
        oh += c ? 1 : 0;
        const bool c2 = oh >= g.OH;
        oh = c2 ? 0 : oh;
        bi += c2 ? 1 : 0;
      }
    }
  };
  auto load_big = [&](int kw) {
#pragma unroll
    for (int j = 0; j < LX; ++j) {
      const unsigned off = (unsigned)(px0[j] + kw) < (unsigned)g.IW ? pbase[j] + (unsigned)(kw * CI * 4) : 0x80000000u;
      px[j] = __builtin_amdgcn_raw_buffer_load_b128(brs, off, 0, 0);
    }
  };
  auto load_small = [&](uint32_t row0) {
#pragma unroll
    for (int j = 0; j < LG; ++j) {
      const uint32_t mm = row0 + j * RPG + gr;
      const unsigned off = mm < m_end ? (mm * CO + gc4 * 4) * 4u : 0x80000000u;
      pg[j] = __builtin_amdgcn_raw_buffer_load_b128(srs, off, 0, 0);
    }
  };

  uint32_t row0 = m_begin + wave * R;
  if (row0 < m_end) {
    decode_tile(row0);
    load_small(row0);
    load_big(0);
  }
  for (; row0 < m_end; row0 += 4 * R) {
    WAVE_LDS_SYNC();       // the previous tile's fragment reads are done
#pragma unroll
    for (int j = 0; j < LG; ++j) *reinterpret_cast<u32x4*>(&sG[(j * RPG + gr) * CO + gc4 * 4]) = pg[j];
    float bv[R / 2][NT];
    const bool more = row0 + 4 * R < m_end;
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      if (t > 0) WAVE_LDS_SYNC();                 // tap t-1's fragments are in registers
#pragma unroll
      for (int j = 0; j < LX; ++j) *reinterpret_cast<u32x4*>(&sX[(j * RPX + xr) * CI + xc4 * 4]) = px[j];
      WAVE_LDS_SYNC();
      // prefetch: the next tap of this tile, or tap 0 (+ the small tile) of the wave's next tile
      if (t + 1 < TG) {
        load_big(t + 1);
      } else if (more) {
        decode_tile(row0 + 4 * R);
        load_small(row0 + 4 * R);
        load_big(0);
      }
      if (t == 0) {
#pragma unroll
        for (int tt = 0; tt < R / 2; ++tt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            bv[tt][nt] = sG[(h * (R / 2) + tt) * CO + nt * 32 + i];
            bsum[nt] += bv[tt][nt];
          }
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {      // fragments in two batches of R/4 row pairs: bounds the registers
        float av[R / 4][KT];
#pragma unroll
        for (int tt = 0; tt < R / 4; ++tt)
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) av[tt][kt] = sX[(h * (R / 2) + half * (R / 4) + tt) * CI + kt * 32 + i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < R / 4; ++tt)
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[t][kt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tt][kt], bv[half * (R / 4) + tt][nt],
                                                                    acc[t][kt][nt], 0, 0, 0);
      }
    }
  }
  // ---- per tap: reduce the 4 waves through LDS, then one coalesced float-atomic set per block
  float* red = lds;
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
            for (int r = 0; r < 16; ++r) {
              int ci = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
              int idx = ci * CO + nt * 32 + i;
              red[idx] = (wv == 0 ? 0.f : red[idx]) + acc[t][kt][nt][r];
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
    for (int nt = 0; nt < NT; ++nt) red[(wave * 2 + h) * CO + nt * 32 + i] = bsum[nt];
    __syncthreads();
    if (threadIdx.x < CO) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += red[q * CO + threadIdx.x];
      atomicAdd(&db[threadIdx.x], t);
    }
  }
}

template <int CI, int CO>
static bool run_wgrad_taprow(const float* big, const float* small, float* dW, float* db, const ConvGeom& g,
                             hipStream_t s) {
  if constexpr (CI + CO > 96) {                      // 5 taps x [CI x CO] accumulators must fit the register file
    return false;
  } else {
  const int64_t M = (int64_t)g.B * g.OH * g.OW;
  if (g.KW != 5 || M * CO * 4 >= (1ll << 31) || (int64_t)g.B * g.IH * g.IW * CI * 4 >= (1ll << 31)) return false;
  // two resident 4-wave blocks per CU = 64 per XCD, and an XCD runs the KH blocks of each of its row chunks: at most
  // 64 / KH chunks per XCD, or the extra block runs as a second round on its own (520 blocks: 210 us instead of 140)
  int64_t chunks = 8 * (64 / g.KH);
  if (chunks < 8) chunks = 8;
  int64_t rpb = (M + chunks - 1) / chunks;
  rpb = (rpb + 63) / 64 * 64;
  if (rpb < 64) rpb = 64;
  chunks = (M + rpb - 1) / rpb;
  const unsigned groups = (unsigned)((chunks + 7) / 8);
  hipLaunchKernelGGL((k_wgrad_taprow<CI, CO, 5>), dim3(groups * 8u * g.KH), dim3(256), 0, s, big, small, dW, db, g, M, rpb);
  return true;
  }
}

template <int CI, int CO>
static void run_wgrad_rows(const float* big, const float* small, float* dW, float* db, const ConvGeom& g, PreOp pre,
                           GradSlots sl, hipStream_t s) {
  const int64_t M = (int64_t)g.B * g.OH * g.OW;
  const int taps = g.KH * g.KW;
  // ~768 blocks in total keeps every CU busy (3 resident blocks) while bounding the float-atomic traffic
  int64_t chunks = 768 / taps;
  if (chunks < 1) chunks = 1;
  int64_t rpb = (M + chunks - 1) / chunks;
  rpb = (rpb + 63) / 64 * 64;
  if (rpb < 64) rpb = 64;
  chunks = (M + rpb - 1) / rpb;
  if (taps != 1 && !det_mode()) sl = GradSlots();      // only the small 1x1 gradients are slotted (runtime: slot_chunks)
  hipLaunchKernelGGL((k_wgrad_rows<CI, CO>), dim3((unsigned)chunks, taps), dim3(256), 0, s, big, small, sl.at(dW),
                     sl.at(db), g, pre, M, rpb, sl.count(), sl.stride);
}

// the float32-MFMA 5 x 5 kernel by itself (tools/conv_probe_s.cpp times it next to the split-bf16 one)
bool launch_conv_wgrad_taprow_f32(const float* big, const float* small, float* dW, float* db, const ConvGeom& g, hipStream_t s) {
  if (g.CI == 64 && g.CO == 32) return run_wgrad_taprow<64, 32>(big, small, dW, db, g, s);
  if (g.CI == 32 && g.CO == 64) return run_wgrad_taprow<32, 64>(big, small, dW, db, g, s);
  return false;
}

bool launch_conv_wgrad_mfma(const float* big, const float* small, float* dW, float* db, const ConvGeom& g, PreOp pre, GradSlots sl,
                            hipStream_t s) {
  if (g.CI < 4 || 2 * g.OW < 64 / (g.CI / 4)) return false;   // the kernel's branch-free row stepping wraps at most twice
#define MVAE_WG(A, B_)                                                          \
  if (g.CI == A && g.CO == B_) {                                                \
    if (!(g.KW == 5 && !pre.scale && !pre.gate && !det_mode() &&                                                      \
          (launch_conv_wgrad_split(big, small, dW, db, g, s) || run_wgrad_taprow<A, B_>(big, small, dW, db, g, s))))    \
      run_wgrad_rows<A, B_>(big, small, dW, db, g, pre, sl, s);                     \
    return true;                                                                \
  }
  MVAE_WG(64, 64) MVAE_WG(32, 32) MVAE_WG(64, 32) MVAE_WG(32, 64)
#undef MVAE_WG
  return false;
}


// =================================================================================================
// k x k strided SAME convolution as a loop of per-tap rank-KC updates on the MFMA.
//   TFORM = false (gather / F-form):  out = small[B,OH,OW,NC=CO],  in = big[B,IH,IW,KC=CI]
//        small[p, n] = bias[n] + sum_tap sum_k big[b, oh*SH+kh-PT, ow*SW+kw-PL, k] * W[tap][k][n]
//   TFORM = true  (transposed / T-form = conv backward-data = Conv2DTranspose forward):
//        out = big[B,IH,IW,NC=CI],  in = small[B,OH,OW,KC=CO]
//        big[b,y,x,n] = bias[n] + sum_{tap: (y+PT-kh) % SH == 0, ...} sum_k small[b,(y+PT-kh)/SH,(x+PL-kw)/SW,k] * W[tap][n][k]
//     Output pixels are processed per sub-pixel phase (py,px) = (y % SH, x % SW) (blockIdx.y): all pixels of a phase
//     share the same set of valid taps, so a wave never diverges (the "4 phase convs" of a stride-2 ConvT).
// One wave = 32 output pixels x NC channels; per tap the lane's input pixel supplies a contiguous run of KC/2
// channels straight from global memory (L2-resident re-reads across taps), the tap's KC x NC weight slice is
// double-buffered in LDS for the whole block.
// =================================================================================================
// fp32 MFMAs share the SIMD's issue slots with ordinary VALU work: measured (tools/mfma_mix.hip) every VALU instruction
// next to v_mfma_f32_32x32x2_f32 costs the MFMA 2.5 - 5 cycles, with any number of waves per SIMD.  The first version
// of this kernel spent ~300 VALU instructions per tap (address arithmetic, padding selects, 64-bit pointers) next to
// its 32 MFMAs and ran at 85 TFLOP/s of 155.  So the tap loop is written to need almost none:
//   * input pixels come through a raw buffer resource: address = 32-bit byte offset, and a tap that falls into the SAME
//     padding gets bit 31 of its offset set -- out of range for the buffer, which returns zeros (no selects, no masks
//     on the data).  Per slot and tap: and, compare, select, add.
//   * every lane fetches whole rows (CPP = KC/4 lanes x 16 B cover one pixel) and the wave turns them into MFMA A
//     fragments (lane = pixel) through a wave-private, XOR-swizzled LDS tile; LDS addresses are loop constants.
//   * WB = 1 keeps ONE weight slice in LDS (two barriers per tap): 40 KB per block, 4 blocks = 16 waves per CU, which is
//     what the 1024-block launches of the headline batch need to run as a single round.
template <int KC, int NC, bool TFORM, int WB>
__global__ void __launch_bounds__(256, WB == 1 ? 4 : 2)
k_conv_taps(const float* __restrict__ in, const float* __restrict__ W, const float* __restrict__ bias,
            float* __restrict__ out, ConvGeom g, unsigned in_bytes, unsigned out_bytes) {
  constexpr int KHF = KC / 2, NT = NC / 32, Q = KHF / 4;
  // T-form stages W[tap][n][k] transposed (consecutive threads -> consecutive k): a row pitch of NC + 1 keeps those
  // stores conflict-free (pitch NC put all 32 lanes of a store on one bank: 85 % of the kernel's LDS cycles were bank
  // conflicts); the fragment reads (lanes along n) are conflict-free for either pitch
  constexpr int WP = TFORM ? NC + 1 : NC;
  constexpr int CPP = KC / 4, PPI = 64 / CPP, SWZ = KC == 32 ? 1 : 0;
  static_assert(32 / PPI == Q, "one load slot per fragment quad");
  __shared__ __attribute__((aligned(16))) float sW[WB][KC * WP];
  __shared__ __attribute__((aligned(16))) float sA[4][32 * KC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  // ---- which output pixels
  // T-form: one sub-pixel phase per blockIdx.y.  (Dispatching the SH*SW phases of a tile back to back on one XCD, so
  // that they share the input in its L2, halves the kernel's fabric reads -- 210 -> 107 MB -- but costs the step 0.1 ms
  // in a same-box A/B; the phases stay a tensor sweep apart.)
  int py = 0, px = 0, CH = g.OH, CW = g.OW;
  const unsigned tile = blockIdx.x;
  if (TFORM) {
    py = blockIdx.y / g.SW; px = blockIdx.y % g.SW;
    CH = (g.IH - py + g.SH - 1) / g.SH; CW = (g.IW - px + g.SW - 1) / g.SW;
  }
  const unsigned Mc = (unsigned)(g.B * CH * CW);           // < 2^31 (launcher)
  const unsigned p0 = tile * 128u;
  if (p0 >= Mc) return;                         // block-uniform
  const int SHh = TFORM ? g.OH : g.IH, SWw = TFORM ? g.OW : g.IW;      // the tensor the taps read
  // ---- tap list (block-uniform)
  int kh0 = 0, kw0 = 0, khs = 1, kws = 1;
  if (TFORM) { kh0 = (py + g.PT) % g.SH; kw0 = (px + g.PL) % g.SW; khs = g.SH; kws = g.SW; }
  const int nkh = kh0 < g.KH ? (g.KH - kh0 + khs - 1) / khs : 0;      // <= 8 (launcher)
  const int nkw = kw0 < g.KW ? (g.KW - kw0 + kws - 1) / kws : 0;
  const int ntaps = nkh * nkw;
  // F-form with stride 2 reads even input columns for even kw and odd ones for odd kw: the tap list walks a kernel row as
  // kw = 0, 2, 4, 1, 3, so consecutive taps touch the same cache lines shifted by one output pixel while they are still
  // in L2 (in natural order the re-use distance is two taps of the whole XCD's traffic, more than the 4 MB L2 holds)
  const int kw_even = (nkw + 1) / 2;
  auto col_of = [&](int t) { return (TFORM || g.SW != 2) ? t : (t < kw_even ? 2 * t : 2 * (t - kw_even) + 1); };
  // ---- the pixels this lane FETCHES: slot j = pixel lp + PPI * j of the wave, 16-byte chunk ch.
  // base[j] = byte offset of (window origin pixel, chunk); inv[j] bit t = tap row t leaves the image, bit 8 + t = tap
  // column t does
  const int lp = lane / CPP, ch = lane % CPP;
  unsigned base[Q], inv[Q];
  const bool pow2 = (CW & (CW - 1)) == 0 && (CH & (CH - 1)) == 0;       // block-uniform: shifts instead of divisions
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
    const int y0 = TFORM ? cy : cy * g.SH, x0 = TFORM ? cx : cx * g.SW;
    base[j] = (unsigned)(((b * SHh + y0) * SWw + x0) * KC + ch * 4) * 4u;
    inv[j] = p < Mc ? (unsigned)(y0 << 16 | x0) : 0xFFFFFFFFu;           // coordinates for now, masks below
  }
  {
    unsigned m[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) m[j] = inv[j] == 0xFFFFFFFFu ? 0xFFFFu : 0u;
    for (int t = 0; t < nkh; ++t) {                                      // run-time trip counts: scalar loops
      const int kh = kh0 + t * khs;
      const int dy = TFORM ? (py + g.PT - kh) / g.SH : kh - g.PT;        // T: exact, kh is in this phase's residue class
#pragma unroll
      for (int j = 0; j < Q; ++j)
        if ((unsigned)((int)(inv[j] >> 16) + dy) >= (unsigned)SHh) m[j] |= 1u << t;
    }
    for (int t = 0; t < nkw; ++t) {
      const int kw = kw0 + col_of(t) * kws;
      const int dx = TFORM ? (px + g.PL - kw) / g.SW : kw - g.PL;
#pragma unroll
      for (int j = 0; j < Q; ++j)
        if ((unsigned)((int)(inv[j] & 0xFFFFu) + dx) >= (unsigned)SWw) m[j] |= 0x100u << t;
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) inv[j] = m[j];
  }
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)in_bytes, 0x00020000);

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

  // The next tap's weight slice and input pixels travel global -> registers (issued BEFORE this tap's MFMAs) -> LDS
  // (written AFTER them): their latency hides under the MFMAs instead of stalling them at an LDS write.
  constexpr int WPT = KC * NC / 256;                     // weights per thread per tap
  float wtmp[WPT];
  auto fetch_w = [&](int th, int tw) {
    const int kh = kh0 + th * khs, kw = kw0 + col_of(tw) * kws;
    const float* wt = W + (int64_t)(kh * g.KW + kw) * KC * NC;
    if (TFORM) {
#pragma unroll
      for (int u = 0; u < WPT; ++u) wtmp[u] = wt[threadIdx.x + u * 256];
    } else {                                             // F-form copies the slice as it lies: 16 B per lane
#pragma unroll
      for (int u = 0; u < WPT / 4; ++u) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wt + (threadIdx.x + u * 256) * 4);
        wtmp[u * 4] = v[0]; wtmp[u * 4 + 1] = v[1]; wtmp[u * 4 + 2] = v[2]; wtmp[u * 4 + 3] = v[3];
      }
    }
  };
  auto store_w = [&](int buf) {
    if (TFORM) {
#pragma unroll
      for (int u = 0; u < WPT; ++u) {
        const int idx = threadIdx.x + u * 256;           // W[tap][n][k] -> sW[k][n]
        const int n = idx / KC, k = idx % KC;
        sW[buf][k * WP + n] = wtmp[u];
      }
    } else {
#pragma unroll
      for (int u = 0; u < WPT / 4; ++u)
        *reinterpret_cast<f32x4*>(&sW[buf][(threadIdx.x + u * 256) * 4]) =
            f32x4{wtmp[u * 4], wtmp[u * 4 + 1], wtmp[u * 4 + 2], wtmp[u * 4 + 3]};
    }
  };
  u32x4 atmp[Q];
  auto fetch_a = [&](int th, int tw) {
    const int kh = kh0 + th * khs, kw = kw0 + col_of(tw) * kws;
    const int dy = TFORM ? (py + g.PT - kh) / g.SH : kh - g.PT;
    const int dx = TFORM ? (px + g.PL - kw) / g.SW : kw - g.PL;
    const unsigned delta = (unsigned)((dy * SWw + dx) * KC * 4);
    const unsigned sel = (1u << th) | (0x100u << tw);
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      const unsigned off = (inv[j] & sel) ? 0x80000000u : base[j] + delta;
      atmp[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    }
  };
  float* myA = sA[wave];
  auto store_a = [&]() {
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      const int pl = lp + PPI * j;
      *reinterpret_cast<u32x4*>(myA + pl * KC + ((ch ^ ((pl >> SWZ) & (CPP - 1))) * 4)) = atmp[j];
    }
  };

  if (ntaps > 0) { fetch_w(0, 0); fetch_a(0, 0); store_w(0); store_a(); }
  int th = 0, tw = 0;                     // the NEXT tap's row / column in the tap list
  for (int it = 0; it < ntaps; ++it) {
    __syncthreads();       // sW[it % WB] and sA are complete (WB = 2: and sW[(it + 1) & 1] is free)
    const bool more = it + 1 < ntaps;
    if (++tw == nkw) { tw = 0; ++th; }
    if (more) { fetch_w(th, tw); fetch_a(th, tw); }
    __builtin_amdgcn_sched_barrier(0);
    const float* w = sW[WB == 2 ? (it & 1) : 0];
    // A and B fragments one group of 4 k-steps ahead of the MFMAs that use them
    f32x4 af[2];
    float bw[2][4][NT];
    auto load_ab = [&](int q, int slot) {
      af[slot] = *reinterpret_cast<const f32x4*>(myA + i * KC + (((h * Q + q) ^ ((i >> SWZ) & (CPP - 1))) * 4));
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bw[slot][e][nt] = w[(h * KHF + q * 4 + e) * WP + nt * 32 + i];
    };
    load_ab(0, 0);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (q + 1 < Q) load_ab(q + 1, (q + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 1][e], bw[q & 1][e][nt], acc[nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // sA is wave-private and this wave's fragment reads of it are done (their data fed the MFMAs above); a single
    // weight buffer must wait until every wave is through with it
    if (WB == 1 && more) __syncthreads();
    if (more) { store_w(WB == 2 ? ((it + 1) & 1) : 0); store_a(); }
  }
  // epilogue: the bias comes from an UNCONDITIONAL load and the 16 row offsets are read from LDS up front.  With the
  // bias load inside `if (bias)`, every exec-masked store block below carried its own s_waitcnt vmcnt(0) -- i.e. each of
  // the 32 stores of a wave waited for the previous store's acknowledgement.
  float bv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bv[nt] = (bias ? bias : W)[nt * 32 + i];
  // byte offsets of the wave's 32 output pixels, through the (now idle) wave-private A tile; pixels past the end get
  // bit 31, which the buffer store drops
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

template <int KC, int NC, bool TFORM>
static void run_conv_taps(const float* in, const float* W, const float* bias, float* out, const ConvGeom& g,
                          hipStream_t s) {
  int64_t Mc;
  int classes = 1;
  if (TFORM) {
    classes = g.SH * g.SW;
    Mc = (int64_t)g.B * ((g.IH + g.SH - 1) / g.SH) * ((g.IW + g.SW - 1) / g.SW);   // largest phase
  } else {
    Mc = (int64_t)g.B * g.OH * g.OW;
  }
  const unsigned in_bytes = (unsigned)((int64_t)g.B * (TFORM ? g.OH * g.OW : g.IH * g.IW) * KC * 4);
  const dim3 grid((unsigned)((Mc + 127) / 128), classes);
  const unsigned out_bytes = (unsigned)((int64_t)g.B * (TFORM ? g.IH * g.IW : g.OH * g.OW) * NC * 4);
  // one weight buffer where that is what brings the block down to 40 KB of LDS (4 blocks per CU), two otherwise
  constexpr int WP = TFORM ? NC + 1 : NC;
  constexpr bool kOneBuf = (2 * KC * WP + 4 * 32 * KC) * 4 > 40 * 1024 && (KC * WP + 4 * 32 * KC) * 4 <= 40 * 1024;
  if (kOneBuf)
    hipLaunchKernelGGL((k_conv_taps<KC, NC, TFORM, kOneBuf ? 1 : 2>), grid, dim3(256), 0, s, in, W, bias, out, g, in_bytes,
                       out_bytes);
  else
    hipLaunchKernelGGL((k_conv_taps<KC, NC, TFORM, 2>), grid, dim3(256), 0, s, in, W, bias, out, g, in_bytes, out_bytes);
}

bool launch_conv_taps_mfma(bool transposed, const float* in, const float* w, const float* bias, float* out,
                           const ConvGeom& g, hipStream_t s) {
  const int KC = transposed ? g.CO : g.CI, NC = transposed ? g.CI : g.CO;
  // the kernel indexes pixels with 32-bit integers
  // the kernel addresses its input with 31-bit byte offsets and lists at most 8 tap rows / columns per phase
  if ((int64_t)g.B * g.IH * g.IW * g.CI * 4 >= (1LL << 31) || (int64_t)g.B * g.OH * g.OW * g.CO * 4 >= (1LL << 31)) return false;
  if (g.KH > 8 || g.KW > 8) return false;
#define MVAE_CT(A, B_)                                                              \
  if (KC == A && NC == B_) {                                                        \
    if (transposed) run_conv_taps<A, B_, true>(in, w, bias, out, g, s);             \
    else run_conv_taps<A, B_, false>(in, w, bias, out, g, s);                       \
    return true;                                                                    \
  }
  MVAE_CT(32, 64) MVAE_CT(64, 32) MVAE_CT(32, 32) MVAE_CT(64, 64)
#undef MVAE_CT
  return false;
}


// =================================================================================================
// Dual kernel for the MobileNetV3 backward (layer_blocks.py:594-641 inverted): from ONE staged pair of tiles
//     Y[M,C]  = X[M,C] . Wt + residual            (T-form 1x1: Wt[k][n] = W[n*C + k], W = Conv2D kernel [ci][co])
//     dW[ci][co] += sum_m (aux[m,ci] * gate[b,ci]) * X[m,co] ;   db[co] += sum_m X[m,co]
//     dot_out[b,c] += sum_{m in image b} Y[m,c] * aux[m,c]          (optional: squeeze-excite gate gradient)
// which are conv2's pair  (dt2 = dout.W2^T, dW2 = (t1*g)^T dout, dg)  and conv0's pair  (da = dt0.W0^T + dout,
// dW0 = a^T dt0): 3 tensor passes instead of 5 (the separate kernels re-read both operands).
// Both products are MFMA-heavy (AI 32 FLOP/B > the fp32 ridge), so the kernel is built for MFMA occupancy: the block's
// 4 waves are WR row groups x WN = C/32 column groups; a wave owns 32 rows x 32 output columns of Y and a
// [C x 32] column slab of dW, which keeps it under 256 registers -> two blocks (8 waves) per CU, one block's
// staging / epilogue under the other's MFMAs.  LDS tiles are unpadded and XOR-swizzled per 16-byte chunk
// (chunk' = chunk ^ (row & (C/4-1))): conflict-free for the row-major float4 staging, the k-permuted ds_read_b128
// GEMM fragments and the channel-on-lane wgrad reads alike.  Y leaves straight from the accumulator layout
// (each store instruction = two full 128-byte row segments), residual and dot source are read in the same layout.
// =================================================================================================
// MODE 0: general (run-time flags, ragged M).  MODE 1 / 2 are the two ways the MobileNetV3 backward calls it, for
// M % TR == 0, as straight-line tile loops (see k_gemm_rows FAST: counted waits, first tile peeled):
//   MODE 1 = conv2 pair: gate + dot, no residual;   MODE 2 = conv0 pair: residual, no gate, no dot.
// In MODE 2 the next tile's residual is requested after this tile's outputs are formed but BEFORE they are stored:
// vmcnt is one in-order queue, so a wait for any load issued behind the stores also waits for the stores' acknowledgement
// (several us under load) -- every load a later wait needs is therefore issued ahead of the tile's stores.
template <int C, int MODE>
__global__ void __launch_bounds__(256, 2) k_gemm_dual(const float* __restrict__ X, const float* __restrict__ W,
                                                      const float* __restrict__ aux, const float* __restrict__ gate,
                                                      const float* __restrict__ residual, float* __restrict__ Y,
                                                      float* __restrict__ dW, float* __restrict__ db,
                                                      float* __restrict__ dot_out, int64_t M, int64_t rows_per_image,
                                                      int nslots, int64_t slot_stride, int dslots, int64_t dstride) {
  // dot_out has `dslots` copies `dstride` floats apart (block b adds into copy b % dslots): with large feature maps
  // thousands of tiles add into the 256 bytes of one image's gate gradient (C256-nb: 8192 adds per line, a ~200 us tail)
  if (dot_out) dot_out += (int64_t)(blockIdx.x % dslots) * dstride;
  const ImgOf img_of(rows_per_image);
  constexpr bool FAST = MODE != 0;
  constexpr int KH = C / 2, NT = C / 32, C4 = C / 4, MASK = C4 - 1;
  constexpr int WN = NT, WR = 4 / WN, TR = 32 * WR;        // block tile = TR rows
  constexpr int LD = TR * C4 / 256;                        // float4 per thread per tensor (= 4)
  __shared__ __attribute__((aligned(16))) float sX[TR * C];
  __shared__ __attribute__((aligned(16))) float sA[TR * C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = wave % WN, rw = wave / WN, n0 = nw * 32;
  const int i = lane & 31, h = lane >> 5;
  const bool has_gate = MODE == 1 || (MODE == 0 && gate != nullptr);
  const bool has_res = MODE == 2 || (MODE == 0 && residual != nullptr);
  const bool has_dot = MODE == 1 || (MODE == 0 && dot_out != nullptr);
  float breg[KH];                                          // Wt[k = h*KH + t][n = n0 + i], resident in registers
#pragma unroll
  for (int t = 0; t < KH; ++t) breg[t] = W[(int64_t)(n0 + i) * C + h * KH + t];
  const int64_t ntiles = (M + TR - 1) / TR;
  const f32x4* X4 = reinterpret_cast<const f32x4*>(X);
  const f32x4* A4 = reinterpret_cast<const f32x4*>(aux);
#define SWZ4(r, c4) ((r) * C4 + ((c4) ^ ((r) & MASK)))                    /* float4 index */
#define SWZ1(r, c) ((r) * C + ((((c) >> 2) ^ ((r) & MASK)) << 2) + ((c) & 3)) /* float index */

  f32x16 accw[NT];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[kt][r] = 0.f;
  float bsum = 0.f;

  // MODE 1 (no residual registers) at C = 32 runs TWO tiles of prefetch ahead, buffers S0 / S1 alternating (the conv2
  // pair has only its two input streams in flight; k_gemm_rows D2, same finding)
  constexpr bool D2 = MODE == 1 && C <= 32;                // at C = 64 the second buffer does not fit (61 spills)
  struct Stage { f32x4 x[LD], a[LD]; };
  Stage S0, S1;
  auto load_tile = [&](int64_t tile, Stage& S) {           // a tile is one contiguous run of TR*C floats
    f32x4 (&stx)[LD] = S.x;
    f32x4 (&sta)[LD] = S.a;
    const f32x4* px = X4 + tile * (TR * C4) + threadIdx.x;
    const f32x4* pa = A4 + tile * (TR * C4) + threadIdx.x;
    if constexpr (FAST) {
#pragma unroll
      for (int j = 0; j < LD; ++j) { stx[j] = px[j * 256]; sta[j] = pa[j * 256]; }
    } else {
      const int64_t left = M - tile * TR;
      const int lim = left < TR ? (int)left : TR;          // valid rows in this tile
#pragma unroll
      for (int j = 0; j < LD; ++j) {
        const bool ok = (j * 256 + (int)threadIdx.x) / C4 < lim;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        stx[j] = ok ? px[j * 256] : z;
        sta[j] = ok ? pa[j * 256] : z;
      }
    }
  };
  // residual in the accumulator layout, one tile ahead (into the registers the previous tile has finished with)
  float res[16];
  auto load_res = [&](int64_t tile) {
    const int64_t row0 = tile * TR + rw * 32;
    const float* pr = residual + (row0 + 4 * h) * C + n0 + i;
    if constexpr (FAST) {
#pragma unroll
      for (int r = 0; r < 16; ++r) res[r] = pr[((r & 3) + 8 * (r >> 2)) * C];
    } else {
      const int64_t left = M - row0 - 4 * h;
      const int lim = left < 32 ? (int)left : 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) res[r] = pr[((r & 3) + 8 * (r >> 2)) < lim ? ((r & 3) + 8 * (r >> 2)) * C : 0];
    }
  };
  int64_t tile = blockIdx.x;
  const int64_t g1 = gridDim.x, ahead = D2 ? 2 * g1 : g1;
  if (tile < ntiles) {
    load_tile(tile, S0);
    if constexpr (D2) load_tile(tile + g1 < ntiles ? tile + g1 : tile, S1);
    if constexpr (MODE == 2) load_res(tile);
    else if constexpr (MODE == 0) { if (residual) load_res(tile); }
  }
  auto body = [&](int64_t tile, Stage& S) {
    const int64_t row0 = tile * TR + rw * 32;              // this wave's 32 rows
    const int64_t next = tile + ahead < ntiles ? tile + ahead : tile;
    // squeeze-excite gate: lane half h reads fragment rows 16h .. 16h+15 of the wave's 32 -- one image per half when
    // rows_per_image % 16 == 0.  Requested first thing: the loads then sit AHEAD of the next tile's prefetch in the
    // in-order vmcnt queue, and the wait before the weight-gradient MFMAs leaves that prefetch in flight.
    const int64_t rowh = FAST ? row0 + 16 * h : (row0 + 16 * h < M ? row0 + 16 * h : row0);
    float gl[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      if constexpr (MODE == 1) gl[kt] = gate[(int64_t)img_of(rowh) * C + kt * 32 + i];
      else if constexpr (MODE == 2) gl[kt] = 1.0f;
      else gl[kt] = (gate && row0 < M) ? gate[(int64_t)img_of(rowh) * C + kt * 32 + i] : 1.0f;
    }
    __syncthreads();                                       // previous tile fully consumed by all waves
#pragma unroll
    for (int j = 0; j < LD; ++j) {
      const int idx = j * 256 + threadIdx.x;
      const int r = idx / C4, c4 = idx % C4;
      reinterpret_cast<f32x4*>(sX)[SWZ4(r, c4)] = S.x[j];
      reinterpret_cast<f32x4*>(sA)[SWZ4(r, c4)] = S.a[j];
    }
    __syncthreads();
    if constexpr (FAST) load_tile(next, S);                // prefetch under the MFMAs (past the end: refetch, unused)
    else if (tile + g1 < ntiles) load_tile(tile + g1, S);
    const int64_t left = M - row0 - 4 * h;
    const int lim = FAST ? 32 : (left < 32 ? (int)left : 32);   // row (r&3)+8(r>>2) of this lane half is valid below lim
    const int64_t ebase = (row0 + 4 * h) * C + n0 + i;
    // ---- Y tile = X . Wt   (32 rows x 32 columns per wave)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int QH = KH / 8 > 0 ? KH / 8 : 1, NB = (KH / 4) / QH;      // fragments in two batches: half the registers
#pragma unroll
    for (int bq = 0; bq < NB; ++bq) {
      f32x4 afr[QH];
#pragma unroll
      for (int q = 0; q < QH; ++q)
        afr[q] = reinterpret_cast<const f32x4*>(sX)[SWZ4(rw * 32 + i, h * (C4 / 2) + bq * QH + q)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < QH; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[q][e], breg[(bq * QH + q) * 4 + e], acc, 0, 0, 0);
    }
    // ---- dW[:, n0..n0+31] += (aux * gate)^T X over the wave's 32 rows (lane half h: rows 16h .. 16h+15)
    constexpr int TB = D2 ? 4 : 8;                          // row pairs per operand batch (D2 is short of registers)
#pragma unroll
    for (int part = 0; part < 16 / TB; ++part) {
      float av[TB][NT], bv[TB];
#pragma unroll
      for (int tt = 0; tt < TB; ++tt) {
        const int r = rw * 32 + h * 16 + part * TB + tt;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) av[tt][kt] = sA[SWZ1(r, kt * 32 + i)] * gl[kt];
        bv[tt] = sX[SWZ1(r, n0 + i)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tt = 0; tt < TB; ++tt) {
        bsum += bv[tt];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
          accw[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tt][kt], bv[tt], accw[kt], 0, 0, 0);
      }
    }
    // ---- epilogue straight from the accumulator layout
    if (has_res) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += res[r];
    }
    if constexpr (MODE == 2) load_res(next);               // ahead of the stores (see the header comment)
    float dsum[2] = {0.f, 0.f};                            // accumulator rows < 16 / >= 16: one image each
    float* py = Y + ebase;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rc = (r & 3) + 8 * (r >> 2);
      if (FAST || rc < lim) {
        if (has_dot) {
          // conv2 pair: aux is t1 (the depthwise ReLU output).  The consumer of Y (k_dw_bwd_ring) needs Y and the ReLU
          // mask t1 > 0 only: the mask rides in the mantissa LSB of Y (a perturbation of at most one ulp, 6e-8
          // relative), which saves that kernel a whole pass over t1.
          const float a = sA[SWZ1(rw * 32 + rc + 4 * h, n0 + i)];
          dsum[r >> 3] += acc[r] * a;
          py[rc * C] = __uint_as_float((__float_as_uint(acc[r]) & ~1u) | (a > 0.f ? 1u : 0u));
        } else {
          py[rc * C] = acc[r];
        }
      }
    }
    if (has_dot) {
      dsum[0] += __shfl_xor(dsum[0], 32, 64);
      dsum[1] += __shfl_xor(dsum[1], 32, 64);
      if (h == 0 && (FAST || row0 < M)) atomicAdd(dot_out + (int64_t)img_of(row0) * C + n0 + i, dsum[0]);
      if (h == 1 && (FAST || row0 + 16 < M)) atomicAdd(dot_out + (int64_t)img_of(row0 + 16) * C + n0 + i, dsum[1]);
    }
    if constexpr (MODE == 0) {
      if (residual && tile + g1 < ntiles) load_res(tile + g1);
    }
  };
  if constexpr (D2) {                                      // loops reachable only through the peeled first tile(s)
    if (tile < ntiles) {
      body(tile, S0);
      tile += g1;
      if (tile < ntiles) {
        body(tile, S1);
        tile += g1;
        while (tile < ntiles) {
          body(tile, S0);
          tile += g1;
          if (tile >= ntiles) break;
          body(tile, S1);
          tile += g1;
        }
      }
    }
  } else if constexpr (FAST) {
    if (tile < ntiles) {
      body(tile, S0);
      for (tile += g1; tile < ntiles; tile += g1) body(tile, S0);
    }
  } else {
    for (; tile < ntiles; tile += g1) body(tile, S0);
  }
  // ---- reduce the row groups' dW slabs through LDS, then one coalesced float-atomic set per block
  float* red = sX;                                         // C*C floats <= TR*C
  for (int step = 0; step < WR; ++step) {
    __syncthreads();
    if (rw == step) {
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const int idx = ci * C + n0 + i;
          red[idx] = (step == 0 ? 0.f : red[idx]) + accw[kt][r];
        }
    }
  }
  bsum += __shfl_xor(bsum, 32, 64);
  if (h == 0) sA[rw * C + n0 + i] = bsum;
  __syncthreads();
  const int64_t slot = (int64_t)(blockIdx.x % nslots) * slot_stride;
  for (int idx = threadIdx.x; idx < C * C; idx += 256) atomicAdd(&dW[slot + idx], red[idx]);
  if (db != nullptr && threadIdx.x < C) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < WR; ++q) t += sA[q * C + threadIdx.x];
    atomicAdd(&db[slot + threadIdx.x], t);
  }
#undef SWZ4
#undef SWZ1
}

template <int C>
static void run_gemm_dual(const float* X, const float* W, const float* aux, const float* gate, const float* residual,
                          float* Y, float* dW, float* db, float* dot_out, int64_t M, int64_t rpi, GradSlots sl,
                          int dslots, int64_t dstride, hipStream_t s) {
  constexpr int TR = 32 * (4 / (C / 32));
  int64_t ntiles = (M + TR - 1) / TR;
  const int cap = 2 * big_grid_cus();                      // 2 resident blocks per CU
  if (C == 64 && launch_gemm_dual_split(X, W, aux, gate, residual, Y, dW, db, dot_out, M, rpi, C, sl, dslots, dstride, cap, s))
    return;
  int grid = (int)(ntiles < cap ? ntiles : cap);
#define MVAE_DUAL(MODE)                                                                                              \
  hipLaunchKernelGGL((k_gemm_dual<C, MODE>), dim3(grid), dim3(256), 0, s, X, W, aux, gate, residual, Y, sl.at(dW),   \
                     sl.at(db), dot_out, M, rpi, sl.count(), sl.stride, dslots < 1 ? 1 : dslots, dstride)
  const bool full = M % TR == 0;
  if (full && gate && dot_out && !residual) MVAE_DUAL(1);
  else if (full && residual && !gate && !dot_out) MVAE_DUAL(2);
  else MVAE_DUAL(0);
#undef MVAE_DUAL
}

// =================================================================================================
// conv0 of the MobileNetV3 block (1x1, C -> C, bias, ReLU; layer_blocks.py:594-602) in the dual kernel's tiling:
// a 4-wave block shares ONE staged X tile (TR rows), a wave owns 32 rows x 32 output columns.  Against k_gemm_rows
// (wave-private 32 x C tiles, 64 weight registers) this needs ~110 registers -> four blocks per CU instead of two, with
// two tiles of prefetch each: the launch has a single input stream and was limited by bytes in flight (3.8 TB/s).
// M % TR == 0; straight-line tile loop with the first pair of tiles peeled (counted vmcnt waits).
// =================================================================================================
// GR = false: conv0 (bias + ReLU).  GR = true: conv2 of the block (layer_blocks.py:625-641): squeeze-excite gate folded
// into the staged input, bias, residual add, no activation; rows_per_image % 16 == 0 (a thread's staged float4 j
// belongs to rows 16j .. 16j+15 of the tile: one image), residual fetched one tile ahead in the accumulator layout.
template <int C, bool GR>
__global__ void __launch_bounds__(256, GR ? 2 : 4) k_conv0_tile(const float* __restrict__ X, const float* __restrict__ W,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ gate,
                                                                const float* __restrict__ residual,
                                                                float* __restrict__ Y, int64_t M, int64_t rows_per_image) {
  constexpr int KH = C / 2, NT = C / 32, C4 = C / 4, MASK = C4 - 1;
  constexpr int WN = NT, WR = 4 / WN, TR = 32 * WR;
  constexpr int LD = TR * C4 / 256;
  static_assert(256 / C4 == 16 || !GR, "gated form: one staged float4 per 16-row group");
  const ImgOf img_of(rows_per_image);
  __shared__ __attribute__((aligned(16))) float sX[TR * C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = wave % WN, rw = wave / WN, n0 = nw * 32;
  const int i = lane & 31, h = lane >> 5;
  float breg[KH];                                          // Wm[k = h*KH + t][n = n0 + i] = W[k*C + n]
#pragma unroll
  for (int t = 0; t < KH; ++t) breg[t] = W[(int64_t)(h * KH + t) * C + n0 + i];
  const float bz = bias ? bias[n0 + i] : 0.f;
  const int64_t ntiles = M / TR;
  const f32x4* X4 = reinterpret_cast<const f32x4*>(X);
  const f32x4* G4 = reinterpret_cast<const f32x4*>(gate);
#define SWZ4(r, c4) ((r) * C4 + ((c4) ^ ((r) & MASK)))
  struct Stage { f32x4 x[LD]; f32x4 g[GR ? LD : 1]; };
  Stage S0, S1;
  auto load_tile = [&](int64_t tile, Stage& S) {
    const f32x4* px = X4 + tile * (TR * C4) + threadIdx.x;
#pragma unroll
    for (int j = 0; j < LD; ++j) S.x[j] = px[j * 256];
    if constexpr (GR) {
#pragma unroll
      for (int j = 0; j < LD; ++j)
        S.g[j] = G4[(int64_t)img_of(tile * TR + j * (256 / C4)) * C4 + threadIdx.x % C4];
    }
  };
  float res[GR ? 16 : 1];
  auto load_res = [&](int64_t tile) {
    const float* pr = residual + (tile * TR + rw * 32 + 4 * h) * C + n0 + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) res[GR ? r : 0] = pr[((r & 3) + 8 * (r >> 2)) * C];
  };
  int64_t tile = blockIdx.x;
  const int64_t g1 = gridDim.x, g2 = 2 * g1;
  if (tile < ntiles) {
    load_tile(tile, S0);
    load_tile(tile + g1 < ntiles ? tile + g1 : tile, S1);
    if constexpr (GR) load_res(tile);
  }
  auto body = [&](int64_t tile, Stage& S, int64_t next_res) {
    const int64_t row0 = tile * TR + rw * 32;
    __syncthreads();                                       // previous tile's fragments are in registers everywhere
#pragma unroll
    for (int j = 0; j < LD; ++j) {
      const int idx = j * 256 + threadIdx.x;
      f32x4 v = S.x[j];
      if constexpr (GR) v = v * S.g[j];
      reinterpret_cast<f32x4*>(sX)[SWZ4(idx / C4, idx % C4)] = v;
    }
    __syncthreads();
    load_tile(tile + g2 < ntiles ? tile + g2 : tile, S);   // two tiles ahead, ahead of this tile's stores
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int QH = KH / 8 > 0 ? KH / 8 : 1, NB = (KH / 4) / QH;
#pragma unroll
    for (int bq = 0; bq < NB; ++bq) {
      f32x4 afr[QH];
#pragma unroll
      for (int q = 0; q < QH; ++q)
        afr[q] = reinterpret_cast<const f32x4*>(sX)[SWZ4(rw * 32 + i, h * (C4 / 2) + bq * QH + q)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < QH; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[q][e], breg[(bq * QH + q) * 4 + e], acc, 0, 0, 0);
    }
    float* py = Y + (row0 + 4 * h) * C + n0 + i;           // accumulator layout: two 128-byte row segments per store
    if constexpr (GR) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += bz + res[r];
      load_res(next_res);                                  // next tile's residual: requested BEFORE this tile's stores
#pragma unroll
      for (int r = 0; r < 16; ++r) py[((r & 3) + 8 * (r >> 2)) * C] = acc[r];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) py[((r & 3) + 8 * (r >> 2)) * C] = fmaxf(acc[r] + bz, 0.f);
    }
  };
  // This block's tiles are tile, tile + g1, ...: `mine` of them.  They are processed in pairs by a loop with a FIXED trip
  // count and no exit in the middle (an early `break` between the two halves let LLVM sink the first half's prefetch
  // below it -- behind the stores); the first pair is peeled (counted vmcnt waits), an odd last tile follows the loop.
  const int64_t mine = tile < ntiles ? (ntiles - tile + g1 - 1) / g1 : 0;
  const int64_t last = tile + (mine > 0 ? mine - 1 : 0) * g1;
  auto nxt = [&](int64_t t) { return t + g1 <= last ? t + g1 : t; };
  if (mine >= 2) {
    body(tile, S0, tile + g1);
    body(tile + g1, S1, nxt(tile + g1));
    tile += g2;
    for (int64_t p = 1; p < mine / 2; ++p) {
      body(tile, S0, tile + g1);
      body(tile + g1, S1, nxt(tile + g1));
      tile += g2;
    }
  }
  if (mine & 1) body(tile, S0, tile);
#undef SWZ4
}

// conv2 of a MobileNetV3 block chained with conv0 of the NEXT block (both 64 -> 64; layer_blocks.py:625-641, then :594-602
// of the following block): Y = (X * gate) . W + bias + residual is stored (it is the next block's residual and a saved
// tensor) and goes back into the LDS tile, where the four waves read it as the A operand of Y2 = relu(Y . W2 + bias2):
// the next block's conv0 does not read Y from HBM (one tensor pass less per block).
// MID: a 1x1 convolution 64 -> 32 (WTM: the Conv2DTranspose layout [out][in]) sits between the two blocks and the next
// block is 32 wide: Ym = Y . Wm + bm (stored) and Y2 = relu(Ym . W2 + b2) [32 -> 32], both by the waves of column group 0
// on their own 32 rows (Wm / W2 as B operands from LDS: [k][n], lanes along n).
template <bool MID, bool WTM>
__global__ void __launch_bounds__(256, 2) k_conv2_chain(const float* __restrict__ X, const float* __restrict__ W,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ gate,
                                                                const float* __restrict__ residual,
                                                                float* __restrict__ Y, const float* __restrict__ Wm,
                                                                const float* __restrict__ biasm, float* __restrict__ Ym,
                                                                const float* __restrict__ W2,
                                                                const float* __restrict__ bias2, float* __restrict__ Y2,
                                                                int64_t M, int64_t rows_per_image) {
  constexpr int C = 64;
  constexpr bool GR = true;
  constexpr int KH = C / 2, NT = C / 32, C4 = C / 4, MASK = C4 - 1;
  constexpr int WN = NT, WR = 4 / WN, TR = 32 * WR;
  constexpr int LD = TR * C4 / 256;
  static_assert(256 / C4 == 16 || !GR, "gated form: one staged float4 per 16-row group");
  const ImgOf img_of(rows_per_image);
  __shared__ __attribute__((aligned(16))) float sX[TR * C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = wave % WN, rw = wave / WN, n0 = nw * 32;
  const int i = lane & 31, h = lane >> 5;
  float breg[KH];                                          // Wm[k = h*KH + t][n = n0 + i] = W[k*C + n]
#pragma unroll
  for (int t = 0; t < KH; ++t) breg[t] = W[(int64_t)(h * KH + t) * C + n0 + i];
  const float bz = bias ? bias[n0 + i] : 0.f;
  float breg2[MID ? 1 : KH];
  __shared__ float sWm[MID ? 64 * 32 : 1], sW2[MID ? 32 * 32 : 1];   // MID: Wm[k][n] (64 x 32), W2[k][n] (32 x 32)
  float bz2, bzm = 0.f;
  if constexpr (MID) {
    for (int idx = threadIdx.x; idx < 64 * 32; idx += 256) {
      const int kk = idx >> 5, nn = idx & 31;
      sWm[idx] = WTM ? Wm[nn * 64 + kk] : Wm[kk * 32 + nn];
    }
    for (int idx = threadIdx.x; idx < 32 * 32; idx += 256) sW2[idx] = W2[idx];
    bzm = biasm ? biasm[i] : 0.f;
    bz2 = bias2 ? bias2[i] : 0.f;
    breg2[0] = 0.f;
  } else {
#pragma unroll
    for (int t = 0; t < KH; ++t) breg2[t] = W2[(int64_t)(h * KH + t) * C + n0 + i];
    bz2 = bias2 ? bias2[n0 + i] : 0.f;
  }
#define SWZ1(r, c) ((r) * C + ((((c) >> 2) ^ ((r) & MASK)) << 2) + ((c) & 3))
  const int64_t ntiles = M / TR;
  const f32x4* X4 = reinterpret_cast<const f32x4*>(X);
  const f32x4* G4 = reinterpret_cast<const f32x4*>(gate);
#define SWZ4(r, c4) ((r) * C4 + ((c4) ^ ((r) & MASK)))
  struct Stage { f32x4 x[LD]; f32x4 g[GR ? LD : 1]; };
  Stage S0, S1;
  auto load_tile = [&](int64_t tile, Stage& S) {
    const f32x4* px = X4 + tile * (TR * C4) + threadIdx.x;
#pragma unroll
    for (int j = 0; j < LD; ++j) S.x[j] = px[j * 256];
    if constexpr (GR) {
#pragma unroll
      for (int j = 0; j < LD; ++j)
        S.g[j] = G4[(int64_t)img_of(tile * TR + j * (256 / C4)) * C4 + threadIdx.x % C4];
    }
  };
  float res[GR ? 16 : 1];
  auto load_res = [&](int64_t tile) {
    const float* pr = residual + (tile * TR + rw * 32 + 4 * h) * C + n0 + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) res[GR ? r : 0] = pr[((r & 3) + 8 * (r >> 2)) * C];
  };
  int64_t tile = blockIdx.x;
  const int64_t g1 = gridDim.x, g2 = 2 * g1;
  if (tile < ntiles) {
    load_tile(tile, S0);
    load_tile(tile + g1 < ntiles ? tile + g1 : tile, S1);
    if constexpr (GR) load_res(tile);
  }
  auto body = [&](int64_t tile, Stage& S, int64_t next_res) {
    const int64_t row0 = tile * TR + rw * 32;
    __syncthreads();                                       // previous tile's fragments are in registers everywhere
#pragma unroll
    for (int j = 0; j < LD; ++j) {
      const int idx = j * 256 + threadIdx.x;
      f32x4 v = S.x[j];
      if constexpr (GR) v = v * S.g[j];
      reinterpret_cast<f32x4*>(sX)[SWZ4(idx / C4, idx % C4)] = v;
    }
    __syncthreads();
    load_tile(tile + g2 < ntiles ? tile + g2 : tile, S);   // two tiles ahead, ahead of this tile's stores
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int QH = KH / 8 > 0 ? KH / 8 : 1, NB = (KH / 4) / QH;
#pragma unroll
    for (int bq = 0; bq < NB; ++bq) {
      f32x4 afr[QH];
#pragma unroll
      for (int q = 0; q < QH; ++q)
        afr[q] = reinterpret_cast<const f32x4*>(sX)[SWZ4(rw * 32 + i, h * (C4 / 2) + bq * QH + q)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < QH; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[q][e], breg[(bq * QH + q) * 4 + e], acc, 0, 0, 0);
    }
    float* py = Y + (row0 + 4 * h) * C + n0 + i;           // accumulator layout: two 128-byte row segments per store
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += bz + res[r];
    load_res(next_res);                                    // next tile's residual: requested BEFORE this tile's stores
#pragma unroll
    for (int r = 0; r < 16; ++r) py[((r & 3) + 8 * (r >> 2)) * C] = acc[r];
    // ---- the Y tile back into the LDS tile (all waves have their X fragments), then Y2 = relu(Y . W2 + bias2)
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) sX[SWZ1(rw * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, n0 + i)] = acc[r];
    __syncthreads();
    if constexpr (MID) {
      if (nw == 0) {                                       // wave-uniform: column group 0 carries the two narrow products
        // Ym[32 rows][32] = Y . Wm + bm : k = h*32 + t (lane half h takes its 32 k), B = sWm[k][i]
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int bq = 0; bq < NB; ++bq) {
          f32x4 afr[QH];
          float bfr[QH * 4];
#pragma unroll
          for (int q = 0; q < QH; ++q) {
            afr[q] = reinterpret_cast<const f32x4*>(sX)[SWZ4(rw * 32 + i, h * (C4 / 2) + bq * QH + q)];
#pragma unroll
            for (int e = 0; e < 4; ++e) bfr[q * 4 + e] = sWm[(h * KH + (bq * QH + q) * 4 + e) * 32 + i];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < QH; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[q][e], bfr[q * 4 + e], acc, 0, 0, 0);
        }
        float* pym = Ym + (row0 + 4 * h) * 32 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[r] += bzm;
          pym[((r & 3) + 8 * (r >> 2)) * 32] = acc[r];
        }
        // Ym -> columns 0..31 of this wave's own rows of the tile (only this wave reads them: wave-local ordering)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int r = 0; r < 16; ++r) sX[SWZ1(rw * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, i)] = acc[r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // Y2 = relu(Ym . W2 + b2): K = 32, lane half h takes k = 16 h .. 16 h + 15 (float4 chunks 4 h .. 4 h + 3)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
          f32x4 afr[4];
          float bfr[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            afr[q] = reinterpret_cast<const f32x4*>(sX)[SWZ4(rw * 32 + i, h * 4 + q)];
#pragma unroll
            for (int e = 0; e < 4; ++e) bfr[q * 4 + e] = sW2[(h * 16 + q * 4 + e) * 32 + i];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[q][e], bfr[q * 4 + e], acc, 0, 0, 0);
        }
        float* py2 = Y2 + (row0 + 4 * h) * 32 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) py2[((r & 3) + 8 * (r >> 2)) * 32] = fmaxf(acc[r] + bz2, 0.f);
      }
    } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int bq = 0; bq < NB; ++bq) {
      f32x4 afr[QH];
#pragma unroll
      for (int q = 0; q < QH; ++q)
        afr[q] = reinterpret_cast<const f32x4*>(sX)[SWZ4(rw * 32 + i, h * (C4 / 2) + bq * QH + q)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < QH; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[q][e], breg2[(bq * QH + q) * 4 + e], acc, 0, 0, 0);
    }
    float* py2 = Y2 + (row0 + 4 * h) * C + n0 + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) py2[((r & 3) + 8 * (r >> 2)) * C] = fmaxf(acc[r] + bz2, 0.f);
    }
  };
  // This block's tiles are tile, tile + g1, ...: `mine` of them.  They are processed in pairs by a loop with a FIXED trip
  // count and no exit in the middle (an early `break` between the two halves let LLVM sink the first half's prefetch
  // below it -- behind the stores); the first pair is peeled (counted vmcnt waits), an odd last tile follows the loop.
  const int64_t mine = tile < ntiles ? (ntiles - tile + g1 - 1) / g1 : 0;
  const int64_t last = tile + (mine > 0 ? mine - 1 : 0) * g1;
  auto nxt = [&](int64_t t) { return t + g1 <= last ? t + g1 : t; };
  if (mine >= 2) {
    body(tile, S0, tile + g1);
    body(tile + g1, S1, nxt(tile + g1));
    tile += g2;
    for (int64_t p = 1; p < mine / 2; ++p) {
      body(tile, S0, tile + g1);
      body(tile + g1, S1, nxt(tile + g1));
      tile += g2;
    }
  }
  if (mine & 1) body(tile, S0, tile);
#undef SWZ4
#undef SWZ1
}


// conv0: Y = relu(X . W + b);  conv2 (gate != null): Y = (X * gate[image]) . W + b + residual.   C -> C.
// false = shape not covered (the caller uses k_gemm_rows)
bool launch_conv0_tile(const float* X, const float* W, const float* bias, const float* gate, const float* residual,
                       float* Y, int64_t M, int64_t rows_per_image, int C, hipStream_t s) {
  if (C != 64 && C != 32) return false;
  const int TR = C == 64 ? 64 : 128;
  if (M % TR != 0 || M < 256 * TR) return false;          // small launches: k_gemm_rows' wave-private tiles start faster
  const bool gr = gate != nullptr;
  if (gr && (C != 64 || !residual || rows_per_image % 16 != 0)) return false;
  if (!gr && residual) return false;
  const int64_t ntiles = M / TR;
  const int cap = (gr ? 2 : 4) * big_grid_cus();           // resident blocks per CU (the gated form needs > 168 registers)
  const int grid = (int)(ntiles < cap ? ntiles : cap);
  if (gr) hipLaunchKernelGGL((k_conv0_tile<64, true>), dim3(grid), dim3(256), 0, s, X, W, bias, gate, residual, Y, M, rows_per_image);
  else if (C == 64) hipLaunchKernelGGL((k_conv0_tile<64, false>), dim3(grid), dim3(256), 0, s, X, W, bias, gate, residual, Y, M, rows_per_image);
  else hipLaunchKernelGGL((k_conv0_tile<32, false>), dim3(grid), dim3(256), 0, s, X, W, bias, gate, residual, Y, M, rows_per_image);
  return true;
}

// conv2 of a block + conv0 of the next in one launch (k_conv2_chain); with wm: through the 1x1 convolution 64 -> 32 between
// them into a 32-wide block (mid_transposed: Conv2DTranspose weight layout).  false = shape not covered / switched off.
bool launch_conv2_chain(const float* X, const float* W, const float* bias, const float* gate, const float* residual,
                        float* Y, const float* wm, const float* biasm, float* Ym, bool mid_transposed, const float* W2,
                        const float* bias2, float* Y2, int64_t M, int64_t rows_per_image, int C, hipStream_t s) {
  static const bool on = [] { const char* e = getenv("MVAE_FUSE_PW_CHAIN"); return e ? atoi(e) != 0 : true; }();
  if (!on || C != 64 || !gate || !residual || rows_per_image % 16 != 0) return false;
  if (M % 64 != 0 || M < 256 * 64) return false;
  const int64_t ntiles = M / 64;
  const int cap = 2 * big_grid_cus();
  const int grid = (int)(ntiles < cap ? ntiles : cap);
#define MVAE_C2C(MID_, WT_)                                                                                          \
  hipLaunchKernelGGL((k_conv2_chain<MID_, WT_>), dim3(grid), dim3(256), 0, s, X, W, bias, gate, residual, Y, wm, biasm, Ym, \
                     W2, bias2, Y2, M, rows_per_image)
  if (!wm) MVAE_C2C(false, false);
  else if (mid_transposed) MVAE_C2C(true, true);
  else MVAE_C2C(true, false);
#undef MVAE_C2C
  return true;
}

// conv (1x1, C -> C) backward pair in one pass; false = shape not covered
bool launch_gemm_dual_mfma(const float* X, const float* W, const float* aux, const float* gate, const float* residual,
                           float* Y, float* dW, float* db, float* dot_out, int64_t M, int64_t rows_per_image, int C,
                           GradSlots sl, int dslots, int64_t dstride, hipStream_t s) {
  if ((gate || dot_out) && (rows_per_image % 16) != 0) return false;
  if (C == 64) { run_gemm_dual<64>(X, W, aux, gate, residual, Y, dW, db, dot_out, M, rows_per_image, sl, dslots, dstride, s); return true; }
  if (C == 32) { run_gemm_dual<32>(X, W, aux, gate, residual, Y, dW, db, dot_out, M, rows_per_image, sl, dslots, dstride, s); return true; }
  return false;
}

}  // namespace mvae
