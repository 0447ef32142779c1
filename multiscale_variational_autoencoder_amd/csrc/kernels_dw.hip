// kernels_dw.hip -- depthwise 3x3 (layer_blocks.py:604-614) as sliding-row kernels: one workgroup walks an image
// (or a column strip of it) top to bottom with a 4-row ring in LDS, so every input row comes from HBM exactly once
// and the 3x3 neighbourhood is served from LDS.
//   forward : t1 = relu(dw(t0) + b)  and  gap[b,c] = mean_hw t1            (DepthwiseConv2D + GlobalAveragePooling2D)
//   backward: d1 = (dt2 * g[b,c] + dgap[b,c]/hw) * (t1 > 0)                (through Multiply, GAP, ReLU -- on the fly)
//             dt0 = dwT(d1) * (t0 > 0) ;  dW[a][e][c] += sum t0 * d1(shifted) ;  db[c] += sum d1
// replacing five separate full-tensor launches (dw_fwd, spatial_sum; mn_dt1pre, dw_wgrad, dw_bwd_data).
#include "kernels.h"
#include "act16.h"

namespace mvae {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// geometry shared by both kernels: a block owns image b = blockIdx.y and columns [x0, x0 + XS) of it
struct DwGeom { int H, W, C4, XS, strips; };

// ring[slot][xs][c4] as float4, xs in [0, XS + 2): column xs <-> image column x0 - 1 + xs
#define RING(slot, xs, c4) ring[((slot) * XSP + (xs)) * g.C4 + (c4)]

template <bool FUSE_GAP, typename T>
__global__ void __launch_bounds__(256) k_dw_fwd_ring(const V4<T> in, const f32x4* __restrict__ w,
                                                     const f32x4* __restrict__ bias, const V4<T> out,
                                                     float* __restrict__ gap, DwGeom g, float inv_hw, int nseg, int RS) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  f32x4* ring = reinterpret_cast<f32x4*>(dyn_lds);
  const int XSP = g.XS + 2;
  // work item = (image b, row segment seg): rows [ya, ya + RS), RS % 4 == 0 (a block used to walk a whole image: at batch
  // 64 the 64 .. 128 wide maps of the 256x256 configuration then gave 128 / 256 blocks for 256 CUs)
  const int b = blockIdx.y / nseg, ya = (blockIdx.y % nseg) * RS, x0 = blockIdx.x * g.XS;
  const int xs_n = min(g.XS, g.W - x0);               // columns this strip really has
  const int c4 = threadIdx.x % g.C4;
  const int items = g.XS * g.C4;                      // <= 512: at most two items per thread
  f32x4 wt[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wt[k] = w[k * g.C4 + c4];
  const f32x4 bs = bias[c4];
  const V4<T> img = in + (int64_t)b * g.H * g.W * g.C4;
  const V4<T> oimg = out + (int64_t)b * g.H * g.W * g.C4;
  const int row_items = XSP * g.C4;

  // Rows travel HBM -> registers -> LDS ring.  A row is fetched FOUR iterations before it is stored to the ring (a
  // register FIFO of four rows, statically indexed by unrolling the row loop by four): with one or two resident blocks
  // per CU the bytes in flight per CU -- not the arithmetic -- set the rate of this kernel.
  typedef typename V4<T>::raw raw_t;
  raw_t rv[4][3], rvm[3];
  auto fetch_row = [&](int y, raw_t (&r)[3]) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int t = threadIdx.x + 256 * u;
      const int x = x0 - 1 + t / g.C4;
      const bool ok = t < row_items && x >= 0 && x < g.W;
      r[u] = img.ld(ok ? ((int64_t)y * g.W + x) * g.C4 + c4 : 0);            // RAW: clamped, unconditional, unconverted
    }
  };
  auto store_row = [&](int y, const raw_t (&r)[3]) {  // row y (with column halo) -> ring slot y & 3; halo zeroed here
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int t = threadIdx.x + 256 * u;
      const int x = x0 - 1 + t / g.C4;
      const bool ok = x >= 0 && x < g.W;
      if (t < row_items) RING(y & 3, t / g.C4, c4) = ok ? V4<T>::cv(r[u]) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  f32x4 gsum[2];
  gsum[0] = gsum[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  // H % 4 == 0, ya % 4 == 0 (launcher): the body below is straight-line code -- no uniform branches around the loads, so
  // the compiler counts the outstanding loads exactly (s_waitcnt vmcnt(N), N > 0) instead of draining them.  Rows past
  // the image are clamped to the last row: fetched and stored to a ring slot nobody reads; row ya - 1 of the first
  // segment likewise (slot 3, the taps with yy < 0 are skipped).
  const int yl = g.H - 1;
  fetch_row(max(ya - 1, 0), rvm);
  fetch_row(ya, rv[0]);
  fetch_row(ya + 1, rv[1]);
  fetch_row(ya + 2, rv[2]);
  fetch_row(ya + 3, rv[3]);
  store_row(ya - 1, rvm);
  store_row(ya, rv[0]);
  fetch_row(min(ya + 4, yl), rv[0]);
  for (int yb = ya; yb < ya + RS; yb += 4) {
#pragma unroll
    for (int u4 = 0; u4 < 4; ++u4) {
      const int y = yb + u4;
      {
        store_row(y + 1, rv[(u4 + 1) & 3]);
        fetch_row(min(y + 5, yl), rv[(u4 + 1) & 3]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int idx = threadIdx.x + 256 * k;
          if (idx < items) {
            const int xl = idx / g.C4;                // local column, image column x0 + xl, ring column xl + 1
            if (xl < xs_n) {
              f32x4 acc = bs;
#pragma unroll
              for (int a = 0; a < 3; ++a) {
                const int yy = y + a - 1;
                if (yy < 0 || yy >= g.H) continue;
#pragma unroll
                for (int e = 0; e < 3; ++e) acc += wt[a * 3 + e] * RING(yy & 3, xl + e, c4);
              }
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[q] = acc[q] > 0.f ? acc[q] : 0.f;
              oimg.st(((int64_t)y * g.W + x0 + xl) * g.C4 + c4, acc);
              if (FUSE_GAP) gsum[k] += acc;
            }
          }
        }
      }
    }
  }
  if (FUSE_GAP) {
    // reduce over the threads that share c4 (tid % C4): through LDS (the ring is free once every thread is done)
    __syncthreads();
    f32x4* red = ring;
    red[threadIdx.x] = gsum[0] + gsum[1];
    __syncthreads();
    if (threadIdx.x < g.C4) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      for (int r = threadIdx.x; r < 256; r += g.C4) t += red[r];
      t = t * inv_hw;
      float* gp = gap + (int64_t)b * g.C4 * 4 + threadIdx.x * 4;
      if (g.strips * nseg == 1) {
        *reinterpret_cast<f32x4*>(gp) = t;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicAdd(gp + q, t[q]);
      }
    }
  }
}

// ---- small feature maps (W <= 16): whole images in LDS --------------------------------------------------------------
// The ring kernels keep (FIFO depth) x (items per thread per row) 16-byte loads in flight per thread; a 16-wide row is
// ONE item per thread, so a CU had 32 KB in flight and the 16x16 maps of B = 512 ran at 3.3 TB/s (Little's law: 5 TB/s
// x ~2.5 us needs ~50 KB per CU).  Here a block takes IPB = 256 / (H * C4) whole images (256 * W float4 = W loads per
// thread, all issued at once), and thread (image, y, c4) then walks its output row with a 3x3 register window fed from
// LDS.  One memory round trip per image instead of a pipeline of H of them; the GAP needs no atomics.
template <int W_, int C4, typename T>
__global__ void __launch_bounds__(256, 2) k_dw_fwd_img(const V4<T> in, const f32x4* __restrict__ w,
                                                       const f32x4* __restrict__ bias, const V4<T> out,
                                                       float* __restrict__ gap, int H, float inv_hw) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  f32x4* tile = reinterpret_cast<f32x4*>(dyn_lds);             // [IPB][H][W_][C4]
  constexpr int total = 256 * W_;
  const int64_t base = (int64_t)blockIdx.x * total;
  typename V4<T>::raw ld[W_];
#pragma unroll
  for (int j = 0; j < W_; ++j) ld[j] = in.ld(base + threadIdx.x + 256 * j);
  const int c4 = threadIdx.x % C4, y = (threadIdx.x / C4) % H, img = threadIdx.x / (C4 * H);
  f32x4 wt[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wt[k] = w[k * C4 + c4];
  const f32x4 bs = bias[c4];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < W_; ++j) tile[threadIdx.x + 256 * j] = V4<T>::cv(ld[j]);
  __syncthreads();
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const int ipi = H * W_ * C4;
  const f32x4* rowp[3];
  bool rok[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int yy = y + a - 1;
    rok[a] = yy >= 0 && yy < H;
    rowp[a] = tile + img * ipi + (rok[a] ? yy : y) * W_ * C4 + c4;
  }
  // window columns x-1, x, x+1 of the three rows; column -1 / W_ are zero
  f32x4 win[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { win[a][1] = zero; win[a][2] = rok[a] ? rowp[a][0] : zero; }
  f32x4 gsum = zero;
  const V4<T> orow = out + (base + img * ipi + y * W_ * C4 + c4);
#pragma unroll
  for (int x = 0; x < W_; ++x) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      win[a][0] = win[a][1]; win[a][1] = win[a][2];
      win[a][2] = (x + 1 < W_ && rok[a]) ? rowp[a][(x + 1) * C4] : zero;
    }
    f32x4 acc = bs;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int e = 0; e < 3; ++e) acc += wt[a * 3 + e] * win[a][e];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = acc[q] > 0.f ? acc[q] : 0.f;
    orow.st(x * C4, acc);
    gsum += acc;
  }
  // GAP: the H rows of (image, c4) through LDS (the tile is free once every thread has its outputs)
  __syncthreads();
  tile[threadIdx.x] = gsum;
  __syncthreads();
  if (y == 0) {
    f32x4 t = zero;
    for (int r = 0; r < H; ++r) t += tile[(img * H + r) * C4 + c4];
    *reinterpret_cast<f32x4*>(gap + ((int64_t)blockIdx.x * (256 / (C4 * H)) + img) * C4 * 4 + c4 * 4) = t * inv_hw;
  }
}

static bool dw_img_shape(int B, int H, int W, int C) {
  if ((C != 32 && C != 64) || (W != 4 && W != 8 && W != 16)) return false;   // channel count is a template parameter
  const int rows = H * (C / 4);                                // (y, c4) pairs per image
  if (rows < 1 || rows > 256 || 256 % rows) return false;
  return B % (256 / rows) == 0;
}

// LSB = the ReLU mask (t1 > 0) arrives in the mantissa LSB of dt2 (written by k_gemm_dual's conv2 pair): t1 is not read
template <bool LSB, typename ST>
__global__ void __launch_bounds__(256) k_dw_bwd_ring(const V4<ST> dt2, const V4<ST> t1,
                                                     const V4<ST> t0, const f32x4* __restrict__ w,
                                                     const f32x4* __restrict__ gate, const f32x4* __restrict__ dgap,
                                                     const V4<ST> dt0, float* __restrict__ dW,
                                                     float* __restrict__ db, DwGeom g, float inv_hw, int B, int RS,
                                                     int nseg, int nslots, int64_t slot_stride) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  f32x4* ring = reinterpret_cast<f32x4*>(dyn_lds);
  const int XSP = g.XS + 2;
  const int x0 = blockIdx.x * g.XS;
  const int xs_n = min(g.XS, g.W - x0);
  const int c4 = threadIdx.x % g.C4;
  const int items = g.XS * g.C4;
  const int row_items = XSP * g.C4;
  f32x4 wt[9], aw[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) { wt[k] = w[k * g.C4 + c4]; aw[k] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  f32x4 ab = {0.f, 0.f, 0.f, 0.f};
  const int xl0 = threadIdx.x / g.C4, xl1 = (threadIdx.x + 256) / g.C4;     // this thread's (at most) two columns
  const bool has0 = threadIdx.x < items && xl0 < xs_n, has1 = threadIdx.x + 256 < items && xl1 < xs_n;

  // work item = (image b, row segment seg): rows [ya, yb) need d1 rows ya-1 .. yb
  for (int item = blockIdx.y; item < B * nseg; item += gridDim.y) {
    const int b = item / nseg, seg = item % nseg;
    const int ya = seg * RS, yb = min(g.H, ya + RS);
    const int64_t ioff = (int64_t)b * g.H * g.W * g.C4;
    // A row of d1 takes two steps: RAW loads of (dt2, t1) into a two-slot register FIFO, issued two iterations before
    // the row is needed, and -- at the top of the iteration that needs it -- the gate / GAP / ReLU arithmetic and the
    // LDS store.  t0 rows ride a second two-slot FIFO.  The row loop is unrolled by two (static slots) and is
    // straight-line code (clamped addresses, RS even): the compiler then waits with exact vmcnt(N) counts and the
    // loads of the next two rows stay in flight under the current row's arithmetic.
    typedef typename V4<ST>::raw raw_t;
    raw_t Fd[2][3], Fa[2][3], T[2][2];
    const f32x4 gg_c = gate[(int64_t)b * g.C4 + c4];       // (row_items % C4 == 0 and 256 % C4 == 0: cc == c4)
    const f32x4 dg_c = dgap[(int64_t)b * g.C4 + c4] * inv_hw;
    const int ybc = min(yb, g.H - 1);                      // last d1 row this item needs
    auto fetch_row = [&](int y, raw_t (&rd)[3], raw_t (&ra)[3]) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int t = threadIdx.x + 256 * u;
        const int x = x0 - 1 + t / g.C4;
        const bool ok = t < row_items && x >= 0 && x < g.W;
        const int64_t o = ok ? ioff + ((int64_t)y * g.W + x) * g.C4 + c4 : 0;
        rd[u] = dt2.ld(o);
        if constexpr (!LSB) ra[u] = t1.ld(o);
      }
    };
    auto store_row = [&](int y, const raw_t (&rdr)[3], const raw_t (&rar)[3]) {   // d1 row y (+ halo) -> ring slot y & 3
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int t = threadIdx.x + 256 * u;
        const int x = x0 - 1 + t / g.C4;
        const bool ok = x >= 0 && x < g.W;
        if (t < row_items) {
          const f32x4 rd = V4<ST>::cv(rdr[u]);
          f32x4 ra = rd;
          if constexpr (!LSB) ra = V4<ST>::cv(rar[u]);
          // mask bit: the mantissa LSB of the STORED value (float: bit 0, kept in the value, <= 1 ulp; bf16: bit 16 of
          // the widened pattern, cleared -- the producer rounded one bit shorter to make room for it)
          constexpr unsigned kBit = sizeof(ST) == 2 ? 0x10000u : 1u;
          f32x4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned bits = __float_as_uint(rd[q]);
            const bool on = LSB ? (bits & kBit) != 0u : ra[q] > 0.f;
            const float val = (LSB && sizeof(ST) == 2) ? __uint_as_float(bits & ~kBit) : rd[q];
            v[q] = (ok && on) ? val * gg_c[q] + dg_c[q] : 0.f;
          }
          RING(y & 3, t / g.C4, c4) = v;
        }
      }
    };
    auto fetch_t0 = [&](int y, raw_t (&tv)[2]) {
      tv[0] = t0.ld(has0 ? ioff + ((int64_t)y * g.W + x0 + xl0) * g.C4 + c4 : 0);
      tv[1] = t0.ld(has1 ? ioff + ((int64_t)y * g.W + x0 + xl1) * g.C4 + c4 : 0);
    };
    __syncthreads();                                  // previous item's ring reads are done
    // prologue: d1 rows ya-1 (for ya = 0: a copy of row 0 in a slot nobody reads) and ya go to the ring at once
    fetch_row(max(ya - 1, 0), Fd[0], Fa[0]);
    fetch_row(ya, Fd[1], Fa[1]);
    fetch_t0(ya, T[0]);
    fetch_t0(min(ya + 1, yb - 1), T[1]);
    store_row(ya - 1, Fd[0], Fa[0]);
    store_row(ya, Fd[1], Fa[1]);
    fetch_row(min(ya + 1, ybc), Fd[1], Fa[1]);
    fetch_row(min(ya + 2, ybc), Fd[0], Fa[0]);
    for (int y2 = ya; y2 < yb; y2 += 2) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int y = y2 + kk;
        store_row(y + 1, Fd[(kk + 1) & 1], Fa[(kk + 1) & 1]);
        fetch_row(min(y + 3, ybc), Fd[(kk + 1) & 1], Fa[(kk + 1) & 1]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const bool has = k == 0 ? has0 : has1;
          const int xl = k == 0 ? xl0 : xl1;
          const f32x4 tvk = has ? V4<ST>::cv(T[kk][k]) : f32x4{0.f, 0.f, 0.f, 0.f};
          if (has) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // position (y,x) was read by output pixel (y-a+1, x-e+1) through tap (a,e)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              const int yy = y - a + 1;
              if (yy < 0 || yy >= g.H) continue;
#pragma unroll
              for (int e = 0; e < 3; ++e) {
                const f32x4 sv = RING(yy & 3, xl + 2 - e, c4);     // column x - e + 1 -> ring column xl + 2 - e
                acc += wt[a * 3 + e] * sv;
                aw[a * 3 + e] += tvk * sv;
              }
            }
            ab += RING(y & 3, xl + 1, c4);
            f32x4 r;
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = tvk[q] > 0.f ? acc[q] : 0.f;
            dt0.st(ioff + ((int64_t)y * g.W + x0 + xl) * g.C4 + c4, r);
          }
        }
        fetch_t0(min(y + 2, yb - 1), T[kk]);
      }
    }
  }
  // ---- block reduction of the 10 float4 accumulators over the threads that share c4 (wave shuffles, then the 4 waves
  //      through LDS); the block's sums leave as one atomic set into gradient slot (block % nslots) (kernels.h)
  __syncthreads();
  f32x4* red = ring;                                  // needs 4 * 10 * C4 float4 <= ring size (checked by launcher)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    f32x4 v = k < 9 ? aw[k < 9 ? k : 0] : ab;
    for (int off = g.C4; off < 64; off <<= 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] += __shfl_xor(v[q], off, 64);
    }
    if (lane < g.C4 && lane < 64) red[(wave * 10 + k) * g.C4 + lane] = v;
  }
  __syncthreads();
  const int wpc = 4;                                  // C4 <= 64 (launcher): every wave holds all C4 columns
  const int64_t slot = (int64_t)((blockIdx.y * gridDim.x + blockIdx.x) % nslots) * slot_stride;
  for (int idx = threadIdx.x; idx < 10 * g.C4 * 4; idx += 256) {
    const int q = idx & 3, cc = (idx >> 2) % g.C4, k = (idx >> 2) / g.C4;
    float t = 0.f;
    for (int wv = 0; wv < wpc; ++wv) t += red[(wv * 10 + k) * g.C4 + cc][q];
    atomicAdd((k < 9 ? dW + slot + (int64_t)k * g.C4 * 4 : db + slot) + cc * 4 + q, t);
  }
}

#undef RING

// Backward for the same small maps (ReLU mask in the LSB of dt2, see k_dw_bwd_ring<true>): the block's images of raw
// dt2 go to LDS, thread (image, y, c4) fetches its own t0 row straight into registers (no neighbourhood needed) and walks
// the row with a 3x3 window of d1 = (dt2 * gate + dgap / hw) * mask, formed as the window is filled.
// XS = 2: two threads share an output row (half of it each), 512 threads per block -- the 16-wide x 64-channel case, where
// one thread per row needs more than 256 registers
template <int W_, int C4, int XS, typename T>
__global__ void __launch_bounds__(256 * XS, XS == 2 ? 1 : 2) k_dw_bwd_img(const V4<T> dt2, const V4<T> t0,
                                                       const f32x4* __restrict__ w, const f32x4* __restrict__ gate,
                                                       const f32x4* __restrict__ dgap, const V4<T> dt0,
                                                       float* __restrict__ dW, float* __restrict__ db, int H,
                                                       float inv_hw, int nslots, int64_t slot_stride) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  f32x4* tile = reinterpret_cast<f32x4*>(dyn_lds);             // [IPB][H][W_][C4] raw dt2
  constexpr int total = 256 * W_, NTHR = 256 * XS, XL = W_ / XS;       // XL = columns per thread
  const int64_t base = (int64_t)blockIdx.x * total;
  const int rowid = threadIdx.x % 256, x0 = (threadIdx.x / 256) * XL;
  const int c4 = rowid % C4, y = (rowid / C4) % H, img = rowid / (C4 * H);
  const int ipi = H * W_ * C4, ipb = 256 / (C4 * H);
  // the t0 row rides a 4-deep register FIFO (fetched four steps ahead of its use): the whole row next to the 19 float4
  // accumulators and the window does not fit 256 registers
  constexpr int FD = XL < 4 ? XL : 4;
  typename V4<T>::raw ld[XL], t0q[FD];
#pragma unroll
  for (int j = 0; j < XL; ++j) ld[j] = dt2.ld(base + threadIdx.x + NTHR * j);
  const V4<T> t0p = t0 + (base + img * ipi + (y * W_ + x0) * C4 + c4);
#pragma unroll
  for (int x = 0; x < FD; ++x) t0q[x] = t0p.ld(x * C4);
  const int64_t bimg = (int64_t)blockIdx.x * ipb + img;
  const f32x4 gg = gate[bimg * C4 + c4];
  const f32x4 dg = dgap[bimg * C4 + c4] * inv_hw;
  f32x4 wt[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wt[k] = w[k * C4 + c4];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < XL; ++j) tile[threadIdx.x + NTHR * j] = V4<T>::cv(ld[j]);
  __syncthreads();
  f32x4 aw[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) aw[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 ab = {0.f, 0.f, 0.f, 0.f};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // win[a][e] = d1(y - a + 1, x - e + 1): the position read by output pixel (y,x)'s transposed tap (a,e)
  const f32x4* rowp[3];
  bool rok[3];
  unsigned rokm[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int yy = y - a + 1;
    rok[a] = yy >= 0 && yy < H;
    rokm[a] = rok[a] ? (sizeof(T) == 2 ? 0x10000u : 1u) : 0u;
    rowp[a] = tile + img * ipi + ((rok[a] ? yy : y) * W_ + x0) * C4 + c4;      // column x0 of the row
  }
  auto d1_at = [&](int a, int xx) {                            // column x0 + xx, inside the row
    const f32x4 r = rowp[a][xx * C4];
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {                              // branch-free: one select on (row valid) & (mask bit)
      const unsigned bits = __float_as_uint(r[q]);
      const float val = sizeof(T) == 2 ? __uint_as_float(bits & ~0x10000u) : r[q];   // bf16: the bit is not part of the value
      const float t = val * gg[q] + dg[q];
      v[q] = (bits & rokm[a]) != 0u ? t : 0.f;
    }
    return v;
  };
  const bool left = x0 > 0, right = x0 + XL < W_;              // a neighbour column beyond this thread's run exists
  f32x4 win[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const f32x4 l = d1_at(a, left ? -1 : 0);
    win[a][1] = left ? l : zero;
    win[a][0] = d1_at(a, 0);
  }
  const V4<T> orow = dt0 + (base + img * ipi + (y * W_ + x0) * C4 + c4);
#pragma unroll
  for (int x = 0; x < XL; ++x) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      win[a][2] = win[a][1]; win[a][1] = win[a][0];
      if (x + 1 < XL) win[a][0] = d1_at(a, x + 1);
      else if (XS == 1) win[a][0] = zero;
      else { const f32x4 r = d1_at(a, right ? XL : XL - 1); win[a][0] = right ? r : zero; }
    }
    const f32x4 tv = V4<T>::cv(t0q[x % FD]);
    if (x + FD < XL) t0q[x % FD] = t0p.ld((x + FD) * C4);
    f32x4 acc = zero;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        acc += wt[a * 3 + e] * win[a][e];
        aw[a * 3 + e] += tv * win[a][e];
      }
    ab += win[1][1];
    f32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = tv[q] > 0.f ? acc[q] : 0.f;
    orow.st(x * C4, r);
    __builtin_amdgcn_sched_barrier(0);                 // keeps the scheduler from hoisting every later window read up here
  }
  // ---- block reduction of the 10 float4 accumulators over the threads that share c4, one atomic set per block
  __syncthreads();
  f32x4* red = tile;                                   // 4 * XS * 10 * C4 float4 <= 256 * W_
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    f32x4 v = k < 9 ? aw[k < 9 ? k : 0] : ab;
    for (int off = C4; off < 64; off <<= 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] += __shfl_xor(v[q], off, 64);
    }
    if (lane < C4) red[(wave * 10 + k) * C4 + lane] = v;
  }
  __syncthreads();
  const int64_t slot = (int64_t)(blockIdx.x % nslots) * slot_stride;
  for (int idx = threadIdx.x; idx < 10 * C4 * 4; idx += NTHR) {
    const int q = idx & 3, cc = (idx >> 2) % C4, k = (idx >> 2) / C4;
    float t = 0.f;
    for (int wv = 0; wv < 4 * XS; ++wv) t += red[(wv * 10 + k) * C4 + cc][q];
    atomicAdd((k < 9 ? dW + slot + (int64_t)k * C4 * 4 : db + slot) + cc * 4 + q, t);
  }
}

static bool dw_geom(int H, int W, int C, DwGeom* g, size_t* lds) {
  if (C % 4) return false;
  const int C4 = C / 4;
  if (C4 > 256 || (256 % C4) != 0) return false;
  int XS = 512 / C4;
  if (XS > W) XS = W;
  if (XS < 1) return false;
  g->H = H; g->W = W; g->C4 = C4; g->XS = XS; g->strips = (W + XS - 1) / XS;
  *lds = (size_t)4 * (XS + 2) * C4 * sizeof(f32x4);
  if (*lds < 256 * sizeof(f32x4)) *lds = 256 * sizeof(f32x4);
  return *lds <= 64 * 1024;
}

bool dw_uses_img(bool backward, bool mask_in_lsb, int B, int H, int W, int C) {
  if (!dw_img_shape(B, H, W, C)) return false;
  if (!backward) return true;
  return mask_in_lsb && (W == 16 ? 80 : 40) * (C / 4) <= 256 * W;   // the reduction scratch (waves x 10 x C4 float4) fits the tile
}

// t1 = relu(dw(t0) + b) and gap = mean_hw(t1) in one pass.  false = shape not covered.
template <typename T>
static bool run_dw_fwd_gap(const T* in, const float* w, const float* b, T* out, float* gap, int B, int H, int W, int C,
                           hipStream_t s) {
  if (dw_uses_img(false, false, B, H, W, C)) {
    const int ipb = 256 / (H * (C / 4));
    const dim3 grid((unsigned)(B / ipb));
    const size_t bytes = (size_t)256 * W * sizeof(f32x4);
    const float inv = 1.0f / (float)(H * W);
#define MVAE_DWI(W_)                                                                                                  \
  do {                                                                                                                \
    if (C == 64) hipLaunchKernelGGL((k_dw_fwd_img<W_, 16, T>), grid, dim3(256), bytes, s, V4<T>(in),                 \
                                    (const f32x4*)w, (const f32x4*)b, V4<T>(out), gap, H, inv);                      \
    else hipLaunchKernelGGL((k_dw_fwd_img<W_, 8, T>), grid, dim3(256), bytes, s, V4<T>(in), (const f32x4*)w,         \
                            (const f32x4*)b, V4<T>(out), gap, H, inv);                                               \
  } while (0)
    if (W == 16) MVAE_DWI(16); else if (W == 8) MVAE_DWI(8); else MVAE_DWI(4);
#undef MVAE_DWI
    return true;
  }
  DwGeom g;
  size_t lds;
  if (!dw_geom(H, W, C, &g, &lds) || (H % 4) != 0) return false;
  // row segments until enough work items exist (bf16: ~1024 = four blocks per CU by LDS, half the bytes per block in
  // flight; f32: 512, the launch shape the 32x32 headline configuration was tuned on); every segment re-reads two halo rows
  static const int fwd_items = [] { const char* e = getenv("MVAE_DW_FWD_ITEMS"); return e ? atoi(e) : 0; }();
  const int64_t target = fwd_items > 0 ? fwd_items : (sizeof(T) == 2 ? 1024 : 512);
  int nseg = 1;
  while (!det_mode() && (int64_t)B * g.strips * nseg < target && (H / (nseg * 2)) % 4 == 0 && H / (nseg * 2) >= 8) nseg *= 2;
  if (det_mode() && g.strips > 1) return false;          // several blocks per image would add their GAP shares atomically
  if ((int64_t)B * nseg > 65535) return false;
  if (g.strips * nseg > 1) launch_zero(gap, (int64_t)B * C, s);
  hipLaunchKernelGGL((k_dw_fwd_ring<true, T>), dim3(g.strips, B * nseg), dim3(256), lds, s, V4<T>(in), (const f32x4*)w,
                     (const f32x4*)b, V4<T>(out), gap, g, 1.0f / (float)(H * W), nseg, H / nseg);
  return true;
}
bool launch_dw_fwd_gap(const float* in, const float* w, const float* b, float* out, float* gap, int B, int H, int W,
                       int C, hipStream_t s, bool bf) {
  if (bf) return run_dw_fwd_gap<bf16_t>((const bf16_t*)in, w, b, (bf16_t*)out, gap, B, H, W, C, s);
  return run_dw_fwd_gap<float>(in, w, b, out, gap, B, H, W, C, s);
}

// fused backward through Multiply/GAP/ReLU + depthwise backward-data + depthwise weight/bias gradients.
template <typename T>
static bool run_dw_bwd_fused(const T* dt2, const T* t1, const T* t0, const float* w, const float* gate,
                             const float* dgap, T* dt0, float* dW, float* db, GradSlots sl, bool mask_in_lsb, int B, int H,
                             int W, int C, hipStream_t s) {
  if (dw_uses_img(true, mask_in_lsb, B, H, W, C)) {
    const int ipb = 256 / (H * (C / 4));
    const dim3 grid((unsigned)(B / ipb));
    if (det_mode() && (int)grid.x > sl.count()) return false;   // one gradient slot per block
    const size_t bytes = (size_t)256 * W * sizeof(f32x4);
    const float inv = 1.0f / (float)(H * W);
#define MVAE_DWB(W_)                                                                                                  \
  do {                                                                                                                \
    constexpr int XS = W_ == 16 ? 2 : 1;                    /* 16-wide rows: two threads per row, 512 per block */    \
    if (C == 64) hipLaunchKernelGGL((k_dw_bwd_img<W_, 16, XS, T>), grid, dim3(256 * XS), bytes, s, V4<T>(dt2),       \
                                    V4<T>(t0), (const f32x4*)w, (const f32x4*)gate, (const f32x4*)dgap,              \
                                    V4<T>(dt0), sl.at(dW), sl.at(db), H, inv, sl.count(), sl.stride);                \
    else hipLaunchKernelGGL((k_dw_bwd_img<W_, 8, XS, T>), grid, dim3(256 * XS), bytes, s, V4<T>(dt2),                \
                            V4<T>(t0), (const f32x4*)w, (const f32x4*)gate, (const f32x4*)dgap, V4<T>(dt0),          \
                            sl.at(dW), sl.at(db), H, inv, sl.count(), sl.stride);                                    \
  } while (0)
    if (W == 16) MVAE_DWB(16); else if (W == 8) MVAE_DWB(8); else MVAE_DWB(4);
#undef MVAE_DWB
    return true;
  }
  DwGeom g;
  size_t lds;
  if (!dw_geom(H, W, C, &g, &lds) || g.C4 > 64) return false;
  const size_t red_bytes = (size_t)40 * g.C4 * sizeof(f32x4);          // 4 waves x 10 accumulators x C4
  if (lds < red_bytes) lds = red_bytes;
  // split images into row segments until 512 work items exist = one round of two blocks per CU (each segment costs two
  // halo rows of re-reads and one exposed prologue: 1024 items measured 82 us against 76 us at B = 512, 32x32x64); the
  // kernel needs an even number of rows per segment
  if (H % 2) return false;
  static const int bwd_items = [] { const char* e = getenv("MVAE_DW_BWD_ITEMS"); return e ? atoi(e) : 0; }();
  const int64_t btarget = bwd_items > 0 ? bwd_items : 512;
  int nseg = 1;
  while ((int64_t)B * g.strips * nseg < btarget && (H / (nseg * 2)) % 2 == 0 && H / (nseg * 2) >= 4) nseg *= 2;
  const int RS = H / nseg;
  int64_t work = (int64_t)B * nseg;
  int gy = (int)(work < kDwMaxBlocks / g.strips ? work : kDwMaxBlocks / g.strips);
  if (gy < 1) return false;
  if (mask_in_lsb)
    hipLaunchKernelGGL((k_dw_bwd_ring<true, T>), dim3(g.strips, gy), dim3(256), lds, s, V4<T>(dt2), V4<T>(t1),
                       V4<T>(t0), (const f32x4*)w, (const f32x4*)gate, (const f32x4*)dgap, V4<T>(dt0), sl.at(dW),
                       sl.at(db), g, 1.0f / (float)(H * W), B, RS, nseg, sl.count(), sl.stride);
  else
    hipLaunchKernelGGL((k_dw_bwd_ring<false, T>), dim3(g.strips, gy), dim3(256), lds, s, V4<T>(dt2), V4<T>(t1),
                       V4<T>(t0), (const f32x4*)w, (const f32x4*)gate, (const f32x4*)dgap, V4<T>(dt0), sl.at(dW),
                       sl.at(db), g, 1.0f / (float)(H * W), B, RS, nseg, sl.count(), sl.stride);
  return true;
}
// bf: bfloat16 storage (mask_in_lsb: the producer k16_dual rounded dt2 one bit shorter and put the mask in the bf16 LSB)
bool launch_dw_bwd_fused(const float* dt2, const float* t1, const float* t0, const float* w, const float* gate,
                         const float* dgap, float* dt0, float* dW, float* db, GradSlots sl, bool mask_in_lsb, int B, int H,
                         int W, int C, hipStream_t s, bool bf) {
  if (bf)
    return run_dw_bwd_fused<bf16_t>((const bf16_t*)dt2, (const bf16_t*)t1, (const bf16_t*)t0, w, gate, dgap,
                                    (bf16_t*)dt0, dW, db, sl, mask_in_lsb, B, H, W, C, s);
  return run_dw_bwd_fused<float>(dt2, t1, t0, w, gate, dgap, dt0, dW, db, sl, mask_in_lsb, B, H, W, C, s);
}

}  // namespace mvae
