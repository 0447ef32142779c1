// kernels_dw.hip -- depthwise 3x3 (layer_blocks.py:604-614) as sliding-row kernels: one workgroup walks an image
// (or a column strip of it) top to bottom with a 4-row ring in LDS, so every input row comes from HBM exactly once
// and the 3x3 neighbourhood is served from LDS.
//   forward : t1 = relu(dw(t0) + b)  and  gap[b,c] = mean_hw t1            (DepthwiseConv2D + GlobalAveragePooling2D)
//   backward: d1 = (dt2 * g[b,c] + dgap[b,c]/hw) * (t1 > 0)                (through Multiply, GAP, ReLU -- on the fly)
//             dt0 = dwT(d1) * (t0 > 0) ;  dW[a][e][c] += sum t0 * d1(shifted) ;  db[c] += sum d1
// replacing five separate full-tensor launches (dw_fwd, spatial_sum; mn_dt1pre, dw_wgrad, dw_bwd_data).
#include "kernels.h"

namespace mvae {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// geometry shared by both kernels: a block owns image b = blockIdx.y and columns [x0, x0 + XS) of it
struct DwGeom { int H, W, C4, XS, strips; };

// ring[slot][xs][c4] as float4, xs in [0, XS + 2): column xs <-> image column x0 - 1 + xs
#define RING(slot, xs, c4) ring[((slot) * XSP + (xs)) * g.C4 + (c4)]

template <bool FUSE_GAP>
__global__ void __launch_bounds__(256) k_dw_fwd_ring(const f32x4* __restrict__ in, const f32x4* __restrict__ w,
                                                     const f32x4* __restrict__ bias, f32x4* __restrict__ out,
                                                     float* __restrict__ gap, DwGeom g, float inv_hw) {
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
  f32x4* ring = reinterpret_cast<f32x4*>(dyn_lds);
  const int XSP = g.XS + 2;
  const int b = blockIdx.y, x0 = blockIdx.x * g.XS;
  const int xs_n = min(g.XS, g.W - x0);               // columns this strip really has
  const int c4 = threadIdx.x % g.C4;
  const int items = g.XS * g.C4;                      // <= 512: at most two items per thread
  f32x4 wt[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wt[k] = w[k * g.C4 + c4];
  const f32x4 bs = bias[c4];
  const f32x4* img = in + (int64_t)b * g.H * g.W * g.C4;
  f32x4* oimg = out + (int64_t)b * g.H * g.W * g.C4;
  const int row_items = XSP * g.C4;

  // rows are fetched into registers one iteration before they are stored to the ring (see k_dw_bwd_ring)
  f32x4 rv[3];
  auto fetch_row = [&](int y) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int t = threadIdx.x + 256 * u;
      rv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t < row_items) {
        const int x = x0 - 1 + t / g.C4;
        if (x >= 0 && x < g.W) rv[u] = img[((int64_t)y * g.W + x) * g.C4 + c4];
      }
    }
  };
  auto store_row = [&](int y) {                       // row y (with column halo) -> ring slot y & 3
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int t = threadIdx.x + 256 * u;
      if (t < row_items) RING(y & 3, t / g.C4, c4) = rv[u];
    }
  };
  f32x4 gsum[2];
  gsum[0] = gsum[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  fetch_row(0); store_row(0);
  if (g.H > 1) fetch_row(1);
  for (int y = 0; y < g.H; ++y) {
    if (y + 1 < g.H) store_row(y + 1);
    if (y + 2 < g.H) fetch_row(y + 2);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = threadIdx.x + 256 * k;
      if (idx < items) {
        const int xl = idx / g.C4;                    // local column, image column x0 + xl, ring column xl + 1
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
          oimg[((int64_t)y * g.W + x0 + xl) * g.C4 + c4] = acc;
          if (FUSE_GAP) gsum[k] += acc;
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
      if (g.strips == 1) {
        *reinterpret_cast<f32x4*>(gp) = t;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicAdd(gp + q, t[q]);
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_dw_bwd_ring(const f32x4* __restrict__ dt2, const f32x4* __restrict__ t1,
                                                     const f32x4* __restrict__ t0, const f32x4* __restrict__ w,
                                                     const f32x4* __restrict__ gate, const f32x4* __restrict__ dgap,
                                                     f32x4* __restrict__ dt0, f32x4* __restrict__ partial, DwGeom g,
                                                     float inv_hw, int B, int RS, int nseg) {
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
    // a row of d1 is fetched in two steps: raw loads into registers (issued two rows ahead), and -- one iteration
    // later, after the compute of the current row -- the gate/ReLU arithmetic and the LDS store.  row_items <= 3*256.
    f32x4 rd[3], ra[3];
    const f32x4 gg_c = gate[(int64_t)b * g.C4 + c4];       // (row_items % C4 == 0 and 256 % C4 == 0: cc == c4)
    const f32x4 dg_c = dgap[(int64_t)b * g.C4 + c4] * inv_hw;
    auto fetch_row = [&](int y) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int t = threadIdx.x + 256 * u;
        rd[u] = ra[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < row_items) {
          const int x = x0 - 1 + t / g.C4;
          if (x >= 0 && x < g.W) {
            const int64_t o = ioff + ((int64_t)y * g.W + x) * g.C4 + c4;
            rd[u] = dt2[o]; ra[u] = t1[o];
          }
        }
      }
    };
    auto store_row = [&](int y) {                     // d1 row y (with column halo) -> ring slot y & 3
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int t = threadIdx.x + 256 * u;
        if (t < row_items) {
          f32x4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = ra[u][q] > 0.f ? rd[u][q] * gg_c[q] + dg_c[q] : 0.f;
          RING(y & 3, t / g.C4, c4) = v;
        }
      }
    };
    __syncthreads();                                  // previous item's ring reads are done
    if (ya > 0) { fetch_row(ya - 1); store_row(ya - 1); }
    fetch_row(ya); store_row(ya);
    if (ya + 1 < g.H) fetch_row(ya + 1);              // in flight across the first iteration
    for (int y = ya; y < yb; ++y) {
      // t0 of this row is independent of the ring: issue it before the barrier so both latencies overlap
      const int64_t o0 = ioff + ((int64_t)y * g.W + x0 + xl0) * g.C4 + c4;
      const int64_t o1 = ioff + ((int64_t)y * g.W + x0 + xl1) * g.C4 + c4;
      f32x4 tv[2];
      tv[0] = has0 ? t0[o0] : f32x4{0.f, 0.f, 0.f, 0.f};
      tv[1] = has1 ? t0[o1] : f32x4{0.f, 0.f, 0.f, 0.f};
      if (y + 1 < g.H) store_row(y + 1);              // fetched during the previous iteration
      if (y + 2 < g.H && y + 1 < yb) fetch_row(y + 2);
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (k == 0 ? has0 : has1) {
          const int xl = k == 0 ? xl0 : xl1;
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
              aw[a * 3 + e] += tv[k] * sv;
            }
          }
          ab += RING(y & 3, xl + 1, c4);
          f32x4 r;
#pragma unroll
          for (int q = 0; q < 4; ++q) r[q] = tv[k][q] > 0.f ? acc[q] : 0.f;
          dt0[k == 0 ? o0 : o1] = r;
        }
      }
    }
  }
  // ---- block reduction of the 10 float4 accumulators over the threads that share c4; the block's partial goes
  //      out with plain stores (summed by k_dw_partials: deterministic, no contended float atomics)
  f32x4* red = ring;                                  // needs 256 float4 = 4 KB <= ring size (checked by launcher)
  f32x4* pout = partial + (int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 10 * g.C4;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    __syncthreads();
    red[threadIdx.x] = k < 9 ? aw[k < 9 ? k : 0] : ab;
    __syncthreads();
    if (threadIdx.x < g.C4) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      for (int r = threadIdx.x; r < 256; r += g.C4) t += red[r];
      pout[k * g.C4 + threadIdx.x] = t;
    }
  }
}

// dW[k][c] += sum_blocks partial[blk][k][c] (k < 9) ; db[c] += sum_blocks partial[blk][9][c]
// blockIdx.y splits the block axis into 32 chunks (each thread sums <= nblk/32 partials, then one float atomic)
__global__ void __launch_bounds__(256) k_dw_partials(const float* __restrict__ partial, int nblk, float* __restrict__ dW,
                                                     float* __restrict__ db, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 10 * C) return;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  int bk = b0;
  for (; bk + 3 < b1; bk += 4) {
    t0 += partial[(int64_t)bk * 10 * C + i];
    t1 += partial[(int64_t)(bk + 1) * 10 * C + i];
    t2 += partial[(int64_t)(bk + 2) * 10 * C + i];
    t3 += partial[(int64_t)(bk + 3) * 10 * C + i];
  }
  for (; bk < b1; ++bk) t0 += partial[(int64_t)bk * 10 * C + i];
  const float t = (t0 + t1) + (t2 + t3);
  if (b1 > b0) atomicAdd(i < 9 * C ? &dW[i] : &db[i - 9 * C], t);
}
#undef RING

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

// t1 = relu(dw(t0) + b) and gap = mean_hw(t1) in one pass.  false = shape not covered.
bool launch_dw_fwd_gap(const float* in, const float* w, const float* b, float* out, float* gap, int B, int H, int W,
                       int C, hipStream_t s) {
  DwGeom g;
  size_t lds;
  if (!dw_geom(H, W, C, &g, &lds) || B > 65535) return false;
  if (g.strips > 1) launch_zero(gap, (int64_t)B * C, s);
  hipLaunchKernelGGL(k_dw_fwd_ring<true>, dim3(g.strips, B), dim3(256), lds, s, (const f32x4*)in, (const f32x4*)w,
                     (const f32x4*)b, (f32x4*)out, gap, g, 1.0f / (float)(H * W));
  return true;
}

// fused backward through Multiply/GAP/ReLU + depthwise backward-data + depthwise weight/bias gradients.
// `partial` is a scratch buffer of kDwMaxBlocks * 10 * C floats.
bool launch_dw_bwd_fused(const float* dt2, const float* t1, const float* t0, const float* w, const float* gate,
                         const float* dgap, float* dt0, float* dW, float* db, float* partial, int B, int H, int W,
                         int C, hipStream_t s) {
  DwGeom g;
  size_t lds;
  if (!dw_geom(H, W, C, &g, &lds) || !partial) return false;
  // split images into row segments until ~2048 work items exist (each costs two halo rows of re-reads)
  int nseg = 1;
  while ((int64_t)B * g.strips * nseg < 2048 && H / (nseg * 2) >= 2) nseg *= 2;
  const int RS = (H + nseg - 1) / nseg;
  nseg = (H + RS - 1) / RS;
  int64_t work = (int64_t)B * nseg;
  int gy = (int)(work < kDwMaxBlocks / g.strips ? work : kDwMaxBlocks / g.strips);
  if (gy < 1) return false;
  hipLaunchKernelGGL(k_dw_bwd_ring, dim3(g.strips, gy), dim3(256), lds, s, (const f32x4*)dt2, (const f32x4*)t1,
                     (const f32x4*)t0, (const f32x4*)w, (const f32x4*)gate, (const f32x4*)dgap, (f32x4*)dt0,
                     (f32x4*)partial, g, 1.0f / (float)(H * W), B, RS, nseg);
  hipLaunchKernelGGL(k_dw_partials, dim3((10 * C + 255) / 256, 32), dim3(256), 0, s, partial, g.strips * gy, dW, db, C);
  return true;
}

}  // namespace mvae
