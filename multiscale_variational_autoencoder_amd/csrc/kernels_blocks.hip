// kernels_blocks.hip -- glue kernels of the stand-alone block library: mobilenetV2_block
// (layer_blocks.py:468-550) and resnet_block (:789-887) are assembled in runtime.cpp from the convolution / depthwise
// launchers of the hot path (kernels.h); what those do not offer -- ReLU backward as its own pass, an activation applied
// after a residual add, the depthwise backward-data without a ReLU mask behind it -- lives here.
#include "kernels.h"
#include "prof.h"

namespace mvae {

namespace {
constexpr int kBlk = 256;
inline unsigned grid_of(int64_t n) { int64_t g = (n + kBlk - 1) / kBlk; return (unsigned)(g < 1 ? 1 : (g > 65535 * 16 ? 65535 * 16 : g)); }
}

// out = dy * (y > 0)
__global__ void __launch_bounds__(256) k_relu_bwd(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out,
                                                  int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk)
    out[i] = y[i] > 0.f ? dy[i] : 0.f;
}
// y = act(y) (+ r): ReLU in place, optionally followed by an add
__global__ void __launch_bounds__(256) k_relu_add(float* __restrict__ y, const float* __restrict__ r, int relu, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    float v = y[i];
    if (relu) v = v > 0.f ? v : 0.f;
    y[i] = r ? v + r[i] : v;
  }
}
// out = a + b
__global__ void __launch_bounds__(256) k_add2(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                              int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) out[i] = a[i] + b[i];
}
// dx = dw^T(dy): the transposed 3x3 depthwise stencil (SAME, stride 1), no mask
__global__ void __launch_bounds__(256) k_dw_bwd_plain(const float* __restrict__ dy, const float* __restrict__ w,
                                                      float* __restrict__ dx, int B, int H, int W, int C) {
  const int64_t n = (int64_t)B * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int x = (int)(p % W), y = (int)((p / W) % H);
    const int64_t b = p / ((int64_t)W * H);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int yy = y - a + 1;                       // output pixel (yy, xx) read this position through tap (a, e)
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int xx = x - e + 1;
        if (xx < 0 || xx >= W) continue;
        acc += w[(a * 3 + e) * C + c] * dy[((b * H + yy) * W + xx) * C + c];
      }
    }
    dx[i] = acc;
  }
}

void launch_relu_bwd(const float* dy, const float* y, float* out, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_relu_bwd, dim3(grid_of(n)), dim3(kBlk), 0, s, dy, y, out, n);
}
void launch_relu_add(float* y, const float* r, bool relu, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_relu_add, dim3(grid_of(n)), dim3(kBlk), 0, s, y, r, relu ? 1 : 0, n);
}
void launch_add2(const float* a, const float* b, float* out, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_add2, dim3(grid_of(n)), dim3(kBlk), 0, s, a, b, out, n);
}
void launch_dw_bwd_plain(const float* dy, const float* w, float* dx, int B, int H, int W, int C, hipStream_t s) {
  hipLaunchKernelGGL(k_dw_bwd_plain, dim3(grid_of((int64_t)B * H * W * C)), dim3(kBlk), 0, s, dy, w, dx, B, H, W, C);
}

}  // namespace mvae
