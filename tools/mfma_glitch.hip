// Does a heavy bf16-MFMA kernel on one stream disturb the results of a plain VALU kernel running beside it?
// (kernels_split.hip debugging: dense_expand outputs came back wrong in one 16-lane group of one register when the
// split-bf16 convolution of another pyramid scale ran concurrently -- and only then.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 2) burner(float* sink, int iters, int use_lds) {
  __shared__ __attribute__((aligned(16))) char lds[49152];
  for (int i = threadIdx.x; i < 49152 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + i;
  __syncthreads();
  f32x16 acc[2];
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  for (int it = 0; it < iters; ++it) {
    if (use_lds) {
      a = *reinterpret_cast<const u32x4*>(lds + ((threadIdx.x * 16 + it * 1024) % 49152));
      b = *reinterpret_cast<const u32x4*>(lds + ((threadIdx.x * 16 + it * 2048 + 512) % 49152));
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), acc[1], 0, 0, 0);
    }
  }
  float t = 0.f;
  for (int r = 0; r < 16; ++r) t += acc[0][r] + acc[1][r];
  if (t == 12345.678f) sink[0] = t;
}

// the dense_expand pattern: out[b, n4] = bias[n4] + sum_k W[k][n4] * z[b][k]   (float4 per thread, LDS-broadcast z)
__global__ void __launch_bounds__(256) victim(const float* __restrict__ z, const f32x4* __restrict__ W,
                                              const f32x4* __restrict__ bias, f32x4* __restrict__ out, int B, int Z, int N4) {
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
  f32x4 acc[4] = {bv, bv, bv, bv};
  for (int k = 0; k < Z; ++k) {
    const f32x4 w = W[k * N4 + n4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += w * sz[r][k];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (b0 + r < B) out[(b0 + r) * N4 + n4] = acc[r];
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200, use_lds = argc > 2 ? atoi(argv[2]) : 1, burn = argc > 3 ? atoi(argv[3]) : 1;
  const int B = 2, Z = 16, N = 32768, N4 = N / 4;
  std::vector<float> hz(B * Z), hW((size_t)Z * N), hb(N), ref((size_t)B * N), got((size_t)B * N);
  srand(1);
  for (auto& v : hz) v = (rand() / (float)RAND_MAX - 0.5f);
  for (auto& v : hW) v = (rand() / (float)RAND_MAX - 0.5f) * 0.03f;
  for (auto& v : hb) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
  float *dz, *dW, *db, *dout, *sink;
  hipMalloc(&dz, hz.size() * 4); hipMalloc(&dW, hW.size() * 4); hipMalloc(&db, hb.size() * 4);
  hipMalloc(&dout, ref.size() * 4); hipMalloc(&sink, 64);
  hipMemcpy(dz, hz.data(), hz.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  const dim3 grid((N4 + 255) / 256, (B + 3) / 4);
  // reference = the same kernel on an idle GPU
  hipLaunchKernelGGL(victim, grid, dim3(256), 0, s2, dz, (const f32x4*)dW, (const f32x4*)db, (f32x4*)dout, B, Z, N4);
  hipStreamSynchronize(s2);
  hipMemcpy(ref.data(), dout, ref.size() * 4, hipMemcpyDeviceToHost);
  long bad_events = 0, bad_elems = 0;
  const int NV = 64;
  float* dmany; hipMalloc(&dmany, (size_t)NV * ref.size() * 4);
  std::vector<float> many((size_t)NV * ref.size());
  for (int rep = 0; rep < reps; ++rep) {
    hipMemsetAsync(dmany, 0, many.size() * 4, s2);
    hipStreamSynchronize(s2);
    if (burn) hipLaunchKernelGGL(burner, dim3(512), dim3(256), 0, s1, sink, 6000, use_lds);     // ~2 ms of MFMAs
    for (int k = 0; k < NV; ++k)         // victims back to back while the burner runs, each into its own buffer
      hipLaunchKernelGGL(victim, grid, dim3(256), 0, s2, dz, (const f32x4*)dW, (const f32x4*)db, (f32x4*)(dmany + (size_t)k * ref.size()), B, Z, N4);
    hipStreamSynchronize(s2);
    hipStreamSynchronize(s1);
    hipMemcpy(many.data(), dmany, many.size() * 4, hipMemcpyDeviceToHost);
    for (int k = 0; k < NV; ++k) {
      long nb = 0; long first = -1;
      const float* got = many.data() + (size_t)k * ref.size();
      for (size_t i = 0; i < ref.size(); ++i) if (got[i] != ref[i]) { if (first < 0) first = (long)i; ++nb; }
      if (nb) { ++bad_events; bad_elems += nb; if (bad_events <= 8) printf("rep %d launch %d: %ld wrong elements, first at %ld (got %.8g ref %.8g)\n", rep, k, nb, first, got[first], ref[first]); }
    }
  }
  printf("reps %d use_lds %d burn %d: bad launches %ld, wrong elements %ld\n", reps, use_lds, burn, bad_events, bad_elems);
  return 0;
}
