// act16.h -- activation storage types.  MVAE_ACT_BF16 keeps the wide [M,c] activations, saved tensors and activation
// gradients as bfloat16 in HBM (include/mvae_hip.h); parameters, gradients, BatchNorm statistics, squeeze-excite
// vectors, latents and losses stay float32, and every kernel computes in float32 (or bf16 MFMA with f32 accumulation).
// The VALU kernels are templated on the storage type T (float or bf16_t) and touch memory only through V4<T>: a pointer
// that is indexed in units of FOUR elements (16 B of float, 8 B of bf16) and hands out / takes f32x4 values.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvae {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
struct bf16_t { uint16_t v; };

__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
// two floats -> packed bf16 pair (round to nearest even, NaN stays NaN): v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 p = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ f32x4_t unpack4(uint2 u) {
  return f32x4_t{bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y)};
}
__device__ __forceinline__ uint2 pack4(f32x4_t v) { return uint2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])}; }

// V4<T>::raw is what ONE load returns (f32x4 / uint2); raw(i) issues the load only, cv() converts.  Kernels keep their
// prefetch registers RAW and convert at the point of use: arithmetic between the loads of a batch (the bf16 -> f32
// shifts) makes hipcc wait for each load before it issues the next (load scheduling rule) -- the depthwise
// forward ran at 1.9 TB/s with converting loads.
template <typename T> struct V4;
template <> struct V4<float> {
  typedef f32x4_t raw;
  __device__ __forceinline__ raw ld(int64_t i) const { return reinterpret_cast<const f32x4_t*>(p)[i]; }
  static __device__ __forceinline__ f32x4_t cv(raw r) { return r; }
  float* p;
  __device__ __host__ V4(float* q = nullptr) : p(q) {}
  __device__ __host__ V4(const float* q) : p(const_cast<float*>(q)) {}
  __device__ __forceinline__ f32x4_t operator[](int64_t i) const { return reinterpret_cast<const f32x4_t*>(p)[i]; }
  __device__ __forceinline__ void st(int64_t i, f32x4_t v) const { reinterpret_cast<f32x4_t*>(p)[i] = v; }
  __device__ __host__ V4 operator+(int64_t i) const { return V4(p + 4 * i); }
  __device__ __host__ explicit operator bool() const { return p != nullptr; }
};
template <> struct V4<bf16_t> {
  typedef uint2 raw;
  __device__ __forceinline__ raw ld(int64_t i) const { return reinterpret_cast<const uint2*>(p)[i]; }
  static __device__ __forceinline__ f32x4_t cv(raw r) { return unpack4(r); }
  bf16_t* p;
  __device__ __host__ V4(bf16_t* q = nullptr) : p(q) {}
  __device__ __host__ V4(const bf16_t* q) : p(const_cast<bf16_t*>(q)) {}
  __device__ __forceinline__ f32x4_t operator[](int64_t i) const { return unpack4(reinterpret_cast<const uint2*>(p)[i]); }
  __device__ __forceinline__ void st(int64_t i, f32x4_t v) const { reinterpret_cast<uint2*>(p)[i] = pack4(v); }
  __device__ __host__ V4 operator+(int64_t i) const { return V4(p + 4 * i); }
  __device__ __host__ explicit operator bool() const { return p != nullptr; }
};

// scalar element access (slow paths, epilogues)
__device__ __forceinline__ float ld1(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ float ld1(const bf16_t* p, int64_t i) { return __uint_as_float((unsigned)p[i].v << 16); }
__device__ __forceinline__ void st1(float* p, int64_t i, float v) { p[i] = v; }
__device__ __forceinline__ void st1(bf16_t* p, int64_t i, float v) { p[i].v = (uint16_t)(pack_bf16(v, 0.f) & 0xFFFFu); }

// Launcher-side view of an activation tensor: the runtime keeps one arena of floats, a bf16 tensor simply occupies half
// the floats.  `bf` selects the instantiation.
struct ActPtr {
  void* p = nullptr;
  bool bf = false;
  const float* f32() const { return static_cast<const float*>(p); }
  float* f32w() const { return static_cast<float*>(p); }
  const bf16_t* b16() const { return static_cast<const bf16_t*>(p); }
  bf16_t* b16w() const { return static_cast<bf16_t*>(p); }
};

}  // namespace mvae
