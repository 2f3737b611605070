// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ldm_hip.h"

#define LDM_CHECK_LAUNCH()                          \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return (int)e__;         \
  } while (0)

typedef unsigned short bf16_t;  // raw bfloat16 storage

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((unsigned)v) << 16);
}

// round-to-nearest-even; NaN stays NaN (quiet)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

template <typename T> struct Elt;
template <> struct Elt<float> {
  static __device__ __forceinline__ float ld(const float *p) { return *p; }
  static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct Elt<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t *p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t *p, float v) { *p = f32_to_bf16(v); }
};

// 4-element vector access (16 B for float, 8 B for bf16)
struct f4 { float v[4]; };
__device__ __forceinline__ f4 ld4(const float *p) {
  float4 t = *reinterpret_cast<const float4 *>(p);
  return f4{{t.x, t.y, t.z, t.w}};
}
__device__ __forceinline__ f4 ld4(const bf16_t *p) {
  uint2 t = *reinterpret_cast<const uint2 *>(p);
  return f4{{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u),
             __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u)}};
}
__device__ __forceinline__ void st4(float *p, const f4 &a) {
  *reinterpret_cast<float4 *>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
}
__device__ __forceinline__ void st4(bf16_t *p, const f4 &a) {
  uint2 t;
  t.x = (unsigned)f32_to_bf16(a.v[0]) | ((unsigned)f32_to_bf16(a.v[1]) << 16);
  t.y = (unsigned)f32_to_bf16(a.v[2]) | ((unsigned)f32_to_bf16(a.v[3]) << 16);
  *reinterpret_cast<uint2 *>(p) = t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline bool aligned_to(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
