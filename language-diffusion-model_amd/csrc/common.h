// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>

#include "../../include/ldm_hip.h"

#define LDM_CHECK_LAUNCH()                          \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return (int)e__;         \
  } while (0)

typedef unsigned short bf16_t;  // raw bfloat16 storage

// Attention-probability dropout (src/transformer.py:278 of the reference applies nn.Dropout to the
// softmax output).  Element (batch*head, query, key) is KEPT iff a 32-bit hash of its coordinates
// and the call's 64-bit seed is >= thresh = round(p * 2^32); kept elements are scaled by
// 1 / (1 - thresh / 2^32).  The mask is a pure function of the coordinates, so the forward and the
// backward kernels regenerate it in whatever layout they hold the probabilities in, and nothing is
// stored.  oracle/dropout_oracle.py restates these three functions.
__device__ __forceinline__ unsigned ldm_mix32(unsigned h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ unsigned ldm_drop_bh(unsigned seed_lo, unsigned bh) { return ldm_mix32(seed_lo + bh * 0x9e3779b1u); }
__device__ __forceinline__ unsigned ldm_drop_row(unsigned bhkey, unsigned seed_hi, unsigned q) {
  return ldm_mix32(bhkey ^ (seed_hi + q * 0x85ebca77u));
}
// The per-element step is one xorshift and ONE 24-bit multiply (v_mul_u32_u24, full rate; a 32-bit integer
// multiply is a quarter-rate instruction and the full mix above has two): it runs once per attention
// probability, where the full mix cost four times the softmax itself.  The row key in front of it is a full
// mix.  Constants chosen by search: keep rate and lag-1..64 correlations along keys, queries and heads within
// sampling noise of the full mix (2e-3 over 2M samples, p = 0.1 / 0.3 / 0.5).
__device__ __forceinline__ bool ldm_drop_keep(unsigned rowkey, unsigned k, unsigned thresh) {
  const unsigned x = rowkey + k * 0x5ed85e17u;
  return __umul24(x ^ (x >> 13), 0xcb5e35u) >= thresh;
}

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((unsigned)v) << 16);
}

// round-to-nearest-even, NaN stays a quiet NaN: the gfx950 packed convert (v_cvt_pk_bf16_f32)
typedef __bf16 ldm_bf2_t __attribute__((ext_vector_type(2)));
typedef float ldm_fl2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const ldm_fl2_t f{lo, hi};
  const ldm_bf2_t h = __builtin_convertvector(f, ldm_bf2_t);
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  const __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
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
  t.x = pack_bf16x2(a.v[0], a.v[1]);
  t.y = pack_bf16x2(a.v[2], a.v[3]);
  *reinterpret_cast<uint2 *>(p) = t;
}

// 8 consecutive elements kept in their storage form (16 B for bf16, 32 B for float)
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
  uint4 u;
  __device__ __forceinline__ void load(const bf16_t *p) { u = *reinterpret_cast<const uint4 *>(p); }
  // streaming forms: the line is not wanted in the caches after this access
  __device__ __forceinline__ void load_nt(const bf16_t *p) {
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    const u4_t t = __builtin_nontemporal_load(reinterpret_cast<const u4_t *>(p));
    u = make_uint4(t[0], t[1], t[2], t[3]);
  }
  static __device__ __forceinline__ void store_nt(bf16_t *p, const float (&o)[8]) {
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    u4_t t = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
    __builtin_nontemporal_store(t, reinterpret_cast<u4_t *>(p));
  }
  // opaque use: values derived from the registers before this point are not kept alive past it
  // (a multi-pass kernel re-derives them from the 16 stored bytes instead of holding 8 floats)
  __device__ __forceinline__ void pin() { asm volatile("" : "+v"(u.x), "+v"(u.y), "+v"(u.z), "+v"(u.w)); }
  // elements j >= j0 become -infinity (j0 >= 8: nothing changes; j0 <= 0: all of them)
  __device__ __forceinline__ void neg_inf_from(int j0) {
    unsigned *w = &u.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned x = k == 0 ? u.x : k == 1 ? u.y : k == 2 ? u.z : u.w;
      if (2 * k >= j0) x = (x & 0xffff0000u) | 0x0000ff80u;
      if (2 * k + 1 >= j0) x = (x & 0x0000ffffu) | 0xff800000u;
      if (k == 0) u.x = x; else if (k == 1) u.y = x; else if (k == 2) u.z = x; else u.w = x;
    }
    (void)w;
  }
  __device__ __forceinline__ float get(int j) const {
    const unsigned w = j < 2 ? u.x : j < 4 ? u.y : j < 6 ? u.z : u.w;
    return (j & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
  }
  static __device__ __forceinline__ void store(bf16_t *p, const float (&o)[8]) {
    uint4 t;
    t.x = pack_bf16x2(o[0], o[1]);
    t.y = pack_bf16x2(o[2], o[3]);
    t.z = pack_bf16x2(o[4], o[5]);
    t.w = pack_bf16x2(o[6], o[7]);
    *reinterpret_cast<uint4 *>(p) = t;
  }
};
template <> struct Raw8<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float *p) {
    a = *reinterpret_cast<const float4 *>(p);
    b = *reinterpret_cast<const float4 *>(p + 4);
  }
  __device__ __forceinline__ void load_nt(const float *p) { load(p); }
  static __device__ __forceinline__ void store_nt(float *p, const float (&o)[8]) { store(p, o); }
  __device__ __forceinline__ void pin() {}
  __device__ __forceinline__ void neg_inf_from(int j0) {
    const float ninf = -__builtin_inff();
    if (0 >= j0) a.x = ninf;
    if (1 >= j0) a.y = ninf;
    if (2 >= j0) a.z = ninf;
    if (3 >= j0) a.w = ninf;
    if (4 >= j0) b.x = ninf;
    if (5 >= j0) b.y = ninf;
    if (6 >= j0) b.z = ninf;
    if (7 >= j0) b.w = ninf;
  }
  __device__ __forceinline__ float get(int j) const {
    return j == 0 ? a.x : j == 1 ? a.y : j == 2 ? a.z : j == 3 ? a.w : j == 4 ? b.x : j == 5 ? b.y : j == 6 ? b.z : b.w;
  }
  static __device__ __forceinline__ void store(float *p, const float (&o)[8]) {
    *reinterpret_cast<float4 *>(p) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4 *>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
  }
};

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

// Per-kernel, per-device "attribute has been set" mask for launchers that raise a kernel's dynamic LDS limit:
// hipFuncSetAttribute acts on the current device's copy of the kernel.  A thread that finds its device's
// bit clear sets the attribute itself before launching (two racing threads both do; the call is
// idempotent), so no launch can precede the attribute on its device.
static inline bool ldm_attr_pending(std::atomic<uint64_t> &done, uint64_t *bit) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  *bit = 1ull << (dev & 63);
  return (done.load(std::memory_order_acquire) & *bit) == 0;
}

static inline bool aligned_to(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
