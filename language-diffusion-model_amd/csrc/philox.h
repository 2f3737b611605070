// Philox4x32-10 counter-based generator shared by the masking and the sampling kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace {

// Philox4x32-10 (Salmon et al. 2011), counter = (c0,c1,c2,c3), key = (k0,k1).
struct U4 { unsigned x, y, z, w; };

__host__ __device__ inline U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
  const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    unsigned long long p0 = (unsigned long long)M0 * c.x;
    unsigned long long p1 = (unsigned long long)M1 * c.z;
    U4 n;
    n.x = (unsigned)(p1 >> 32) ^ c.y ^ k0;
    n.y = (unsigned)p1;
    n.z = (unsigned)(p0 >> 32) ^ c.w ^ k1;
    n.w = (unsigned)p0;
    c = n;
    k0 += W0;
    k1 += W1;
  }
  return c;
}

__host__ __device__ inline float u01(unsigned x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

}  // namespace
