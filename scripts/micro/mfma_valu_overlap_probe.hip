// Probe: do MFMAs of one wave and vector-ALU work of ANOTHER wave on the same SIMD run at the same time?
// 512 threads per workgroup, one workgroup per CU: waves 0-3 (one per SIMD) issue chains of 32x32x16 bf16 MFMAs,
// waves 4-7 (the second wave of each SIMD) issue v_exp_f32 or v_fma_f32 chains.  Three runs per kind: MFMA waves
// alone, ALU waves alone, both.  If the units overlap, "both" takes about the longer of the two; if the SIMD
// issues one or the other, about their sum.  No memory traffic inside the timed loops.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/micro/mfma_valu_overlap_probe.hip -o scripts/micro/bin/mfma_valu_overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef float f16_t __attribute__((ext_vector_type(16)));

// MODE bit 0: MFMA waves work; bit 1: ALU waves work.  KIND 0: v_exp_f32, 1: v_fma_f32, 2: v_pk_fma_f32
template <int MODE, int KIND>
__global__ __launch_bounds__(512, 1) void k(float *out, int iters) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float res = 0.f;
  if (wave < 4) {
    if (MODE & 1) {
      f16_t acc[4];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
      bf8_t a, b;
      for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)((lane + i) & 3); b[i] = (__bf16)(float)((lane - i) & 3); }
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);   // 32 per iteration
      }
      for (int i = 0; i < 4; ++i) res += acc[i][0] + acc[i][15];
    }
  } else {
    if (MODE & 2) {
      float x[8];
      for (int i = 0; i < 8; ++i) x[i] = 0.001f * (float)(lane + i);
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int i = 0; i < 8; ++i) {     // 64 per iteration, eight independent chains
            if (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
          }
        if (KIND == 2) {
#pragma unroll
          for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < 8; i += 2) {   // 64 per iteration as well, two lanes' worth each
              typedef float f2_t __attribute__((ext_vector_type(2)));
              f2_t v = {x[i], x[i + 1]};
              asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(v));
              x[i] = v[0]; x[i + 1] = v[1];
            }
        }
      }
      for (int i = 0; i < 8; ++i) res += x[i];
    }
  }
  if (res == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = res;   // keeps the work alive, never true in practice
}

template <int MODE, int KIND>
static float run(float *out, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE, KIND>), dim3(256), dim3(512), 0, 0, out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k<MODE, KIND>), dim3(256), dim3(512), 0, 0, out, iters);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms;
}

int main() {
  float *out;
  CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
  const int iters = 4000;      // 128k MFMAs per wave (32 cycles each at one per 8 passes: ~4M cycles) / 256k ALU instructions
  const char *names[3] = {"v_exp_f32", "v_fma_f32", "v_pk_fma_f32"};
  const float m = run<1, 0>(out, iters);
  printf("MFMA waves alone: %.2f ms (%d x 32 MFMAs of 32x32x16 per wave: %.0f TFLOP/s)\n", m, iters,
         256.0 * 4 * iters * 32.0 * 2 * 32 * 32 * 16 / (m * 1e-3) / 1e12);
  const float a0 = run<2, 0>(out, iters), b0 = run<3, 0>(out, iters);
  const float a1 = run<2, 1>(out, iters), b1 = run<3, 1>(out, iters);
  const float a2 = run<2, 2>(out, iters), b2 = run<3, 2>(out, iters);
  const float al[3] = {a0, a1, a2}, bo[3] = {b0, b1, b2};
  for (int kd = 0; kd < 3; ++kd)
    printf("%-13s ALU waves alone %.2f ms, both %.2f ms  (longer of the two %.2f, sum %.2f)\n", names[kd], al[kd], bo[kd],
           al[kd] > m ? al[kd] : m, al[kd] + m);
  CK(hipFree(out));
  return 0;
}
