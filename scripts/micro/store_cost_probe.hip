// Probe: what a 16-byte-per-lane global store costs the wave that issues it between MFMAs (one wave per SIMD),
// by the shape of the 1 KiB it writes.  Each wave runs ITERS x { NM MFMAs, one store }.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/micro/store_cost_probe.hip -o scripts/micro/bin/store_cost_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));

constexpr unsigned WINDOW = 1u << 20;     // bytes per wave
constexpr unsigned PITCH = 4608;          // row pitch of the strided shapes (a [*, 2304] bf16 matrix)

// lane offset inside the window by shape: 0: 1 KiB contiguous; 1: 16 rows x 64 B; 2: 4 rows x 256 B; 3: 8 rows x 128 B
__host__ __device__ inline unsigned lane_off(int shape, int lane) {
  if (shape == 0) return lane * 16;
  if (shape == 1) return (lane & 15) * PITCH + (lane >> 4) * 16;
  if (shape == 2) return (lane >> 4) * PITCH + (lane & 15) * 16;
  return (lane >> 3) * PITCH + (lane & 7) * 16;
}
// iteration offset: moves along the rows (shapes 1-3) or through the window (shape 0); always a multiple of 256
__host__ __device__ inline unsigned iter_off(int shape, int it) {
  if (shape == 0) return (unsigned)(it & 63) * 1024;
  return (unsigned)(it & 15) * 256 + (unsigned)((it >> 4) & 7) * 16 * PITCH;
}

// NT: non-temporal store
template <int SHAPE, int NT, int NM, int STORE>
__global__ __launch_bounds__(256, 1) void k(char *out, int iters, unsigned long long *cyc) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f4_t acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f4_t{0.f, 0.f, 0.f, 0.f};
  bf8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
  char *base = out + ((size_t)blockIdx.x * 4 + wave) * WINDOW;
  const unsigned off = lane_off(SHAPE, lane);
  u4_t v = {1u, 2u, 3u, (unsigned)lane};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m % 16]) : "v"(a), "v"(b));
    if (STORE) {
      const unsigned o = off + iter_off(SHAPE, it);
      if (NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 0" ::"v"(o), "v"(v), "s"(base) : "memory");
      else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 0" ::"v"(o), "v"(v), "s"(base) : "memory");
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  if (s == 1234.5f) out[0] = 1;
  if (lane == 0 && blockIdx.x == 17) cyc[wave] = t1 - t0;
}

template <int SHAPE, int NT, int NM, int STORE>
static void run(const char *name, char *buf, unsigned long long *dcyc, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<SHAPE, NT, NM, STORE>), dim3(256), dim3(256), 0, 0, buf, iters, dcyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<SHAPE, NT, NM, STORE>), dim3(256), dim3(256), 0, 0, buf, iters, dcyc);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long c[4];
  CK(hipMemcpy(c, dcyc, 32, hipMemcpyDeviceToHost));
  printf("%-52s %8.1f us  cycles/iter: %7.1f %7.1f %7.1f %7.1f  (MFMA %d)\n", name, ms * 1e3, (double)c[0] / iters, (double)c[1] / iters,
         (double)c[2] / iters, (double)c[3] / iters, NM * 16);
}

int main() {
  // every offset a lane can form stays inside its wave's window
  for (int shape = 0; shape < 4; ++shape)
    for (int lane = 0; lane < 64; ++lane)
      for (int it = 0; it < 4096; ++it)
        if ((size_t)lane_off(shape, lane) + iter_off(shape, it) + 16 > WINDOW) { printf("offset out of window\n"); return 1; }
  char *buf; unsigned long long *dcyc;
  CK(hipMalloc(&buf, (size_t)256 * 4 * WINDOW));
  CK(hipMalloc(&dcyc, 64));
  const int iters = 4000;
  run<0, 0, 16, 0>("16 MFMA, no store", buf, dcyc, iters);
  run<0, 0, 16, 1>("16 MFMA + store, 1 KiB contiguous", buf, dcyc, iters);
  run<0, 1, 16, 1>("16 MFMA + nt store, 1 KiB contiguous", buf, dcyc, iters);
  run<1, 0, 16, 1>("16 MFMA + store, 16 rows x 64 B", buf, dcyc, iters);
  run<1, 1, 16, 1>("16 MFMA + nt store, 16 rows x 64 B", buf, dcyc, iters);
  run<2, 0, 16, 1>("16 MFMA + store, 4 rows x 256 B", buf, dcyc, iters);
  run<2, 1, 16, 1>("16 MFMA + nt store, 4 rows x 256 B", buf, dcyc, iters);
  run<3, 0, 16, 1>("16 MFMA + store, 8 rows x 128 B", buf, dcyc, iters);
  run<3, 1, 16, 1>("16 MFMA + nt store, 8 rows x 128 B", buf, dcyc, iters);
  run<0, 0, 48, 0>("48 MFMA, no store", buf, dcyc, iters);
  run<1, 1, 48, 1>("48 MFMA + nt store, 16 rows x 64 B", buf, dcyc, iters);
  run<2, 1, 48, 1>("48 MFMA + nt store, 4 rows x 256 B", buf, dcyc, iters);
  run<3, 1, 48, 1>("48 MFMA + nt store, 8 rows x 128 B", buf, dcyc, iters);
  return 0;
}
