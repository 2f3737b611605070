// Micro-benchmark: L2 -> LDS streaming rate of 16-byte LDS-DMA loads by access shape.
// Every workgroup (8 waves) re-reads a tile set that fits in L2; the figure printed is the
// aggregate GB/s and bytes per cycle per CU (at the nominal 2.4 GHz).
//   mode 0: 16 rows x 64 B per instruction (K-contiguous operand, 32-deep half-tile), row stride ld
//   mode 1:  8 rows x 128 B per instruction (64-deep tile)
//   mode 2:  2 rows x 512 B per instruction (strided operand)
//   mode 3:  1 KiB contiguous per instruction
//   mode 4: as mode 0, but the two 64-B halves of each 128-B line are requested by consecutive instructions
//   mode 5: as mode 0, but the second halves follow after the wave's four first-half instructions
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

template <int MODE>
__global__ __launch_bounds__(512) void stream_k(const char *base, long ld_bytes, int iters, int rows_per_wg,
                                                int share) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // workgroups that share a tile set: `share` consecutive-by-8 blocks (same XCD) read the same rows
  const int tile = (blockIdx.x / 8 / share) * 8 + (blockIdx.x & 7);
  const char *wg = base + (long)tile * rows_per_wg * ld_bytes;
  long off;
  int rows_per_instr;
  if (MODE == 0 || MODE >= 4) { off = (long)(lane >> 2) * ld_bytes + (lane & 3) * 16; rows_per_instr = 16; }
  else if (MODE == 1) { off = (long)(lane >> 3) * ld_bytes + (lane & 7) * 16; rows_per_instr = 8; }
  else if (MODE == 2) { off = (long)(lane >> 5) * ld_bytes + (lane & 31) * 16; rows_per_instr = 2; }
  else { off = lane * 16; rows_per_instr = 1; }
  const int instr_per_pass = rows_per_wg / rows_per_instr / 8;   // per wave
  char *dst = smem + wave * 16384;
  if (MODE >= 4) {
    for (int it = 0; it < iters; ++it) {
      const long kshift = (long)(it & 1) * 128;
      if (MODE == 4) {
        for (int i = 0; i < instr_per_pass; ++i) {
          const long row = (long)(wave * instr_per_pass + i) * rows_per_instr;
          const char *src = wg + row * ld_bytes + kshift + off;
          __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(dst + (2 * i & 15) * 1024), 16, 0, 0);
          __builtin_amdgcn_global_load_lds(GLB_PTR(src + 64), LDS_PTR(dst + ((2 * i + 1) & 15) * 1024), 16, 0, 0);
          asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        }
      } else {
        for (int h = 0; h < 2; ++h)
          for (int i = 0; i < instr_per_pass; ++i) {
            const long row = (long)(wave * instr_per_pass + i) * rows_per_instr;
            const char *src = wg + row * ld_bytes + kshift + off + h * 64;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(dst + ((h * instr_per_pass + i) & 15) * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
          }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  for (int it = 0; it < iters; ++it) {
    const long kshift = (MODE == 3) ? 0 : (long)(it & 3) * (MODE == 0 ? 64 : MODE == 1 ? 128 : 512);
    for (int i = 0; i < instr_per_pass; ++i) {
      const long row = (long)(wave * instr_per_pass + i) * rows_per_instr;
      const char *src = wg + (MODE == 3 ? row * 1024 : row * ld_bytes + kshift) + off;
      __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(dst + (i & 15) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main(int argc, char **argv) {
  const int iters = 400;
  const long ld_bytes = 2048;           // 1024 bf16 per row
  const size_t bytes = 512ull << 20;
  char *buf;
  if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
  hipMemset(buf, 1, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int share = 1; share <= 32; share *= 4) {
    for (int mode = 0; mode < 6; ++mode) {
      const int rows_per_wg = mode == 3 ? 32 : 512;   // mode 3: 32 KiB contiguous per pass
      const long pass_bytes = mode == 3 ? 32 * 1024 : (long)rows_per_wg * (mode == 0 ? 64 : (mode == 1 || mode >= 4) ? 128 : 512);
      auto launch = [&]() {
        switch (mode) {
          case 0: hipLaunchKernelGGL(stream_k<0>, dim3(256), dim3(512), 131072, 0, buf, ld_bytes, iters, rows_per_wg, share); break;
          case 1: hipLaunchKernelGGL(stream_k<1>, dim3(256), dim3(512), 131072, 0, buf, ld_bytes, iters, rows_per_wg, share); break;
          case 2: hipLaunchKernelGGL(stream_k<2>, dim3(256), dim3(512), 131072, 0, buf, ld_bytes, iters, rows_per_wg, share); break;
          case 4: hipLaunchKernelGGL(stream_k<4>, dim3(256), dim3(512), 131072, 0, buf, ld_bytes, iters, rows_per_wg, share); break;
          case 5: hipLaunchKernelGGL(stream_k<5>, dim3(256), dim3(512), 131072, 0, buf, ld_bytes, iters, rows_per_wg, share); break;
          default: hipLaunchKernelGGL(stream_k<3>, dim3(256), dim3(512), 131072, 0, buf, ld_bytes, iters, rows_per_wg, share); break;
        }
      };
      hipFuncSetAttribute((const void *)stream_k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      hipFuncSetAttribute((const void *)stream_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      hipFuncSetAttribute((const void *)stream_k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      hipFuncSetAttribute((const void *)stream_k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      hipFuncSetAttribute((const void *)stream_k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      hipFuncSetAttribute((const void *)stream_k<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double total = 256.0 * iters * pass_bytes;
      printf("share %2d mode %d: %8.1f GB/s total, %6.1f GB/s/CU, %5.1f B/cyc/CU @2.4GHz  (%.3f ms)\n", share, mode,
             total / ms / 1e6, total / ms / 1e6 / 256, total / ms / 1e6 / 256 / 2.4, ms);
    }
  }
  return 0;
}
