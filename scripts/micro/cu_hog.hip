// Occupies `nwg` workgroups (256 threads each) for about `usec` microseconds: a stand-in for a
// communication kernel that holds a few CUs while compute kernels run on another stream.
// Every wave leaves when the constant 100 MHz wall clock passes its deadline.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void cu_hog_k(uint64_t ticks, unsigned *sink) {
  const uint64_t t0 = wall_clock64();
  unsigned x = threadIdx.x;
  while (wall_clock64() - t0 < ticks) x = x * 1664525u + 1013904223u;
  if (x == 0xdeadbeefu) sink[0] = x;
}
extern "C" int cu_hog(int nwg, double usec, void *sink, void *stream) {
  if (nwg <= 0 || usec <= 0 || usec > 2e5) return -1;     // at most 0.2 s
  hipLaunchKernelGGL(cu_hog_k, dim3(nwg), dim3(256), 0, (hipStream_t)stream, (uint64_t)(usec * 100.0), (unsigned *)sink);
  return (int)hipGetLastError();
}
