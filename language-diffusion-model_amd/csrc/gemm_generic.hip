// Generic strided, batched GEMM with fp32 FMA accumulation.  This is the exact-fp32
// arithmetic path (used for the F32 parity mode and for shapes the MFMA kernel does
// not take); the bf16 throughput path is gemm_mfma.hip.
//   C[m,n] = alpha * sum_k A(m,k) * B(k,n) (+ bias[n]) (+ R[m,n])
#include "common.h"
#include "gemm.h"

namespace {

constexpr int GT = 64;   // tile edge (M and N)
constexpr int GK = 16;   // K step

struct GArgs {
  int64_t M, N, K;
  float alpha;
  const void *A; int64_t a_rs, a_cs;     // element strides of A(m,k): m*a_rs + k*a_cs
  const void *B; int64_t b_rs, b_cs;     // element strides of B(k,n): k*b_rs + n*b_cs
  void *C; int64_t ldc;
  const float *bias;
  const void *R; int64_t ldr;
  ldm_bstride sa, sb, sc;
  int64_t batch_inner;
  int accumulate;   // C += result (C is read as TO)
};

// A_KFAST: A is contiguous along k (a_cs == 1); else contiguous along m.
// B_NFAST: B is contiguous along n (b_cs == 1); else contiguous along k.
template <typename TI, typename TO, bool A_KFAST, bool B_NFAST>
__global__ __launch_bounds__(256) void gemm_generic_k(GArgs g) {
  __shared__ float As[GK][GT + 4];
  __shared__ float Bs[GK][GT + 4];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int64_t m0 = (int64_t)blockIdx.y * GT, n0 = (int64_t)blockIdx.x * GT;
  const int64_t bo = blockIdx.z / g.batch_inner, bi = blockIdx.z % g.batch_inner;
  const TI *A = (const TI *)g.A + bo * g.sa.outer + bi * g.sa.inner;
  const TI *B = (const TI *)g.B + bo * g.sb.outer + bi * g.sb.inner;
  TO *C = (TO *)g.C + bo * g.sc.outer + bi * g.sc.inner;

  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int64_t k0 = 0; k0 < g.K; k0 += GK) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int m, k;
      if (A_KFAST) { m = tid >> 2; k = (tid & 3) * 4 + j; }
      else { k = tid >> 4; m = (tid & 15) * 4 + j; }
      const int64_t gm = m0 + m, gk = k0 + k;
      As[k][m] = (gm < g.M && gk < g.K) ? Elt<TI>::ld(A + gm * g.a_rs + gk * g.a_cs) : 0.f;
      int n, kb;
      if (B_NFAST) { kb = tid >> 4; n = (tid & 15) * 4 + j; }
      else { n = tid >> 2; kb = (tid & 3) * 4 + j; }
      const int64_t gn = n0 + n, gkb = k0 + kb;
      Bs[kb][n] = (gn < g.N && gkb < g.K) ? Elt<TI>::ld(B + gkb * g.b_rs + gn * g.b_cs) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GK; ++k) {
      const float4 a = *reinterpret_cast<const float4 *>(&As[k][ty * 4]);
      const float4 b = *reinterpret_cast<const float4 *>(&Bs[k][tx * 4]);
      const float av[4] = {a.x, a.y, a.z, a.w};
      const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t gm = m0 + ty * 4 + i;
    if (gm >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t gn = n0 + tx * 4 + j;
      if (gn >= g.N) continue;
      float v = acc[i][j] * g.alpha;
      if (g.bias) v += g.bias[gn];
      if (g.R) v += Elt<TO>::ld((const TO *)g.R + gm * g.ldr + gn);
      if (g.accumulate) v += Elt<TO>::ld(C + gm * g.ldc + gn);
      Elt<TO>::st(C + gm * g.ldc + gn, v);
    }
  }
}

template <typename TI, typename TO>
int launch_generic(const GArgs &g, int transA, int transB, int64_t batch, hipStream_t s) {
  dim3 grid((unsigned)ceil_div(g.N, GT), (unsigned)ceil_div(g.M, GT), (unsigned)batch);
  if (grid.y > 65535 || grid.z > 65535) return LDM_E_SHAPE;
  if (!transA && !transB) hipLaunchKernelGGL((gemm_generic_k<TI, TO, true, true>), grid, dim3(256), 0, s, g);
  else if (!transA && transB) hipLaunchKernelGGL((gemm_generic_k<TI, TO, true, false>), grid, dim3(256), 0, s, g);
  else if (transA && !transB) hipLaunchKernelGGL((gemm_generic_k<TI, TO, false, true>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_generic_k<TI, TO, false, false>), grid, dim3(256), 0, s, g);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

}  // namespace

int ldm_gemm_generic(int in_dtype, int out_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K,
                     float alpha, const void *A, int64_t lda, ldm_bstride sa, const void *B, int64_t ldb,
                     ldm_bstride sb, void *C, int64_t ldc, ldm_bstride sc, const float *bias, const void *R,
                     int64_t ldr, int64_t batch, int64_t batch_inner, hipStream_t s) {
  GArgs g;
  g.M = M; g.N = N; g.K = K; g.alpha = alpha;
  g.A = A; g.a_rs = transA ? 1 : lda; g.a_cs = transA ? lda : 1;
  g.B = B; g.b_rs = transB ? 1 : ldb; g.b_cs = transB ? ldb : 1;
  g.C = C; g.ldc = ldc; g.bias = bias; g.R = R; g.ldr = ldr;
  g.sa = sa; g.sb = sb; g.sc = sc; g.batch_inner = batch_inner > 0 ? batch_inner : 1;
  g.accumulate = 0;
  if (in_dtype == LDM_F32 && out_dtype == LDM_F32) return launch_generic<float, float>(g, transA, transB, batch, s);
  if (in_dtype == LDM_BF16 && out_dtype == LDM_BF16) return launch_generic<bf16_t, bf16_t>(g, transA, transB, batch, s);
  if (in_dtype == LDM_BF16 && out_dtype == LDM_F32) return launch_generic<bf16_t, float>(g, transA, transB, batch, s);
  if (in_dtype == LDM_F32 && out_dtype == LDM_BF16) return launch_generic<float, bf16_t>(g, transA, transB, batch, s);
  return LDM_E_DTYPE;
}

int ldm_gemm_generic_acc(int in_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                         const void *A, int64_t lda, const void *B, int64_t ldb, float *C, int64_t ldc,
                         hipStream_t s) {
  GArgs g;
  g.M = M; g.N = N; g.K = K; g.alpha = alpha;
  g.A = A; g.a_rs = transA ? 1 : lda; g.a_cs = transA ? lda : 1;
  g.B = B; g.b_rs = transB ? 1 : ldb; g.b_cs = transB ? ldb : 1;
  g.C = C; g.ldc = ldc; g.bias = nullptr; g.R = nullptr; g.ldr = 0;
  g.sa = g.sb = g.sc = ldm_bstride{0, 0}; g.batch_inner = 1; g.accumulate = 1;
  if (in_dtype == LDM_F32) return launch_generic<float, float>(g, transA, transB, 1, s);
  if (in_dtype == LDM_BF16) return launch_generic<bf16_t, float>(g, transA, transB, 1, s);
  return LDM_E_DTYPE;
}
