// C-ABI GEMM dispatch: bf16 problems that fit the MFMA kernel run there, everything
// else (F32 parity mode, odd shapes, batched attention products) on the FMA kernel.
#include "common.h"
#include "gemm.h"
#include <stdlib.h>

// LDM_GEMM_TILE=128 pins the 128x128 kernel (A/B measurements); default: 256-row tiles first.
static int gemm_kernel_choice() {
  static int choice = -1;
  if (choice < 0) {
    const char *e = getenv("LDM_GEMM_TILE");
    choice = (e && atoi(e) == 128) ? 128 : 256;
  }
  return choice;
}

extern "C" int ldm_gemm(int in_dtype, int out_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K,
                        float alpha, const void *A, int64_t lda, const void *B, int64_t ldb, void *C,
                        int64_t ldc, const float *bias, const void *R, int64_t ldr, void *stream) {
  if (M < 0 || N < 0 || K < 0) return LDM_E_BADARG;
  if (M == 0 || N == 0) return LDM_OK;
  if (!C || (K > 0 && (!A || !B))) return LDM_E_BADARG;
  if (in_dtype != LDM_F32 && in_dtype != LDM_BF16) return LDM_E_DTYPE;
  if (out_dtype != LDM_F32 && out_dtype != LDM_BF16) return LDM_E_DTYPE;
  hipStream_t s = (hipStream_t)stream;
  if (in_dtype == LDM_BF16) {
    int rc = LDM_E_SHAPE;
    if (gemm_kernel_choice() != 128)
      rc = ldm_gemm_mfma256_bf16(out_dtype, transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, R, ldr, s);
    if (rc != LDM_E_SHAPE && rc != LDM_E_ALIGN) return rc;
    rc = ldm_gemm_mfma_bf16(out_dtype, transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, R, ldr, s);
    if (rc != LDM_E_SHAPE && rc != LDM_E_ALIGN) return rc;
  }
  const ldm_bstride z{0, 0};
  return ldm_gemm_generic(in_dtype, out_dtype, transA, transB, M, N, K, alpha, A, lda, z, B, ldb, z, C, ldc, z,
                          bias, R, ldr, 1, 1, s);
}

extern "C" int ldm_gemm_acc(int in_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                            const void *A, int64_t lda, const void *B, int64_t ldb, float *C, int64_t ldc,
                            void *stream) {
  if (M < 0 || N < 0 || K < 0) return LDM_E_BADARG;
  if (M == 0 || N == 0 || K == 0) return LDM_OK;
  if (!A || !B || !C) return LDM_E_BADARG;
  if (in_dtype != LDM_F32 && in_dtype != LDM_BF16) return LDM_E_DTYPE;
  hipStream_t s = (hipStream_t)stream;
  if (in_dtype == LDM_BF16) {
    int rc = LDM_E_SHAPE;
    if (gemm_kernel_choice() != 128)
      rc = ldm_gemm_mfma256_bf16_acc(transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, nullptr, s);
    if (rc != LDM_E_SHAPE && rc != LDM_E_ALIGN) return rc;
    rc = ldm_gemm_mfma_bf16_acc(transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, nullptr, s);
    if (rc != LDM_E_SHAPE && rc != LDM_E_ALIGN) return rc;
  }
  return ldm_gemm_generic_acc(in_dtype, transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, s);
}

extern "C" int ldm_gemm_acc_rowsum(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                                   const void *A, int64_t lda, const void *B, int64_t ldb, float *C, int64_t ldc,
                                   float *a_rowsum, void *stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || !a_rowsum) return LDM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  int rc = LDM_E_SHAPE;
  if (gemm_kernel_choice() != 128)
    rc = ldm_gemm_mfma256_bf16_acc(transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, a_rowsum, s);
  if (rc != LDM_E_SHAPE && rc != LDM_E_ALIGN) return rc;
  return ldm_gemm_mfma_bf16_acc(transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, a_rowsum, s);
}

extern "C" int ldm_gemm_rows_dyn(int out_dtype, int transB, int64_t M, int64_t N, int64_t K, const void *A,
                                 int64_t lda, const void *B, int64_t ldb, void *C, int64_t ldc,
                                 const float *bias, const int32_t *m_dev, const int32_t *row_map,
                                 void *stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || !m_dev) return LDM_E_BADARG;
  return ldm_gemm_mfma256_bf16_dyn(out_dtype, 0, transB, M, N, K, 1.0f, A, lda, B, ldb, C, ldc, bias, m_dev,
                                   row_map, (hipStream_t)stream);
}

extern "C" int ldm_gemm_acc_kdyn(int transA, int transB, int64_t M, int64_t N, int64_t K, const void *A,
                                 int64_t lda, const void *B, int64_t ldb, float *C, int64_t ldc,
                                 const int32_t *k_dev, float *a_rowsum, void *stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || !k_dev) return LDM_E_BADARG;
  return ldm_gemm_mfma256_bf16_acc_dynk(transA, transB, M, N, K, 1.0f, A, lda, B, ldb, C, ldc, k_dev, a_rowsum,
                                        (hipStream_t)stream);
}

extern "C" int ldm_gemm_batched(int in_dtype, int out_dtype, int transA, int transB, int64_t M, int64_t N,
                                int64_t K, float alpha, const void *A, int64_t lda, ldm_bstride sa,
                                const void *B, int64_t ldb, ldm_bstride sb, void *C, int64_t ldc,
                                ldm_bstride sc, int64_t batch, int64_t batch_inner, void *stream) {
  if (M < 0 || N < 0 || K < 0 || batch < 0) return LDM_E_BADARG;
  if (M == 0 || N == 0 || batch == 0) return LDM_OK;
  if (!C || (K > 0 && (!A || !B))) return LDM_E_BADARG;
  return ldm_gemm_generic(in_dtype, out_dtype, transA, transB, M, N, K, alpha, A, lda, sa, B, ldb, sb, C, ldc,
                          sc, nullptr, nullptr, 0, batch, batch_inner, (hipStream_t)stream);
}

extern "C" int ldm_gemm_dropout_residual(int64_t M, int64_t N, int64_t K, const void *A, int64_t lda, const void *W,
                                         int64_t ldw, const float *bias, const void *R, int64_t ldr, void *C,
                                         int64_t ldc, float p, uint64_t seed, void *stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !R || !C || !(p >= 0.0f) || !(p < 1.0f)) return LDM_E_BADARG;
  // threshold and scale exactly as ldm_dropout_add derives them
  const double t = (double)p * 4294967296.0;
  const unsigned thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
  const float inv = (float)(4294967296.0 / (4294967296.0 - (double)thresh));
  return ldm_gemm_mfma256_bf16_drop_res(M, N, K, A, lda, W, ldw, C, ldc, bias, R, ldr, thresh, inv, seed,
                                        (hipStream_t)stream);
}

extern "C" int ldm_gemm_swiglu_bwd(int64_t T, int64_t F, int64_t K, const void *dy, int64_t ld_dy, const void *W,
                                   int64_t ldw, const void *ag, int64_t ld_ag, void *dag, int64_t ld_dag,
                                   void *stream) {
  if (T <= 0 || F <= 0 || K <= 0 || !dy || !W || !ag || !dag) return LDM_E_BADARG;
  return ldm_gemm_mfma256_bf16_swiglu_bwd(T, F, K, dy, ld_dy, W, ldw, ag, ld_ag, dag, ld_dag, (hipStream_t)stream);
}

extern "C" int ldm_gemm_swiglu_fwd(int64_t T, int64_t F, int64_t K, const void *x, int64_t ldx, const void *W12,
                                   int64_t ldw, const float *bias, void *ag, int64_t ld_ag, void *h, int64_t ld_h,
                                   void *stream) {
  if (T <= 0 || F <= 0 || K <= 0 || !x || !W12 || !ag || !h) return LDM_E_BADARG;
  return ldm_gemm_mfma256_bf16_swiglu_fwd(T, F, K, x, ldx, W12, ldw, bias, ag, ld_ag, h, ld_h, (hipStream_t)stream);
}

extern "C" int ldm_gemm_qkv_rope(int64_t T, int64_t D, int64_t K, int64_t hd, int64_t seq, const void *x, int64_t ldx,
                                 const void *Wqkv, int64_t ldw, const float *bias, const float *cos_t,
                                 const float *sin_t, const float *step, float q_scale, void *qkv, int64_t ld_qkv,
                                 void *stream) {
  if (T <= 0 || D <= 0 || K <= 0 || hd <= 0 || seq <= 0) return LDM_E_BADARG;
  if (!x || !Wqkv || !qkv || !cos_t || !sin_t || !step) return LDM_E_BADARG;
  return ldm_gemm_mfma256_bf16_qkv_rope(T, D, K, hd, seq, x, ldx, Wqkv, ldw, bias, cos_t, sin_t, step, q_scale, qkv,
                                        ld_qkv, (hipStream_t)stream);
}

extern "C" int ldm_abi_version(void) { return 21; }

extern "C" const char *ldm_error_string(int code) {
  switch (code) {
    case LDM_OK: return "ok";
    case LDM_E_BADARG: return "bad argument (null pointer or negative size)";
    case LDM_E_ALIGN: return "pointer or leading dimension not aligned";
    case LDM_E_SHAPE: return "shape not supported by the kernels";
    case LDM_E_DTYPE: return "unknown element type";
    case LDM_E_WORKSPACE: return "workspace too small";
    case LDM_E_COMM: return "RCCL not loadable or an RCCL call failed (ldm_comm_last_error() has the text)";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}
