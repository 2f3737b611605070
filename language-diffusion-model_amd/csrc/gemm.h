// Internal GEMM entry points shared by the C-ABI dispatcher and the attention code.
#pragma once
#include "common.h"

// exact-fp32-accumulate FMA kernel, any strides / alignment / batch
int ldm_gemm_generic(int in_dtype, int out_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K,
                     float alpha, const void *A, int64_t lda, ldm_bstride sa, const void *B, int64_t ldb,
                     ldm_bstride sb, void *C, int64_t ldc, ldm_bstride sc, const float *bias, const void *R,
                     int64_t ldr, int64_t batch, int64_t batch_inner, hipStream_t s);

// bf16 MFMA kernel; returns LDM_E_SHAPE / LDM_E_ALIGN when the problem does not fit it
int ldm_gemm_mfma_bf16(int out_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                       const void *A, int64_t lda, const void *B, int64_t ldb, void *C, int64_t ldc,
                       const float *bias, const void *R, int64_t ldr, hipStream_t s);

// C(fp32) += alpha * opA(A) * opB(B) with the K range split across workgroups (float atomics);
// rowsum (nullable): rowsum[m] += alpha * sum_k opA[m, k]
int ldm_gemm_mfma_bf16_acc(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const void *A,
                           int64_t lda, const void *B, int64_t ldb, float *C, int64_t ldc, float *rowsum,
                           hipStream_t s);
int ldm_gemm_generic_acc(int in_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                         const void *A, int64_t lda, const void *B, int64_t ldb, float *C, int64_t ldc,
                         hipStream_t s);

// 256-row-tile kernels (gemm_mfma256.hip); same contracts as the 128-tile forms above
int ldm_gemm_mfma256_bf16(int out_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                          const void *A, int64_t lda, const void *B, int64_t ldb, void *C, int64_t ldc,
                          const float *bias, const void *R, int64_t ldr, hipStream_t s);
int ldm_gemm_mfma256_bf16_acc(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const void *A,
                              int64_t lda, const void *B, int64_t ldb, float *C, int64_t ldc, float *rowsum,
                              hipStream_t s);
int ldm_gemm_mfma256_bf16_dyn(int out_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                              const void *A, int64_t lda, const void *B, int64_t ldb, void *C, int64_t ldc,
                              const float *bias, const int *dyn_m, const int *row_map, hipStream_t s);
int ldm_gemm_mfma256_bf16_acc_dynk(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                                   const void *A, int64_t lda, const void *B, int64_t ldb, float *C, int64_t ldc,
                                   const int *dyn_k, float *rowsum, hipStream_t s);
// residual + dropout(projection + bias) from one GEMM (full, aligned 256 x 256 tiles only)
int ldm_gemm_mfma256_bf16_drop_res(int64_t M, int64_t N, int64_t K, const void *A, int64_t lda, const void *W,
                                   int64_t ldw, void *C, int64_t ldc, const float *bias, const void *R, int64_t ldr,
                                   unsigned thresh, float inv, uint64_t seed, hipStream_t s);
// SwiGLU backward fused into the down-projection's input-gradient GEMM (full 256 x 256 tiles only)
int ldm_gemm_mfma256_bf16_swiglu_bwd(int64_t T, int64_t F, int64_t K, const void *dy, int64_t ld_dy, const void *W,
                                     int64_t ldw, const void *ag, int64_t ld_ag, void *dag, int64_t ld_dag,
                                     hipStream_t s);
// SwiGLU forward fused into the gate/up projection: [a | g] and h = a * silu(g) from one GEMM
int ldm_gemm_mfma256_bf16_swiglu_fwd(int64_t T, int64_t F, int64_t K, const void *x, int64_t ldx, const void *W12,
                                     int64_t ldw, const float *bias, void *ag, int64_t ld_ag, void *h, int64_t ld_h,
                                     hipStream_t s);
int ldm_gemm_mfma256_bf16_qkv_rope(int64_t T, int64_t D, int64_t K, int64_t hd, int64_t S, const void *x, int64_t ldx,
                                   const void *W, int64_t ldw, const float *bias, const float *cos_t, const float *sin_t,
                                   const float *step, float q_scale, void *qkv, int64_t ld_qkv, hipStream_t s);

// four-wave kernel (gemm_mfma4w.hip): A [M, K], B [N, K] K-contiguous, bf16 output, full 256 x 256 tiles,
// optional bias; LDM_E_SHAPE / LDM_E_ALIGN for anything else
bool ldm_gemm_4w_enabled();
int ldm_gemm_mfma4w_bf16(int64_t M, int64_t N, int64_t K, const void *A, int64_t lda, const void *B, int64_t ldb,
                         void *C, int64_t ldc, const float *bias, const void *R, const int *dyn_m, hipStream_t s);
