/*
 * ldm_hip.h -- C ABI of the MI355X (gfx950) masked-diffusion training-step library.
 *
 * Every entry point takes plain device pointers, sizes and a hipStream_t passed
 * as void*.  Nothing here allocates, frees or synchronises: outputs and
 * workspaces are caller-provided (PyTorch owns all memory), work is enqueued on
 * the given stream and the call returns immediately.
 *
 * Return value: 0 on success, a negative LDM_E_* code for a rejected argument,
 * or a positive hipError_t if a launch failed.  Nothing throws across the ABI.
 *
 * Element types: LDM_F32 = float, LDM_BF16 = bfloat16 (raw uint16 storage).
 * All matrices are row-major; "ld" arguments are row strides in ELEMENTS.
 *
 * Each declaration names the reference code it stands in for
 * (paths relative to the Usefulmaths/Language-Diffusion-Model checkout).
 */
#ifndef LDM_HIP_H
#define LDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { LDM_F32 = 0, LDM_BF16 = 1 };

enum {
  LDM_OK = 0,
  LDM_E_BADARG = -1,   /* null pointer / negative size */
  LDM_E_ALIGN = -2,    /* pointer or leading dimension not aligned as required */
  LDM_E_SHAPE = -3,    /* shape outside what the kernels support */
  LDM_E_DTYPE = -4,    /* unknown element type */
  LDM_E_WORKSPACE = -5,/* workspace too small */
  LDM_E_COMM = -6      /* RCCL not loadable, or an RCCL call failed: see ldm_comm_last_error() */
};

int ldm_abi_version(void);
const char *ldm_error_string(int code);

/* ---------------------------------------------------------------- K1: forward masking
 * Replaces RandomMaskingStrategy.apply_random_mask (src/masking_strategy.py:88-114)
 * together with the draw of t in DiffusionLanguageModel.forward (src/model.py:59-68).
 *   eligible = id not in {bos,eos,pad};  mask = (u < t) & eligible  (fp32 compare)
 *   masked_out = mask ? mask_id : id;    mask_out = mask (1 byte, 0/1)
 * u comes from `draw` (fp32 [B*S]) when non-null (parity mode: the caller injects
 * the reference's torch.rand_like stream), otherwise from Philox4x32-10 keyed by
 * (seed, offset) inside the kernel (one HBM pass, 17 B/token).
 * t: t_mode 0 = the scalar t_scalar for every sequence (reference behaviour),
 *    1 = per-sequence fp32 t_seq[B] (read), 2 = per-sequence t ~ U(eps,1) drawn
 *    in-kernel from Philox and written to t_seq[B]. */
int ldm_mask_fwd(const int64_t *ids, const float *draw, int t_mode, float t_scalar,
                 float *t_seq, float eps, int64_t batch, int64_t seq,
                 int64_t bos_id, int64_t eos_id, int64_t pad_id, int64_t mask_id,
                 uint64_t seed, uint64_t offset,
                 int64_t *masked_out, uint8_t *mask_out, void *stream);

/* ---------------------------------------------------------------- embedding
 * nn.Embedding forward/backward (src/transformer.py:382,426).
 * table is fp32 [vocab, d]; out is [T, d] of out_dtype.  Backward scatter-adds
 * dout rows into the PRE-ZEROED fp32 dtable; rows of `hot_id` (the mask token,
 * about half of all rows) are summed inside each workgroup first. */
int ldm_embed_fwd(const int64_t *ids, const float *table, int64_t T, int64_t d, int64_t vocab,
                  void *out, int out_dtype, void *stream);
int ldm_embed_bwd(const int64_t *ids, const void *dout, int dtype, int64_t T, int64_t d,
                  int64_t vocab, int64_t hot_id, float *dtable, void *stream);

/* ---------------------------------------------------------------- RMSNorm
 * RMSNorm.forward (src/transformer.py:44-60): y = w * x / sqrt(mean(x^2) + eps).
 * x,y: [T, d] of dtype; w fp32 [d]; rstd fp32 [T] is saved for backward.
 * Backward: dx = rstd*(dy*w) - x*rstd^3*mean(dy*w*x) (+ dres when non-null, the
 * residual-stream gradient), dw = sum_t dy*x*rstd.  dw_partial is a workspace of
 * ldm_rmsnorm_bwd_ws_floats(T,d) floats; dw (fp32 [d]) is overwritten. */
int ldm_rmsnorm_fwd(const void *x, const float *w, int64_t T, int64_t d, float eps, int dtype,
                    void *y, float *rstd, void *stream);
int64_t ldm_rmsnorm_bwd_ws_floats(int64_t T, int64_t d);
int ldm_rmsnorm_bwd(const void *x, const float *w, const float *rstd, const void *dy,
                    const void *dres, int64_t T, int64_t d, int dtype, void *dx, float *dw,
                    float *dw_partial, void *stream);

/* ldm_rmsnorm_bwd with a second output dx_drop = keep * dx / (1 - p), where keep is the mask that
 * ldm_dropout_add(y, res, out, T, d, dtype, p, seed) applied when it formed the norm's input x = out
 * (src/transformer.py:352,361 feed `src + self.dropout(...)` to the next norm): dx is the gradient of
 * that sum, dx_drop the gradient of the dropout's operand, so the projection behind the dropout needs no
 * masking pass of its own.  dx_drop is formed from dx as stored (rounded to dtype).  d % 8 == 0 and
 * d <= 4096 and 16-byte-aligned rows, else LDM_E_SHAPE / LDM_E_ALIGN (use the two calls). */
int ldm_rmsnorm_bwd_dropgrad(const void *x, const float *w, const float *rstd, const void *dy,
                             const void *dres, int64_t T, int64_t d, int dtype, void *dx,
                             void *dx_drop, float p, uint64_t seed, float *dw, float *dw_partial,
                             void *stream);

/* RMSNorm between an fp32 residual stream and bf16 projections: x fp32 -> y bf16, statistics and products
 * in fp32 -- what the reference's norm is under bf16 autocast (src/transformer.py:54-58 run in fp32, the
 * Linear behind it casts its input).  Backward: x, dres (nullable) and dx fp32, dy bf16; dw / dw_partial as
 * ldm_rmsnorm_bwd.  d % 8 == 0 and d <= 4096, else LDM_E_SHAPE. */
int ldm_rmsnorm_fwd_f32_bf16(const float *x, const float *w, int64_t T, int64_t d, float eps, void *y,
                             float *rstd, void *stream);
int ldm_rmsnorm_bwd_f32_bf16(const float *x, const float *w, const float *rstd, const void *dy,
                             const float *dres, int64_t T, int64_t d, float *dx, float *dw,
                             float *dw_partial, void *stream);

/* ldm_rmsnorm_bwd for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): the same kernels and results, but the dw partials live in a
 * fixed 8 MiB device array inside the library instead of a caller-provided workspace, so the call
 * needs no buffer beyond its inputs and outputs and allocates nothing.  No counterpart in the
 * reference.  Limits: d <= ldm_rmsnorm_bwd_noalloc_max_d() (4096; larger returns LDM_E_WORKSPACE).
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two
 * threads) race; stream-ordered calls are safe.  With d <= 1024 the result is bit-identical to
 * ldm_rmsnorm_bwd; above that fewer workgroups share the array, so dw is summed in a different
 * order (dx is identical). */
int64_t ldm_rmsnorm_bwd_noalloc_max_d(void);
int ldm_rmsnorm_bwd_noalloc(const void *x, const float *w, const float *rstd, const void *dy,
                            const void *dres, int64_t T, int64_t d, int dtype, void *dx, float *dw,
                            void *stream);

/* ---------------------------------------------------------------- RoPE (+ query scale)
 * RotaryPositionEmbedding.apply_rotary_pos_emb + rotate_half + "q / sqrt(hd)"
 * (src/transformer.py:98-101,178-185,261).  Works in place on the q and k thirds of
 * a fused qkv activation [B*S, 3*H*hd] (ld elements per row).  cos/sin are fp32
 * [S, hd/2] (the reference's tables are cat(freqs,freqs), i.e. the two halves are
 * equal).  q is multiplied by q_scale after rotation.  inverse != 0 applies the
 * transpose (backward): the inverse rotation, then the same q_scale. */
int ldm_rope_qk(void *qkv, int dtype, int64_t batch, int64_t seq, int64_t nhead, int64_t hd,
                int64_t ld, const float *cos_t, const float *sin_t, float q_scale, int inverse,
                void *stream);

/* ---------------------------------------------------------------- SwiGLU elementwise
 * SwiGLU.forward middle line (src/transformer.py:93): h = a * silu(g), where the
 * fused activation ag is [T, 2F]: a = ag[:, :F] (w1 branch), g = ag[:, F:] (w2 branch).
 * Backward: da = dh*silu(g); dg = dh*a*sigmoid(g)*(1+g*(1-sigmoid(g))), written to dag. */
int ldm_swiglu_fwd(const void *ag, int64_t T, int64_t F, int dtype, void *h, void *stream);
int ldm_swiglu_bwd(const void *ag, const void *dh, int64_t T, int64_t F, int dtype, void *dag,
                   void *stream);

/* ---------------------------------------------------------------- GEMM (nn.Linear fwd/bwd)
 * Replaces the nn.Linear calls of src/transformer.py:242-244,288,91-95,441 and their
 * autograd backward.  C[M,N] = alpha * opA(A)[M,K] * opB(B)[K,N] (+ bias[N]) (+ R[M,N]).
 *   transA == 0: A stored [M,K] (lda >= K);  transA != 0: A stored [K,M] (lda >= M)
 *   transB == 0: B stored [K,N] (ldb >= N);  transB != 0: B stored [N,K] (ldb >= K)
 * in_dtype is the type of A and B; out_dtype the type of C and R; bias is fp32.
 * Accumulation is fp32.  BF16 inputs run on MFMA and need lda, ldb multiples of 8
 * and 16-byte aligned A, B; F32 inputs have no alignment requirement.
 * Batched form: `batch` problems, problem i = (i / batch_inner, i % batch_inner)
 * offsets its A/B/C by the two element strides given (0 strides broadcast). */
typedef struct {
  int64_t outer, inner; /* element strides for the outer and inner batch index */
} ldm_bstride;

int ldm_gemm(int in_dtype, int out_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K,
             float alpha, const void *A, int64_t lda, const void *B, int64_t ldb, void *C,
             int64_t ldc, const float *bias, const void *R, int64_t ldr, void *stream);
/* Accumulating form for weight gradients: C(fp32)[M,N] += alpha * opA(A) * opB(B).  BF16
 * problems split the K range over workgroups (float atomics into C), so a small output
 * with a long reduction (K = number of tokens) still fills the chip; the sum order, and so
 * the last bits, can differ between runs.  F32 problems accumulate in a fixed order. */
int ldm_gemm_acc(int in_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                 const void *A, int64_t lda, const void *B, int64_t ldb, float *C, int64_t ldc,
                 void *stream);

/* Weight AND bias gradient of a projection in one pass (nn.Linear backward under autograd,
 * e.g. src/transformer.py:212-215, 441): C(fp32) += alpha * opA(A) * opB(B) and
 * a_rowsum[m] (fp32, pre-zeroed by the caller like C) += alpha * sum_k opA(A)[m, k].
 * With opA(A) = dy^T this is dW += dy^T x and db += column sums of dy.  BF16 MFMA kernels
 * only: returns LDM_E_SHAPE / LDM_E_ALIGN for problems they do not take (the caller then uses
 * ldm_gemm_acc + ldm_colsum). */
int ldm_gemm_acc_rowsum(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                        const void *A, int64_t lda, const void *B, int64_t ldb, float *C, int64_t ldc,
                        float *a_rowsum, void *stream);
int ldm_gemm_batched(int in_dtype, int out_dtype, int transA, int transB, int64_t M, int64_t N,
                     int64_t K, float alpha, const void *A, int64_t lda, ldm_bstride sa,
                     const void *B, int64_t ldb, ldm_bstride sb, void *C, int64_t ldc,
                     ldm_bstride sc, int64_t batch, int64_t batch_inner, void *stream);

/* The SwiGLU backward as the epilogue of the down-projection's input-gradient GEMM (BF16, MFMA):
 * SwiGLU.forward (src/transformer.py:91-95) is w3(a * silu(g)) with ag = [a | g] = [w1(x) | w2(x)]; under
 * autograd the gradient of the hidden activation is dh = dy * W3 (dy [T, K], W3 stored [K, F]) and
 *   dag[:, :F] = dh * silu(g),   dag[:, F:] = dh * a * sigmoid(g) * (1 + g * (1 - sigmoid(g))).
 * This entry point writes dag [T, 2F] without writing dh: exactly what ldm_gemm (dh) followed by
 * ldm_swiglu_bwd leaves, up to the bf16 rounding of dh that the two-call form goes through.
 * T and F must be multiples of 256, row strides multiples of 8; otherwise LDM_E_SHAPE / LDM_E_ALIGN
 * (the caller then makes the two calls). */
int ldm_gemm_swiglu_bwd(int64_t T, int64_t F, int64_t K, const void *dy, int64_t ld_dy, const void *W,
                        int64_t ldw, const void *ag, int64_t ld_ag, void *dag, int64_t ld_dag, void *stream);
/* SwiGLU forward inside the gate/up projection (src/transformer.py:91-95: w1(x) * silu(w2(x))):
 * ag[T, 2F] = x W12^T + bias (columns [a | g], what ldm_gemm with the packed weight gives) AND
 * h[T, F] = a * silu(g) from ONE GEMM, so the activation is never a separate pass.  W12 = [w1; w2],
 * [2F, K] with K contiguous; bias fp32 [2F] or NULL.  BF16; T a multiple of 256, F of 128, K of 32,
 * row strides multiples of 8; otherwise LDM_E_SHAPE / LDM_E_ALIGN (use ldm_gemm + ldm_swiglu_fwd). */
int ldm_gemm_swiglu_fwd(int64_t T, int64_t F, int64_t K, const void *x, int64_t ldx, const void *W12,
                        int64_t ldw, const float *bias, void *ag, int64_t ld_ag, void *h, int64_t ld_h,
                        void *stream);

/* The q/k/v projections, the rotary embedding and the query scale from ONE GEMM
 * (src/transformer.py:242-244 then :255-261): qkv[T, 3D] = x Wqkv^T + bias with Wqkv = [Wq; Wk; Wv]
 * ([3D, K], K contiguous), then, in the epilogue and in fp32, apply_rotary_pos_emb (rotate-half,
 * :98-101,178-185) to every head of the q and k thirds -- the position of row r is r % seq -- and
 * q *= q_scale: what ldm_gemm followed by ldm_rope_qk(inverse = 0) leaves, without the second pass
 * (and without rounding q and k to bf16 before the rotation).  cos_t / sin_t: fp32 [seq, hd/2];
 * step: fp32 [2, hd/2] = cos(16 theta_d) | sin(16 theta_d), theta_d = the angle per position of pair d
 * (the epilogue walks down a wave's rows 16 positions at a time by angle addition, so that it needs the
 * tables only once per tile).  BF16; T a multiple of 256, D of 256, hd in {32, 64, 96, 128},
 * seq a multiple of 128 that divides T; otherwise LDM_E_SHAPE / LDM_E_ALIGN (use the two calls). */
int ldm_gemm_qkv_rope(int64_t T, int64_t D, int64_t K, int64_t hd, int64_t seq, const void *x, int64_t ldx,
                      const void *Wqkv, int64_t ldw, const float *bias, const float *cos_t, const float *sin_t,
                      const float *step, float q_scale, void *qkv, int64_t ld_qkv, void *stream);

/* Column sums (bias gradients): out[n] = sum_m X[m,n]; out fp32 [N], overwritten.
 * ws must hold ldm_colsum_ws_floats(M,N) floats. */
int64_t ldm_colsum_ws_floats(int64_t M, int64_t N);
int ldm_colsum(const void *X, int dtype, int64_t M, int64_t N, int64_t ld, float *out, float *ws,
               void *stream);

/* Elementwise cast fp32 -> bf16 (weight shadows), n elements. */
int ldm_cast_f32_to_bf16(const float *src, void *dst, int64_t n, void *stream);

/* dst[c, r] = src[r, c] for bf16 (any 2-byte) elements, src [rows, cols] with row stride ld_src, dst
 * [cols, rows] with row stride ld_dst.  Keeps a K-contiguous copy of each projection weight for the
 * input-gradient product dx = dy W of nn.Linear's backward (src/transformer.py:208-212,288,91-95 under
 * autograd): with W^T stored [in, out] that product reads both operands along K, like the forward. */
int ldm_transpose_bf16(const void *src, int64_t ld_src, void *dst, int64_t ld_dst, int64_t rows, int64_t cols,
                       void *stream);

/* ---------------------------------------------------------------- attention
 * RotaryMultiheadAttention.forward lines 255-285 (src/transformer.py) after the
 * projections and RoPE: non-causal softmax(q k^T) v per (batch, head) with padded
 * keys excluded (key_valid: uint8 [B,S], 1 = attend, NULL = all valid; the
 * reference's masked_fill(-inf) at :271-274).  q is expected pre-scaled.
 * q, k, v are read from a fused [B*S, 3*H*hd] activation (ld_qkv elements per row;
 * q at column 0, k at H*hd, v at 2*H*hd); o is [B*S, H*hd] (ld_o).
 * `saved` keeps what backward needs: BF16: fp32 log-sum-exp [B,H,S] followed by the
 * rounding residual of o (o_exact - o as fp8, in a kernel-private layout; the backward forms rowsum(do * o) from o plus
 * this residual -- with the rounded o alone the q/k gradients of deep layers carry about three times
 * the error of the reference under bf16 autocast); F32: the probabilities [B,H,S,S].
 * Size: ldm_attn_saved_bytes().  `saved` must be 16-byte aligned.
 * Backward writes dq,dk,dv into dqkv (same layout as qkv) from do_ and o.
 * ws: ldm_attn_bwd_ws_bytes() bytes of scratch. */
int64_t ldm_attn_saved_bytes(int dtype, int64_t batch, int64_t nhead, int64_t seq, int64_t hd);
int64_t ldm_attn_bwd_ws_bytes(int dtype, int64_t batch, int64_t nhead, int64_t seq, int64_t hd);
int ldm_attn_fwd(int dtype, const void *qkv, int64_t ld_qkv, const uint8_t *key_valid,
                 int64_t batch, int64_t nhead, int64_t seq, int64_t hd, void *o, int64_t ld_o,
                 void *saved, void *stream);
int ldm_attn_bwd(int dtype, const void *qkv, int64_t ld_qkv, const uint8_t *key_valid,
                 int64_t batch, int64_t nhead, int64_t seq, int64_t hd, const void *o,
                 const void *do_, int64_t ld_o, const void *saved, void *dqkv, void *ws,
                 void *stream);
/* ldm_attn_bwd followed by the backward of ldm_rope_qk in one pass: dq and dk are written as the
 * gradients w.r.t. the q and k BEFORE apply_rotary_pos_emb and the query scale
 * (src/transformer.py:178-185, 255-258 under autograd), i.e. exactly what
 * ldm_attn_bwd + ldm_rope_qk(inverse = 1, q_scale) leave in dqkv.  cos_t / sin_t: fp32 [seq, hd/2].
 * BF16 with hd in {32, 64, 96, 128} only; otherwise LDM_E_SHAPE (use the two calls). */
int ldm_attn_bwd_unrope(int dtype, const void *qkv, int64_t ld_qkv, const uint8_t *key_valid,
                        int64_t batch, int64_t nhead, int64_t seq, int64_t hd, const void *o,
                        const void *do_, int64_t ld_o, const void *saved, void *dqkv, void *ws,
                        const float *cos_t, const float *sin_t, float q_scale, void *stream);
/* The same with dropout on the attention probabilities, `attn_weights = self.dropout(attn_weights)`
 * (src/transformer.py:278): O = (mask * softmax / (1 - p)) v.  Element (batch*head, query, key) is
 * kept iff a 32-bit hash of (seed, its coordinates) is >= round(p * 2^32) (csrc/common.h), so the
 * backward regenerates the forward's mask from (p, seed) and nothing is stored; p in [0, 1), p == 0
 * is ldm_attn_fwd / ldm_attn_bwd exactly.  The reference draws its mask from torch's generator: the
 * two masks agree in distribution, not element by element.
 * ws: F32 forward needs ldm_attn_bwd_ws_bytes() of scratch (NULL for BF16).  Backward: cos_t / sin_t
 * both NULL, or both given to fuse the rotary backward as ldm_attn_bwd_unrope does. */
int ldm_attn_fwd_dropout(int dtype, const void *qkv, int64_t ld_qkv, const uint8_t *key_valid,
                         int64_t batch, int64_t nhead, int64_t seq, int64_t hd, void *o, int64_t ld_o,
                         void *saved, void *ws, float p, uint64_t seed, void *stream);
int ldm_attn_bwd_dropout(int dtype, const void *qkv, int64_t ld_qkv, const uint8_t *key_valid,
                         int64_t batch, int64_t nhead, int64_t seq, int64_t hd, const void *o,
                         const void *do_, int64_t ld_o, const void *saved, void *dqkv, void *ws,
                         const float *cos_t, const float *sin_t, float q_scale, float p, uint64_t seed,
                         void *stream);
/* The forward with the additive mask of `scores = scores + attn_mask` (src/transformer.py:267-268;
 * the model itself passes none, :435).  LDM_F32 only (LDM_E_DTYPE otherwise): the path that
 * materialises the scores.  add_mask: fp32, element (b, h, q, k) at
 * add_mask[b*mask_sb + h*mask_sh + q*mask_sr + k] -- strides in elements, 0 broadcasts that axis, the
 * key axis is contiguous -- added before the key mask and the softmax.  p may be 0 (ws then unused).
 * `saved` holds the resulting probabilities, so ldm_attn_bwd / ldm_attn_bwd_dropout (same p, seed)
 * are its backward unchanged; no gradient flows to the mask. */
int ldm_attn_fwd_addmask(int dtype, const void *qkv, int64_t ld_qkv, const uint8_t *key_valid,
                         const float *add_mask, int64_t mask_sb, int64_t mask_sh, int64_t mask_sr,
                         int64_t batch, int64_t nhead, int64_t seq, int64_t hd, void *o, int64_t ld_o,
                         void *saved, void *ws, float p, uint64_t seed, void *stream);

/* ---------------------------------------------------------------- residual dropout
 * out = res + keep * y / (1 - p): the `src + self.dropout(...)` of the encoder layer
 * (src/transformer.py:352,361) in one pass; res == NULL gives plain dropout, which is also the
 * backward (gy = keep * g / (1 - p) with the forward's p and seed).  y/res/out: [rows, d] contiguous.
 * keep(row, col) is the coordinate hash of the attention dropout (csrc/common.h) with the batch*head
 * slot fixed at 0xffffffff: regenerated, never stored.  p in [0, 1). */
int ldm_dropout_add(const void *y, const void *res, void *out, int64_t rows, int64_t d, int dtype,
                    float p, uint64_t seed, void *stream);
/* The same line of the encoder layer with its projection (src/transformer.py:352 with :288, :361 with
 * :95): C[M, N] = R + keep * (A[M, K] W[N, K]^T + bias) / (1 - p) from ONE GEMM, the mask applied in the
 * epilogue before the residual joins.  keep(row, col) is that of ldm_dropout_add with the same p and seed,
 * so the backward is ldm_dropout_add(g, NULL, ...) followed by the projection's usual backward.  BF16; A,
 * W K-contiguous; M and N multiples of 256, row strides multiples of 8, C and R 16-byte aligned, bias
 * fp32 [N] or NULL; otherwise LDM_E_SHAPE / LDM_E_ALIGN (the caller then runs ldm_gemm and
 * ldm_dropout_add). */
int ldm_gemm_dropout_residual(int64_t M, int64_t N, int64_t K, const void *A, int64_t lda, const void *W,
                              int64_t ldw, const float *bias, const void *R, int64_t ldr, void *C, int64_t ldc,
                              float p, uint64_t seed, void *stream);

/* ---------------------------------------------------------------- K4: masked cross-entropy
 * masked_cross_entropy_loss / variational_lower_bound_loss (src/loss.py:8-65).
 * A row r of logits [T, V] (ld elements per row) takes part when
 *   (mask_bool ? mask_bool[r] : masked_ids[r] == mask_token_id) and
 *   (att_valid == NULL or att_valid[r] != 0).
 * loss = sum_r w_r * (logsumexp(logits_r) - logits_r[target_r]) / norm, where
 *   w_r = seq_weight ? seq_weight[r / seq] : 1      (reference: 1)
 *   norm = norm_mode 0: number of participating rows (reference: mean over masked)
 *          norm_mode 1: T (the paper's B*L)
 * and loss = 0 when no row participates.  Only participating rows are read.
 * Outputs: row_lse fp32 [T] (valid for participating rows), loss_out fp32[1],
 * count_out fp32[1] (= number of participating rows).  row_tmp: fp32 [T] scratch.
 * Backward writes dlogits[T, V] (dtype, ld_d per row; zero for other rows and for
 * padding columns V..ld_d-1): (softmax - onehot) * w_r * gout[0] / norm. */
int ldm_ce_fwd(const void *logits, int dtype, int64_t ld, int64_t T, int64_t V, int64_t seq,
               const int64_t *targets, const int64_t *masked_ids, int64_t mask_token_id,
               const uint8_t *mask_bool, const uint8_t *att_valid, const float *seq_weight,
               int norm_mode, float *row_lse, float *row_tmp, float *loss_out, float *count_out,
               void *stream);
int ldm_ce_bwd(const void *logits, int dtype, int64_t ld, int64_t T, int64_t V, int64_t seq,
               const int64_t *targets, const int64_t *masked_ids, int64_t mask_token_id,
               const uint8_t *mask_bool, const uint8_t *att_valid, const float *seq_weight,
               int norm_mode, const float *row_lse, const float *count, const float *gout,
               void *dlogits, int64_t ld_d, void *stream);

/* ---------------------------------------------------------------- LM-head backward over masked rows only
 * The cross-entropy gradient is zero on rows that do not participate (about half of all rows
 * for t ~ U(0,1)), so the two backward GEMMs of the vocabulary projection
 * (src/transformer.py:441 under autograd) can skip them.  The row count is known only on the
 * device, so these forms take it as a device pointer and never synchronise with the host.
 *   ldm_ce_compact_rows : rows[0..count) = ascending indices of participating rows
 *   ldm_ce_bwd_compact  : as ldm_ce_bwd but writes compact row i (= source row rows[i]) of
 *                         dlogits [T, ld_d]; rows count..roundup(count,64) are zeroed
 *   ldm_gather_rows     : out[i,:] = x[rows[i],:] (same zero rounding rows); out has
 *                         roundup(max_rows,64) rows
 *   ldm_scatter_rows    : out[rows[i],:] = x[i,:] for i < *count (x has max_rows rows, out has
 *                         out_rows; the other rows of out are left as they are)
 *   ldm_gemm_rows_dyn   : C[row_map[m], :] = A[m, :K] * opB(B) (+ bias) for m < *m_dev; row_map may
 *                         be NULL (identity), bias may be NULL                 (BF16, MFMA).
 *                         A holds all M rows (rows >= *m_dev may be read, their products are dropped);
 *                         rows >= *m_dev of C are not written
 *   ldm_gemm_acc_kdyn   : C(fp32) += opA(A) * opB(B) with K = roundup(*k_dev, 32) (BF16, MFMA);
 *                         a_rowsum (nullable) as in ldm_gemm_acc_rowsum
 *   ldm_colsum_rows_dyn : ldm_colsum over the first *m_dev rows
 * The GEMM forms return LDM_E_SHAPE / LDM_E_ALIGN for problems the MFMA kernel does not take. */
int ldm_ce_compact_rows(int64_t T, int64_t seq, const int64_t *masked_ids, int64_t mask_token_id,
                        const uint8_t *mask_bool, const uint8_t *att_valid, int32_t *rows, int32_t *count,
                        void *stream);
int ldm_ce_bwd_compact(const void *logits, int dtype, int64_t ld, int64_t T, int64_t V, int64_t seq,
                       const int64_t *targets, const float *seq_weight, int norm_mode, const float *row_lse,
                       const float *count, const float *gout, const int32_t *rows, const int32_t *count_i,
                       void *dlogits, int64_t ld_d, void *stream);
int ldm_gather_rows(const void *x, int dtype, int64_t ld, const int32_t *rows, const int32_t *count,
                    int64_t max_rows, int64_t d, void *out, void *stream);
int ldm_scatter_rows(const void *x, int dtype, int64_t ld, const int32_t *rows, const int32_t *count,
                     int64_t max_rows, int64_t out_rows, int64_t d, void *out, int64_t ld_out, void *stream);
int ldm_gemm_rows_dyn(int out_dtype, int transB, int64_t M, int64_t N, int64_t K, const void *A, int64_t lda,
                      const void *B, int64_t ldb, void *C, int64_t ldc, const float *bias, const int32_t *m_dev,
                      const int32_t *row_map, void *stream);
int ldm_gemm_acc_kdyn(int transA, int transB, int64_t M, int64_t N, int64_t K, const void *A, int64_t lda,
                      const void *B, int64_t ldb, float *C, int64_t ldc, const int32_t *k_dev, float *a_rowsum,
                      void *stream);
int ldm_colsum_rows_dyn(const void *X, int dtype, int64_t M, int64_t N, int64_t ld, float *out, float *ws,
                        const int32_t *m_dev, void *stream);

/* Loss and its gradient in ONE pass over compact logit rows (training step, LM head computed on the
 * participating rows only): masked_cross_entropy_loss (src/loss.py:12-28) and the backward of
 * F.cross_entropy (src/loss.py:22 under autograd) together.  Rows 0..*count-1 of logits
 * [rows_alloc, ld] all take part, targets[i] is the label of compact row i.  Each row is read once:
 *   row_tmp[i] = w_i (logsumexp(row) - row[target]),   row := w_i (softmax(row) - onehot(target)) / norm
 * with w_i = row_weight ? row_weight[i] : 1 (reference: 1; the paper-form objective hands in 1/t of the
 * row's sequence) and norm = *count (norm_mode 0, the reference's mean over masked positions) or norm_T
 * (norm_mode 1, the paper's B*L); columns V..ld-1 are written as zero and
 * rows *count..roundup(*count, 64)-1 are zeroed.  loss_out[0] = sum_i row_tmp[i] / norm (0 when
 * *count == 0), count_out[0] = *count.  The gradient left in logits is for an upstream gradient of
 * 1; ldm_scale_rows_dev applies another one (x[i, :ncols] *= *scale for i < *count; a launch that
 * returns at once when *scale == 1).  V up to 131072; 16-byte aligned rows (ld % 8 == 0). */
int ldm_ce_fwd_grad(void *logits, int dtype, int64_t ld, int64_t rows_alloc, int64_t V,
                    const int64_t *targets, const int32_t *count, int norm_mode, float norm_T,
                    const float *row_weight, float *row_lse, float *row_tmp, float *loss_out, float *count_out,
                    void *stream);
int ldm_scale_rows_dev(void *x, int dtype, int64_t ld, int64_t rows_alloc, int64_t ncols,
                       const int32_t *count, const float *scale, void *stream);

/* ---------------------------------------------------------------- optimizer step
 * torch.optim.AdamW as the reference builds it (main.py:87-91: lr, weight_decay; betas (0.9, 0.999),
 * eps 1e-8 by default) and steps it (src/trainer.py:75,137), for `count` tensors in one pass:
 *   g' = g * grad_scale;  p *= 1 - lr * weight_decay;  m += (g' - m)(1 - beta1);
 *   v = beta2 v + (1 - beta2) g'^2;  p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * `step` counts from 1.  grad_scale folds the 1 / world_size of a summing gradient all-reduce into the
 * update.  shadow (nullable): a second copy of the updated parameter written by the same pass --
 * shadow_kind 1 = bf16 (the weight copy the bf16 GEMMs read), 2 = fp32.  `tensors` is a HOST array;
 * every pointer inside is a 16-byte aligned device pointer (bf16 shadow: 8-byte). */
typedef struct {
  float *p;
  const float *g;
  float *m, *v;
  void *shadow;
  int64_t n;
  int32_t shadow_kind;
} ldm_adamw_tensor;
int ldm_adamw_step(const ldm_adamw_tensor *tensors, int64_t count, int64_t step, float lr, float beta1,
                   float beta2, float eps, float weight_decay, float grad_scale, void *stream);
/* The same update behind a device-side gate, for the step whose loss had no participating position:
 * the reference returns a constant there (src/loss.py:18-19), every .grad stays None and
 * optimizer.step() (src/trainer.py:75,137) changes nothing -- no decay, no moment update, no step
 * count.  gate: device fp32[1], the count_out of the loss kernels; when *gate == 0 the launch leaves
 * p, m, v and the shadows as they are.  skipped (nullable): device int32[1], the number of updates
 * gated off so far; `step` is the host's count including those, and the bias corrections use
 * step - *skipped.  ldm_adamw_gate_commit adds 1 to *skipped when *gate == 0: call it once after the
 * last ldm_adamw_step_gated of an optimizer step.  No host synchronisation anywhere. */
int ldm_adamw_step_gated(const ldm_adamw_tensor *tensors, int64_t count, int64_t step, float lr, float beta1,
                         float beta2, float eps, float weight_decay, float grad_scale, const float *gate,
                         const int32_t *skipped, void *stream);
int ldm_adamw_gate_commit(const float *gate, int32_t *skipped, void *stream);

/* ---------------------------------------------------------------- gradient all-reduce (RCCL)
 * What DistributedDataParallel does for the reference (main.py:105-108 wraps the model, the bucketed
 * all-reduce runs inside accelerator.backward, src/trainer.py:136): one process per GPU, a flat bucket
 * of gradients summed in place over the ranks while backward goes on.
 * RCCL is found at run time (dlopen; an instance the process already holds -- PyTorch's -- is the one
 * used; LDM_RCCL_LIB names another file), so a communicator made by the caller's own ncclCommInitRank
 * can be handed in as `comm` (an ncclComm_t).  The three ldm_comm_* calls make one without touching
 * RCCL's headers: rank 0 fills a 128-byte id (ldm_comm_unique_id), hands it to the other ranks by any
 * channel, and every rank calls ldm_comm_init_rank with its GPU current.
 * ldm_allreduce_bucket: sum of buf[0..count) (dtype LDM_F32 or LDM_BF16, 16-byte aligned) over the
 *   ranks, in place, queued on comm_stream behind everything producer_stream holds at the call (the
 *   kernels that wrote the gradients).  Returns at once; calls on one communicator must come in the same
 *   order on every rank.
 * ldm_allreduce_join: what consumer_stream gets from now on (the optimizer step) runs behind what
 *   comm_stream holds now.
 * Errors: LDM_E_COMM (text in ldm_comm_last_error()), HIP error codes as positive values. */
int ldm_comm_unique_id(void *id128);
int ldm_comm_init_rank(void **comm, int nranks, const void *id128, int rank);
int ldm_comm_size(void *comm, int *nranks);
int ldm_comm_destroy(void *comm);
const char *ldm_comm_last_error(void);
int ldm_allreduce_bucket(void *comm, void *buf, int64_t count, int dtype, void *producer_stream,
                         void *comm_stream);
int ldm_allreduce_join(void *comm_stream, void *consumer_stream);

/* ---------------------------------------------------------------- batch producer
 * The batch contract of the training step, {"input_ids", "attention_mask"}: int64 [batch, smax] each
 * (src/dataset.py:48, 126-127; src/trainer.py:120-121).
 * ldm_collate_pad: collate_batch (src/dataset.py:26-48: pad_sequence of the ids and of the masks, padding
 *   value 0) from the UNPADDED tokens: flat = the sequences of the batch back to back, offsets[batch+1]
 *   their starts (both device pointers); ids_out[b, s] = flat[offsets[b] + s] or pad_id, mask_out = 1 / 0.
 *   Only the real tokens cross the host link; smax >= the longest sequence.
 * ldm_synth_tokens: synthetic batches made on the device.  Flat position i = b * seq + s uses word i & 3
 *   of Philox4x32-10 block (counter = (i >> 2, offset), key = seed): id = id_lo + (word * (id_hi - id_lo)) >> 32;
 *   column 0 = bos_id, column len-1 = eos_id, columns >= len = pad_id with mask 0, where
 *   len = seq when min_len >= seq, else min_len + (word0(counter = (b, offset ^ 2^31)) * (seq - min_len + 1)) >> 32
 *   (min_len is raised to 2).  No counterpart in the reference (its data comes from a text file). */
int ldm_collate_pad(const int64_t *flat, const int64_t *offsets, int64_t batch, int64_t smax,
                    int64_t pad_id, int64_t *ids_out, int64_t *mask_out, void *stream);
int ldm_synth_tokens(int64_t batch, int64_t seq, int64_t id_lo, int64_t id_hi, int64_t bos_id,
                     int64_t eos_id, int64_t pad_id, int64_t min_len, uint64_t seed, uint64_t offset,
                     int64_t *ids_out, int64_t *mask_out, void *stream);

/* ---------------------------------------------------------------- reverse process: token selection
 * ldm_sample_tokens replaces what DiffusionLanguageModel.generate / sample do with the logits
 * (src/model.py:93-105 and :167-177): for every row whose id is mask_id (all_rows != 0: for
 * every row) pick a column of logits[T, V] (row stride ld):
 *   temperature == 0: argmax, the first maximum as torch.argmax returns it;
 *   temperature  > 0: argmax(logit / temperature + Gumbel noise), a draw from
 *                     softmax(logits / temperature) -- the distribution the reference hands to
 *                     torch.multinomial; the noise is Philox4x32-10 keyed by (seed, offset), so
 *                     the STREAM differs from torch's.
 * filled_out[T] (may be null) = ids with the mask positions replaced by the pick;
 * pred_out[T]   (may be null) = the pick of every visited row (other rows are not written).
 *
 * ldm_remask_eligible + ldm_remask_apply replace RandomRemaskingStrategy.apply_remask
 * (src/masking_strategy.py:160-198): pos_out[0..*count_out) = flat indices, ascending (the order
 * of Tensor.nonzero()), of the positions that were masked in this step, are not bos/eos/pad and
 * are not prompt positions (prompt may be null); then out = ids with out[pos[perm[j]]] = mask_id
 * for j < k, where perm is the caller's permutation of 0..num_eligible-1 (torch.randperm in the
 * reference, :189) and k = int(num_eligible * remask_ratio) (:182). */
int ldm_sample_tokens(const void *logits, int dtype, int64_t T, int64_t V, int64_t ld, const int64_t *ids,
                      int64_t mask_id, int all_rows, float temperature, uint64_t seed, uint64_t offset,
                      int64_t *filled_out, int64_t *pred_out, void *stream);
int ldm_remask_eligible(const int64_t *ids, const uint8_t *was_masked, const uint8_t *prompt, int64_t n,
                        int64_t bos_id, int64_t eos_id, int64_t pad_id, int64_t *pos_out, int32_t *count_out,
                        void *stream);
int ldm_remask_apply(const int64_t *ids, int64_t n, const int64_t *pos, int64_t num_eligible,
                     const int64_t *perm, int64_t k, int64_t mask_id, int64_t *out, void *stream);

/* ---------------------------------------------------------------- fp32 softmax helpers
 * Used by the F32 attention path (materialised scores, as the reference does at
 * src/transformer.py:264-281).  Exposed for testing. */
int ldm_softmax_rows_f32(float *scores, const uint8_t *key_valid, int64_t batch, int64_t nhead,
                         int64_t seq_q, int64_t seq_k, void *stream);
int ldm_softmax_bwd_rows_f32(const float *probs, float *dprobs_inout, int64_t rows, int64_t cols,
                             void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LDM_HIP_H */
