// F32 attention (parity mode): scores are materialised exactly as the reference does
// (src/transformer.py:264-281): S = q k^T, masked_fill(-inf) on padded keys, softmax,
// O = P v; backward by the transposed products plus the softmax Jacobian.  Built from
// the strided batched FMA GEMM, so every accumulation is plain fp32.
#include "common.h"
#include "gemm.h"
#include "attn.h"

namespace {

// the additive mask of one launch: element (b, h, q, k) at add[b*sb + h*sh + q*sr + k] (strides in
// elements; 0 broadcasts), added to the scores before the key mask as src/transformer.py:267-274 does
struct AddMask {
  const float *add;
  int64_t sb, sh, sr, nhead, seq_q;
};

// one wave per score row, in place
__global__ __launch_bounds__(256) void softmax_rows_k(float *scores, const uint8_t *key_valid, int64_t rows,
                                                      int64_t rows_per_batch, int64_t seq_k, AddMask am) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < rows; r += nwaves) {
    float *row = scores + r * seq_k;
    const uint8_t *kv = key_valid ? key_valid + (r / rows_per_batch) * seq_k : nullptr;
    if (am.add) {   // row r = ((b * nhead) + h) * seq_q + q
      const int64_t q = r % am.seq_q, bh = r / am.seq_q;
      const float *ar = am.add + (bh / am.nhead) * am.sb + (bh % am.nhead) * am.sh + q * am.sr;
      for (int64_t c = lane; c < seq_k; c += 64) row[c] += ar[c];   // each lane re-reads only its own columns
    }
    float mx = -INFINITY;
    for (int64_t c = lane; c < seq_k; c += 64) {
      const float v = (kv && !kv[c]) ? -INFINITY : row[c];
      mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int64_t c = lane; c < seq_k; c += 64) {
      const float v = (kv && !kv[c]) ? -INFINITY : row[c];
      sum += expf(v - mx);
    }
    sum = wave_sum(sum);
    for (int64_t c = lane; c < seq_k; c += 64) {
      const float v = (kv && !kv[c]) ? -INFINITY : row[c];
      row[c] = expf(v - mx) / sum;
    }
  }
}

// dS = P * (dP - sum_k P*dP), in place on dP.  With dropout (thresh != 0; row r is query r % seq_q of
// batch*head r / seq_q) the incoming dP is first masked and scaled like the probabilities were.
__global__ __launch_bounds__(256) void softmax_bwd_rows_k(const float *probs, float *dprobs, int64_t rows,
                                                          int64_t cols, int64_t seq_q, unsigned thresh, float inv,
                                                          unsigned seed_lo, unsigned seed_hi) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float *p = probs + r * cols;
    float *d = dprobs + r * cols;
    unsigned rowkey = 0;
    if (thresh) {
      rowkey = ldm_drop_row(ldm_drop_bh(seed_lo, (unsigned)(r / seq_q)), seed_hi, (unsigned)(r % seq_q));
      for (int64_t c = lane; c < cols; c += 64) d[c] = ldm_drop_keep(rowkey, (unsigned)c, thresh) ? d[c] * inv : 0.f;
    }
    float dot = 0.f;
    for (int64_t c = lane; c < cols; c += 64) dot = fmaf(p[c], d[c], dot);
    dot = wave_sum(dot);
    for (int64_t c = lane; c < cols; c += 64) d[c] = p[c] * (d[c] - dot);
  }
}

// out = mask * P / (1 - p): the probabilities as the value product sees them under dropout
__global__ __launch_bounds__(256) void dropout_probs_k(const float *probs, float *out, int64_t rows, int64_t cols,
                                                       int64_t seq_q, unsigned thresh, float inv, unsigned seed_lo,
                                                       unsigned seed_hi) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const unsigned rowkey = ldm_drop_row(ldm_drop_bh(seed_lo, (unsigned)(r / seq_q)), seed_hi, (unsigned)(r % seq_q));
    for (int64_t c = lane; c < cols; c += 64)
      out[r * cols + c] = ldm_drop_keep(rowkey, (unsigned)c, thresh) ? probs[r * cols + c] * inv : 0.f;
  }
}

inline float drop_inv_of(unsigned thresh) { return (float)(4294967296.0 / (4294967296.0 - (double)thresh)); }

inline unsigned wave_grid(int64_t rows) {
  int64_t b = ceil_div(rows, 4);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

static int softmax_rows_launch(float *scores, const uint8_t *key_valid, int64_t batch, int64_t nhead, int64_t seq_q,
                               int64_t seq_k, AddMask am, void *stream) {
  if (batch < 0 || nhead < 0 || seq_q < 0 || seq_k < 0) return LDM_E_BADARG;
  const int64_t rows = batch * nhead * seq_q;
  if (rows == 0 || seq_k == 0) return LDM_OK;
  if (!scores) return LDM_E_BADARG;
  hipLaunchKernelGGL(softmax_rows_k, dim3(wave_grid(rows)), dim3(256), 0, (hipStream_t)stream, scores, key_valid,
                     rows, nhead * seq_q, seq_k, am);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

extern "C" int ldm_softmax_rows_f32(float *scores, const uint8_t *key_valid, int64_t batch, int64_t nhead,
                                    int64_t seq_q, int64_t seq_k, void *stream) {
  return softmax_rows_launch(scores, key_valid, batch, nhead, seq_q, seq_k, AddMask{nullptr, 0, 0, 0, 1, 1}, stream);
}

extern "C" int ldm_softmax_bwd_rows_f32(const float *probs, float *dprobs_inout, int64_t rows, int64_t cols,
                                        void *stream) {
  if (rows < 0 || cols < 0) return LDM_E_BADARG;
  if (rows == 0 || cols == 0) return LDM_OK;
  if (!probs || !dprobs_inout) return LDM_E_BADARG;
  hipLaunchKernelGGL(softmax_bwd_rows_k, dim3(wave_grid(rows)), dim3(256), 0, (hipStream_t)stream, probs,
                     dprobs_inout, rows, cols, (int64_t)1, 0u, 1.0f, 0u, 0u);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

int ldm_attn_fwd_f32(const float *qkv, int64_t ld, const uint8_t *key_valid, int64_t B, int64_t H, int64_t S,
                     int64_t hd, float *o, int64_t ld_o, float *probs, float *dropped, unsigned drop_thresh,
                     uint64_t seed, hipStream_t s, const float *add_mask, int64_t mask_sb, int64_t mask_sh,
                     int64_t mask_sr) {
  if (drop_thresh && !dropped) return LDM_E_WORKSPACE;
  const int64_t D = H * hd;
  const ldm_bstride s_qkv{S * ld, hd}, s_p{H * S * S, S * S}, s_o{S * ld_o, hd};
  // S = q k^T : A = q [S,hd], B = k stored [S_k, hd] (transB)
  int rc = ldm_gemm_generic(LDM_F32, LDM_F32, 0, 1, S, S, hd, 1.0f, qkv, ld, s_qkv, qkv + D, ld, s_qkv, probs, S,
                            s_p, nullptr, nullptr, 0, B * H, H, s);
  if (rc) return rc;
  rc = softmax_rows_launch(probs, key_valid, B, H, S, S, AddMask{add_mask, mask_sb, mask_sh, mask_sr, H, S}, s);
  if (rc) return rc;
  const float *pv = probs;
  if (drop_thresh) {   // the saved probabilities stay undropped (the softmax Jacobian needs them)
    hipLaunchKernelGGL(dropout_probs_k, dim3(wave_grid(B * H * S)), dim3(256), 0, s, probs, dropped, B * H * S, S, S,
                       drop_thresh, drop_inv_of(drop_thresh), (unsigned)seed, (unsigned)(seed >> 32));
    LDM_CHECK_LAUNCH();
    pv = dropped;
  }
  // O = P v : B = v stored [S_k, hd]
  return ldm_gemm_generic(LDM_F32, LDM_F32, 0, 0, S, hd, S, 1.0f, pv, S, s_p, qkv + 2 * D, ld, s_qkv, o, ld_o,
                          s_o, nullptr, nullptr, 0, B * H, H, s);
}

int ldm_attn_bwd_f32(const float *qkv, int64_t ld, int64_t B, int64_t H, int64_t S, int64_t hd, const float *dout,
                     int64_t ld_o, const float *probs, float *dqkv, float *dprobs, unsigned drop_thresh, uint64_t seed,
                     hipStream_t s) {
  const int64_t D = H * hd;
  const ldm_bstride s_qkv{S * ld, hd}, s_p{H * S * S, S * S}, s_o{S * ld_o, hd};
  const float inv = drop_inv_of(drop_thresh);
  const float *pv = probs;
  if (drop_thresh) {   // the workspace first holds the dropped probabilities for dV, then dP
    hipLaunchKernelGGL(dropout_probs_k, dim3(wave_grid(B * H * S)), dim3(256), 0, s, probs, dprobs, B * H * S, S, S,
                       drop_thresh, inv, (unsigned)seed, (unsigned)(seed >> 32));
    LDM_CHECK_LAUNCH();
    pv = dprobs;
  }
  // dV = P^T dO : A stored [S_q(k), S_k(m)] = P (transA), B = dO [S_q, hd]
  int rc = ldm_gemm_generic(LDM_F32, LDM_F32, 1, 0, S, hd, S, 1.0f, pv, S, s_p, dout, ld_o, s_o, dqkv + 2 * D,
                            ld, s_qkv, nullptr, nullptr, 0, B * H, H, s);
  if (rc) return rc;
  // dP = dO v^T : B = v stored [S_k(n), hd(k)] (transB)
  rc = ldm_gemm_generic(LDM_F32, LDM_F32, 0, 1, S, S, hd, 1.0f, dout, ld_o, s_o, qkv + 2 * D, ld, s_qkv, dprobs, S,
                        s_p, nullptr, nullptr, 0, B * H, H, s);
  if (rc) return rc;
  hipLaunchKernelGGL(softmax_bwd_rows_k, dim3(wave_grid(B * H * S)), dim3(256), 0, s, probs, dprobs, B * H * S, S, S,
                     drop_thresh, inv, (unsigned)seed, (unsigned)(seed >> 32));
  LDM_CHECK_LAUNCH();
  // dQ = dS k : B = k stored [S_k(k), hd(n)]
  rc = ldm_gemm_generic(LDM_F32, LDM_F32, 0, 0, S, hd, S, 1.0f, dprobs, S, s_p, qkv + D, ld, s_qkv, dqkv, ld,
                        s_qkv, nullptr, nullptr, 0, B * H, H, s);
  if (rc) return rc;
  // dK = dS^T q : A stored [S_q(k), S_k(m)] (transA), B = q [S_q(k), hd(n)]
  return ldm_gemm_generic(LDM_F32, LDM_F32, 1, 0, S, hd, S, 1.0f, dprobs, S, s_p, qkv, ld, s_qkv, dqkv + D, ld,
                          s_qkv, nullptr, nullptr, 0, B * H, H, s);
}
