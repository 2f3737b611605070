// C-ABI attention dispatch: BF16 -> flash kernels, F32 -> materialised-score path.
#include "common.h"
#include "attn.h"
#include <stdlib.h>

// BF16 `saved`: fp32 log-sum-exp [B,H,S], then (256-byte aligned) the bf16 rounding residual of O in a
// layout private to the kernels, which the backward adds to O when it forms delta = rowsum(dO * O)
// LDM_ATTN_OLO=0: do not write / use the residual of O (A/B measurements of its cost and of its effect)
static bool olo_enabled() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("LDM_ATTN_OLO");
    v = (e && atoi(e) == 0) ? 0 : 1;
  }
  return v == 1;
}

static int64_t lse_bytes(int64_t batch, int64_t nhead, int64_t seq) {
  return (batch * nhead * seq * (int64_t)sizeof(float) + 255) / 256 * 256;
}

extern "C" int64_t ldm_attn_saved_bytes(int dtype, int64_t batch, int64_t nhead, int64_t seq, int64_t hd) {
  if (dtype == LDM_BF16) {
    // residual of O: one slot of DP (head dim rounded up to 32/64/96/128) per row of every 128-row block
    const int64_t dp = hd <= 32 ? 32 : hd <= 64 ? 64 : hd <= 96 ? 96 : 128;
    return lse_bytes(batch, nhead, seq) + batch * nhead * ((seq + 127) / 128) * 128 * dp;   // one byte per element
  }
  return batch * nhead * seq * seq * (int64_t)sizeof(float);
}

extern "C" int64_t ldm_attn_bwd_ws_bytes(int dtype, int64_t batch, int64_t nhead, int64_t seq, int64_t hd) {
  (void)hd;
  if (dtype == LDM_BF16) return batch * nhead * seq * (int64_t)sizeof(float);   // delta
  return batch * nhead * seq * seq * (int64_t)sizeof(float);
}

// keep threshold of the dropout hash for probability p (see common.h); -1 for p outside [0, 1)
static int64_t drop_threshold(float p) {
  if (!(p >= 0.0f) || !(p < 1.0f)) return -1;
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 4294967295ll : (int64_t)t;
}

static int attn_fwd_impl(int dtype, const void *qkv, int64_t ld_qkv, const uint8_t *key_valid, int64_t batch,
                         int64_t nhead, int64_t seq, int64_t hd, void *o, int64_t ld_o, void *saved, void *ws,
                         unsigned thresh, uint64_t seed, void *stream, const float *add_mask = nullptr,
                         int64_t mask_sb = 0, int64_t mask_sh = 0, int64_t mask_sr = 0) {
  if (batch < 0 || nhead <= 0 || seq < 0 || hd <= 0) return LDM_E_BADARG;
  if (batch == 0 || seq == 0) return LDM_OK;
  if (!qkv || !o || !saved) return LDM_E_BADARG;
  if (ld_qkv < 3 * nhead * hd || ld_o < nhead * hd) return LDM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (add_mask && dtype != LDM_F32) return LDM_E_DTYPE;
  if (dtype == LDM_BF16)
    return ldm_flash_fwd_bf16((const bf16_t *)qkv, ld_qkv, key_valid, batch, nhead, seq, hd, (bf16_t *)o, ld_o,
                              (float *)saved, olo_enabled() ? (bf16_t *)((char *)saved + lse_bytes(batch, nhead, seq)) : nullptr,
                              thresh, seed, s);
  if (dtype == LDM_F32)
    return ldm_attn_fwd_f32((const float *)qkv, ld_qkv, key_valid, batch, nhead, seq, hd, (float *)o, ld_o,
                            (float *)saved, (float *)ws, thresh, seed, s, add_mask, mask_sb, mask_sh, mask_sr);
  return LDM_E_DTYPE;
}

extern "C" int ldm_attn_fwd_addmask(int dtype, const void *qkv, int64_t ld_qkv, const uint8_t *key_valid,
                                    const float *add_mask, int64_t mask_sb, int64_t mask_sh, int64_t mask_sr,
                                    int64_t batch, int64_t nhead, int64_t seq, int64_t hd, void *o, int64_t ld_o,
                                    void *saved, void *ws, float p, uint64_t seed, void *stream) {
  const int64_t t = drop_threshold(p);
  if (t < 0 || !add_mask || mask_sb < 0 || mask_sh < 0 || mask_sr < 0) return LDM_E_BADARG;
  if (dtype != LDM_F32) return LDM_E_DTYPE;
  if (t > 0 && !ws) return LDM_E_WORKSPACE;
  return attn_fwd_impl(dtype, qkv, ld_qkv, key_valid, batch, nhead, seq, hd, o, ld_o, saved, ws, (unsigned)t, seed,
                       stream, add_mask, mask_sb, mask_sh, mask_sr);
}

extern "C" int ldm_attn_fwd(int dtype, const void *qkv, int64_t ld_qkv, const uint8_t *key_valid, int64_t batch,
                            int64_t nhead, int64_t seq, int64_t hd, void *o, int64_t ld_o, void *saved,
                            void *stream) {
  return attn_fwd_impl(dtype, qkv, ld_qkv, key_valid, batch, nhead, seq, hd, o, ld_o, saved, nullptr, 0u, 0, stream);
}

extern "C" int ldm_attn_fwd_dropout(int dtype, const void *qkv, int64_t ld_qkv, const uint8_t *key_valid,
                                    int64_t batch, int64_t nhead, int64_t seq, int64_t hd, void *o, int64_t ld_o,
                                    void *saved, void *ws, float p, uint64_t seed, void *stream) {
  const int64_t t = drop_threshold(p);
  if (t < 0) return LDM_E_BADARG;
  if (t > 0 && dtype == LDM_F32 && !ws) return LDM_E_WORKSPACE;
  return attn_fwd_impl(dtype, qkv, ld_qkv, key_valid, batch, nhead, seq, hd, o, ld_o, saved, ws, (unsigned)t, seed,
                       stream);
}

static int attn_bwd_impl(int dtype, const void *qkv, int64_t ld_qkv, const uint8_t *key_valid, int64_t batch,
                         int64_t nhead, int64_t seq, int64_t hd, const void *o, const void *do_, int64_t ld_o,
                         const void *saved, void *dqkv, void *ws, const float *rope_cos, const float *rope_sin,
                         float rope_scale, unsigned thresh, uint64_t seed, void *stream) {
  if (batch < 0 || nhead <= 0 || seq < 0 || hd <= 0) return LDM_E_BADARG;
  if (batch == 0 || seq == 0) return LDM_OK;
  if (!qkv || !o || !do_ || !saved || !dqkv || !ws) return LDM_E_BADARG;
  if (ld_qkv < 3 * nhead * hd || ld_o < nhead * hd) return LDM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == LDM_BF16)
    return ldm_flash_bwd_bf16((const bf16_t *)qkv, ld_qkv, key_valid, batch, nhead, seq, hd, (const bf16_t *)o,
                              (const bf16_t *)do_, ld_o, (const float *)saved,
                              olo_enabled() ? (const bf16_t *)((const char *)saved + lse_bytes(batch, nhead, seq)) : nullptr,
                              (bf16_t *)dqkv,
                              (float *)ws, rope_cos, rope_sin, rope_scale, thresh, seed, s);
  if (rope_cos) return LDM_E_SHAPE;
  if (dtype == LDM_F32)
    return ldm_attn_bwd_f32((const float *)qkv, ld_qkv, batch, nhead, seq, hd, (const float *)do_, ld_o,
                            (const float *)saved, (float *)dqkv, (float *)ws, thresh, seed, s);
  return LDM_E_DTYPE;
}

extern "C" int ldm_attn_bwd(int dtype, const void *qkv, int64_t ld_qkv, const uint8_t *key_valid, int64_t batch,
                            int64_t nhead, int64_t seq, int64_t hd, const void *o, const void *do_, int64_t ld_o,
                            const void *saved, void *dqkv, void *ws, void *stream) {
  return attn_bwd_impl(dtype, qkv, ld_qkv, key_valid, batch, nhead, seq, hd, o, do_, ld_o, saved, dqkv, ws, nullptr,
                       nullptr, 1.0f, 0u, 0, stream);
}

extern "C" int ldm_attn_bwd_unrope(int dtype, const void *qkv, int64_t ld_qkv, const uint8_t *key_valid,
                                   int64_t batch, int64_t nhead, int64_t seq, int64_t hd, const void *o,
                                   const void *do_, int64_t ld_o, const void *saved, void *dqkv, void *ws,
                                   const float *cos_t, const float *sin_t, float q_scale, void *stream) {
  if (!cos_t || !sin_t) return LDM_E_BADARG;
  return attn_bwd_impl(dtype, qkv, ld_qkv, key_valid, batch, nhead, seq, hd, o, do_, ld_o, saved, dqkv, ws, cos_t,
                       sin_t, q_scale, 0u, 0, stream);
}

extern "C" int ldm_attn_bwd_dropout(int dtype, const void *qkv, int64_t ld_qkv, const uint8_t *key_valid,
                                    int64_t batch, int64_t nhead, int64_t seq, int64_t hd, const void *o,
                                    const void *do_, int64_t ld_o, const void *saved, void *dqkv, void *ws,
                                    const float *cos_t, const float *sin_t, float q_scale, float p, uint64_t seed,
                                    void *stream) {
  const int64_t t = drop_threshold(p);
  if (t < 0 || (cos_t == nullptr) != (sin_t == nullptr)) return LDM_E_BADARG;
  return attn_bwd_impl(dtype, qkv, ld_qkv, key_valid, batch, nhead, seq, hd, o, do_, ld_o, saved, dqkv, ws, cos_t,
                       sin_t, cos_t ? q_scale : 1.0f, (unsigned)t, seed, stream);
}
