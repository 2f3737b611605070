// Internal attention entry points (C-ABI wrappers live in attn_api.hip).
#pragma once
#include "common.h"

int ldm_attn_fwd_f32(const float *qkv, int64_t ld, const uint8_t *key_valid, int64_t B, int64_t H, int64_t S,
                     int64_t hd, float *o, int64_t ld_o, float *probs, float *dropped, unsigned drop_thresh,
                     uint64_t seed, hipStream_t s, const float *add_mask, int64_t mask_sb, int64_t mask_sh,
                     int64_t mask_sr);
int ldm_attn_bwd_f32(const float *qkv, int64_t ld, int64_t B, int64_t H, int64_t S, int64_t hd, const float *dout,
                     int64_t ld_o, const float *probs, float *dqkv, float *dprobs, unsigned drop_thresh, uint64_t seed,
                     hipStream_t s);

int ldm_flash_fwd_bf16(const bf16_t *qkv, int64_t ld, const uint8_t *key_valid, int64_t B, int64_t H, int64_t S,
                       int64_t hd, bf16_t *o, int64_t ld_o, float *lse, bf16_t *o_lo, unsigned drop_thresh,
                       uint64_t seed, hipStream_t s);
int ldm_flash_bwd_bf16(const bf16_t *qkv, int64_t ld, const uint8_t *key_valid, int64_t B, int64_t H, int64_t S,
                       int64_t hd, const bf16_t *o, const bf16_t *dout, int64_t ld_o, const float *lse,
                       const bf16_t *o_lo, bf16_t *dqkv, float *delta, const float *rope_cos, const float *rope_sin,
                       float rope_scale, unsigned drop_thresh, uint64_t seed, hipStream_t s);
