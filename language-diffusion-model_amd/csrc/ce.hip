// K4: masked cross-entropy with a streamed log-sum-exp (HBM-bound: one read of the
// participating logit rows forward, one read + one write backward).
// Follows masked_cross_entropy_loss / variational_lower_bound_loss (src/loss.py:8-65).
#include "common.h"

namespace {

struct CeArgs {
  int64_t ld, T, V, seq;
  const int64_t *targets;
  const int64_t *masked_ids;
  int64_t mask_token_id;
  const uint8_t *mask_bool;
  const uint8_t *att_valid;
  const float *seq_weight;
  int norm_mode;
};

__device__ __forceinline__ bool row_active(const CeArgs &a, int64_t r) {
  bool m = a.mask_bool ? (a.mask_bool[r] != 0) : (a.masked_ids[r] == a.mask_token_id);
  if (a.att_valid) m = m && (a.att_valid[r] != 0);
  return m;
}

// (max, sum of exp relative to max) pair and its merge
struct MS { float m, s; };
__device__ __forceinline__ MS ms_merge(MS a, MS b) {
  const float m = fmaxf(a.m, b.m);
  if (m == -INFINITY) return MS{m, 0.f};
  return MS{m, a.s * __expf(a.m - m) + b.s * __expf(b.m - m)};
}
__device__ __forceinline__ void ms_push(MS &a, float x) {
  if (x > a.m) {
    a.s = a.s * __expf(a.m - x) + 1.0f;
    a.m = x;
  } else {
    a.s += __expf(x - a.m);
  }
}

// One 256-thread block per row; inactive rows leave after one flag read.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void ce_fwd_rows_k(const T *logits, CeArgs a, float *row_lse,
                                                     float *row_loss) {
  const int64_t r = blockIdx.x;
  if (!row_active(a, r)) {
    if (threadIdx.x == 0) row_loss[r] = 0.f;
    return;
  }
  const T *row = logits + r * a.ld;
  MS acc{-INFINITY, 0.f};
  if (VEC) {
    const int64_t nv = a.V >> 2;
    for (int64_t i = threadIdx.x; i < nv; i += 256) {
      f4 v = ld4(row + i * 4);
      const float mx = fmaxf(fmaxf(v.v[0], v.v[1]), fmaxf(v.v[2], v.v[3]));
      if (mx > acc.m) {
        acc.s *= __expf(acc.m - mx);
        acc.m = mx;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc.s += __expf(v.v[j] - acc.m);
    }
    for (int64_t i = (nv << 2) + threadIdx.x; i < a.V; i += 256) ms_push(acc, Elt<T>::ld(row + i));
  } else {
    for (int64_t i = threadIdx.x; i < a.V; i += 256) ms_push(acc, Elt<T>::ld(row + i));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    MS other{__shfl_xor(acc.m, o, 64), __shfl_xor(acc.s, o, 64)};
    acc = ms_merge(acc, other);
  }
  __shared__ float sm_m[4], sm_s[4];
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sm_m[wid] = acc.m;
    sm_s[wid] = acc.s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    MS t{sm_m[0], sm_s[0]};
    for (int w = 1; w < 4; ++w) t = ms_merge(t, MS{sm_m[w], sm_s[w]});
    const float lse = t.m + logf(t.s);
    const int64_t tgt = a.targets[r];
    const float xt = (tgt >= 0 && tgt < a.V) ? Elt<T>::ld(row + tgt) : 0.f;
    const float w = a.seq_weight ? a.seq_weight[r / a.seq] : 1.0f;
    row_lse[r] = lse;
    row_loss[r] = w * (lse - xt);
  }
}

// Single block: deterministic sum of the per-row losses and the active-row count.
__global__ __launch_bounds__(1024) void ce_finish_k(const float *row_loss, CeArgs a, float *loss_out,
                                                    float *count_out) {
  __shared__ double sm_l[16];
  __shared__ float sm_c[16];
  double l = 0.0;
  float c = 0.f;
  for (int64_t r0 = threadIdx.x; r0 < a.T; r0 += 8 * 1024) {
    bool f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = r0 + j * 1024 < a.T && row_active(a, r0 + j * 1024);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = f[j] ? row_loss[r0 + j * 1024] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      l += (double)v[j];
      c += f[j] ? 1.0f : 0.f;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    l += __shfl_xor(l, o, 64);
    c += __shfl_xor(c, o, 64);
  }
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sm_l[wid] = l;
    sm_c[wid] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tl = 0.0;
    float tc = 0.f;
    for (int w = 0; w < 16; ++w) {
      tl += sm_l[w];
      tc += sm_c[w];
    }
    const float norm = a.norm_mode == 0 ? tc : (float)a.T;
    loss_out[0] = (tc > 0.f && norm > 0.f) ? (float)(tl / (double)norm) : 0.f;
    count_out[0] = tc;
  }
}

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void ce_bwd_rows_k(const T *logits, CeArgs a, const float *row_lse,
                                                     const float *count, const float *gout, T *dl,
                                                     int64_t ld_d) {
  const int64_t r = blockIdx.x;
  T *drow = dl + r * ld_d;
  const bool act = row_active(a, r);
  if (!act) {
    if (VEC) {
      const f4 z{{0.f, 0.f, 0.f, 0.f}};
      for (int64_t i = threadIdx.x; i < (ld_d >> 2); i += 256) st4(drow + i * 4, z);
    } else {
      for (int64_t i = threadIdx.x; i < ld_d; i += 256) Elt<T>::st(drow + i, 0.f);
    }
    return;
  }
  const T *row = logits + r * a.ld;
  const float cnt = count[0];
  const float norm = a.norm_mode == 0 ? cnt : (float)a.T;
  const float w = a.seq_weight ? a.seq_weight[r / a.seq] : 1.0f;
  const float coef = gout[0] * w / norm;
  const float lse = row_lse[r];
  const int64_t tgt = a.targets[r];
  if (VEC) {
    const int64_t nv = a.V >> 2;
    for (int64_t i = threadIdx.x; i < (ld_d >> 2); i += 256) {
      f4 o{{0.f, 0.f, 0.f, 0.f}};
      if (i < nv) {
        f4 v = ld4(row + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p = __expf(v.v[j] - lse);
          o.v[j] = (p - ((i * 4 + j) == tgt ? 1.0f : 0.f)) * coef;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t c = i * 4 + j;
          if (c < a.V) {
            const float p = __expf(Elt<T>::ld(row + c) - lse);
            o.v[j] = (p - (c == tgt ? 1.0f : 0.f)) * coef;
          }
        }
      }
      st4(drow + i * 4, o);
    }
  } else {
    for (int64_t c = threadIdx.x; c < ld_d; c += 256) {
      float o = 0.f;
      if (c < a.V) {
        const float p = __expf(Elt<T>::ld(row + c) - lse);
        o = (p - (c == tgt ? 1.0f : 0.f)) * coef;
      }
      Elt<T>::st(drow + c, o);
    }
  }
}

// Compaction: rows[0..count) = indices of the participating rows in ascending order.
// One 1024-thread block walks the rows 1024 at a time (coalesced flag reads): ballot + popcount
// inside a wave, the 16 wave counts through LDS.
__global__ __launch_bounds__(1024) void ce_compact_k(CeArgs a, int *rows, int *count) {
  __shared__ int wsum[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int base = 0;   // rows found before this chunk (the same value in every thread)
  constexpr int G = 8;   // chunks whose flags are fetched together (independent loads in flight)
  for (int64_t g0 = 0; g0 < a.T; g0 += (int64_t)G * 1024) {
    bool acts[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int64_t r = g0 + j * 1024 + threadIdx.x;
      acts[j] = r < a.T && row_active(a, r);
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      if (g0 + j * 1024 >= a.T) break;                    // block-uniform
      const int64_t r = g0 + j * 1024 + threadIdx.x;
      const unsigned long long m = __ballot(acts[j]);
      if (lane == 0) wsum[wid] = __popcll(m);
      __syncthreads();
      int before = 0, total = 0;
#pragma unroll
      for (int w = 0; w < 16; ++w) {
        const int c = wsum[w];
        before += w < wid ? c : 0;
        total += c;
      }
      if (acts[j]) rows[base + before + __popcll(m & ((1ull << lane) - 1ull))] = (int)r;
      base += total;
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) count[0] = base;
}

// Multi-block compaction: block b owns `sub` consecutive 1024-row pieces and counts the
// participating rows BEFORE its first row itself (independent flag reads, eight in flight), so no
// block waits for another; the last block also writes the total.  Same output as ce_compact_k.
__global__ __launch_bounds__(1024) void ce_compact_mb_k(CeArgs a, int *rows, int *count, int sub) {
  __shared__ int wsum[16];
  __shared__ int wpre[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t start = (int64_t)blockIdx.x * sub * 1024;
  int mine = 0;
  int64_t r = threadIdx.x;
  for (; r + 7 * 1024 < start; r += 8 * 1024) {
    bool f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = row_active(a, r + j * 1024);
#pragma unroll
    for (int j = 0; j < 8; ++j) mine += f[j] ? 1 : 0;
  }
  for (; r < start; r += 1024) mine += row_active(a, r) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if (lane == 0) wpre[wid] = mine;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) base += wpre[w];
  for (int s = 0; s < sub; ++s) {
    const int64_t rr = start + (int64_t)s * 1024 + threadIdx.x;
    if (start + (int64_t)s * 1024 >= a.T) break;          // block-uniform
    const bool act = rr < a.T && row_active(a, rr);
    const unsigned long long m = __ballot(act);
    if (lane == 0) wsum[wid] = __popcll(m);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int c = wsum[w];
      before += w < wid ? c : 0;
      total += c;
    }
    if (act) rows[base + before + __popcll(m & ((1ull << lane) - 1ull))] = (int)rr;
    base += total;
    __syncthreads();
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) count[0] = base;
}

// Backward into the COMPACT layout: block i handles compact row i (source row rows[i]);
// rows count .. roundup(count, 64) are zero-filled so that a 32/64-deep reduction over the
// compact rows can run past the count; later rows are not touched.
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_compact_k(const T *logits, CeArgs a, const float *row_lse,
                                                        const float *countf, const float *gout, const int *rows,
                                                        const int *count, T *dl, int64_t ld_d) {
  const int cnt = count[0];
  const int64_t i = blockIdx.x;
  if (i >= (((int64_t)cnt + 63) / 64) * 64) return;
  T *drow = dl + i * ld_d;
  const int64_t nv_d = ld_d >> 2;
  if (i >= cnt) {
    const f4 z{{0.f, 0.f, 0.f, 0.f}};
    for (int64_t c = threadIdx.x; c < nv_d; c += 256) st4(drow + c * 4, z);
    return;
  }
  const int64_t r = rows[i];
  const T *row = logits + r * a.ld;
  const float norm = a.norm_mode == 0 ? countf[0] : (float)a.T;
  const float w = a.seq_weight ? a.seq_weight[r / a.seq] : 1.0f;
  const float coef = gout[0] * w / norm;
  const float lse = row_lse[r];
  const int64_t tgt = a.targets[r];
  const int64_t nv = a.V >> 2;
  for (int64_t c = threadIdx.x; c < nv_d; c += 256) {
    f4 o{{0.f, 0.f, 0.f, 0.f}};
    if (c < nv) {
      const f4 v = ld4(row + c * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) o.v[j] = (__expf(v.v[j] - lse) - ((c * 4 + j) == tgt ? 1.0f : 0.f)) * coef;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t cc = c * 4 + j;
        if (cc < a.V) o.v[j] = (__expf(Elt<T>::ld(row + cc) - lse) - (cc == tgt ? 1.0f : 0.f)) * coef;
      }
    }
    st4(drow + c * 4, o);
  }
}

// ---------------------------------------------------------------------------------------------
// 16-byte-vector forms.  A workgroup WALKS rows (blockIdx.x, blockIdx.x + gridDim.x, ...) instead
// of one workgroup per row: rows that do not take part cost one flag read shared by 64 candidate
// rows (one ballot), not a workgroup launch each.  Every thread keeps UNR row vectors in flight.
constexpr int CE_UNR = 4;

template <typename T>
__device__ __forceinline__ void ce_row_lse8(const T *row, int64_t V, MS &acc) {
  const int nv = (int)(V >> 3);
  for (int i0 = threadIdx.x; i0 < nv; i0 += 256 * CE_UNR) {
    Raw8<T> v[CE_UNR];
#pragma unroll
    for (int u = 0; u < CE_UNR; ++u)
      if (i0 + u * 256 < nv) v[u].load(row + (int64_t)(i0 + u * 256) * 8);
#pragma unroll
    for (int u = 0; u < CE_UNR; ++u)
      if (i0 + u * 256 < nv) {
        float mx = v[u].get(0);
#pragma unroll
        for (int j = 1; j < 8; ++j) mx = fmaxf(mx, v[u].get(j));
        if (mx > acc.m) {
          acc.s *= __expf(acc.m - mx);
          acc.m = mx;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.s += __expf(v[u].get(j) - acc.m);
      }
  }
  for (int64_t i = ((int64_t)nv << 3) + threadIdx.x; i < V; i += 256) ms_push(acc, Elt<T>::ld(row + i));
}

// The rows this workgroup looks at are blockIdx.x + k * gridDim.x; their flags are read 64 at a
// time by wave 0 and shared through LDS as a bit mask.
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_walk8_k(const T *logits, CeArgs a, float *row_lse,
                                                      float *row_loss) {
  __shared__ unsigned long long sm_mask;
  __shared__ float sm_m[4], sm_s[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t G = gridDim.x;
  for (int64_t k0 = 0; blockIdx.x + k0 * G < a.T; k0 += 64) {
    if (wid == 0) {
      const int64_t r = blockIdx.x + (k0 + lane) * G;
      const unsigned long long m = __ballot(r < a.T && row_active(a, r));
      if (lane == 0) sm_mask = m;
    }
    __syncthreads();
    unsigned long long mask = sm_mask;
    while (mask) {
      const int bit = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const int64_t r = blockIdx.x + (k0 + bit) * G;
      const T *row = logits + r * a.ld;
      MS acc{-INFINITY, 0.f};
      ce_row_lse8<T>(row, a.V, acc);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        MS other{__shfl_xor(acc.m, o, 64), __shfl_xor(acc.s, o, 64)};
        acc = ms_merge(acc, other);
      }
      if (lane == 0) {
        sm_m[wid] = acc.m;
        sm_s[wid] = acc.s;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        MS t{sm_m[0], sm_s[0]};
        for (int w = 1; w < 4; ++w) t = ms_merge(t, MS{sm_m[w], sm_s[w]});
        const float lse = t.m + logf(t.s);
        const int64_t tgt = a.targets[r];
        const float xt = (tgt >= 0 && tgt < a.V) ? Elt<T>::ld(row + tgt) : 0.f;
        const float w = a.seq_weight ? a.seq_weight[r / a.seq] : 1.0f;
        row_lse[r] = lse;
        row_loss[r] = w * (lse - xt);
      }
      __syncthreads();   // sm_m / sm_s are free again
    }
    __syncthreads();     // everybody has read sm_mask
  }
}

// dlogits of one row, 16-byte vectors: vector i of the output (ld_d elements, zero beyond V) comes
// from vector i of the logits row and is written by the thread that read it (in-place safe).
template <typename T>
__device__ __forceinline__ void ce_row_grad8(const T *row, T *drow, int64_t V, int64_t ld_d, float lse, float coef,
                                             int64_t tgt) {
  const int nv_full = (int)(V >> 3);          // vectors entirely inside the row
  const int nv_d = (int)(ld_d >> 3);
  for (int i0 = threadIdx.x; i0 < nv_d; i0 += 256 * CE_UNR) {
    Raw8<T> v[CE_UNR];
#pragma unroll
    for (int u = 0; u < CE_UNR; ++u)
      if (i0 + u * 256 < nv_full) v[u].load(row + (int64_t)(i0 + u * 256) * 8);
#pragma unroll
    for (int u = 0; u < CE_UNR; ++u) {
      const int i = i0 + u * 256;
      if (i >= nv_d) break;
      float o[8];
      if (i < nv_full) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = __expf(v[u].get(j) - lse) * coef;
        const int64_t dt = tgt - (int64_t)i * 8;
        if (dt >= 0 && dt < 8) {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j == dt) o[j] -= coef;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int64_t c = (int64_t)i * 8 + j;
          o[j] = c < V ? (__expf(Elt<T>::ld(row + c) - lse) - (c == tgt ? 1.0f : 0.f)) * coef : 0.f;
        }
      }
      Raw8<T>::store(drow + (int64_t)i * 8, o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_compact8_k(const T *logits, CeArgs a, const float *row_lse,
                                                         const float *countf, const float *gout, const int *rows,
                                                         const int *count, T *dl, int64_t ld_d) {
  const int cnt = count[0];
  const int64_t lim = (((int64_t)cnt + 63) / 64) * 64;
  const float norm = a.norm_mode == 0 ? countf[0] : (float)a.T;
  const float g0 = gout[0];
  for (int64_t i = blockIdx.x; i < lim && i < a.T + 63; i += gridDim.x) {
    T *drow = dl + i * ld_d;
    if (i >= cnt) {
      const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int64_t c = threadIdx.x; c < (ld_d >> 3); c += 256) Raw8<T>::store(drow + c * 8, z);
      continue;
    }
    const int64_t r = rows[i];
    const float w = a.seq_weight ? a.seq_weight[r / a.seq] : 1.0f;
    ce_row_grad8<T>(logits + r * a.ld, drow, a.V, ld_d, row_lse[r], g0 * w / norm, a.targets[r]);
  }
}

// ---------------------------------------------------------------------------------------------
// Forward AND gradient in one pass over compact rows [0, *count): a row is read once into
// registers (NV 16-byte vectors per thread), its log-sum-exp and loss term come from the
// registers, and (softmax - onehot) / norm is written back over the row -- one read and one
// write of the logits instead of read (forward) + read + write (backward).  Rows
// *count .. roundup(*count, 64) are zeroed (the weight-gradient GEMM reduces over them).
template <int BT>
__device__ __forceinline__ float block_reduce(float v, float *sm, bool is_max) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  v = is_max ? wave_max(v) : wave_sum(v);
  if (lane == 0) sm[wid] = v;
  __syncthreads();
  float r = sm[0];
#pragma unroll
  for (int w = 1; w < BT / 64; ++w) r = is_max ? fmaxf(r, sm[w]) : r + sm[w];
  __syncthreads();
  return r;
}

template <typename T, int BT, int NV>
__global__ __launch_bounds__(BT) void ce_fwd_grad_k(T *logits, int64_t ld, int64_t V, int64_t rows_alloc,
                                                    const int64_t *targets, const int *count, float norm_T,
                                                    int norm_mode, const float *row_weight, float *row_lse,
                                                    float *row_loss) {
  __shared__ float sm[BT / 64];
  __shared__ float sm_xt;
  const int cnt = count[0];
  int64_t lim = (((int64_t)cnt + 63) / 64) * 64;
  if (lim > rows_alloc) lim = rows_alloc;
  const float norm = norm_mode == 0 ? (float)cnt : norm_T;
  const float inv = norm > 0.f ? 1.0f / norm : 0.f;
  const int nv_d = (int)(ld >> 3);                 // vectors per stored row
  const int nv = (int)((V + 7) >> 3);              // vectors holding logits (the last may be partial)
  const int nu = (nv + BT - 1) / BT;               // vector rounds of the block (uniform)
  const int vlast = (int)(V >> 3), jlast = (int)(V & 7);   // the partial vector and its first invalid element
  for (int64_t i = blockIdx.x; i < lim; i += gridDim.x) {
    T *row = logits + i * ld;
    if (i >= cnt) {
      const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int c = threadIdx.x; c < nv_d; c += BT) Raw8<T>::store(row + (int64_t)c * 8, z);
      continue;
    }
    // columns >= V (and vectors past the row) hold -infinity in registers: every pass then runs
    // the same code on every vector (exp(-inf) = 0 is also what their gradient must be)
    Raw8<T> v[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      if (u < nu) {
        const int c = threadIdx.x + u * BT;
        v[u].load_nt(row + (c < nv ? c : 0) * 8);   // read once, rewritten in place below: nothing to keep in cache
        if (c >= vlast) v[u].neg_inf_from(c == vlast ? jlast : 0);
      }
    }
    const int64_t tgt = targets[i];
    const int tvec = (tgt >= 0 && tgt < V) ? (int)(tgt >> 3) : -1, tj = (int)(tgt & 7);
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < NV; ++u)
      if (u < nu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, v[u].get(j));
      }
    mx = block_reduce<BT>(mx, sm, true);
#pragma unroll
    for (int u = 0; u < NV; ++u) v[u].pin();
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u)
      if (u < nu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += __expf(v[u].get(j) - mx);
        if ((int)threadIdx.x + u * BT == tvec) {   // one thread of the block
          float xt = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (tj == j) xt = v[u].get(j);
          sm_xt = xt;
        }
      }
    sum = block_reduce<BT>(sum, sm, false);      // its barriers also publish sm_xt
    const float lse = mx + logf(sum);
    const float wr = row_weight ? row_weight[i] : 1.0f;   // per-row weight (1/t of the row's sequence)
    const float winv = wr * inv;
    if (threadIdx.x == 0) {
      row_lse[i] = lse;
      row_loss[i] = wr * (lse - (tvec >= 0 ? sm_xt : 0.f));
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) v[u].pin();
#pragma unroll
    for (int u = 0; u < NV; ++u)
      if (u < nu) {
        const int c = threadIdx.x + u * BT;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = __expf(v[u].get(j) - lse) * winv;
        if (c == tvec) {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (tj == j) o[j] -= winv;
        }
        if (c < nv) Raw8<T>::store_nt(row + (int64_t)c * 8, o);
      }
    __syncthreads();   // sm_xt is free for the next row
  }
}

// loss = sum_{i < count} row_loss[i] / norm (deterministic order), count_out = count
__global__ __launch_bounds__(1024) void ce_finish_compact_k(const float *row_loss, const int *count, float norm_T,
                                                            int norm_mode, float *loss_out, float *count_out) {
  __shared__ double sm_l[16];
  const int cnt = count[0];
  double l = 0.0;
  for (int r0 = threadIdx.x; r0 < cnt; r0 += 8 * 1024) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = r0 + j * 1024 < cnt ? row_loss[r0 + j * 1024] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) l += (double)v[j];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
  if ((threadIdx.x & 63) == 0) sm_l[threadIdx.x >> 6] = l;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tl = 0.0;
    for (int w = 0; w < 16; ++w) tl += sm_l[w];
    const float norm = norm_mode == 0 ? (float)cnt : norm_T;
    loss_out[0] = (cnt > 0 && norm > 0.f) ? (float)(tl / (double)norm) : 0.f;
    count_out[0] = (float)cnt;
  }
}

// x[i, 0..ncols) *= *scale for i < *count -- nothing is touched when *scale == 1
template <typename T>
__global__ __launch_bounds__(256) void scale_rows_dev_k(T *x, int64_t ld, int64_t ncols, const int *count,
                                                        const float *scale) {
  const float sc = scale[0];
  if (sc == 1.0f) return;
  const int cnt = count[0];
  const int nv = (int)(ncols >> 3);
  for (int64_t i = blockIdx.x; i < cnt; i += gridDim.x) {
    T *row = x + i * ld;
    for (int c = threadIdx.x; c < nv; c += 256) {
      Raw8<T> v;
      v.load(row + (int64_t)c * 8);
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = v.get(j) * sc;
      Raw8<T>::store(row + (int64_t)c * 8, o);
    }
  }
}

}  // namespace

static int ce_args(CeArgs &a, int64_t ld, int64_t T, int64_t V, int64_t seq, const int64_t *targets,
                   const int64_t *masked_ids, int64_t mask_token_id, const uint8_t *mask_bool,
                   const uint8_t *att_valid, const float *seq_weight, int norm_mode) {
  if (T < 0 || V <= 0 || ld < V) return LDM_E_BADARG;
  if (!targets || (!masked_ids && !mask_bool)) return LDM_E_BADARG;
  if (norm_mode != 0 && norm_mode != 1) return LDM_E_BADARG;
  if (seq <= 0) seq = 1;
  a.ld = ld; a.T = T; a.V = V; a.seq = seq; a.targets = targets; a.masked_ids = masked_ids;
  a.mask_token_id = mask_token_id; a.mask_bool = mask_bool; a.att_valid = att_valid;
  a.seq_weight = seq_weight; a.norm_mode = norm_mode;
  return LDM_OK;
}

extern "C" int ldm_ce_fwd(const void *logits, int dtype, int64_t ld, int64_t T, int64_t V, int64_t seq,
                          const int64_t *targets, const int64_t *masked_ids, int64_t mask_token_id,
                          const uint8_t *mask_bool, const uint8_t *att_valid, const float *seq_weight,
                          int norm_mode, float *row_lse, float *row_tmp, float *loss_out,
                          float *count_out, void *stream) {
  CeArgs a;
  int rc = ce_args(a, ld, T, V, seq, targets, masked_ids, mask_token_id, mask_bool, att_valid, seq_weight, norm_mode);
  if (rc) return rc;
  if (!loss_out || !count_out) return LDM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (T > 0) {
    if (!logits || !row_lse || !row_tmp) return LDM_E_BADARG;
    if (T > 0x7fffffff) return LDM_E_SHAPE;
    const size_t esz = dtype == LDM_F32 ? 4 : 2;
    const bool vec = aligned_to(logits, 4 * esz) && ((ld * esz) % (4 * esz) == 0);
    if (aligned_to(logits, 8 * esz) && ld % 8 == 0 && V >= 2048 && V < (1ll << 33) &&
        (dtype == LDM_F32 || dtype == LDM_BF16)) {
      const unsigned grid = (unsigned)(T < 2048 ? T : 2048);
      if (dtype == LDM_F32)
        hipLaunchKernelGGL(ce_fwd_walk8_k<float>, dim3(grid), dim3(256), 0, s, (const float *)logits, a, row_lse, row_tmp);
      else
        hipLaunchKernelGGL(ce_fwd_walk8_k<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t *)logits, a, row_lse, row_tmp);
    } else if (dtype == LDM_F32) {
      if (vec) hipLaunchKernelGGL((ce_fwd_rows_k<float, true>), dim3((unsigned)T), dim3(256), 0, s, (const float *)logits, a, row_lse, row_tmp);
      else hipLaunchKernelGGL((ce_fwd_rows_k<float, false>), dim3((unsigned)T), dim3(256), 0, s, (const float *)logits, a, row_lse, row_tmp);
    } else if (dtype == LDM_BF16) {
      if (vec) hipLaunchKernelGGL((ce_fwd_rows_k<bf16_t, true>), dim3((unsigned)T), dim3(256), 0, s, (const bf16_t *)logits, a, row_lse, row_tmp);
      else hipLaunchKernelGGL((ce_fwd_rows_k<bf16_t, false>), dim3((unsigned)T), dim3(256), 0, s, (const bf16_t *)logits, a, row_lse, row_tmp);
    } else {
      return LDM_E_DTYPE;
    }
    LDM_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(ce_finish_k, dim3(1), dim3(1024), 0, s, row_tmp, a, loss_out, count_out);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

extern "C" int ldm_ce_bwd(const void *logits, int dtype, int64_t ld, int64_t T, int64_t V, int64_t seq,
                          const int64_t *targets, const int64_t *masked_ids, int64_t mask_token_id,
                          const uint8_t *mask_bool, const uint8_t *att_valid, const float *seq_weight,
                          int norm_mode, const float *row_lse, const float *count, const float *gout,
                          void *dlogits, int64_t ld_d, void *stream) {
  CeArgs a;
  int rc = ce_args(a, ld, T, V, seq, targets, masked_ids, mask_token_id, mask_bool, att_valid, seq_weight, norm_mode);
  if (rc) return rc;
  if (T == 0) return LDM_OK;
  if (!logits || !row_lse || !count || !gout || !dlogits || ld_d < V) return LDM_E_BADARG;
  if (T > 0x7fffffff) return LDM_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype == LDM_F32 ? 4 : 2;
  const bool vec = aligned_to(logits, 4 * esz) && ((ld * esz) % (4 * esz) == 0) &&
                   aligned_to(dlogits, 4 * esz) && ((ld_d * esz) % (4 * esz) == 0);
  if (dtype == LDM_F32) {
    if (vec) hipLaunchKernelGGL((ce_bwd_rows_k<float, true>), dim3((unsigned)T), dim3(256), 0, s, (const float *)logits, a, row_lse, count, gout, (float *)dlogits, ld_d);
    else hipLaunchKernelGGL((ce_bwd_rows_k<float, false>), dim3((unsigned)T), dim3(256), 0, s, (const float *)logits, a, row_lse, count, gout, (float *)dlogits, ld_d);
  } else if (dtype == LDM_BF16) {
    if (vec) hipLaunchKernelGGL((ce_bwd_rows_k<bf16_t, true>), dim3((unsigned)T), dim3(256), 0, s, (const bf16_t *)logits, a, row_lse, count, gout, (bf16_t *)dlogits, ld_d);
    else hipLaunchKernelGGL((ce_bwd_rows_k<bf16_t, false>), dim3((unsigned)T), dim3(256), 0, s, (const bf16_t *)logits, a, row_lse, count, gout, (bf16_t *)dlogits, ld_d);
  } else {
    return LDM_E_DTYPE;
  }
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

extern "C" int ldm_ce_compact_rows(int64_t T, int64_t seq, const int64_t *masked_ids, int64_t mask_token_id,
                                   const uint8_t *mask_bool, const uint8_t *att_valid, int32_t *rows,
                                   int32_t *count, void *stream) {
  if (T < 0 || T > 0x7fffffff || !rows || !count || (!masked_ids && !mask_bool)) return LDM_E_BADARG;
  CeArgs a{};
  a.T = T; a.V = 1; a.ld = 1; a.seq = seq > 0 ? seq : 1; a.masked_ids = masked_ids; a.mask_token_id = mask_token_id;
  a.mask_bool = mask_bool; a.att_valid = att_valid;
  if (T > 1024 && T <= (1 << 24)) {
    // at most 64 blocks: the rows before a block are re-counted by it, which is quadratic in the block count
    const int sub = (int)((T + 64 * 1024 - 1) / (64 * 1024));
    const unsigned nb = (unsigned)((T + (int64_t)sub * 1024 - 1) / ((int64_t)sub * 1024));
    hipLaunchKernelGGL(ce_compact_mb_k, dim3(nb), dim3(1024), 0, (hipStream_t)stream, a, rows, count, sub);
  } else
    hipLaunchKernelGGL(ce_compact_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, rows, count);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

extern "C" int ldm_ce_bwd_compact(const void *logits, int dtype, int64_t ld, int64_t T, int64_t V, int64_t seq,
                                  const int64_t *targets, const float *seq_weight, int norm_mode,
                                  const float *row_lse, const float *count, const float *gout,
                                  const int32_t *rows, const int32_t *count_i, void *dlogits, int64_t ld_d,
                                  void *stream) {
  if (T <= 0) return T == 0 ? LDM_OK : LDM_E_BADARG;
  if (!logits || !targets || !row_lse || !count || !gout || !rows || !count_i || !dlogits || ld_d < V)
    return LDM_E_BADARG;
  if (T > 0x7fffffff) return LDM_E_SHAPE;
  const size_t esz = dtype == LDM_F32 ? 4 : 2;
  if (!aligned_to(logits, 4 * esz) || (ld % 4) || !aligned_to(dlogits, 4 * esz) || (ld_d % 4)) return LDM_E_ALIGN;
  CeArgs a{};
  a.ld = ld; a.T = T; a.V = V; a.seq = seq > 0 ? seq : 1; a.targets = targets; a.seq_weight = seq_weight;
  a.norm_mode = norm_mode;
  hipStream_t s = (hipStream_t)stream;
  if (aligned_to(logits, 8 * esz) && ld % 8 == 0 && aligned_to(dlogits, 8 * esz) && ld_d % 8 == 0 && V >= 2048 &&
      V < (1ll << 33) && (dtype == LDM_F32 || dtype == LDM_BF16)) {
    const int64_t tp = ((T + 63) / 64) * 64;
    const unsigned grid = (unsigned)(tp < 2048 ? tp : 2048);
    if (dtype == LDM_F32)
      hipLaunchKernelGGL(ce_bwd_compact8_k<float>, dim3(grid), dim3(256), 0, s, (const float *)logits, a, row_lse, count, gout, rows, count_i, (float *)dlogits, ld_d);
    else
      hipLaunchKernelGGL(ce_bwd_compact8_k<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t *)logits, a, row_lse, count, gout, rows, count_i, (bf16_t *)dlogits, ld_d);
    LDM_CHECK_LAUNCH();
    return LDM_OK;
  }
  if (dtype == LDM_F32)
    hipLaunchKernelGGL(ce_bwd_compact_k<float>, dim3((unsigned)(((T + 63) / 64) * 64)), dim3(256), 0, s, (const float *)logits, a, row_lse, count, gout, rows, count_i, (float *)dlogits, ld_d);
  else if (dtype == LDM_BF16)
    hipLaunchKernelGGL(ce_bwd_compact_k<bf16_t>, dim3((unsigned)(((T + 63) / 64) * 64)), dim3(256), 0, s, (const bf16_t *)logits, a, row_lse, count, gout, rows, count_i, (bf16_t *)dlogits, ld_d);
  else
    return LDM_E_DTYPE;
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

template <typename T>
static int ce_fwd_grad_launch(void *logits, int64_t ld, int64_t rows_alloc, int64_t V, const int64_t *targets,
                              const int32_t *count, float norm_T, int norm_mode, const float *row_weight,
                              float *row_lse, float *row_tmp, hipStream_t s) {
  const int64_t nv = (V + 7) / 8;
  if (nv <= 256 * 16) {
    const unsigned grid = (unsigned)(rows_alloc < 2048 ? rows_alloc : 2048);
    hipLaunchKernelGGL((ce_fwd_grad_k<T, 256, 16>), dim3(grid), dim3(256), 0, s, (T *)logits, ld, V, rows_alloc, targets,
                       count, norm_T, norm_mode, row_weight, row_lse, row_tmp);
  } else if (nv <= 1024 * 16) {
    const unsigned grid = (unsigned)(rows_alloc < 512 ? rows_alloc : 512);
    hipLaunchKernelGGL((ce_fwd_grad_k<T, 1024, 16>), dim3(grid), dim3(1024), 0, s, (T *)logits, ld, V, rows_alloc,
                       targets, count, norm_T, norm_mode, row_weight, row_lse, row_tmp);
  } else {
    return LDM_E_SHAPE;
  }
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

extern "C" int ldm_ce_fwd_grad(void *logits, int dtype, int64_t ld, int64_t rows_alloc, int64_t V,
                               const int64_t *targets, const int32_t *count, int norm_mode, float norm_T,
                               const float *row_weight, float *row_lse, float *row_tmp, float *loss_out,
                               float *count_out, void *stream) {
  if (rows_alloc < 0 || V <= 0 || ld < V) return LDM_E_BADARG;
  if (!targets || !count || !loss_out || !count_out) return LDM_E_BADARG;
  if (norm_mode != 0 && norm_mode != 1) return LDM_E_BADARG;
  if (rows_alloc > 0x7fffffff) return LDM_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (rows_alloc > 0) {
    if (!logits || !row_lse || !row_tmp) return LDM_E_BADARG;
    const size_t esz = dtype == LDM_F32 ? 4 : 2;
    if (!aligned_to(logits, 8 * esz) || ld % 8) return LDM_E_ALIGN;
    int rc;
    if (dtype == LDM_F32) rc = ce_fwd_grad_launch<float>(logits, ld, rows_alloc, V, targets, count, norm_T, norm_mode, row_weight, row_lse, row_tmp, s);
    else if (dtype == LDM_BF16) rc = ce_fwd_grad_launch<bf16_t>(logits, ld, rows_alloc, V, targets, count, norm_T, norm_mode, row_weight, row_lse, row_tmp, s);
    else return LDM_E_DTYPE;
    if (rc) return rc;
  }
  hipLaunchKernelGGL(ce_finish_compact_k, dim3(1), dim3(1024), 0, s, row_tmp, count, norm_T, norm_mode, loss_out, count_out);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

extern "C" int ldm_scale_rows_dev(void *x, int dtype, int64_t ld, int64_t rows_alloc, int64_t ncols,
                                  const int32_t *count, const float *scale, void *stream) {
  if (rows_alloc < 0 || ncols < 0 || ld < ncols) return LDM_E_BADARG;
  if (rows_alloc == 0 || ncols == 0) return LDM_OK;
  if (!x || !count || !scale) return LDM_E_BADARG;
  const size_t esz = dtype == LDM_F32 ? 4 : 2;
  if (!aligned_to(x, 8 * esz) || ld % 8 || ncols % 8) return LDM_E_ALIGN;
  const unsigned grid = (unsigned)(rows_alloc < 2048 ? rows_alloc : 2048);
  if (dtype == LDM_F32)
    hipLaunchKernelGGL(scale_rows_dev_k<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float *)x, ld, ncols, count, scale);
  else if (dtype == LDM_BF16)
    hipLaunchKernelGGL(scale_rows_dev_k<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (bf16_t *)x, ld, ncols, count, scale);
  else
    return LDM_E_DTYPE;
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}
