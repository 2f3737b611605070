// HBM-bound row kernels: embedding gather / scatter-add, RMSNorm fwd/bwd, RoPE,
// SwiGLU, column sums, casts.  All move 8-16 B per lane per access.
#include "common.h"
#include <mutex>

namespace {

// ------------------------------------------------------------------ embedding
// One wave per token row; lanes sweep the row in 4-element vectors.
template <typename TO>
__global__ __launch_bounds__(256) void embed_fwd_k(const int64_t *ids, const float *table,
                                                   int64_t T, int64_t d, int64_t vocab, TO *out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t t = wave; t < T; t += nwaves) {
    int64_t id = ids[t];
    const bool ok = (id >= 0) & (id < vocab);
    const float *src = table + (ok ? id : 0) * d;
    TO *dst = out + t * d;
    for (int64_t c = lane * 4; c < d; c += 256) {
      f4 v = ld4(src + c);
      if (!ok) v = f4{{0.f, 0.f, 0.f, 0.f}};
      st4(dst + c, v);
    }
  }
}

// Scatter-add of dout rows into dtable.  Each block owns ROWS consecutive tokens.
// Rows whose id == hot_id (the mask token: about half of all rows in a training
// step) are first summed in registers across the block's tokens and added with
// one atomic per column per block; other rows go straight to float atomics
// (256 contiguous bytes per wave-instruction).
template <typename TI, int ROWS>
__global__ __launch_bounds__(256) void embed_bwd_k(const int64_t *ids, const TI *dout, int64_t T,
                                                   int64_t d, int64_t vocab, int64_t hot_id,
                                                   float *dtable) {
  const int64_t t0 = (int64_t)blockIdx.x * ROWS;
  const int64_t t1 = (t0 + ROWS < T) ? t0 + ROWS : T;
  for (int64_t c = threadIdx.x; c < d; c += blockDim.x) {
    float hot = 0.f;
    bool any_hot = false;
    for (int64_t t = t0; t < t1; ++t) {
      const int64_t id = ids[t];          // wave-uniform: served from one cache line
      const float g = Elt<TI>::ld(dout + t * d + c);
      if (id == hot_id) {
        hot += g;
        any_hot = true;
      } else if (id >= 0 && id < vocab) {
        atomicAdd(dtable + id * d + c, g);
      }
    }
    if (any_hot && hot_id >= 0 && hot_id < vocab) atomicAdd(dtable + hot_id * d + c, hot);
  }
}

// ------------------------------------------------------------------ RMSNorm
// One wave per row.  NV = 4-element vectors per lane held in registers.
template <typename T, int NV>
__global__ __launch_bounds__(256) void rmsnorm_fwd_k(const T *x, const float *w, int64_t rows,
                                                     int d, float eps, T *y, float *rstd) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int nvec = d >> 2;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const T *xr = x + r * d;
    f4 v[NV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        v[i] = ld4(xr + vi * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) ss = fmaf(v[i].v[j], v[i].v[j], ss);
      }
    }
    ss = wave_sum(ss);
    const float rs = 1.0f / sqrtf(ss / (float)d + eps);
    if (lane == 0) rstd[r] = rs;
    T *yr = y + r * d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        f4 g = ld4(w + vi * 4);
        f4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o.v[j] = g.v[j] * (v[i].v[j] * rs);
        st4(yr + vi * 4, o);
      }
    }
  }
}

// Backward: each wave walks a strided set of rows, keeps its dw partial sums in
// registers, then the block's 4 waves are reduced through LDS and the block writes
// one row of dw_partial[gridDim.x][d].
template <typename T, int NV>
__global__ __launch_bounds__(256) void rmsnorm_bwd_k(const T *x, const float *w, const float *rstd,
                                                     const T *dy, const T *dres, int64_t rows, int d,
                                                     T *dx, float *dw_partial) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [4][d]
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int nvec = d >> 2;
  f4 wv[NV], acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = lane + i * 64;
    acc[i] = f4{{0.f, 0.f, 0.f, 0.f}};
    wv[i] = (vi < nvec) ? ld4(w + vi * 4) : f4{{0.f, 0.f, 0.f, 0.f}};
  }
  const float inv_d = 1.0f / (float)d;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const T *xr = x + r * d;
    const T *gr = dy + r * d;
    const float rs = rstd[r];
    f4 xv[NV], gv[NV];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        xv[i] = ld4(xr + vi * 4);
        gv[i] = ld4(gr + vi * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i].v[j] = fmaf(gv[i].v[j], xv[i].v[j] * rs, acc[i].v[j]);
          gv[i].v[j] *= wv[i].v[j];
          dot = fmaf(gv[i].v[j], xv[i].v[j], dot);
        }
      }
    }
    dot = wave_sum(dot);
    const float k = dot * rs * rs * rs * inv_d;
    T *dr = dx + r * d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        f4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o.v[j] = fmaf(rs, gv[i].v[j], -xv[i].v[j] * k);
        if (dres) {
          f4 rr = ld4(dres + r * d + vi * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) o.v[j] += rr.v[j];
        }
        st4(dr + vi * 4, o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = lane + i * 64;
    if (vi < nvec) st4(smem + wid * d + vi * 4, acc[i]);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += blockDim.x)
    dw_partial[(int64_t)blockIdx.x * d + c] = smem[c] + smem[d + c] + smem[2 * d + c] + smem[3 * d + c];
}

// 8-element-vector forms (d % 8 == 0): 16-byte accesses for bf16, every load of a row issued
// before the first use, rows kept in storage form in registers.  NV = vectors per lane.
// TO: the type of y (T, or bf16 under an fp32 x: the norm between an fp32 residual stream and a bf16 projection)
template <typename T, int NV, typename TO = T>
__global__ __launch_bounds__(256) void rmsnorm_fwd8_k(const T *x, const float *w, int64_t rows, int d, float eps,
                                                      TO *y, float *rstd) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int nvec = d >> 3;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const T *xr = x + r * d;
    Raw8<T> v[NV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + i * 64 < nvec) v[i].load(xr + (lane + i * 64) * 8);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + i * 64 < nvec) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ss = fmaf(v[i].get(j), v[i].get(j), ss);
      }
    ss = wave_sum(ss);
    const float rs = 1.0f / sqrtf(ss / (float)d + eps);
    if (lane == 0) rstd[r] = rs;
    TO *yr = y + r * d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        const f4 g0 = ld4(w + vi * 8), g1 = ld4(w + vi * 8 + 4);
        float o[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = g0.v[j] * (v[i].get(j) * rs);
          o[j + 4] = g1.v[j] * (v[i].get(j + 4) * rs);
        }
        Raw8<TO>::store(yr + vi * 8, o);
      }
    }
  }
}

// dx_drop (nullable): a second output, keep * dx / (1 - p) with the residual-dropout mask of the projection
// whose output this norm read (row = token, column = feature; the mask of ldm_dropout_add): the gradient that
// projection's backward needs, written here instead of by a masking pass of its own over dx.
struct RmsDrop { void *dx_drop; unsigned thresh, seed_lo, seed_hi; float inv; };

// TG: the type of dy (T, or bf16 under fp32 x / dres / dx: the fp32 residual stream's norm, see rmsnorm_fwd8_k)
template <typename T, int NV, bool RES, bool DROP, typename TG = T>
__global__ __launch_bounds__(256) void rmsnorm_bwd8_k(const T *x, const float *w, const float *rstd, const TG *dy,
                                                      const T *dres, int64_t rows, int d, T *dx,
                                                      float *dw_partial, RmsDrop dr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [4][d]
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int nvec = d >> 3;
  float wv[NV][8], acc[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = lane + i * 64;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[i][j] = 0.f;
      wv[i][j] = (vi < nvec) ? w[vi * 8 + j] : 0.f;
    }
  }
  const float inv_d = 1.0f / (float)d;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const float rs = rstd[r];
    Raw8<T> xv[NV], rv[NV];
    Raw8<TG> gv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        // streaming accesses (-8 % alone, -4.5 % in the step): none of the four arrays is read again soon
        xv[i].load_nt(x + r * d + vi * 8);
        gv[i].load_nt(dy + r * d + vi * 8);
        if (RES) rv[i].load_nt(dres + r * d + vi * 8);
      }
    }
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + i * 64 < nvec) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xx = xv[i].get(j), gg = gv[i].get(j);
          acc[i][j] = fmaf(gg, xx * rs, acc[i][j]);
          dot = fmaf(gg * wv[i][j], xx, dot);
        }
      }
    dot = wave_sum(dot);
    const float k = dot * rs * rs * rs * inv_d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int vi = lane + i * 64;
      if (vi < nvec) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          o[j] = fmaf(rs, gv[i].get(j) * wv[i][j], -xv[i].get(j) * k);
          if (RES) o[j] += rv[i].get(j);
        }
        Raw8<T>::store_nt(dx + r * d + vi * 8, o);
        if constexpr (DROP) {
          const unsigned rk = ldm_drop_row(ldm_drop_bh(dr.seed_lo, 0xffffffffu), dr.seed_hi, (unsigned)r);
          float o2[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            // the masked copy is formed from the STORED value of dx (rounded to T), as a pass over dx would
            float v = o[j];
            if constexpr (sizeof(T) == 2) v = bf16_to_f32(f32_to_bf16(v));
            o2[j] = ldm_drop_keep(rk, (unsigned)(vi * 8 + j), dr.thresh) ? v * dr.inv : 0.f;
          }
          Raw8<T>::store_nt((T *)dr.dx_drop + r * d + vi * 8, o2);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int vi = lane + i * 64;
    if (vi < nvec) {
#pragma unroll
      for (int j = 0; j < 8; ++j) smem[wid * d + vi * 8 + j] = acc[i][j];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += blockDim.x)
    dw_partial[(int64_t)blockIdx.x * d + c] = smem[c] + smem[d + c] + smem[2 * d + c] + smem[3 * d + c];
}

// out[c] = sum_p partial[p][c]   (deterministic second stage)
// Block = 32 columns x 32 partial lanes (1024 threads); each lane strides over the partials with
// eight loads in flight, LDS joins the 32 lanes in a fixed order.
__global__ __launch_bounds__(1024) void reduce_partials_k(const float *partial, int64_t nparts, int64_t d,
                                                          float *out) {
  __shared__ float sm[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t c = (int64_t)blockIdx.x * 32 + tx;
  float s = 0.f;
  if (c < d) {
    int64_t p = ty;
    for (; p + 7 * 32 < nparts; p += 8 * 32) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = partial[(p + j * 32) * d + c];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; p < nparts; p += 32) s += partial[p * d + c];
  }
  sm[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < d) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += sm[i][tx];
    out[c] = t;
  }
}

// ------------------------------------------------------------------ RoPE on fused qkv
// One thread per (row, head, pair-vector): rotates element i with i + hd/2.
template <typename T>
__global__ __launch_bounds__(256) void rope_qk_k(T *qkv, int64_t rows, int64_t seq, int nhead, int hd,
                                                 int64_t ld, const float *cos_t, const float *sin_t,
                                                 float q_scale, int inverse) {
  const int half = hd >> 1;
  const int vec_per_head = half >> 2;            // 4 pairs per thread
  const int64_t per_row = (int64_t)2 * nhead * vec_per_head;   // q heads then k heads
  const int64_t total = rows * per_row;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / per_row;
    int rem = (int)(idx - row * per_row);
    const int which = rem / (nhead * vec_per_head);   // 0 = q, 1 = k
    rem -= which * nhead * vec_per_head;
    const int h = rem / vec_per_head;
    const int i0 = (rem - h * vec_per_head) * 4;
    const int64_t pos = row % seq;
    T *p = qkv + row * ld + (int64_t)which * nhead * hd + (int64_t)h * hd + i0;
    f4 a = ld4(p), b = ld4(p + half);
    f4 c = ld4(cos_t + pos * half + i0), s = ld4(sin_t + pos * half + i0);
    const float sc = which == 0 ? q_scale : 1.0f;
    f4 oa, ob;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!inverse) {
        oa.v[j] = (a.v[j] * c.v[j] - b.v[j] * s.v[j]) * sc;
        ob.v[j] = (b.v[j] * c.v[j] + a.v[j] * s.v[j]) * sc;
      } else {
        oa.v[j] = (a.v[j] * c.v[j] + b.v[j] * s.v[j]) * sc;
        ob.v[j] = (b.v[j] * c.v[j] - a.v[j] * s.v[j]) * sc;
      }
    }
    st4(p, oa);
    st4(p + half, ob);
  }
}

// 8-pair form (hd/2 a multiple of 8): a thread rotates 8 consecutive elements with their partners
// hd/2 further on, 16-byte accesses for bf16.  A workgroup walks groups of RG consecutive rows;
// inside a group the (row, q|k, head, vector) items are spread flat over the threads.  The two
// divisions that take an item index apart are multiplications by ceil(2^32 / divisor) (m_row,
// m_vph; exact while index * divisor < 2^32, which the launcher checks; m_vph == 0 stands for a
// divisor of 1), and the position in the sequence is carried from the group's first row: integer
// division by a run-time value costs more VALU time than the rotation itself.
template <typename T, int RG>
__global__ __launch_bounds__(256) void rope_qk8_k(T *__restrict__ qkv, int rows, int seq, int nhead, int hd,
                                                  int64_t ld, const float *__restrict__ cos_t,
                                                  const float *__restrict__ sin_t, float q_scale, int inverse,
                                                  unsigned m_row, unsigned m_vph) {
  const int half = hd >> 1;
  const unsigned vph = (unsigned)half >> 3;               // vectors per head half
  const unsigned per_head_set = (unsigned)nhead * vph;    // items of the q (or k) heads of one row
  const unsigned per_row = 2u * per_head_set;
  const int ngroups = (rows + RG - 1) / RG;
  for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int row0 = g * RG;
    const int nr = rows - row0 < RG ? rows - row0 : RG;
    const unsigned items = (unsigned)nr * per_row;
    const int pos0 = row0 % seq;
    for (unsigned it = threadIdx.x; it < items; it += 256) {
      const unsigned lr = __umulhi(it, m_row);
      unsigned rem = it - lr * per_row;
      const unsigned which = rem >= per_head_set ? 1u : 0u;
      rem -= which * per_head_set;
      const unsigned h = m_vph ? __umulhi(rem, m_vph) : rem;
      const unsigned i0 = (rem - h * vph) * 8u;
      const int row = row0 + (int)lr;
      int pos = pos0 + (int)lr;
      while (pos >= seq) pos -= seq;
      T *p = qkv + (int64_t)row * ld + (int64_t)which * nhead * hd + (int64_t)h * hd + i0;
      Raw8<T> a, b;
      a.load(p);
      b.load(p + half);
      const float *cp = cos_t + (int64_t)pos * half + i0, *sp = sin_t + (int64_t)pos * half + i0;
      const f4 c0 = ld4(cp), c1 = ld4(cp + 4), s0 = ld4(sp), s1 = ld4(sp + 4);
      const float sc = which == 0 ? q_scale : 1.0f;
      float oa[8], ob[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float c = j < 4 ? c0.v[j] : c1.v[j - 4], sn = j < 4 ? s0.v[j] : s1.v[j - 4];
        const float av = a.get(j), bv = b.get(j);
        if (!inverse) {
          oa[j] = (av * c - bv * sn) * sc;
          ob[j] = (bv * c + av * sn) * sc;
        } else {
          oa[j] = (av * c + bv * sn) * sc;
          ob[j] = (bv * c - av * sn) * sc;
        }
      }
      Raw8<T>::store(p, oa);
      Raw8<T>::store(p + half, ob);
    }
  }
}

// scalar form for head dims whose half is not a multiple of 4
template <typename T>
__global__ __launch_bounds__(256) void rope_qk_scalar_k(T *qkv, int64_t rows, int64_t seq, int nhead,
                                                        int hd, int64_t ld, const float *cos_t,
                                                        const float *sin_t, float q_scale, int inverse) {
  const int half = hd >> 1;
  const int64_t per_row = (int64_t)2 * nhead * half;
  const int64_t total = rows * per_row;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / per_row;
    int rem = (int)(idx - row * per_row);
    const int which = rem / (nhead * half);
    rem -= which * nhead * half;
    const int h = rem / half;
    const int i = rem - h * half;
    const int64_t pos = row % seq;
    T *p = qkv + row * ld + (int64_t)which * nhead * hd + (int64_t)h * hd + i;
    const float a = Elt<T>::ld(p), b = Elt<T>::ld(p + half);
    const float c = cos_t[pos * half + i], s = sin_t[pos * half + i];
    const float sc = which == 0 ? q_scale : 1.0f;
    float oa, ob;
    if (!inverse) {
      oa = (a * c - b * s) * sc;
      ob = (b * c + a * s) * sc;
    } else {
      oa = (a * c + b * s) * sc;
      ob = (b * c - a * s) * sc;
    }
    Elt<T>::st(p, oa);
    Elt<T>::st(p + half, ob);
  }
}

// ------------------------------------------------------------------ SwiGLU
template <typename T>
__global__ __launch_bounds__(256) void swiglu_fwd_k(const T *ag, int64_t rows, int64_t F, T *h) {
  const int64_t vec_per_row = F >> 2;
  const int64_t total = rows * vec_per_row;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / vec_per_row;
    const int64_t c = (idx - r * vec_per_row) * 4;
    f4 a = ld4(ag + r * 2 * F + c), g = ld4(ag + r * 2 * F + F + c);
    f4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sg = 1.0f / (1.0f + __expf(-g.v[j]));
      o.v[j] = a.v[j] * (g.v[j] * sg);
    }
    st4(h + r * F + c, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_k(const T *ag, const T *dh, int64_t rows, int64_t F,
                                                    T *dag) {
  const int64_t vec_per_row = F >> 2;
  const int64_t total = rows * vec_per_row;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / vec_per_row;
    const int64_t c = (idx - r * vec_per_row) * 4;
    f4 a = ld4(ag + r * 2 * F + c), g = ld4(ag + r * 2 * F + F + c), d = ld4(dh + r * F + c);
    f4 da, dg;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sg = 1.0f / (1.0f + __expf(-g.v[j]));
      da.v[j] = d.v[j] * (g.v[j] * sg);
      dg.v[j] = d.v[j] * a.v[j] * sg * (1.0f + g.v[j] * (1.0f - sg));
    }
    st4(dag + r * 2 * F + c, da);
    st4(dag + r * 2 * F + F + c, dg);
  }
}

// 8-element-vector forms (F % 8 == 0): one wave per row, 16-byte accesses for bf16.
template <typename T>
__global__ __launch_bounds__(256) void swiglu_fwd8_k(const T *__restrict__ ag, int64_t rows, int F,
                                                     T *__restrict__ h) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int nvec = F >> 3;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const T *ar = ag + r * 2 * F, *gr = ar + F;
    T *hr = h + r * F;
    for (int v0 = lane; v0 < nvec; v0 += 128) {
      const int v1 = v0 + 64;
      const bool two = v1 < nvec;
      Raw8<T> a0, g0, a1, g1;
      a0.load(ar + v0 * 8);
      g0.load(gr + v0 * 8);
      if (two) {
        a1.load(ar + v1 * 8);
        g1.load(gr + v1 * 8);
      }
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = g0.get(j);
        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-g));
        o[j] = a0.get(j) * (g * sg);
      }
      Raw8<T>::store(hr + v0 * 8, o);
      if (two) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float g = g1.get(j);
          const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-g));
          o[j] = a1.get(j) * (g * sg);
        }
        Raw8<T>::store(hr + v1 * 8, o);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd8_k(const T *__restrict__ ag, const T *__restrict__ dh,
                                                     int64_t rows, int F, T *__restrict__ dag) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int nvec = F >> 3;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const T *ar = ag + r * 2 * F, *gr = ar + F, *dr = dh + r * F;
    T *oa = dag + r * 2 * F, *og = oa + F;
    for (int v0 = lane; v0 < nvec; v0 += 128) {
      const int v1 = v0 + 64;
      const bool two = v1 < nvec;
      Raw8<T> a0, g0, d0, a1, g1, d1;
      a0.load(ar + v0 * 8);
      g0.load(gr + v0 * 8);
      d0.load(dr + v0 * 8);
      if (two) {
        a1.load(ar + v1 * 8);
        g1.load(gr + v1 * 8);
        d1.load(dr + v1 * 8);
      }
      float da[8], dg[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = g0.get(j), d = d0.get(j);
        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-g));
        da[j] = d * (g * sg);
        dg[j] = d * a0.get(j) * sg * (1.0f + g * (1.0f - sg));
      }
      Raw8<T>::store(oa + v0 * 8, da);
      Raw8<T>::store(og + v0 * 8, dg);
      if (two) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float g = g1.get(j), d = d1.get(j);
          const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-g));
          da[j] = d * (g * sg);
          dg[j] = d * a1.get(j) * sg * (1.0f + g * (1.0f - sg));
        }
        Raw8<T>::store(oa + v1 * 8, da);
        Raw8<T>::store(og + v1 * 8, dg);
      }
    }
  }
}

// ------------------------------------------------------------------ column sums
// Block (32 x 8): thread x owns 4 adjacent columns (one 4-element vector), 8 row lanes;
// grid.y slices the rows.  Rows are read in full 128-element (bf16: 256 B) runs.
template <typename T>
__global__ __launch_bounds__(256) void colsum_k(const T *X, int64_t M, int64_t N, int64_t ld,
                                                int64_t rows_per_slice, float *partial, const int *m_dev) {
  if (m_dev) M = m_dev[0] < M ? m_dev[0] : M;   // row count known only on the device
  __shared__ float sm[8][32][5];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t c = ((int64_t)blockIdx.x * 32 + tx) * 4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_slice;
  const int64_t r1 = (r0 + rows_per_slice < M) ? r0 + rows_per_slice : M;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  const bool vec_ok = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  if (c + 3 < N && vec_ok) {
#pragma unroll 4
    for (int64_t r = r0 + ty; r < r1; r += 8) {
      const f4 v = ld4(X + r * ld + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += v.v[j];
    }
  } else if (c < N) {
    for (int64_t r = r0 + ty; r < r1; r += 8)
      for (int j = 0; j < 4 && c + j < N; ++j) s[j] += Elt<T>::ld(X + r * ld + c + j);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) sm[ty][tx][j] = s[j];
  __syncthreads();
  if (ty < 4) {
    const int j = ty;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += sm[i][tx][j];
    if (c + j < N) partial[(int64_t)blockIdx.y * N + c + j] = t;
  }
}

// out[i, :] = x[rows[i], :] for i < count; rows count .. roundup(count, 64) are zero-filled.
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_k(const T *x, int64_t ld, const int *rows, const int *count,
                                                     int64_t d, T *out) {
  const int cnt = count[0];
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per output row
  if (i >= (((int64_t)cnt + 63) / 64) * 64) return;
  T *dst = out + i * d;
  if (i >= cnt) {
    const f4 z{{0.f, 0.f, 0.f, 0.f}};
    for (int64_t c = lane * 4; c < d; c += 256) st4(dst + c, z);
    return;
  }
  const T *src = x + (int64_t)rows[i] * ld;
  for (int64_t c = lane * 4; c < d; c += 256) st4(dst + c, ld4(src + c));
}

// out[rows[i], :] = x[i, :] for i < count (the inverse of gather_rows_k; the other rows of out are left alone)
template <typename T>
__global__ __launch_bounds__(256) void scatter_rows_k(const T *x, int64_t ld, const int *rows, const int *count,
                                                      int64_t out_rows, int64_t d, T *out, int64_t ld_out) {
  const int cnt = count[0];
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per source row
  if (i >= cnt) return;
  const int64_t r = rows[i];
  if (r < 0 || r >= out_rows) return;
  const T *src = x + i * ld;
  T *dst = out + r * ld_out;
  for (int64_t c = lane * 4; c < d; c += 256) st4(dst + c, ld4(src + c));
}

__global__ __launch_bounds__(256) void cast_f32_bf16_k(const float *src, bf16_t *dst, int64_t n4,
                                                       int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x)
    st4(dst + i * 4, ld4(src + i * 4));
  if (blockIdx.x == 0)
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) dst[i] = f32_to_bf16(src[i]);
}

inline unsigned stream_grid(int64_t work_items, int per_block) {
  int64_t b = ceil_div(work_items, per_block);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ------------------------------------------------------------------ residual dropout
// out = res + keep * y / (1 - p)   (src/transformer.py:352,361: src + self.dropout(sublayer(src)))
// and its backward gy = keep * g / (1 - p).  keep(row, col) is the hash of common.h with the
// batch*head slot set to 0xffffffff, regenerated by the backward: no mask is stored.
constexpr unsigned DROP_STREAM = 0xffffffffu;

template <typename T, bool ADD>
__global__ __launch_bounds__(256) void dropout8_k(const T *__restrict__ y, const T *__restrict__ res,
                                                  T *__restrict__ out, int64_t rows, int d, unsigned thresh,
                                                  float inv, unsigned seed_lo, unsigned seed_hi) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int nvec = d >> 3;
  const unsigned streamkey = ldm_drop_bh(seed_lo, DROP_STREAM);
  for (int64_t r = wave; r < rows; r += nwaves) {
    const unsigned rowkey = ldm_drop_row(streamkey, seed_hi, (unsigned)r);
    for (int v = lane; v < nvec; v += 64) {
      Raw8<T> a, b;
      a.load(y + r * d + v * 8);
      if (ADD) b.load(res + r * d + v * 8);
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float kept = ldm_drop_keep(rowkey, (unsigned)(v * 8 + j), thresh) ? a.get(j) * inv : 0.f;
        o[j] = ADD ? b.get(j) + kept : kept;
      }
      Raw8<T>::store(out + r * d + v * 8, o);
    }
  }
}

// any width / alignment
template <typename T, bool ADD>
__global__ __launch_bounds__(256) void dropout_scalar_k(const T *y, const T *res, T *out, int64_t rows, int64_t d,
                                                        unsigned thresh, float inv, unsigned seed_lo,
                                                        unsigned seed_hi) {
  const unsigned streamkey = ldm_drop_bh(seed_lo, DROP_STREAM);
  const int64_t total = rows * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / d;
    const unsigned rowkey = ldm_drop_row(streamkey, seed_hi, (unsigned)r);
    const float kept = ldm_drop_keep(rowkey, (unsigned)(i - r * d), thresh) ? Elt<T>::ld(y + i) * inv : 0.f;
    Elt<T>::st(out + i, ADD ? Elt<T>::ld(res + i) + kept : kept);
  }
}

template <typename T, bool ADD>
int dropout_launch(const void *y, const void *res, void *out, int64_t rows, int64_t d, unsigned thresh, float inv,
                   uint64_t seed, hipStream_t s) {
  const unsigned lo = (unsigned)seed, hi = (unsigned)(seed >> 32);
  if (d % 8 == 0 && d < (1 << 28) && aligned_to(y, 16) && aligned_to(out, 16) && (!ADD || aligned_to(res, 16)))
    hipLaunchKernelGGL((dropout8_k<T, ADD>), dim3(stream_grid(rows, 4)), dim3(256), 0, s, (const T *)y, (const T *)res,
                       (T *)out, rows, (int)d, thresh, inv, lo, hi);
  else
    hipLaunchKernelGGL((dropout_scalar_k<T, ADD>), dim3(stream_grid(rows * d, 256)), dim3(256), 0, s, (const T *)y,
                       (const T *)res, (T *)out, rows, d, thresh, inv, lo, hi);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

inline int rms_nv(int64_t d) {
  const int64_t need = ceil_div(d / 4, 64);
  int nv = 1;
  while (nv < need) nv <<= 1;
  return nv;
}

constexpr int kRmsBwdBlocks = 2048;  // rows of the dw partial workspace (8 workgroups per CU resident)

inline int rms_nv8(int64_t d) {
  const int64_t need = ceil_div(d / 8, 64);
  int nv = 1;
  while (nv < need) nv <<= 1;
  return nv;
}

}  // namespace

// ====================================================================== C ABI
extern "C" int ldm_embed_fwd(const int64_t *ids, const float *table, int64_t T, int64_t d,
                             int64_t vocab, void *out, int out_dtype, void *stream) {
  if (T < 0 || d <= 0 || vocab <= 0) return LDM_E_BADARG;
  if (T == 0) return LDM_OK;
  if (!ids || !table || !out) return LDM_E_BADARG;
  if (d % 4) return LDM_E_SHAPE;
  if (!aligned_to(table, 16) || !aligned_to(out, 16)) return LDM_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = stream_grid(T, 4);
  if (out_dtype == LDM_F32)
    hipLaunchKernelGGL(embed_fwd_k<float>, dim3(grid), dim3(256), 0, s, ids, table, T, d, vocab, (float *)out);
  else if (out_dtype == LDM_BF16)
    hipLaunchKernelGGL(embed_fwd_k<bf16_t>, dim3(grid), dim3(256), 0, s, ids, table, T, d, vocab, (bf16_t *)out);
  else
    return LDM_E_DTYPE;
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

extern "C" int ldm_embed_bwd(const int64_t *ids, const void *dout, int dtype, int64_t T, int64_t d,
                             int64_t vocab, int64_t hot_id, float *dtable, void *stream) {
  if (T < 0 || d <= 0 || vocab <= 0) return LDM_E_BADARG;
  if (T == 0) return LDM_OK;
  if (!ids || !dout || !dtable) return LDM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  constexpr int ROWS = 32;
  const unsigned grid = (unsigned)ceil_div(T, ROWS);
  if (dtype == LDM_F32)
    hipLaunchKernelGGL((embed_bwd_k<float, ROWS>), dim3(grid), dim3(256), 0, s, ids, (const float *)dout, T, d, vocab, hot_id, dtable);
  else if (dtype == LDM_BF16)
    hipLaunchKernelGGL((embed_bwd_k<bf16_t, ROWS>), dim3(grid), dim3(256), 0, s, ids, (const bf16_t *)dout, T, d, vocab, hot_id, dtable);
  else
    return LDM_E_DTYPE;
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

template <typename T>
static int rmsnorm_fwd_launch(const void *x, const float *w, int64_t rows, int64_t d, float eps, void *y,
                              float *rstd, hipStream_t s) {
  const unsigned grid = stream_grid(rows, 4);
  const T *xx = (const T *)x;
  T *yy = (T *)y;
  if (d % 8 == 0 && aligned_to(w, 16) && rms_nv8(d) <= 8) {
#define RMS_F8(NV) hipLaunchKernelGGL((rmsnorm_fwd8_k<T, NV>), dim3(grid), dim3(256), 0, s, xx, w, rows, (int)d, eps, yy, rstd)
    switch (rms_nv8(d)) {
      case 1: RMS_F8(1); break;
      case 2: RMS_F8(2); break;
      case 4: RMS_F8(4); break;
      default: RMS_F8(8); break;   // rows of up to 4096 elements (the LLaDA-8B width)
    }
#undef RMS_F8
    LDM_CHECK_LAUNCH();
    return LDM_OK;
  }
#define RMS_F(NV) hipLaunchKernelGGL((rmsnorm_fwd_k<T, NV>), dim3(grid), dim3(256), 0, s, xx, w, rows, (int)d, eps, yy, rstd)
  switch (rms_nv(d)) {
    case 1: RMS_F(1); break;
    case 2: RMS_F(2); break;
    case 4: RMS_F(4); break;
    case 8: RMS_F(8); break;
    case 16: RMS_F(16); break;
    case 32: RMS_F(32); break;
    default: return LDM_E_SHAPE;
  }
#undef RMS_F
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

extern "C" int ldm_rmsnorm_fwd(const void *x, const float *w, int64_t T, int64_t d, float eps, int dtype,
                               void *y, float *rstd, void *stream) {
  if (T < 0 || d <= 0) return LDM_E_BADARG;
  if (T == 0) return LDM_OK;
  if (!x || !w || !y || !rstd) return LDM_E_BADARG;
  if (d % 4 || d > 8192) return LDM_E_SHAPE;
  if (!aligned_to(x, 16) || !aligned_to(y, 16) || !aligned_to(w, 16)) return LDM_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == LDM_F32) return rmsnorm_fwd_launch<float>(x, w, T, d, eps, y, rstd, s);
  if (dtype == LDM_BF16) return rmsnorm_fwd_launch<bf16_t>(x, w, T, d, eps, y, rstd, s);
  return LDM_E_DTYPE;
}

// RMSNorm between an fp32 residual stream and bf16 projections: x fp32 -> y bf16 (statistics and products in
// fp32, one rounding), the form the reference's norm takes under bf16 autocast (src/transformer.py:54-58 run in
// fp32, the Linear behind it casts).  16-byte-vector kernels only: d % 8 == 0, d <= 4096.
extern "C" int ldm_rmsnorm_fwd_f32_bf16(const float *x, const float *w, int64_t T, int64_t d, float eps, void *y,
                                        float *rstd, void *stream) {
  if (T < 0 || d <= 0) return LDM_E_BADARG;
  if (T == 0) return LDM_OK;
  if (!x || !w || !y || !rstd) return LDM_E_BADARG;
  if (d % 8 || rms_nv8(d) > 8) return LDM_E_SHAPE;
  if (!aligned_to(x, 16) || !aligned_to(y, 16) || !aligned_to(w, 16)) return LDM_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = stream_grid(T, 4);
#define RMS_FM(NV) hipLaunchKernelGGL((rmsnorm_fwd8_k<float, NV, bf16_t>), dim3(grid), dim3(256), 0, s, x, w, T, (int)d, eps, (bf16_t *)y, rstd)
  switch (rms_nv8(d)) {
    case 1: RMS_FM(1); break;
    case 2: RMS_FM(2); break;
    case 4: RMS_FM(4); break;
    default: RMS_FM(8); break;
  }
#undef RMS_FM
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

// its backward: x, dres (nullable), dx fp32; dy bf16.  dw / dw_partial as ldm_rmsnorm_bwd.
extern "C" int ldm_rmsnorm_bwd_f32_bf16(const float *x, const float *w, const float *rstd, const void *dy,
                                        const float *dres, int64_t T, int64_t d, float *dx, float *dw,
                                        float *dw_partial, void *stream) {
  if (T < 0 || d <= 0) return LDM_E_BADARG;
  if (!dw) return LDM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (T == 0) return (int)hipMemsetAsync(dw, 0, d * sizeof(float), s);
  if (!x || !w || !rstd || !dy || !dx || !dw_partial) return LDM_E_BADARG;
  if (d % 8 || rms_nv8(d) > 8) return LDM_E_SHAPE;
  if (!aligned_to(x, 16) || !aligned_to(dy, 16) || !aligned_to(dx, 16) || !aligned_to(w, 16) || (dres && !aligned_to(dres, 16)))
    return LDM_E_ALIGN;
  int64_t blocks = ceil_div(T, 4);
  if (blocks > kRmsBwdBlocks) blocks = kRmsBwdBlocks;
  const unsigned grid = (unsigned)blocks;
  const size_t lds = (size_t)4 * d * sizeof(float);
  const RmsDrop nodrop{nullptr, 0u, 0u, 0u, 1.0f};
#define RMS_BM(NV)                                                                                                  \
  do {                                                                                                              \
    if (dres)                                                                                                       \
      hipLaunchKernelGGL((rmsnorm_bwd8_k<float, NV, true, false, bf16_t>), dim3(grid), dim3(256), lds, s, x, w, rstd, \
                         (const bf16_t *)dy, dres, T, (int)d, dx, dw_partial, nodrop);                              \
    else                                                                                                            \
      hipLaunchKernelGGL((rmsnorm_bwd8_k<float, NV, false, false, bf16_t>), dim3(grid), dim3(256), lds, s, x, w, rstd, \
                         (const bf16_t *)dy, dres, T, (int)d, dx, dw_partial, nodrop);                              \
  } while (0)
  switch (rms_nv8(d)) {
    case 1: RMS_BM(1); break;
    case 2: RMS_BM(2); break;
    case 4: RMS_BM(4); break;
    default: RMS_BM(8); break;
  }
#undef RMS_BM
  LDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(reduce_partials_k, dim3((unsigned)ceil_div(d, 32)), dim3(1024), 0, s, dw_partial, (int64_t)grid, d, dw);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

extern "C" int64_t ldm_rmsnorm_bwd_ws_floats(int64_t T, int64_t d) {
  (void)T;
  return (int64_t)kRmsBwdBlocks * d;
}

template <typename T>
static int rmsnorm_bwd_launch(const void *x, const float *w, const float *rstd, const void *dy,
                              const void *dres, int64_t rows, int64_t d, void *dx, float *dw,
                              float *dw_partial, hipStream_t s, int64_t max_blocks = kRmsBwdBlocks,
                              RmsDrop dr = RmsDrop{nullptr, 0u, 0u, 0u, 1.0f}) {
  int64_t blocks = ceil_div(rows, 4);
  if (blocks > max_blocks) blocks = max_blocks;
  const unsigned grid = (unsigned)blocks;
  const size_t lds = (size_t)4 * d * sizeof(float);
  if (d % 8 == 0 && rms_nv8(d) <= 8) {
#define RMS_B8_(NV, RES, DROP)                                                                                    \
  hipLaunchKernelGGL((rmsnorm_bwd8_k<T, NV, RES, DROP>), dim3(grid), dim3(256), lds, s, (const T *)x, w, rstd,    \
                     (const T *)dy, (const T *)dres, rows, (int)d, (T *)dx, dw_partial, dr)
#define RMS_B8(NV)                                                                                                \
  do {                                                                                                            \
    if (dr.dx_drop) {                                                                                             \
      if (dres) RMS_B8_(NV, true, true); else RMS_B8_(NV, false, true);                                           \
    } else {                                                                                                      \
      if (dres) RMS_B8_(NV, true, false); else RMS_B8_(NV, false, false);                                         \
    }                                                                                                             \
  } while (0)
    switch (rms_nv8(d)) {
      case 1: RMS_B8(1); break;
      case 2: RMS_B8(2); break;
      case 4: RMS_B8(4); break;
      default: RMS_B8(8); break;
    }
#undef RMS_B8
#undef RMS_B8_
    LDM_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_partials_k, dim3((unsigned)ceil_div(d, 32)), dim3(1024), 0, s, dw_partial,
                       (int64_t)grid, d, dw);
    LDM_CHECK_LAUNCH();
    return LDM_OK;
  }
  if (dr.dx_drop) return LDM_E_SHAPE;   // the second output exists in the 16-byte-vector kernels only
#define RMS_B(NV)                                                                                      \
  hipLaunchKernelGGL((rmsnorm_bwd_k<T, NV>), dim3(grid), dim3(256), lds, s, (const T *)x, w, rstd,      \
                     (const T *)dy, (const T *)dres, rows, (int)d, (T *)dx, dw_partial)
  switch (rms_nv(d)) {
    case 1: RMS_B(1); break;
    case 2: RMS_B(2); break;
    case 4: RMS_B(4); break;
    case 8: RMS_B(8); break;
    case 16: RMS_B(16); break;
    case 32: RMS_B(32); break;
    default: return LDM_E_SHAPE;
  }
#undef RMS_B
  LDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(reduce_partials_k, dim3((unsigned)ceil_div(d, 32)), dim3(1024), 0, s, dw_partial,
                     (int64_t)grid, d, dw);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

extern "C" int ldm_rmsnorm_bwd(const void *x, const float *w, const float *rstd, const void *dy,
                               const void *dres, int64_t T, int64_t d, int dtype, void *dx, float *dw,
                               float *dw_partial, void *stream) {
  if (T < 0 || d <= 0) return LDM_E_BADARG;
  if (!dw) return LDM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (T == 0) return (int)hipMemsetAsync(dw, 0, d * sizeof(float), s);
  if (!x || !w || !rstd || !dy || !dx || !dw_partial) return LDM_E_BADARG;
  // (the kernels keep four rows of d floats in LDS: 64 KiB, the limit of a launch without a raised
  // dynamic-LDS attribute, at d = 4096)
  if (d % 4 || d > 4096) return LDM_E_SHAPE;
  if (!aligned_to(x, 16) || !aligned_to(dy, 16) || !aligned_to(dx, 16) || !aligned_to(w, 16) ||
      (dres && !aligned_to(dres, 16)))
    return LDM_E_ALIGN;
  if (dtype == LDM_F32) return rmsnorm_bwd_launch<float>(x, w, rstd, dy, dres, T, d, dx, dw, dw_partial, s);
  if (dtype == LDM_BF16) return rmsnorm_bwd_launch<bf16_t>(x, w, rstd, dy, dres, T, d, dx, dw, dw_partial, s);
  return LDM_E_DTYPE;
}

// ldm_rmsnorm_bwd with a second output dx_drop = keep * dx / (1 - p), the mask that
// ldm_dropout_add(y, res, out, T, d, dtype, p, seed) applied to the [T, d] tensor this norm read.
extern "C" int ldm_rmsnorm_bwd_dropgrad(const void *x, const float *w, const float *rstd, const void *dy,
                                        const void *dres, int64_t T, int64_t d, int dtype, void *dx, void *dx_drop,
                                        float p, uint64_t seed, float *dw, float *dw_partial, void *stream) {
  if (T < 0 || d <= 0 || !(p >= 0.0f) || !(p < 1.0f)) return LDM_E_BADARG;
  if (!dw) return LDM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (T == 0) return (int)hipMemsetAsync(dw, 0, d * sizeof(float), s);
  if (!x || !w || !rstd || !dy || !dx || !dx_drop || !dw_partial) return LDM_E_BADARG;
  if (d % 8 || d > 4096) return LDM_E_SHAPE;
  if (!aligned_to(x, 16) || !aligned_to(dy, 16) || !aligned_to(dx, 16) || !aligned_to(dx_drop, 16) || !aligned_to(w, 16) ||
      (dres && !aligned_to(dres, 16)))
    return LDM_E_ALIGN;
  // threshold and scale exactly as ldm_dropout_add derives them
  const double t = (double)p * 4294967296.0;
  RmsDrop dr;
  dr.dx_drop = dx_drop;
  dr.thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
  dr.inv = (float)(4294967296.0 / (4294967296.0 - (double)dr.thresh));
  dr.seed_lo = (unsigned)seed; dr.seed_hi = (unsigned)(seed >> 32);
  if (dtype == LDM_F32) return rmsnorm_bwd_launch<float>(x, w, rstd, dy, dres, T, d, dx, dw, dw_partial, s, kRmsBwdBlocks, dr);
  if (dtype == LDM_BF16) return rmsnorm_bwd_launch<bf16_t>(x, w, rstd, dy, dres, T, d, dx, dw, dw_partial, s, kRmsBwdBlocks, dr);
  return LDM_E_DTYPE;
}

// ---------------------------------------------------------------- RMSNorm backward without a workspace argument
// The per-workgroup dw partials live in a file-scope device array instead of a caller-provided
// buffer, so the call neither allocates nor needs anything allocated for it.  Every GPU holds its
// own copy of the array at its own address: the address is looked up once per device.
constexpr int64_t kNoAllocScratchFloats = (int64_t)2 << 20;   // 8 MiB per GPU, held until the process ends
static __device__ __align__(256) float scratch[kNoAllocScratchFloats];
constexpr int kMaxDevices = 64;
static float *g_scratch_ptr[kMaxDevices];
static std::mutex g_scratch_mutex;

// device address of `scratch` on the CURRENT device (host code must never pass the array's name
// to a kernel: on the host it is only a handle for this lookup)
static int scratch_address(float **out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= kMaxDevices) return LDM_E_SHAPE;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    float *p = nullptr;
    e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return (int)e;
    if (!p) return LDM_E_WORKSPACE;
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return LDM_OK;
}

extern "C" int64_t ldm_rmsnorm_bwd_noalloc_max_d(void) { return 4096; }

extern "C" int ldm_rmsnorm_bwd_noalloc(const void *x, const float *w, const float *rstd, const void *dy,
                                       const void *dres, int64_t T, int64_t d, int dtype, void *dx, float *dw,
                                       void *stream) {
  if (T < 0 || d <= 0) return LDM_E_BADARG;
  if (!dw) return LDM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (T == 0) return (int)hipMemsetAsync(dw, 0, d * sizeof(float), s);
  if (!x || !w || !rstd || !dy || !dx) return LDM_E_BADARG;
  if (d % 4) return LDM_E_SHAPE;
  if (d > ldm_rmsnorm_bwd_noalloc_max_d()) return LDM_E_WORKSPACE;   // the fixed scratch array is too small
  if (!aligned_to(x, 16) || !aligned_to(dy, 16) || !aligned_to(dx, 16) || !aligned_to(w, 16) ||
      (dres && !aligned_to(dres, 16)))
    return LDM_E_ALIGN;
  float *partial = nullptr;
  const int rc = scratch_address(&partial);
  if (rc) return rc;
  // one row of d floats per workgroup: fewer workgroups than the workspace form when d > 1024.
  // Every row the second stage reads is written by the first, so nothing needs clearing.
  int64_t max_blocks = kNoAllocScratchFloats / d;
  if (max_blocks > kRmsBwdBlocks) max_blocks = kRmsBwdBlocks;
  if (dtype == LDM_F32) return rmsnorm_bwd_launch<float>(x, w, rstd, dy, dres, T, d, dx, dw, partial, s, max_blocks);
  if (dtype == LDM_BF16) return rmsnorm_bwd_launch<bf16_t>(x, w, rstd, dy, dres, T, d, dx, dw, partial, s, max_blocks);
  return LDM_E_DTYPE;
}

extern "C" int ldm_rope_qk(void *qkv, int dtype, int64_t batch, int64_t seq, int64_t nhead, int64_t hd,
                           int64_t ld, const float *cos_t, const float *sin_t, float q_scale, int inverse,
                           void *stream) {
  if (batch < 0 || seq < 0 || nhead <= 0 || hd <= 0) return LDM_E_BADARG;
  const int64_t rows = batch * seq;
  if (rows == 0) return LDM_OK;
  if (!qkv || !cos_t || !sin_t) return LDM_E_BADARG;
  if (hd % 2 || ld < 3 * nhead * hd) return LDM_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const int64_t half = hd / 2;
  const size_t esz = dtype == LDM_F32 ? 4 : 2;
  const bool vec = (half % 4 == 0) && (ld % 4 == 0) && aligned_to(qkv, 16) && aligned_to(cos_t, 16) &&
                   aligned_to(sin_t, 16) && ((half * esz) % (4 * esz) == 0);
  if (dtype != LDM_F32 && dtype != LDM_BF16) return LDM_E_DTYPE;
  if (vec && half % 8 == 0 && ld % 8 == 0 && (nhead * hd) % 8 == 0 && rows < (1 << 30) && seq < (1 << 30) &&
      2 * nhead * (half / 8) <= (1 << 12)) {
    constexpr int RG = 8;
    const unsigned grid = stream_grid(ceil_div(rows, RG), 1);
    // ceil(2^32 / d): floor(n / d) == umulhi(n, that) for n * d < 2^32 (here n < RG * per_row <= 2^15)
    const uint64_t per_row = 2 * (uint64_t)nhead * (half / 8), vph = half / 8;
    const unsigned m_row = (unsigned)(((1ull << 32) + per_row - 1) / per_row);
    const unsigned m_vph = vph == 1 ? 0u : (unsigned)(((1ull << 32) + vph - 1) / vph);
    if (dtype == LDM_F32)
      hipLaunchKernelGGL((rope_qk8_k<float, RG>), dim3(grid), dim3(256), 0, s, (float *)qkv, (int)rows, (int)seq, (int)nhead, (int)hd, ld, cos_t, sin_t, q_scale, inverse, m_row, m_vph);
    else
      hipLaunchKernelGGL((rope_qk8_k<bf16_t, RG>), dim3(grid), dim3(256), 0, s, (bf16_t *)qkv, (int)rows, (int)seq, (int)nhead, (int)hd, ld, cos_t, sin_t, q_scale, inverse, m_row, m_vph);
  } else if (vec) {
    const unsigned grid = stream_grid(rows * 2 * nhead * (half / 4), 256);
    if (dtype == LDM_F32)
      hipLaunchKernelGGL(rope_qk_k<float>, dim3(grid), dim3(256), 0, s, (float *)qkv, rows, seq, (int)nhead, (int)hd, ld, cos_t, sin_t, q_scale, inverse);
    else
      hipLaunchKernelGGL(rope_qk_k<bf16_t>, dim3(grid), dim3(256), 0, s, (bf16_t *)qkv, rows, seq, (int)nhead, (int)hd, ld, cos_t, sin_t, q_scale, inverse);
  } else {
    const unsigned grid = stream_grid(rows * 2 * nhead * half, 256);
    if (dtype == LDM_F32)
      hipLaunchKernelGGL(rope_qk_scalar_k<float>, dim3(grid), dim3(256), 0, s, (float *)qkv, rows, seq, (int)nhead, (int)hd, ld, cos_t, sin_t, q_scale, inverse);
    else
      hipLaunchKernelGGL(rope_qk_scalar_k<bf16_t>, dim3(grid), dim3(256), 0, s, (bf16_t *)qkv, rows, seq, (int)nhead, (int)hd, ld, cos_t, sin_t, q_scale, inverse);
  }
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

extern "C" int ldm_swiglu_fwd(const void *ag, int64_t T, int64_t F, int dtype, void *h, void *stream) {
  if (T < 0 || F <= 0) return LDM_E_BADARG;
  if (T == 0) return LDM_OK;
  if (!ag || !h) return LDM_E_BADARG;
  if (F % 4) return LDM_E_SHAPE;
  if (!aligned_to(ag, 16) || !aligned_to(h, 16)) return LDM_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (F % 8 == 0 && F < (1 << 28) && (dtype == LDM_F32 || dtype == LDM_BF16)) {
    const unsigned g8 = stream_grid(T, 4);
    if (dtype == LDM_F32)
      hipLaunchKernelGGL(swiglu_fwd8_k<float>, dim3(g8), dim3(256), 0, s, (const float *)ag, T, (int)F, (float *)h);
    else
      hipLaunchKernelGGL(swiglu_fwd8_k<bf16_t>, dim3(g8), dim3(256), 0, s, (const bf16_t *)ag, T, (int)F, (bf16_t *)h);
    LDM_CHECK_LAUNCH();
    return LDM_OK;
  }
  const unsigned grid = stream_grid(T * (F / 4), 256);
  if (dtype == LDM_F32)
    hipLaunchKernelGGL(swiglu_fwd_k<float>, dim3(grid), dim3(256), 0, s, (const float *)ag, T, F, (float *)h);
  else if (dtype == LDM_BF16)
    hipLaunchKernelGGL(swiglu_fwd_k<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t *)ag, T, F, (bf16_t *)h);
  else
    return LDM_E_DTYPE;
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

extern "C" int ldm_swiglu_bwd(const void *ag, const void *dh, int64_t T, int64_t F, int dtype, void *dag,
                              void *stream) {
  if (T < 0 || F <= 0) return LDM_E_BADARG;
  if (T == 0) return LDM_OK;
  if (!ag || !dh || !dag) return LDM_E_BADARG;
  if (F % 4) return LDM_E_SHAPE;
  if (!aligned_to(ag, 16) || !aligned_to(dh, 16) || !aligned_to(dag, 16)) return LDM_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  if (F % 8 == 0 && F < (1 << 28) && (dtype == LDM_F32 || dtype == LDM_BF16)) {
    const unsigned g8 = stream_grid(T, 4);
    if (dtype == LDM_F32)
      hipLaunchKernelGGL(swiglu_bwd8_k<float>, dim3(g8), dim3(256), 0, s, (const float *)ag, (const float *)dh, T, (int)F, (float *)dag);
    else
      hipLaunchKernelGGL(swiglu_bwd8_k<bf16_t>, dim3(g8), dim3(256), 0, s, (const bf16_t *)ag, (const bf16_t *)dh, T, (int)F, (bf16_t *)dag);
    LDM_CHECK_LAUNCH();
    return LDM_OK;
  }
  const unsigned grid = stream_grid(T * (F / 4), 256);
  if (dtype == LDM_F32)
    hipLaunchKernelGGL(swiglu_bwd_k<float>, dim3(grid), dim3(256), 0, s, (const float *)ag, (const float *)dh, T, F, (float *)dag);
  else if (dtype == LDM_BF16)
    hipLaunchKernelGGL(swiglu_bwd_k<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t *)ag, (const bf16_t *)dh, T, F, (bf16_t *)dag);
  else
    return LDM_E_DTYPE;
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

static inline int64_t colsum_slices(int64_t M) {
  int64_t sl = ceil_div(M, 256);
  if (sl > 64) sl = 64;
  if (sl < 1) sl = 1;
  return sl;
}

extern "C" int64_t ldm_colsum_ws_floats(int64_t M, int64_t N) { return colsum_slices(M) * N; }

extern "C" int ldm_scatter_rows(const void *x, int dtype, int64_t ld, const int32_t *rows, const int32_t *count,
                                int64_t max_rows, int64_t out_rows, int64_t d, void *out, int64_t ld_out, void *stream) {
  if (max_rows < 0 || out_rows < 0 || d <= 0 || ld < d || ld_out < d) return LDM_E_BADARG;
  if (max_rows == 0 || out_rows == 0) return LDM_OK;
  if (!x || !rows || !count || !out) return LDM_E_BADARG;
  if (d % 4 || ld % 4 || ld_out % 4) return LDM_E_SHAPE;
  if (!aligned_to(x, 16) || !aligned_to(out, 16)) return LDM_E_ALIGN;
  const unsigned grid = (unsigned)ceil_div(max_rows, 4);
  if (dtype == LDM_F32)
    hipLaunchKernelGGL(scatter_rows_k<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float *)x, ld, rows, count, out_rows, d, (float *)out, ld_out);
  else if (dtype == LDM_BF16)
    hipLaunchKernelGGL(scatter_rows_k<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)x, ld, rows, count, out_rows, d, (bf16_t *)out, ld_out);
  else
    return LDM_E_DTYPE;
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

static int colsum_impl(const void *X, int dtype, int64_t M, int64_t N, int64_t ld, float *out, float *ws,
                       const int *m_dev, void *stream);

extern "C" int ldm_colsum(const void *X, int dtype, int64_t M, int64_t N, int64_t ld, float *out, float *ws,
                          void *stream) {
  return colsum_impl(X, dtype, M, N, ld, out, ws, nullptr, stream);
}

extern "C" int ldm_colsum_rows_dyn(const void *X, int dtype, int64_t M, int64_t N, int64_t ld, float *out,
                                   float *ws, const int32_t *m_dev, void *stream) {
  if (!m_dev) return LDM_E_BADARG;
  return colsum_impl(X, dtype, M, N, ld, out, ws, m_dev, stream);
}

extern "C" int ldm_gather_rows(const void *x, int dtype, int64_t ld, const int32_t *rows, const int32_t *count,
                               int64_t max_rows, int64_t d, void *out, void *stream) {
  if (max_rows < 0 || d <= 0) return LDM_E_BADARG;
  if (max_rows == 0) return LDM_OK;
  if (!x || !rows || !count || !out) return LDM_E_BADARG;
  if (d % 4 || ld % 4) return LDM_E_SHAPE;
  if (!aligned_to(x, 16) || !aligned_to(out, 16)) return LDM_E_ALIGN;
  const unsigned grid = (unsigned)ceil_div(((max_rows + 63) / 64) * 64, 4);
  if (dtype == LDM_F32)
    hipLaunchKernelGGL(gather_rows_k<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float *)x, ld, rows, count, d, (float *)out);
  else if (dtype == LDM_BF16)
    hipLaunchKernelGGL(gather_rows_k<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)x, ld, rows, count, d, (bf16_t *)out);
  else
    return LDM_E_DTYPE;
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

static int colsum_impl(const void *X, int dtype, int64_t M, int64_t N, int64_t ld, float *out, float *ws,
                       const int *m_dev, void *stream) {
  if (M < 0 || N <= 0 || !out) return LDM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (M == 0) return (int)hipMemsetAsync(out, 0, N * sizeof(float), s);
  if (!X || !ws) return LDM_E_BADARG;
  const int64_t slices = colsum_slices(M);
  const int64_t rps = ceil_div(M, slices);
  dim3 grid((unsigned)ceil_div(N, 128), (unsigned)slices);
  if (dtype == LDM_F32)
    hipLaunchKernelGGL(colsum_k<float>, grid, dim3(256), 0, s, (const float *)X, M, N, ld, rps, ws, m_dev);
  else if (dtype == LDM_BF16)
    hipLaunchKernelGGL(colsum_k<bf16_t>, grid, dim3(256), 0, s, (const bf16_t *)X, M, N, ld, rps, ws, m_dev);
  else
    return LDM_E_DTYPE;
  LDM_CHECK_LAUNCH();
  hipLaunchKernelGGL(reduce_partials_k, dim3((unsigned)ceil_div(N, 32)), dim3(1024), 0, s, ws, slices, N, out);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

extern "C" int ldm_cast_f32_to_bf16(const float *src, void *dst, int64_t n, void *stream) {
  if (n < 0) return LDM_E_BADARG;
  if (n == 0) return LDM_OK;
  if (!src || !dst) return LDM_E_BADARG;
  if (!aligned_to(src, 16) || !aligned_to(dst, 8)) return LDM_E_ALIGN;
  hipLaunchKernelGGL(cast_f32_bf16_k, dim3(stream_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (bf16_t *)dst, n / 4, n);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

// dst[c, r] = src[r, c] for 2-byte elements: 64 x 64 tiles through LDS (rows padded to 66 elements, so that the
// column-wise reads of the second half do not all fall into one bank)
__global__ __launch_bounds__(256) void transpose16_k(const bf16_t *__restrict__ src, int64_t lds_, bf16_t *__restrict__ dst,
                                                       int64_t ldd, int64_t R, int64_t Cn) {
  __shared__ bf16_t tile[64][66];
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int t = threadIdx.x;
  for (int i = t; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < R && c0 + c < Cn) ? src[(r0 + r) * lds_ + c0 + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = t; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < Cn && r0 + r < R) dst[(c0 + c) * ldd + r0 + r] = tile[r][c];
  }
}

extern "C" int ldm_transpose_bf16(const void *src, int64_t ld_src, void *dst, int64_t ld_dst, int64_t rows, int64_t cols,
                                  void *stream) {
  if (rows < 0 || cols < 0) return LDM_E_BADARG;
  if (rows == 0 || cols == 0) return LDM_OK;
  if (!src || !dst || ld_src < cols || ld_dst < rows) return LDM_E_BADARG;
  if (ceil_div(rows, 64) > 65535) return LDM_E_SHAPE;
  hipLaunchKernelGGL(transpose16_k, dim3((unsigned)ceil_div(cols, 64), (unsigned)ceil_div(rows, 64)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t *)src, ld_src, (bf16_t *)dst, ld_dst, rows, cols);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

extern "C" int ldm_dropout_add(const void *y, const void *res, void *out, int64_t rows, int64_t d, int dtype, float p,
                               uint64_t seed, void *stream) {
  if (rows < 0 || d <= 0 || !(p >= 0.0f) || !(p < 1.0f)) return LDM_E_BADARG;
  if (rows == 0) return LDM_OK;
  if (!y || !out) return LDM_E_BADARG;
  if (rows > 0xffffffffll || d > 0xffffffffll) return LDM_E_SHAPE;   // 32-bit coordinates in the mask hash
  const double t = (double)p * 4294967296.0;
  const unsigned thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
  const float inv = (float)(4294967296.0 / (4294967296.0 - (double)thresh));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == LDM_F32)
    return res ? dropout_launch<float, true>(y, res, out, rows, d, thresh, inv, seed, s)
               : dropout_launch<float, false>(y, res, out, rows, d, thresh, inv, seed, s);
  if (dtype == LDM_BF16)
    return res ? dropout_launch<bf16_t, true>(y, res, out, rows, d, thresh, inv, seed, s)
               : dropout_launch<bf16_t, false>(y, res, out, rows, d, thresh, inv, seed, s);
  return LDM_E_DTYPE;
}
