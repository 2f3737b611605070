// Non-causal flash attention, forward and backward, bf16 on v_mfma_f32_32x32x16_bf16.
// Replaces src/transformer.py:255-285 (scores, key-padding mask, softmax, PV) and its
// autograd backward without materialising the S x S scores.
//
// Layout: q, k, v live in one fused activation [B*S, 3*H*hd]; a workgroup is 4 waves
// and owns 128 rows (queries, or keys in the dK/dV kernel) of one (batch, head), 32 per
// wave; the other side streams through LDS in 64-row tiles (global -> registers ->
// LDS, loads issued one tile ahead).
//
// Orientation: scores are computed TRANSPOSED, S^T = K Q^T (keys on the accumulator
// rows, the query on the lane), so the softmax statistics of a query are per-lane
// scalars and the fp32 accumulator, packed to bf16, is directly the B operand of the
// next product (O^T = V^T P^T, dQ^T = K^T dS^T).  The dK/dV kernel uses the other
// orientation (S = Q K^T, key on the lane) for the same reason.  The matching A
// operand needs its k index in the accumulator's row order
// (16s + 8(j>>2) + 4*half + (j&3)); ds_read_b64_tr_b16 delivers exactly that from a
// row-major [row][d] LDS tile.
#include "common.h"
#include "attn.h"
#include <stdlib.h>

namespace {

typedef short s4_t __attribute__((ext_vector_type(4)));
typedef short s8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef float f16_t __attribute__((ext_vector_type(16)));

constexpr float LOG2E = 1.4426950408889634f;

#define LDS_PTR2(p) ((__attribute__((address_space(3))) void *)(p))
#define GLB_PTR2(p) ((const __attribute__((address_space(1))) void *)(p))

template <int DP>
struct AT {
  static constexpr int NK = DP / 16;          // 16-deep k steps over the head dim
  static constexpr int ND = DP / 32;          // 32-wide blocks of the head dim
  static constexpr int RSTR = DP * 2 + 16;    // row stride (bytes) of tiles read by rows: odd multiple of 16 B
  static constexpr int TSTR = ((DP * 2) % 128 == 64) ? DP * 2 : DP * 2 + 64;   // tiles read transposed only
  static constexpr int NCH = DP * 8 / 256;    // 16-B chunks per thread for a 64-row tile
  static constexpr int CPR = DP / 8;          // 16-B chunks per tile row
};

struct FArgs {
  const bf16_t *qkv; int64_t ld;
  const uint8_t *key_valid;
  int B, H, S, hd;
  bf16_t *o; const bf16_t *o_in; const bf16_t *dout; int64_t ld_o;
  float *lse; const float *lse_in;
  // O - bf16(O), as fp8: written by the forward, read by the dQ kernel, so that delta = rowsum(dO * O) is
  // formed from O to ~12 mantissa bits.  Layout: private to the two kernels, one byte per element,
  // [workgroup (batch, head, 128-row block)][wave][lane][DP / 2] in the forward's accumulator order (both
  // kernels give lane (c, half) of wave w the row 128 blk + 32 w + c): DP / 32 16-byte stores per lane.  With the stored
  // bf16 O alone the error of delta is the SAME for every key of a query row and does not average out
  // over the keys in dQ: the q/k projection gradients of deep layers then carry 3x the error the
  // reference has under bf16 autocast (measured; its softmax backward sums P * dP in fp32).
  bf16_t *o_lo; const bf16_t *o_lo_in;
  bf16_t *dqkv; float *delta;
  unsigned long long *timing;   // experiment builds only
  // optional (backward): fp32 [S, hd/2] rotary tables; dq and dk are then written as gradients
  // w.r.t. the PRE-rotation q and k (dq also multiplied by rope_scale, the factor q was scaled by)
  const float *rope_cos, *rope_sin; float rope_scale;
  // attention-probability dropout (0 = none): keep threshold, 1/(1-p), seed halves (see common.h)
  unsigned drop_thresh; float drop_inv; unsigned seed_lo, seed_hi;
};

// Experiment builds (-DLDM_ATTN_TIMING): cycle totals of the forward kernel's per-tile segments of
// workgroup (1,0,0), wave 0, are written to the device address in LDM_ATTN_TIMING_BUF.
#ifdef LDM_ATTN_TIMING
#define ASTAMP(var) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); var = __builtin_amdgcn_s_memtime(); \
                         asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#else
#define ASTAMP(var) do { } while (0)
#endif

__device__ __forceinline__ bool aligned_to_dev(const void *p, unsigned a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
// Workgroup -> (128-row block, head, batch).  The grid is one-dimensional and consecutive
// workgroup ids go round-robin over the 8 XCDs, each with its own L2: the nblk workgroups that
// stream the SAME K/V (or Q/dO) rows of one (batch, head) are therefore given ids that are 8
// apart, so that they run on one XCD, back to back, and share those rows through its L2 instead
// of fetching them from beyond L2 once per XCD.
struct Blk { int blk, head, b; };
__device__ __forceinline__ Blk decode_blk(const int S, const int H) {
  const int nblk = (S + 127) / 128;
  const int w = blockIdx.x, group = 8 * nblk;
  int L = w;
  if (w < (int)(gridDim.x / group) * group) {
    const int j = w >> 3, x = w & 7;
    L = ((j / nblk) * 8 + x) * nblk + j % nblk;
  }
  Blk r;
  r.blk = L % nblk;
  const int bh = L / nblk;
  r.head = bh % H;
  r.b = bh / H;
  return r;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// 64 rows x DP columns, rows >= S and columns >= hd zero-filled
template <int DP>
__device__ __forceinline__ void tile_load(const bf16_t *base, int64_t ld, int row0, int S, int hd, int tid,
                                          uint4 (&r)[AT<DP>::NCH]) {
#pragma unroll
  for (int i = 0; i < AT<DP>::NCH; ++i) {
    const int c = tid + i * 256;
    const int row = c / AT<DP>::CPR, c8 = c - row * AT<DP>::CPR;
    const int grow = row0 + row;
    if (grow < S && c8 * 8 < hd)
      r[i] = *reinterpret_cast<const uint4 *>(base + (int64_t)grow * ld + c8 * 8);
    else
      r[i] = make_uint4(0u, 0u, 0u, 0u);
  }
}
// The same 64-row tile walk with the per-thread addressing done ONCE: chunk i of this thread is
// (row rr[i], 16-byte column cc[i]) of every tile; a tile step adds 64 rows.
template <int DP>
struct TileWalk {
  const bf16_t *p[AT<DP>::NCH];   // address of the chunk in tile 0 (clamped into the matrix when out of range)
  int rr[AT<DP>::NCH];            // tile-relative row, or a huge value for columns >= hd
  int64_t step;                   // 64 rows
  __device__ __forceinline__ void init(const bf16_t *base, int64_t ld, int hd, int tid) {
    step = 64 * ld;
#pragma unroll
    for (int i = 0; i < AT<DP>::NCH; ++i) {
      const int c = tid + i * 256;
      const int row = c / AT<DP>::CPR, c8 = c - row * AT<DP>::CPR;
      const bool col_ok = c8 * 8 < hd;
      rr[i] = col_ok ? row : (1 << 30);
      p[i] = base + (int64_t)row * ld + (col_ok ? c8 * 8 : 0);
    }
  }
  // tile t (rows 64 t ..), rows >= S and columns >= hd zero-filled
  __device__ __forceinline__ void load(int t, int S, uint4 (&r)[AT<DP>::NCH]) const {
    const int64_t off = (int64_t)t * step;
    const int lim = S - 64 * t;
#pragma unroll
    for (int i = 0; i < AT<DP>::NCH; ++i) {
      if (rr[i] < lim) r[i] = *reinterpret_cast<const uint4 *>(p[i] + off);
      else r[i] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
};

template <int DP, int STR>
__device__ __forceinline__ void tile_store(char *tile, int tid, const uint4 (&r)[AT<DP>::NCH]) {
#pragma unroll
  for (int i = 0; i < AT<DP>::NCH; ++i) {
    const int c = tid + i * 256;
    const int row = c / AT<DP>::CPR, c8 = c - row * AT<DP>::CPR;
    *reinterpret_cast<uint4 *>(tile + row * STR + c8 * 16) = r[i];
  }
}

// MFMA operand with the tile ROW on the operand's row/col index and k = head dim
template <int STR>
__device__ __forceinline__ bf8_t row_frag(const char *tile, int row, int kk, int half) {
  return *reinterpret_cast<const bf8_t *>(tile + row * STR + kk * 32 + half * 16);
}
// MFMA A operand with the head dim (block dblk) on the operand's row and k = tile rows
// rbase .. rbase+15 in accumulator-row order
template <int STR>
__device__ __forceinline__ bf8_t tr_frag(const char *tile, int rbase, int dblk, int lane) {
  const int half = lane >> 5, gg = (lane >> 4) & 1, i = lane & 15;
  const char *a0 = tile + (rbase + 4 * half + (i >> 2)) * STR + (dblk * 32 + 16 * gg + 4 * (i & 3)) * 2;
  s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(a0));
  s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(a0 + 8 * STR));
  s8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf8_t, v);
}
// direct-from-global operand: row `row` of a [rows][hd] matrix, k = 16*kk + 8*half + 0..7
__device__ __forceinline__ bf8_t global_frag(const bf16_t *base, int64_t ld, int row, bool row_ok, int kk,
                                             int half, int hd) {
  const int d = 16 * kk + 8 * half;
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (row_ok && d < hd) v = *reinterpret_cast<const uint4 *>(base + (int64_t)row * ld + d);
  return __builtin_bit_cast(bf8_t, v);
}
__device__ __forceinline__ bf8_t pack8(const float *p) {
  bf8_t r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)p[j];
  return r;
}
// accumulator register r of a 32x32 tile sits on row (r&3) + 8*(r>>2) + 4*half
__device__ __forceinline__ constexpr int acc_row(int r) { return (r & 3) + 8 * (r >> 2); }

// store an accumulator tile transposed: lane owns matrix row `row`, registers run over d
template <int DP>
__device__ __forceinline__ void store_rows(bf16_t *dst_row, const f16_t (&acc)[AT<DP>::ND], float scale,
                                           int half, int hd) {
#pragma unroll
  for (int db = 0; db < AT<DP>::ND; ++db)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int d = db * 32 + 8 * r4 + 4 * half;
      if (d < hd) {
        f4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v.v[j] = acc[db][4 * r4 + j] * scale;
        st4(dst_row + d, v);
      }
    }
}

// Store a wave's 32 x DP accumulator tile (lane = row, registers over d) through LDS so that the
// global stores are 16-byte pieces of CONTIGUOUS rows (a row-per-lane store is issue-bound:
// every instruction touches 32 rows).  `wlds`: this wave's private 32 * RSTR bytes; the caller has
// made sure the workgroup is done with whatever lived there.  rows_ok = rows of the tile inside
// the matrix.
template <int DP>
__device__ __forceinline__ void store_tile_lds(char *wlds, bf16_t *dst, int64_t ld, const f16_t (&acc)[AT<DP>::ND],
                                               float scale, int lane, int rows_ok, int hd) {
  using T = AT<DP>;
  const int half = lane >> 5, c = lane & 31;
#pragma unroll
  for (int db = 0; db < T::ND; ++db)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int d = db * 32 + 8 * r4 + 4 * half;
      uint2 v;
      v.x = pack_bf16x2(acc[db][4 * r4 + 0] * scale, acc[db][4 * r4 + 1] * scale);
      v.y = pack_bf16x2(acc[db][4 * r4 + 2] * scale, acc[db][4 * r4 + 3] * scale);
      *reinterpret_cast<uint2 *>(wlds + c * T::RSTR + d * 2) = v;
    }
  constexpr int NPIECE = 32 * T::CPR;
#pragma unroll
  for (int i = 0; i < (NPIECE + 63) / 64; ++i) {
    const int idx = lane + 64 * i;
    const int row = idx / T::CPR, piece = idx - row * T::CPR;
    if (idx < NPIECE && row < rows_ok && piece * 8 < hd) {
      typedef unsigned u4n_t __attribute__((ext_vector_type(4)));
      const u4n_t v = *reinterpret_cast<const u4n_t *>(wlds + row * T::RSTR + piece * 16);
      // (streamed: the consumer is a GEMM several hundred MB later; -1 % on the backward in the step)
      __builtin_nontemporal_store(v, reinterpret_cast<u4n_t *>(dst + (int64_t)row * ld + piece * 8));
    }
  }
}

// The transposed rotary rotation of an accumulator tile, in place (head dim == DP): the pair
// (d, d + DP/2) of one row lives in ONE lane -- registers of block d/32 and of block
// (d + DP/2)/32 -- so it needs no exchange.  Forward (src/transformer.py:178-185):
// y[d] = x[d] cos - x[d+h] sin, y[d+h] = x[d+h] cos + x[d] sin; this applies the transpose.
template <int DP>
__device__ __forceinline__ void unrope_acc(f16_t (&acc)[AT<DP>::ND], float scale, int half, const float *cos_row,
                                           const float *sin_row) {
  constexpr int H2 = DP / 2;
#pragma unroll
  for (int d0 = 0; d0 < H2; d0 += 8) {
    const int lb = d0 / 32, lr = (d0 % 32) / 8;
    const int e0 = d0 + H2;
    const int hb = e0 / 32, hr = (e0 % 32) / 8;
    const int d = d0 + 4 * half;
    const f4 cs = ld4(cos_row + d), sn = ld4(sin_row + d);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gl = acc[lb][4 * lr + j], gh = acc[hb][4 * hr + j];
      acc[lb][4 * lr + j] = (gl * cs.v[j] + gh * sn.v[j]) * scale;
      acc[hb][4 * hr + j] = (gh * cs.v[j] - gl * sn.v[j]) * scale;
    }
  }
}

// O_lo = O - bf16(O) of this lane's share of the output tile (see FArgs::o_lo), stored as OCP fp8 (e4m3)
// of O_lo * 2^12: the residual is at most 2^-9 of |O|, so three more mantissa bits on it take the error
// of delta from 2^-9 to 2^-13 of |O| |dO|, below every other rounding of the backward, at one byte per
// element.  Bytes 8 kk .. 8 kk + 7 of the lane's slot are head-dim positions 16 kk + 8 half + 0..7.
constexpr float OLO_SCALE = 4096.0f, OLO_INV = 1.0f / 4096.0f;
template <int DP>
__device__ __forceinline__ void store_rounding_residual(const FArgs &a, const f16_t (&oacc)[AT<DP>::ND], float inv_l,
                                                        const Blk &blk, int wave, int lane) {
  const int nblk = (a.S + 127) / 128;
  // a lane holds half of its row: DP / 2 bytes per lane, 64 * DP per workgroup
  unsigned char *slot = reinterpret_cast<unsigned char *>(a.o_lo) +
                        (((((int64_t)blk.b * a.H + blk.head) * nblk + blk.blk) * 4 + wave) * 64 + lane) * (DP / 2);
#pragma unroll
  for (int db = 0; db < AT<DP>::ND; ++db) {
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = oacc[db][4 * q + j] * inv_l;
        r[j] = (v - bf16_to_f32(f32_to_bf16(v))) * OLO_SCALE;
      }
      int pk = __builtin_amdgcn_cvt_pk_fp8_f32(r[0], r[1], 0, false);
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(r[2], r[3], pk, true);
      w[q] = (unsigned)pk;
    }
    // w[q] = positions 32 db + 8 q + 4 half + 0..3.  Swapping the upper half-wave of w[2p] with the lower
    // half-wave of w[2p+1] leaves lane (c, half) with positions 32 db + 16 p + 8 half + 0..7 in
    // (w[2p], w[2p+1]): the k-operand order in which the dQ kernel holds dO (16 kk + 8 half + 0..7, kk = 2 db + p)
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
      const auto sw = __builtin_amdgcn_permlane32_swap(w[2 * p2], w[2 * p2 + 1], false, false);
      w[2 * p2] = sw[0];
      w[2 * p2 + 1] = sw[1];
    }
    *reinterpret_cast<uint4 *>(slot + db * 16) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// This lane's part of rowsum(o_lo[row] * dO[row]) (see FArgs::o_lo) against the dO fragments the lane already
// holds (dof[kk] = positions 16 kk + 8 half + 0..7); the caller's cross-half add completes the row.  In two
// steps, so that the caller can request the bytes together with its other prologue loads (one trip to
// memory instead of one after the other).
template <int DP>
__device__ __forceinline__ void delta_residual_load(const FArgs &a, const Blk &blk, int wave, int lane,
                                                    uint4 (&lo)[AT<DP>::ND]) {
#pragma unroll
  for (int db = 0; db < AT<DP>::ND; ++db) lo[db] = make_uint4(0u, 0u, 0u, 0u);   // fp8 zero
  if (a.o_lo_in) {
    const int nblk = (a.S + 127) / 128;
    const unsigned char *slot = reinterpret_cast<const unsigned char *>(a.o_lo_in) +
                                (((((int64_t)blk.b * a.H + blk.head) * nblk + blk.blk) * 4 + wave) * 64 + lane) * (DP / 2);
#pragma unroll
    for (int db = 0; db < AT<DP>::ND; ++db) lo[db] = *reinterpret_cast<const uint4 *>(slot + db * 16);
  }
}
template <int DP>
__device__ __forceinline__ float delta_residual(const uint4 (&lo)[AT<DP>::ND], const bf8_t (&dof)[AT<DP>::NK]) {
  float acc = 0.f;
#pragma unroll
  for (int db = 0; db < AT<DP>::ND; ++db) {
    const unsigned lw[4] = {lo[db].x, lo[db].y, lo[db].z, lo[db].w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kk = 2 * db + (q >> 1), j0 = 4 * (q & 1);
      const auto r01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)lw[q], false);
      const auto r23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)lw[q], true);
      acc = fmaf(r01[0], (float)dof[kk][j0 + 0], acc);
      acc = fmaf(r01[1], (float)dof[kk][j0 + 1], acc);
      acc = fmaf(r23[0], (float)dof[kk][j0 + 2], acc);
      acc = fmaf(r23[1], (float)dof[kk][j0 + 3], acc);
    }
  }
  return acc * OLO_INV;
}

// --------------------------------------------------------------------------- forward
template <int DP, bool DROP>
__global__ __launch_bounds__(256, 2) void flash_fwd_k(FArgs a) {
  using T = AT<DP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *Kt = smem;
  char *Vt = smem + 64 * T::RSTR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, c = lane & 31;
  const Blk blk = decode_blk(a.S, a.H);
  const int b = blk.b, head = blk.head;
  const int S = a.S, hd = a.hd;
  const int64_t D = (int64_t)a.H * hd;
  const bf16_t *qb = a.qkv + (int64_t)b * S * a.ld + (int64_t)head * hd;
  const bf16_t *kb = qb + D, *vb = qb + 2 * D;
  const int q = blk.blk * 128 + wave * 32 + c;

#ifdef LDM_ATTN_TIMING
  unsigned long long tk0;
  ASTAMP(tk0);
#endif
  bf8_t qf[T::NK];
#pragma unroll
  for (int kk = 0; kk < T::NK; ++kk) qf[kk] = global_frag(qb, a.ld, q < S ? q : S - 1, true, kk, half, hd);

  f16_t oacc[T::ND];
#pragma unroll
  for (int i = 0; i < T::ND; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;
  const unsigned drop_row = DROP ? ldm_drop_row(ldm_drop_bh(a.seed_lo, (unsigned)(b * a.H + head)), a.seed_hi, (unsigned)q) : 0u;

  const int nt = (S + 63) / 64;
  uint4 kr[T::NCH], vr[T::NCH];
  TileWalk<DP> kw, vw;
  kw.init(kb, a.ld, hd, tid);
  vw.init(vb, a.ld, hd, tid);
  kw.load(0, S, kr);
  vw.load(0, S, vr);
  tile_store<DP, T::RSTR>(Kt, tid, kr);
  tile_store<DP, T::TSTR>(Vt, tid, vr);
  __syncthreads();

#ifdef LDM_ATTN_TIMING
  unsigned long long ta[6] = {0, 0, 0, 0, 0, 0}, s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
#endif
  for (int t = 0; t < nt; ++t) {
    ASTAMP(s0);
    const int k0 = t * 64;
    if (t + 1 < nt) {
      kw.load(t + 1, S, kr);
      vw.load(t + 1, S, vr);
    }
    bool kok = (k0 + lane) < S;
    if (kok && a.key_valid) kok = a.key_valid[(int64_t)b * S + k0 + lane] != 0;
    const unsigned long long valid = __ballot(kok);
    ASTAMP(s1);

    // S^T = K Q^T: the K fragments are read two MFMAs ahead of their use
    f16_t sacc[2];
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kbk][r] = 0.f;
    {
      constexpr int NF = 2 * T::NK, AHEAD = 2;
      bf8_t kf[NF];
#pragma unroll
      for (int f = 0; f < AHEAD; ++f) kf[f] = row_frag<T::RSTR>(Kt, (f / T::NK) * 32 + c, f % T::NK, half);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f + AHEAD < NF) kf[f + AHEAD] = row_frag<T::RSTR>(Kt, ((f + AHEAD) / T::NK) * 32 + c, (f + AHEAD) % T::NK, half);
        sacc[f / T::NK] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[f], qf[f % T::NK], sacc[f / T::NK], 0, 0, 0);
      }
    }
#ifdef LDM_ATTN_TIMING
    asm volatile("" : "+v"(sacc[0]), "+v"(sacc[1]));
    ASTAMP(s2);
#endif
    if (valid != ~0ull) {
      const unsigned long long vs = valid >> (4 * half);
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (!((vs >> (kbk * 32 + acc_row(r))) & 1ull)) sacc[kbk][r] = -INFINITY;
    }
    float mx = m;
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kbk][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mu = (mx == -INFINITY) ? 0.f : mx;
    const float alpha = fast_exp2((m - mu) * LOG2E);
    m = mx;
    float ps = 0.f;
    const float mneg = -mu * LOG2E;
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(fmaf(sacc[kbk][r], LOG2E, mneg));
        sacc[kbk][r] = p;
        ps += p;
      }
    lsum = lsum * alpha + ps;
    if constexpr (DROP) {   // the row sum above is that of the undropped softmax
      const unsigned kb0 = (unsigned)(k0 + 4 * half);
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sacc[kbk][r] = ldm_drop_keep(drop_row, kb0 + kbk * 32 + acc_row(r), a.drop_thresh) ? sacc[kbk][r] * a.drop_inv : 0.f;
    }
#pragma unroll
    for (int i = 0; i < T::ND; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;

    ASTAMP(s3);
    // O^T += V^T P^T: the V fragments are read two MFMAs ahead of their use
    {
      constexpr int NF = 4 * T::ND, AHEAD = 2;     // fragment f: key block f / ND (16 keys), head-dim block f % ND
      bf8_t vfr[NF], pf[4];
#pragma unroll
      for (int f = 0; f < AHEAD; ++f) vfr[f] = tr_frag<T::TSTR>(Vt, (f / T::ND) * 16, f % T::ND, lane);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float tmp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tmp[j] = sacc[g4 >> 1][8 * (g4 & 1) + j];
        pf[g4] = pack8(tmp);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f + AHEAD < NF) vfr[f + AHEAD] = tr_frag<T::TSTR>(Vt, ((f + AHEAD) / T::ND) * 16, (f + AHEAD) % T::ND, lane);
        oacc[f % T::ND] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[f], pf[f / T::ND], oacc[f % T::ND], 0, 0, 0);
      }
    }
#ifdef LDM_ATTN_TIMING
    asm volatile("" : "+v"(oacc[0]));
    ASTAMP(s4);
#endif
    __syncthreads();
    if (t + 1 < nt) {
      tile_store<DP, T::RSTR>(Kt, tid, kr);
      tile_store<DP, T::TSTR>(Vt, tid, vr);
      __syncthreads();
    }
#ifdef LDM_ATTN_TIMING
    ASTAMP(s5);
    ta[0] += s1 - s0; ta[1] += s2 - s1; ta[2] += s3 - s2; ta[3] += s4 - s3; ta[4] += s5 - s4; ta[5] += 1;
#endif
  }
  const float ltot = lsum + __shfl_xor(lsum, 32, 64);
  if (q < S && half == 0) a.lse[((int64_t)b * a.H + head) * S + q] = m + logf(ltot);
  {
    const int q0 = blk.blk * 128 + wave * 32;   // the loop ended with a barrier: the K/V tiles are dead
    const float inv_l = 1.0f / ltot;
    if (a.ld_o % 8 == 0 && aligned_to_dev(a.o, 16))
      store_tile_lds<DP>(smem + wave * 32 * T::RSTR, a.o + ((int64_t)b * S + q0) * a.ld_o + (int64_t)head * hd, a.ld_o,
                         oacc, inv_l, lane, S - q0, hd);
    else if (q < S)
      store_rows<DP>(a.o + ((int64_t)b * S + q) * a.ld_o + (int64_t)head * hd, oacc, inv_l, half, hd);
    if (a.o_lo) store_rounding_residual<DP>(a, oacc, inv_l, blk, wave, lane);
  }
#ifdef LDM_ATTN_TIMING
  if (a.timing && tid == 0 && blockIdx.x == 8) {
    unsigned long long tk1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ASTAMP(tk1);
    for (int i = 0; i < 6; ++i) a.timing[i] = ta[i];
    a.timing[6] = tk1 - tk0;
  }
#endif
}

// =========================================================================== LDS-DMA variants
// Head dims 96 and 128: the streamed 64-row tiles go global -> LDS by 16-byte LDS-DMA (no
// register staging, no ds_write, one barrier per tile, two buffers).  An LDS-DMA instruction
// writes 1 KiB of CONTIGUOUS LDS (lane l at +16 l), so rows cannot be padded: the row pitch is
// exactly the row (192 B for head dim 96, 256 B for 128) and 16-byte chunk c of row r sits at
// chunk position pos(c, r), an XOR swizzle applied to the SOURCE address of each lane:
//   pitch 256: c ^ (((r & 3) << 2) | ((r >> 2) & 3))
//   pitch 192: c ^ ((r >> 2) & 3)     (rows already step 48 banks; only the low chunk bits move)
// Both are conflict-free for the by-row reads (16 lanes = 16 rows, one chunk) and for the
// transposing reads (32 lanes = 4 rows x 4 chunks), and invariant under r += 16.
__device__ __forceinline__ int swz2(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
template <int DP>
struct Lay {
  static_assert(DP == 96 || DP == 128, "LDS-DMA tiles: head dim 96 or 128");
  static constexpr int PITCH = DP * 2;
  static constexpr int CPR = PITCH / 16;           // chunks per row
  static constexpr int TILE = 64 * PITCH;          // bytes of a 64-row image
  static constexpr int PW = TILE / 1024 / 4;       // LDS-DMA instructions per wave per image
  static __device__ __forceinline__ int pos(int ch, int r) {
    return DP == 96 ? (ch ^ ((r >> 2) & 3)) : (ch ^ swz2(r));
  }
};


// The PW LDS-DMA pieces (1 KiB each) of a 64-row image that one of the 4 waves issues.
template <int DP>
struct TileDma {
  using L = Lay<DP>;
  const char *base;         // wave-uniform: row 0 of tile 0 (scalar registers)
  unsigned off[L::PW];      // byte offset of this lane's source in tile 0, piece i (32 bits: 64 rows of one sequence)
  int64_t step;             // 64 rows, bytes
  static __device__ __forceinline__ int row_of(int i, int wave, int lane) { return ((wave * L::PW + i) * 64 + lane) / L::CPR; }
  __device__ __forceinline__ void init(const bf16_t *b, int64_t ld, int wave, int lane) {
    base = reinterpret_cast<const char *>(b);
    step = 128 * ld;
#pragma unroll
    for (int i = 0; i < L::PW; ++i) {
      const int cidx = (wave * L::PW + i) * 64 + lane;   // chunk slot of this lane in the image
      const int row = cidx / L::CPR, slot = cidx % L::CPR;
      off[i] = (unsigned)(row * (int)(2 * ld) + L::pos(slot, row) * 16);   // pos() is its own inverse
    }
  }
  // piece i of tile t -> image `img`; rows >= S re-read row S-1 (finite values; masked later).
  // Tile t must hold at least one row (64 t < S): the clamp steps back inside the tile only.
  __device__ __forceinline__ void issue_one(int i, int t, int S, int64_t ld, char *img, int wave) const {
    const int lim = S - 64 * t;            // wave-uniform, >= 1
    int64_t toff = (int64_t)t * step;      // a scalar product: the tile's base stays in scalar registers
    asm volatile("" : "+s"(toff));
    const char *tb = base + toff;
    unsigned o = off[i];
    if (__builtin_expect(lim < 64, 0)) {
      // the last, partial tile only.  The empty asm keeps this a branch: as a select the clamp costs every
      // tile of every piece two quarter-rate integer multiplies and a multiply-add per lane
      // (about a third of the vector work of a forward tile at head dim 96).
      asm volatile("" : "+v"(o));
      const int row0 = row_of(i, wave, (int)(threadIdx.x & 63));
      o -= (unsigned)((row0 >= lim ? row0 - (lim - 1) : 0) * (int)(2 * ld));
    }
    __builtin_amdgcn_global_load_lds(GLB_PTR2(tb + o), LDS_PTR2(img + (wave * L::PW + i) * 1024), 16, 0, 0);
  }
  __device__ __forceinline__ void issue(int t, int S, int64_t ld, char *img, int wave) const {
#pragma unroll
    for (int i = 0; i < L::PW; ++i) issue_one(i, t, S, ld, img, wave);
  }
};

// The same with one 64-bit source pointer per piece: the dK/dV kernel at head dim 96 sits at 256 registers and
// the register allocator spills with the form above but not with this one.
template <int DP>
struct TileDmaP {
  using L = Lay<DP>;
  const bf16_t *p[L::PW];   // source of this lane in tile 0, piece i
  int row0[L::PW];          // its tile-relative row
  int64_t step;             // 64 rows
  __device__ __forceinline__ void init(const bf16_t *base, int64_t ld, int wave, int lane) {
    step = 64 * ld;
#pragma unroll
    for (int i = 0; i < L::PW; ++i) {
      const int cidx = (wave * L::PW + i) * 64 + lane;   // chunk slot of this lane in the image
      const int row = cidx / L::CPR, slot = cidx % L::CPR;
      row0[i] = row;
      p[i] = base + (int64_t)row * ld + L::pos(slot, row) * 8;   // pos() is its own inverse
    }
  }
  // piece i of tile t -> image `img`; rows >= S re-read row S-1 (finite values; masked later)
  __device__ __forceinline__ void issue_one(int i, int t, int S, int64_t ld, char *img, int wave) const {
    const int lim = S - 64 * t;            // wave-uniform
    int64_t toff = (int64_t)t * step;      // a scalar product; one 64-bit add per lane below
    asm volatile("" : "+s"(toff));
    const bf16_t *src = p[i] + toff;
    if (__builtin_expect(lim < 64, 0)) {
      // the last, partial tile only.  The empty asm keeps this a branch: as a select the clamp costs every
      // tile of every piece two quarter-rate integer multiplies and a 64-bit multiply-add per lane
      // (about a third of the vector work of a forward tile at head dim 96).
      asm volatile("" : "+v"(src));
      src -= (int64_t)(row0[i] >= lim ? row0[i] - (lim - 1) : 0) * ld;
    }
    __builtin_amdgcn_global_load_lds(GLB_PTR2(src), LDS_PTR2(img + (wave * L::PW + i) * 1024), 16, 0, 0);
  }
  __device__ __forceinline__ void issue(int t, int S, int64_t ld, char *img, int wave) const {
#pragma unroll
    for (int i = 0; i < L::PW; ++i) issue_one(i, t, S, ld, img, wave);
  }
};

template <int DP>
__device__ __forceinline__ bf8_t row_frag2(const char *tile, int row, int kk, int half) {
  return *reinterpret_cast<const bf8_t *>(tile + row * Lay<DP>::PITCH + (Lay<DP>::pos(kk * 2 + half, row) << 4));
}

// Transposing reads issued from inline asm: the compiler orders nothing around them (its own
// transposing-read intrinsic makes it drain every LDS-DMA in flight first), so the caller waits
// with tr_wait2 before it uses the registers.  `base` = per-lane byte address of (r0, dblk)
// within an image; `off` = image + 16-row-block offset, the second read is 8 rows on.
template <int DP>
struct TrAddr {
  unsigned a0, a1;
  __device__ __forceinline__ void init(int dblk, int lane) {
    const int half = lane >> 5, gg = (lane >> 4) & 1, i = lane & 15;
    const int r0 = 4 * half + (i >> 2), r1 = r0 + 8;
    const int d0 = dblk * 32 + 16 * gg + 4 * (i & 3);
    const int ch = d0 >> 3, in = (d0 & 7) * 2;
    a0 = r0 * Lay<DP>::PITCH + (Lay<DP>::pos(ch, r0) << 4) + in;
    a1 = r1 * Lay<DP>::PITCH + (Lay<DP>::pos(ch, r1) << 4) + in;
  }
};
template <int DP>
__device__ __forceinline__ void tr_issue2(const TrAddr<DP> &ta, unsigned off, s4_t &lo, s4_t &hi) {
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(ta.a0 + off));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(ta.a1 + off));
}
// The same with the image / row-block offset as the instruction's immediate: `b0`, `b1` = the two per-lane
// addresses with the image's base already added (once per tile), `imm` a constant once the loops around
// the call are unrolled.  One vector add per read less than tr_issue2.
__device__ __forceinline__ void tr_issue2o(unsigned b0, unsigned b1, int imm, s4_t &lo, s4_t &hi) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(b0), "n"(imm));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(b1), "n"(imm));
}
template <int N>
__device__ __forceinline__ bf8_t tr_wait2(s4_t &lo, s4_t &hi) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(lo), "+v"(hi) : "n"(N));
  s8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf8_t, v);
}
__device__ __forceinline__ unsigned lds_off(const void *p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

template <int DP, bool DROP>
// (head dim 96: three workgroups per CU -- 155 registers, 164 with dropout once the compiler is held to 168)
__global__ __launch_bounds__(256, DP == 96 ? 3 : 2) void flash2_fwd_k(FArgs a) {
#ifdef LDM_ATTN_TIMING
  unsigned long long t_entry; ASTAMP(t_entry);
  const unsigned long long r_entry = __builtin_amdgcn_s_memrealtime();
#endif
  using T = AT<DP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (K image, V image)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, c = lane & 31;
  const Blk blk = decode_blk(a.S, a.H);
  const int b = blk.b, head = blk.head;
  const int S = a.S, hd = a.hd;
  const int64_t D = (int64_t)a.H * hd;
  const bf16_t *qb = a.qkv + (int64_t)b * S * a.ld + (int64_t)head * hd;
  const bf16_t *kb = qb + D, *vb = qb + 2 * D;
  const int q = blk.blk * 128 + wave * 32 + c;

  TrAddr<DP> vta[T::ND];
#pragma unroll
  for (int i = 0; i < T::ND; ++i) {
    // absolute addresses into the V image of the CURRENT buffer; they move to the other buffer at the end
    // of every tile
    vta[i].init(i, lane);
    vta[i].a0 += lds_off(smem) + Lay<DP>::TILE; vta[i].a1 += lds_off(smem) + Lay<DP>::TILE;
  }
  TileDma<DP> kd, vd;
  kd.init(kb, a.ld, wave, lane);
  vd.init(vb, a.ld, wave, lane);
  kd.issue(0, S, a.ld, smem, wave);
  vd.issue(0, S, a.ld, smem + Lay<DP>::TILE, wave);

  // The workgroup's own Q rows reach their lanes through the ring's second buffer (whole rows by LDS-DMA into
  // two 64-row images beside the first K/V tile, fragments by ds_read_b128; loaded 16 bytes per lane straight
  // into the operand layout every instruction touched 32 rows).  Rows past the end re-read the last row.
  bf8_t qf[T::NK];
  // validity byte of this lane's key in the tile about to be processed: requested a whole tile ahead (asked
  // for at the top of its own tile it was a trip to L2 in front of every tile's barrier: 2.5 % of the
  // forward with the step's all-ones mask)
  unsigned key_byte = 1u;
  {
    const int row_blk = blk.blk * 128, s_rem = S - row_blk;
    char *buf1 = smem + 2 * Lay<DP>::TILE;
    TileDma<DP> qd;
    qd.init(qb + (int64_t)row_blk * a.ld, a.ld, wave, lane);
    qd.issue(0, s_rem, a.ld, buf1, wave);
    qd.issue(s_rem > 64 ? 1 : 0, s_rem, a.ld, buf1 + Lay<DP>::TILE, wave);   // (short block: the first image again)
    if (a.key_valid && lane < S) key_byte = a.key_valid[(int64_t)b * S + lane];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(key_byte) : : "memory");
    __syncthreads();
    const char *my1 = buf1 + (wave >> 1) * Lay<DP>::TILE;
#pragma unroll
    for (int kk = 0; kk < T::NK; ++kk) qf[kk] = row_frag2<DP>(my1, (wave & 1) * 32 + c, kk, half);
    // (the barrier at the top of the first tile comes before anything is written to the second buffer)
  }

  f16_t oacc[T::ND];
#pragma unroll
  for (int i = 0; i < T::ND; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;
  const unsigned drop_row = DROP ? ldm_drop_row(ldm_drop_bh(a.seed_lo, (unsigned)(b * a.H + head)), a.seed_hi, (unsigned)q) : 0u;

#ifdef LDM_ATTN_TIMING
  unsigned long long ta[6] = {0, 0, 0, 0, 0, 0}, tk0, tk1, tb = 0;
  ASTAMP(tk0);
#endif
  const int nt = (S + 63) / 64;
  for (int t = 0; t < nt; ++t) {
#ifdef LDM_ATTN_TIMING
    unsigned long long s0, s1, s2, s3, s4;
    ASTAMP(s0);
#endif
    const int k0 = t * 64;
    const char *Kt = smem + (t & 1) * 2 * Lay<DP>::TILE;   // (the V image is one image further on: vta)
    const bool kok = (k0 + lane) < S && key_byte != 0;
    const unsigned long long valid = __ballot(kok);
    // tile t has landed (this wave's pieces); after the barrier it is visible to all and every
    // wave is done with tile t-1, whose buffer the next tile's LDS-DMA then refills
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LDM_ATTN_TIMING
    { unsigned long long sw; ASTAMP(sw); ta[4] += sw - s0; }
#endif
    __syncthreads();
#ifdef LDM_ATTN_TIMING
    { unsigned long long sb; ASTAMP(sb); ta[5] += 0; tb += sb - s0; }
#endif
    // the next tile's LDS-DMA is issued one instruction at a time between the MFMAs of S: an
    // issue that stalls on a full memory queue then overlaps matrix work already in flight
    const bool more = t + 1 < nt;
    char *nb = smem + ((t + 1) & 1) * 2 * Lay<DP>::TILE;
    if (a.key_valid && k0 + 64 + lane < S) key_byte = a.key_valid[(int64_t)b * S + k0 + 64 + lane];

    ASTAMP(s1);
    f16_t sacc[2];
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kbk][r] = 0.f;
    {
      constexpr int NF = 2 * T::NK, AHEAD = 2;
      bf8_t kf[NF];
#pragma unroll
      for (int f = 0; f < AHEAD; ++f) kf[f] = row_frag2<DP>(Kt, (f / T::NK) * 32 + c, f % T::NK, half);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f + AHEAD < NF) kf[f + AHEAD] = row_frag2<DP>(Kt, ((f + AHEAD) / T::NK) * 32 + c, (f + AHEAD) % T::NK, half);
        sacc[f / T::NK] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[f], qf[f % T::NK], sacc[f / T::NK], 0, 0, 0);
        if ((f & 1) == 0 && f / 2 < 2 * Lay<DP>::PW && more) {
          constexpr int PW = Lay<DP>::PW;
          if (f / 2 < PW) kd.issue_one(f / 2, t + 1, S, a.ld, nb, wave);
          else vd.issue_one(f / 2 - PW, t + 1, S, a.ld, nb + Lay<DP>::TILE, wave);
        }
      }
    }
    ASTAMP(s2);
    if (valid != ~0ull) {
      const unsigned long long vs = valid >> (4 * half);
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (!((vs >> (kbk * 32 + acc_row(r))) & 1ull)) sacc[kbk][r] = -INFINITY;
    }
    float mx = m;
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kbk][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mu = (mx == -INFINITY) ? 0.f : mx;
    const float alpha = fast_exp2((m - mu) * LOG2E);
    m = mx;
    float ps = 0.f;
    const float mneg = -mu * LOG2E;
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(fmaf(sacc[kbk][r], LOG2E, mneg));
        sacc[kbk][r] = p;
        ps += p;
      }
    lsum = lsum * alpha + ps;
    if constexpr (DROP) {   // the row sum above is that of the undropped softmax
      const unsigned kb0 = (unsigned)(k0 + 4 * half);
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          sacc[kbk][r] = ldm_drop_keep(drop_row, kb0 + kbk * 32 + acc_row(r), a.drop_thresh) ? sacc[kbk][r] * a.drop_inv : 0.f;
    }
#pragma unroll
    for (int i = 0; i < T::ND; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    ASTAMP(s3);
    {
      constexpr int NF = 4 * T::ND;
      s4_t lo[NF], hi[NF];
      bf8_t pf[4];
      tr_issue2o(vta[0].a0, vta[0].a1, 0, lo[0], hi[0]);
      tr_issue2o(vta[1 % T::ND].a0, vta[1 % T::ND].a1, (1 / T::ND) * 16 * Lay<DP>::PITCH, lo[1], hi[1]);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float tmp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tmp[j] = sacc[g4 >> 1][8 * (g4 & 1) + j];
        pf[g4] = pack8(tmp);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f + 2 < NF) tr_issue2o(vta[(f + 2) % T::ND].a0, vta[(f + 2) % T::ND].a1, ((f + 2) / T::ND) * 16 * Lay<DP>::PITCH, lo[f + 2], hi[f + 2]);
        // reads in flight after this issue: fragments f .. min(f+2, NF-1), two each
        const bf8_t vf = (f + 2 < NF) ? tr_wait2<4>(lo[f], hi[f]) : (f + 1 < NF) ? tr_wait2<2>(lo[f], hi[f]) : tr_wait2<0>(lo[f], hi[f]);
        oacc[f % T::ND] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[f / T::ND], oacc[f % T::ND], 0, 0, 0);
      }
    }
#ifdef LDM_ATTN_TIMING
    ASTAMP(s4);
    ta[0] += s1 - s0; ta[1] += s2 - s1; ta[2] += s3 - s2; ta[3] += s4 - s3; ta[5] += 1;
#endif
    const int flip = (t & 1) ? -2 * Lay<DP>::TILE : 2 * Lay<DP>::TILE;
#pragma unroll
    for (int i = 0; i < T::ND; ++i) { vta[i].a0 += flip; vta[i].a1 += flip; }
  }
#ifdef LDM_ATTN_TIMING
  unsigned long long t_loop1; ASTAMP(t_loop1);
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // every wave is done with the tiles: the epilogue re-uses the LDS
  const float ltot = lsum + __shfl_xor(lsum, 32, 64);
  if (q < S && half == 0) a.lse[((int64_t)b * a.H + head) * S + q] = m + logf(ltot);
  {
    const int q0 = blk.blk * 128 + wave * 32;
    const float inv_l = 1.0f / ltot;
    if (a.ld_o % 8 == 0 && aligned_to_dev(a.o, 16))
      store_tile_lds<DP>(smem + wave * 32 * T::RSTR, a.o + ((int64_t)b * S + q0) * a.ld_o + (int64_t)head * hd, a.ld_o,
                         oacc, inv_l, lane, S - q0, hd);
    else if (q < S)
      store_rows<DP>(a.o + ((int64_t)b * S + q) * a.ld_o + (int64_t)head * hd, oacc, inv_l, half, hd);
    if (a.o_lo) store_rounding_residual<DP>(a, oacc, inv_l, blk, wave, lane);
  }
#ifdef LDM_ATTN_TIMING
  if (a.timing && tid == 0 && blockIdx.x == 8) {
    ASTAMP(tk1);
    for (int i = 0; i < 6; ++i) a.timing[i] = ta[i];
    a.timing[6] = tk1 - tk0;
  }
  // a workgroup from the middle of the grid: cycles entry -> loop, loop, loop -> stores issued; stores
  // drained; and its lifetime in 10-ns ticks
  if (a.timing && tid == 0 && blockIdx.x == gridDim.x / 2 + 8) {
    unsigned long long te; ASTAMP(te);
    a.timing[16] = tk0 - t_entry; a.timing[17] = t_loop1 - tk0; a.timing[18] = te - t_loop1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long td; ASTAMP(td);
    a.timing[19] = td - te;
    a.timing[20] = __builtin_amdgcn_s_memrealtime() - r_entry;
  }
  if (a.timing && lane == 0 && blockIdx.x == 8) { a.timing[8 + wave] = tb; a.timing[12 + wave] = ta[4]; }
#endif
}

// --------------------------------------------------------------------------- dQ, LDS-DMA tiles
template <int DP, bool DROP>
// (two workgroups per CU; the dropout variant at head dim 128 would spill 11 registers and keeps one)
__global__ __launch_bounds__(256, (DP == 128 && DROP) ? 1 : 2) void flash2_bwd_dq_k(FArgs a) {
#ifdef LDM_ATTN_TIMING
  unsigned long long t_entry; ASTAMP(t_entry);
  const unsigned long long r_entry = __builtin_amdgcn_s_memrealtime();
#endif
  using T = AT<DP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (K image, V image)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, c = lane & 31;
  const Blk blk = decode_blk(a.S, a.H);
  const int b = blk.b, head = blk.head;
  const int S = a.S, hd = a.hd;
  const int64_t D = (int64_t)a.H * hd;
  const bf16_t *qb = a.qkv + (int64_t)b * S * a.ld + (int64_t)head * hd;
  const bf16_t *kb = qb + D, *vb = qb + 2 * D;
  const bf16_t *dob = a.dout + (int64_t)b * S * a.ld_o + (int64_t)head * hd;
  const int q = blk.blk * 128 + wave * 32 + c;
  const int qc = q < S ? q : S - 1;

  TrAddr<DP> kta[T::ND];
#pragma unroll
  for (int i = 0; i < T::ND; ++i) {
    // absolute addresses into the K image of the CURRENT buffer; they move to the other buffer at the end
    // of every tile
    kta[i].init(i, lane);
    kta[i].a0 += lds_off(smem); kta[i].a1 += lds_off(smem);
  }
  TileDma<DP> kd, vd;
  kd.init(kb, a.ld, wave, lane);
  vd.init(vb, a.ld, wave, lane);

  // The workgroup's own rows of Q, dO and O reach their lanes through LDS: 1-KiB LDS-DMA pieces of whole rows
  // into 64-row images, fragments by ds_read_b128.  (As 16-byte loads straight into the operand layout every
  // instruction touched 32 rows 4.6 KB apart: ~700 cycles per instruction, 13k cycles before the first tile,
  // a quarter of the workgroup's lifetime at S = 512.)  Two trips, inside the ring's four images: Q into the
  // second buffer and dO into the first; then, once their fragments are in registers, O into the second
  // buffer beside the first K/V tile into the first.  The fp8 rounding residual of O and the row's
  // log-sum-exp are plain loads in the first trip.
  bf8_t qf[T::NK], dof[T::NK];
  const int64_t stat = ((int64_t)b * a.H + head) * S + qc;
  uint4 olo[T::ND];
  delta_residual_load<DP>(a, blk, wave, lane, olo);
  const float lse_row = a.lse_in[stat];
  // validity byte of this lane's key in the tile about to be processed, requested a whole tile ahead
  unsigned key_byte = 1u;
  if (a.key_valid && lane < S) key_byte = a.key_valid[(int64_t)b * S + lane];
  float dsum = 0.f;
  {
    const int row_blk = blk.blk * 128, s_rem = S - row_blk;
    char *buf1 = smem + 2 * Lay<DP>::TILE, *ext = smem;   // (dO borrows the first buffer before K/V arrive)
    TileDma<DP> qd, gd;
    qd.init(qb + (int64_t)row_blk * a.ld, a.ld, wave, lane);
    gd.init(dob + (int64_t)row_blk * a.ld_o, a.ld_o, wave, lane);
    // (a block of at most 64 rows has no second image: those rows are never used, and the image takes the
    // first one's rows again -- the descriptor's 32-bit offsets cannot step back across a whole image)
    const int t1 = s_rem > 64 ? 1 : 0;
    qd.issue(0, s_rem, a.ld, buf1, wave);
    gd.issue(0, s_rem, a.ld_o, ext, wave);
    qd.issue(t1, s_rem, a.ld, buf1 + Lay<DP>::TILE, wave);
    gd.issue(t1, s_rem, a.ld_o, ext + Lay<DP>::TILE, wave);
    const char *my1 = buf1 + (wave >> 1) * Lay<DP>::TILE, *mye = ext + (wave >> 1) * Lay<DP>::TILE;
    const int myrow = (wave & 1) * 32 + c;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < T::NK; ++kk) {
      qf[kk] = row_frag2<DP>(my1, myrow, kk, half);
      dof[kk] = row_frag2<DP>(mye, myrow, kk, half);
    }
    __syncthreads();   // every wave holds its Q and dO fragments: K/V and O may take their place
    kd.issue(0, S, a.ld, smem, wave);
    vd.issue(0, S, a.ld, smem + Lay<DP>::TILE, wave);
    TileDma<DP> od;
    od.init(a.o_in + ((int64_t)b * S + row_blk) * a.ld_o + (int64_t)head * hd, a.ld_o, wave, lane);
    od.issue(0, s_rem, a.ld_o, buf1, wave);
    od.issue(t1, s_rem, a.ld_o, buf1 + Lay<DP>::TILE, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < T::NK; ++kk) {
      const bf8_t of = row_frag2<DP>(my1, myrow, kk, half);
#pragma unroll
      for (int j = 0; j < 8; ++j) dsum = fmaf((float)of[j], (float)dof[kk][j], dsum);
    }
    // (the barrier at the top of the first tile comes before anything is written to the second buffer)
  }
  dsum += delta_residual<DP>(olo, dof);
  const float dlt = dsum + __shfl_xor(dsum, 32, 64);
  const unsigned drop_row = DROP ? ldm_drop_row(ldm_drop_bh(a.seed_lo, (unsigned)(b * a.H + head)), a.seed_hi, (unsigned)q) : 0u;
  const float nlse = -lse_row * LOG2E;
  if (half == 0 && q < S) a.delta[stat] = dlt;

  f16_t dq[T::ND];
#pragma unroll
  for (int i = 0; i < T::ND; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

#ifdef LDM_ATTN_TIMING
  unsigned long long t_loop0; ASTAMP(t_loop0);
  unsigned long long tw_vm = 0, tw_bar = 0;
#endif
  const int nt = (S + 63) / 64;
  for (int t = 0; t < nt; ++t) {
    const int k0 = t * 64;
    const char *Kt = smem + (t & 1) * 2 * Lay<DP>::TILE;
    const char *Vt = Kt + Lay<DP>::TILE;
    const bool kok = (k0 + lane) < S && key_byte != 0;
    const unsigned long long bal = __ballot(kok);
    const bool all_valid = bal == ~0ull;          // wave-uniform: the common tile has no padded key
    const unsigned long long vs = bal >> (4 * half);
#ifdef LDM_ATTN_TIMING
    unsigned long long w0, w1, w2;
    ASTAMP(w0);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LDM_ATTN_TIMING
    ASTAMP(w1);
#endif
    __syncthreads();
#ifdef LDM_ATTN_TIMING
    ASTAMP(w2);
    tw_vm += w1 - w0; tw_bar += w2 - w1;
#endif
    // the next tile's LDS-DMA goes out one instruction at a time between the first MFMAs
    const bool more = t + 1 < nt;
    char *nb = smem + ((t + 1) & 1) * 2 * Lay<DP>::TILE;
    if (a.key_valid && k0 + 64 + lane < S) key_byte = a.key_valid[(int64_t)b * S + k0 + 64 + lane];
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk) {
      f16_t sacc, dpacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
      // the K and V row fragments are requested AHK k-steps ahead of the MFMAs that take them
#ifdef LDM_DQ_AHEAD
      constexpr int AHK = LDM_DQ_AHEAD;
#else
      constexpr int AHK = 2;
#endif
      bf8_t kfr[T::NK], vfr[T::NK];
#pragma unroll
      for (int kk = 0; kk < AHK; ++kk) {
        kfr[kk] = row_frag2<DP>(Kt, kbk * 32 + c, kk, half);
        vfr[kk] = row_frag2<DP>(Vt, kbk * 32 + c, kk, half);
      }
#pragma unroll
      for (int kk = 0; kk < T::NK; ++kk) {
        if (kk + AHK < T::NK) {
          kfr[kk + AHK] = row_frag2<DP>(Kt, kbk * 32 + c, kk + AHK, half);
          vfr[kk + AHK] = row_frag2<DP>(Vt, kbk * 32 + c, kk + AHK, half);
        }
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kk], qf[kk], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[kk], dof[kk], dpacc, 0, 0, 0);
        if (kbk == 0 && kk < 2 * Lay<DP>::PW && more) {
          constexpr int PW = Lay<DP>::PW;
          if (kk < PW) kd.issue_one(kk, t + 1, S, a.ld, nb, wave);
          else vd.issue_one(kk - PW, t + 1, S, a.ld, nb + Lay<DP>::TILE, wave);
        }
      }
      constexpr int NF = 2 * T::ND;
      s4_t lo[NF], hi[NF];
      const int kbo = kbk * 32 * Lay<DP>::PITCH;   // (immediate offsets from here on)
      // all transposed K fragments of this key block are requested before the exponentials:
      // the LDS round trips overlap that VALU work instead of the MFMA chain below
      constexpr int AH = DP == 96 ? NF : 4;
#pragma unroll
      for (int f = 0; f < AH; ++f)
        tr_issue2o(kta[f % T::ND].a0, kta[f % T::ND].a1, kbo + (f / T::ND) * 16 * Lay<DP>::PITCH, lo[f], hi[f]);
      float ds[16];
      // (the validity test is three vector instructions per element: a third of the tile's vector work, so the
      // tile without a padded key -- every tile of an unpadded batch -- takes a branch around it)
      if (all_valid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = fast_exp2(fmaf(sacc[r], LOG2E, nlse));
          if constexpr (DROP)
            ds[r] = p * ((ldm_drop_keep(drop_row, (unsigned)(k0 + 4 * half + kbk * 32 + acc_row(r)), a.drop_thresh) ? dpacc[r] * a.drop_inv : 0.f) - dlt);
          else
            ds[r] = p * (dpacc[r] - dlt);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool ok = (vs >> (kbk * 32 + acc_row(r))) & 1ull;
          const float p = ok ? fast_exp2(fmaf(sacc[r], LOG2E, nlse)) : 0.f;
          if constexpr (DROP)
            ds[r] = p * ((ldm_drop_keep(drop_row, (unsigned)(k0 + 4 * half + kbk * 32 + acc_row(r)), a.drop_thresh) ? dpacc[r] * a.drop_inv : 0.f) - dlt);
          else
            ds[r] = p * (dpacc[r] - dlt);
        }
      }
      const bf8_t dsf[2] = {pack8(ds), pack8(ds + 8)};
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f + AH < NF) tr_issue2o(kta[(f + AH) % T::ND].a0, kta[(f + AH) % T::ND].a1, kbo + ((f + AH) / T::ND) * 16 * Lay<DP>::PITCH, lo[f + AH], hi[f + AH]);
        // reads in flight now: fragments f .. min(f + AH, NF - 1), two each
        constexpr int LAST = NF - 1;
        const int ahead = (f + AH < NF ? f + AH : LAST) - f;
        const bf8_t kf = ahead >= 7 ? tr_wait2<14>(lo[f], hi[f]) : ahead == 6 ? tr_wait2<12>(lo[f], hi[f])
                       : ahead == 5 ? tr_wait2<10>(lo[f], hi[f]) : ahead == 4 ? tr_wait2<8>(lo[f], hi[f])
                       : ahead == 3 ? tr_wait2<6>(lo[f], hi[f]) : ahead == 2 ? tr_wait2<4>(lo[f], hi[f])
                       : ahead == 1 ? tr_wait2<2>(lo[f], hi[f]) : tr_wait2<0>(lo[f], hi[f]);
        dq[f % T::ND] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, dsf[f / T::ND], dq[f % T::ND], 0, 0, 0);
      }
    }
    const int flip = (t & 1) ? -2 * Lay<DP>::TILE : 2 * Lay<DP>::TILE;
#pragma unroll
    for (int i = 0; i < T::ND; ++i) { kta[i].a0 += flip; kta[i].a1 += flip; }
  }
#ifdef LDM_ATTN_TIMING
  unsigned long long t_loop1; ASTAMP(t_loop1);
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    const int q0 = blk.blk * 128 + wave * 32;
    if (a.rope_cos) unrope_acc<DP>(dq, a.rope_scale, half, a.rope_cos + (int64_t)qc * (DP / 2), a.rope_sin + (int64_t)qc * (DP / 2));
    if (aligned_to_dev(a.dqkv, 16))
      store_tile_lds<DP>(smem + wave * 32 * T::RSTR, a.dqkv + ((int64_t)b * S + q0) * a.ld + (int64_t)head * hd, a.ld, dq,
                         1.0f, lane, S - q0, hd);
    else if (q < S)
      store_rows<DP>(a.dqkv + ((int64_t)b * S + q) * a.ld + (int64_t)head * hd, dq, 1.0f, half, hd);
  }
#ifdef LDM_ATTN_TIMING
  if (a.timing && threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 + 8) {
    unsigned long long te; ASTAMP(te);
    a.timing[29] = tw_vm; a.timing[30] = tw_bar;
    a.timing[24] = t_loop0 - t_entry; a.timing[25] = t_loop1 - t_loop0; a.timing[26] = te - t_loop1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long td; ASTAMP(td);
    a.timing[27] = td - te;
    a.timing[28] = __builtin_amdgcn_s_memrealtime() - r_entry;
  }
#endif
}

// --------------------------------------------------------------------------- dK, dV, LDS-DMA tiles
// Q and dO tiles and the two per-row statistics (lse, delta: 64 floats each, one 4-byte LDS-DMA
// instruction of wave 0 / wave 1) land in one of two buffers; one barrier per tile.
template <int DP, bool DROP>
__global__ __launch_bounds__(256, DP == 128 ? 1 : 2) void flash2_bwd_dkv_k(FArgs a) {
#ifdef LDM_ATTN_TIMING
  unsigned long long t_entry; ASTAMP(t_entry);
  const unsigned long long r_entry = __builtin_amdgcn_s_memrealtime();
#endif
  using T = AT<DP>;
  constexpr int BUF = 2 * Lay<DP>::TILE + 512;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (Q image, dO image, lse[64], delta[64]), 2 images for the prologue
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, c = lane & 31;
  const Blk blk = decode_blk(a.S, a.H);
  const int b = blk.b, head = blk.head;
  const int S = a.S, hd = a.hd;
  const int64_t D = (int64_t)a.H * hd;
  const bf16_t *qb = a.qkv + (int64_t)b * S * a.ld + (int64_t)head * hd;
  const bf16_t *kb = qb + D, *vb = qb + 2 * D;
  const bf16_t *dob = a.dout + (int64_t)b * S * a.ld_o + (int64_t)head * hd;
  const int key = blk.blk * 128 + wave * 32 + c;
  const unsigned drop_bh = DROP ? ldm_drop_bh(a.seed_lo, (unsigned)(b * a.H + head)) : 0u;
  const int64_t stat0 = ((int64_t)b * a.H + head) * S;

  TrAddr<DP> ta[T::ND];
#pragma unroll
  for (int i = 0; i < T::ND; ++i) {
    // absolute addresses into the Q image of the CURRENT buffer (the dO image is one image further on);
    // they move to the other buffer at the end of every tile
    ta[i].init(i, lane);
    ta[i].a0 += lds_off(smem); ta[i].a1 += lds_off(smem);
  }
  TileDmaP<DP> qd, gd;
  qd.init(qb, a.ld, wave, lane);
  gd.init(dob, a.ld_o, wave, lane);
  auto issue_stats = [&](int t, char *buf) {
    if (wave < 2) {   // rows past the end re-read the last row; their probabilities are zeroed below
      const int qq = t * 64 + lane;
      const float *src = (wave == 0 ? a.lse_in : a.delta) + stat0 + (qq < S ? qq : S - 1);
      __builtin_amdgcn_global_load_lds(GLB_PTR2(src), LDS_PTR2(buf + 2 * Lay<DP>::TILE + wave * 256), 4, 0, 0);
    }
  };
  qd.issue(0, S, a.ld, smem, wave);
  gd.issue(0, S, a.ld_o, smem + Lay<DP>::TILE, wave);
  issue_stats(0, smem);

  // The workgroup's own K and V rows reach their lanes through LDS (whole rows by LDS-DMA into 64-row images,
  // fragments by ds_read_b128; loaded 16 bytes per lane straight into the operand layout every instruction
  // touched 32 rows): K into the second buffer, V into the two images behind the buffers, beside the first
  // Q/dO tile.  Rows past the end re-read the last row; their keys are masked below.
  bf8_t kf[T::NK], vf[T::NK];
  {
    const int row_blk = blk.blk * 128, s_rem = S - row_blk;
    char *buf1 = smem + BUF, *ext = smem + 2 * BUF;
    TileDma<DP> kd2, vd2;
    kd2.init(kb + (int64_t)row_blk * a.ld, a.ld, wave, lane);
    vd2.init(vb + (int64_t)row_blk * a.ld, a.ld, wave, lane);
    const int t1 = s_rem > 64 ? 1 : 0;   // (a block of at most 64 rows: the second image repeats the first)
    kd2.issue(0, s_rem, a.ld, buf1, wave);
    vd2.issue(0, s_rem, a.ld, ext, wave);
    kd2.issue(t1, s_rem, a.ld, buf1 + Lay<DP>::TILE, wave);
    vd2.issue(t1, s_rem, a.ld, ext + Lay<DP>::TILE, wave);
  }
  // the key's validity byte travels with the loads above (asked for first and compared at once, it was a
  // trip to memory of its own before anything else had been requested)
  unsigned key_byte = 1u;
  if (key < S && a.key_valid) key_byte = a.key_valid[(int64_t)b * S + key];
  {
    const char *myk = smem + BUF + (wave >> 1) * Lay<DP>::TILE, *myv = smem + 2 * BUF + (wave >> 1) * Lay<DP>::TILE;
    const int myrow = (wave & 1) * 32 + c;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(key_byte) : : "memory");
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < T::NK; ++kk) {
      kf[kk] = row_frag2<DP>(myk, myrow, kk, half);
      vf[kk] = row_frag2<DP>(myv, myrow, kk, half);
    }
    // (the barrier at the top of the first tile comes before anything is written to the second buffer)
  }
  const bool key_ok = key < S && key_byte != 0;
  // a padded key starts its score accumulator at -infinity (a select on the 16 initial values, where the
  // negation of the log-sum-exp was anyway): its probabilities come out of the exponential as zeros, with
  // no select per element afterwards.  Not by adding a bias of -infinity: a fully padded sequence has a
  // log-sum-exp of -infinity, and the sum would be a NaN.
  // (the dropout variant sits at the register limit and keeps the select per element)
  const bool key_init_ok = DROP || key_ok;
  f16_t dk[T::ND], dv[T::ND];
#pragma unroll
  for (int i = 0; i < T::ND; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }

#ifdef LDM_ATTN_TIMING
  unsigned long long t_loop0; ASTAMP(t_loop0);
  unsigned long long tw_vm = 0, tw_bar = 0;
#endif
  const int nt = (S + 63) / 64;
  for (int t = 0; t < nt; ++t) {
    const char *Qt = smem + (t & 1) * BUF;
    const char *Gt = Qt + Lay<DP>::TILE;
    const float *lse_s = reinterpret_cast<const float *>(Qt + 2 * Lay<DP>::TILE);
    // (delta follows lse: 64 floats on)
    const int lim = S - t * 64;   // query rows of this tile that exist
#ifdef LDM_ATTN_TIMING
    unsigned long long w0, w1, w2;
    ASTAMP(w0);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LDM_ATTN_TIMING
    ASTAMP(w1);
#endif
    __syncthreads();
#ifdef LDM_ATTN_TIMING
    ASTAMP(w2);
    tw_vm += w1 - w0; tw_bar += w2 - w1;
#endif
    // next tile: the statistics now, the images one LDS-DMA instruction at a time between the first MFMAs
    const bool more = t + 1 < nt;
    char *nb = smem + ((t + 1) & 1) * BUF;
    if (more) issue_stats(t + 1, nb);
    [[maybe_unused]] const unsigned row_keys = DROP ? ldm_drop_row(drop_bh, a.seed_hi, (unsigned)(t * 64 + lane)) : 0u;
#pragma unroll
    for (int qbk = 0; qbk < 2; ++qbk) {
      // the score accumulator starts at -lse and the dP accumulator at -delta: the products then leave
      // s - lse and dP - delta, with no register copy of the statistics beside them
      // (read by asm: behind a read it can see, the compiler drains every LDS-DMA in flight, that is
      // the next tile's statistics just requested and, before the second query block, its images too)
      f16_t s0, dpacc;
      [[maybe_unused]] float nd[16];
      typedef float f4v_t __attribute__((ext_vector_type(4)));
      f4v_t lq[4], dq4[4];
      {
        const unsigned so = lds_off(lse_s) + (qbk * 32 + 4 * half) * 4;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(lq[r4]) : "v"(so), "n"(32 * r4));
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dq4[r4]) : "v"(so), "n"(256 + 32 * r4));
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(lq[0]), "+v"(lq[1]), "+v"(lq[2]), "+v"(lq[3]), "+v"(dq4[0]), "+v"(dq4[1]), "+v"(dq4[2]), "+v"(dq4[3]));
      }
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float4 l4 = make_float4(lq[r4][0], lq[r4][1], lq[r4][2], lq[r4][3]);
        const float4 d4 = make_float4(dq4[r4][0], dq4[r4][1], dq4[r4][2], dq4[r4][3]);
        s0[4 * r4 + 0] = key_init_ok ? -l4.x : -INFINITY; s0[4 * r4 + 1] = key_init_ok ? -l4.y : -INFINITY;
        s0[4 * r4 + 2] = key_init_ok ? -l4.z : -INFINITY; s0[4 * r4 + 3] = key_init_ok ? -l4.w : -INFINITY;
        if constexpr (DROP) {   // the mask applies to dP alone: -delta is added after the products
          nd[4 * r4 + 0] = -d4.x; nd[4 * r4 + 1] = -d4.y; nd[4 * r4 + 2] = -d4.z; nd[4 * r4 + 3] = -d4.w;
          dpacc[4 * r4 + 0] = 0.f; dpacc[4 * r4 + 1] = 0.f; dpacc[4 * r4 + 2] = 0.f; dpacc[4 * r4 + 3] = 0.f;
        } else {
          dpacc[4 * r4 + 0] = -d4.x; dpacc[4 * r4 + 1] = -d4.y; dpacc[4 * r4 + 2] = -d4.z; dpacc[4 * r4 + 3] = -d4.w;
        }
      }
      // the Q and dO row fragments are requested AHQ k-steps ahead of the MFMAs that take them (head dim 96
      // runs two waves per SIMD on 256 registers: one step fits without spilling, two do not)
#ifdef LDM_DKV_AHEAD
      constexpr int AHQ = LDM_DKV_AHEAD;
#else
      constexpr int AHQ = DP == 128 ? 2 : 1;
#endif
      bf8_t qfr[T::NK], gfr[T::NK];
#pragma unroll
      for (int kk = 0; kk < AHQ; ++kk) {
        qfr[kk] = row_frag2<DP>(Qt, qbk * 32 + c, kk, half);
        gfr[kk] = row_frag2<DP>(Gt, qbk * 32 + c, kk, half);
      }
#pragma unroll
      for (int kk = 0; kk < T::NK; ++kk) {
        if (kk + AHQ < T::NK) {
          qfr[kk + AHQ] = row_frag2<DP>(Qt, qbk * 32 + c, kk + AHQ, half);
          gfr[kk + AHQ] = row_frag2<DP>(Gt, qbk * 32 + c, kk + AHQ, half);
        }
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qfr[kk], kf[kk], s0, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfr[kk], vf[kk], dpacc, 0, 0, 0);
        if (qbk == 0 && kk < 2 * Lay<DP>::PW && more) {
          constexpr int PW = Lay<DP>::PW;
          if (kk < PW) qd.issue_one(kk, t + 1, S, a.ld, nb, wave);
          else gd.issue_one(kk - PW, t + 1, S, a.ld_o, nb + Lay<DP>::TILE, wave);
        }
      }
      constexpr int NF = 2 * T::ND;
      s4_t glo[NF], ghi[NF], qlo[NF], qhi[NF];
      const int qbo = qbk * 32 * Lay<DP>::PITCH, gbo = Lay<DP>::TILE + qbo;   // immediates
#ifdef LDM_DKV_AHEAD2
      constexpr int AH2 = LDM_DKV_AHEAD2;
#else
      constexpr int AH2 = 1;
#endif
#pragma unroll
      for (int f = 0; f < AH2; ++f) {
        const int ro = (f / T::ND) * 16 * Lay<DP>::PITCH;
        tr_issue2o(ta[f % T::ND].a0, ta[f % T::ND].a1, gbo + ro, glo[f], ghi[f]);
        tr_issue2o(ta[f % T::ND].a0, ta[f % T::ND].a1, qbo + ro, qlo[f], qhi[f]);
      }
      float p[16], ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = (!DROP || key_ok) ? fast_exp2(s0[r] * LOG2E) : 0.f;
        if constexpr (DROP) {
          // row key of query row t*64 + qbk*32 + acc_row(r) + 4*half: lane L of `row_keys` holds that of row
          // t*64 + L (one hash round per lane and tile instead of one per element here)
          const unsigned rk_lo = __builtin_amdgcn_readlane(row_keys, qbk * 32 + acc_row(r));
          const unsigned rk_hi = __builtin_amdgcn_readlane(row_keys, qbk * 32 + acc_row(r) + 4);
          const bool keep = ldm_drop_keep(half ? rk_hi : rk_lo, (unsigned)key, a.drop_thresh);
          p[r] = keep ? pr * a.drop_inv : 0.f;
          ds[r] = pr * ((keep ? dpacc[r] * a.drop_inv : 0.f) + nd[r]);
        } else {
          p[r] = pr;
          ds[r] = pr * dpacc[r];
        }
      }
      if (lim < 64) {   // workgroup-uniform: the last, partial tile
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (qbk * 32 + acc_row(r) + 4 * half >= lim) { p[r] = 0.f; ds[r] = 0.f; }
      }
      const bf8_t pf[2] = {pack8(p), pack8(p + 8)}, dsf[2] = {pack8(ds), pack8(ds + 8)};
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f + AH2 < NF) {
          const int ro = ((f + AH2) / T::ND) * 16 * Lay<DP>::PITCH;
          tr_issue2o(ta[(f + AH2) % T::ND].a0, ta[(f + AH2) % T::ND].a1, gbo + ro, glo[f + AH2], ghi[f + AH2]);
          tr_issue2o(ta[(f + AH2) % T::ND].a0, ta[(f + AH2) % T::ND].a1, qbo + ro, qlo[f + AH2], qhi[f + AH2]);
        }
        // in flight after the issue: fragment pairs f .. min(f + AH2, NF - 1), four reads each (dO pair first)
        const int yg = (NF - 1 - f) < AH2 ? (NF - 1 - f) : AH2;
        static_assert(AH2 >= 1 && AH2 <= 3, "lgkmcnt is a 4-bit counter");
        const bf8_t gf = yg == 3 ? tr_wait2<14>(glo[f], ghi[f]) : yg == 2 ? tr_wait2<10>(glo[f], ghi[f])
                       : yg == 1 ? tr_wait2<6>(glo[f], ghi[f]) : tr_wait2<2>(glo[f], ghi[f]);
        dv[f % T::ND] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf, pf[f / T::ND], dv[f % T::ND], 0, 0, 0);
        const bf8_t qf2 = yg == 3 ? tr_wait2<12>(qlo[f], qhi[f]) : yg == 2 ? tr_wait2<8>(qlo[f], qhi[f])
                        : yg == 1 ? tr_wait2<4>(qlo[f], qhi[f]) : tr_wait2<0>(qlo[f], qhi[f]);
        dk[f % T::ND] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf2, dsf[f / T::ND], dk[f % T::ND], 0, 0, 0);
      }
    }
    const int flip = (t & 1) ? -BUF : BUF;
#pragma unroll
    for (int i = 0; i < T::ND; ++i) { ta[i].a0 += flip; ta[i].a1 += flip; }
  }
#ifdef LDM_ATTN_TIMING
  unsigned long long t_loop1; ASTAMP(t_loop1);
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    const int k0w = blk.blk * 128 + wave * 32;
    if (a.rope_cos) {
      // (the tables' rows straight from memory: staging them through LDS as the prologues do measured 0.7 % slower)
      const int keyc = key < S ? key : S - 1;
      unrope_acc<DP>(dk, 1.0f, half, a.rope_cos + (int64_t)keyc * (DP / 2), a.rope_sin + (int64_t)keyc * (DP / 2));
    }
    if (aligned_to_dev(a.dqkv, 16)) {
      char *wl = smem + wave * 32 * T::RSTR;
      bf16_t *row0 = a.dqkv + ((int64_t)b * S + k0w) * a.ld + (int64_t)head * hd;
      store_tile_lds<DP>(wl, row0 + D, a.ld, dk, 1.0f, lane, S - k0w, hd);
      store_tile_lds<DP>(wl, row0 + 2 * D, a.ld, dv, 1.0f, lane, S - k0w, hd);
    } else if (key < S) {
      bf16_t *row = a.dqkv + ((int64_t)b * S + key) * a.ld + (int64_t)head * hd;
      store_rows<DP>(row + D, dk, 1.0f, half, hd);
      store_rows<DP>(row + 2 * D, dv, 1.0f, half, hd);
    }
  }
#ifdef LDM_ATTN_TIMING
  if (a.timing && threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 + 8) {
    unsigned long long te; ASTAMP(te);
    a.timing[37] = tw_vm; a.timing[38] = tw_bar;
    a.timing[32] = t_loop0 - t_entry; a.timing[33] = t_loop1 - t_loop0; a.timing[34] = te - t_loop1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long td; ASTAMP(td);
    a.timing[35] = td - te;
    a.timing[36] = __builtin_amdgcn_s_memrealtime() - r_entry;
  }
#endif
}

// --------------------------------------------------------------------------- dQ
template <int DP, bool DROP>
__global__ __launch_bounds__(256, 2) void flash_bwd_dq_k(FArgs a) {
  using T = AT<DP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *Kt = smem;
  char *Vt = smem + 64 * T::RSTR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, c = lane & 31;
  const Blk blk = decode_blk(a.S, a.H);
  const int b = blk.b, head = blk.head;
  const int S = a.S, hd = a.hd;
  const int64_t D = (int64_t)a.H * hd;
  const bf16_t *qb = a.qkv + (int64_t)b * S * a.ld + (int64_t)head * hd;
  const bf16_t *kb = qb + D, *vb = qb + 2 * D;
  const bf16_t *dob = a.dout + (int64_t)b * S * a.ld_o + (int64_t)head * hd;
  const int q = blk.blk * 128 + wave * 32 + c;
  const int qc = q < S ? q : S - 1;

  bf8_t qf[T::NK], dof[T::NK];
  // delta = rowsum(dO * O): each query row is split over the two half-lanes, so the row sum
  // is one cross-half exchange; it is also published for the dK/dV kernel that runs next.
  const bf16_t *ob = a.o_in + (int64_t)b * S * a.ld_o + (int64_t)head * hd;
  // every prologue load is requested before the first of them is used: one trip to memory for Q, dO, O, the
  // rounding residual of O and the row's log-sum-exp (requested one group after the other, each group behind
  // the use of the one before, the prologue was three trips: a quarter of the workgroup's lifetime at S = 512)
  const int64_t stat = ((int64_t)b * a.H + head) * S + qc;
  uint4 olo[T::ND];
  delta_residual_load<DP>(a, blk, wave, lane, olo);
  const float lse_row = a.lse_in[stat];
  bf8_t ofr[T::NK];
#pragma unroll
  for (int kk = 0; kk < T::NK; ++kk) {
    qf[kk] = global_frag(qb, a.ld, qc, true, kk, half, hd);
    dof[kk] = global_frag(dob, a.ld_o, qc, true, kk, half, hd);
    ofr[kk] = global_frag(ob, a.ld_o, qc, true, kk, half, hd);
  }
  float dsum = 0.f;
#pragma unroll
  for (int kk = 0; kk < T::NK; ++kk)
#pragma unroll
    for (int j = 0; j < 8; ++j) dsum = fmaf((float)ofr[kk][j], (float)dof[kk][j], dsum);
  dsum += delta_residual<DP>(olo, dof);
  const float dlt = dsum + __shfl_xor(dsum, 32, 64);
  const unsigned drop_row = DROP ? ldm_drop_row(ldm_drop_bh(a.seed_lo, (unsigned)(b * a.H + head)), a.seed_hi, (unsigned)q) : 0u;
  const float nlse = -lse_row * LOG2E;
  if (half == 0 && q < S) a.delta[stat] = dlt;

  f16_t dq[T::ND];
#pragma unroll
  for (int i = 0; i < T::ND; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;

  const int nt = (S + 63) / 64;
  uint4 kr[T::NCH], vr[T::NCH];
  tile_load<DP>(kb, a.ld, 0, S, hd, tid, kr);
  tile_load<DP>(vb, a.ld, 0, S, hd, tid, vr);
  tile_store<DP, T::RSTR>(Kt, tid, kr);
  tile_store<DP, T::RSTR>(Vt, tid, vr);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int k0 = t * 64;
    if (t + 1 < nt) {
      tile_load<DP>(kb, a.ld, k0 + 64, S, hd, tid, kr);
      tile_load<DP>(vb, a.ld, k0 + 64, S, hd, tid, vr);
    }
    bool kok = (k0 + lane) < S;
    if (kok && a.key_valid) kok = a.key_valid[(int64_t)b * S + k0 + lane] != 0;
    const unsigned long long vs = __ballot(kok) >> (4 * half);

#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk) {
      f16_t sacc, dpacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < T::NK; ++kk) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<T::RSTR>(Kt, kbk * 32 + c, kk, half), qf[kk], sacc, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<T::RSTR>(Vt, kbk * 32 + c, kk, half), dof[kk], dpacc, 0, 0, 0);
      }
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = (vs >> (kbk * 32 + acc_row(r))) & 1ull;
        const float p = ok ? fast_exp2(fmaf(sacc[r], LOG2E, nlse)) : 0.f;
        if constexpr (DROP)
          ds[r] = p * ((ldm_drop_keep(drop_row, (unsigned)(k0 + 4 * half + kbk * 32 + acc_row(r)), a.drop_thresh) ? dpacc[r] * a.drop_inv : 0.f) - dlt);
        else
          ds[r] = p * (dpacc[r] - dlt);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf8_t dsf = pack8(ds + 8 * s);
#pragma unroll
        for (int db = 0; db < T::ND; ++db)
          dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<T::RSTR>(Kt, kbk * 32 + 16 * s, db, lane), dsf,
                                                           dq[db], 0, 0, 0);
      }
    }
    __syncthreads();
    if (t + 1 < nt) {
      tile_store<DP, T::RSTR>(Kt, tid, kr);
      tile_store<DP, T::RSTR>(Vt, tid, vr);
      __syncthreads();
    }
  }
  {
    const int q0 = blk.blk * 128 + wave * 32;
    if (a.rope_cos) unrope_acc<DP>(dq, a.rope_scale, half, a.rope_cos + (int64_t)qc * (DP / 2), a.rope_sin + (int64_t)qc * (DP / 2));
    if (aligned_to_dev(a.dqkv, 16))
      store_tile_lds<DP>(smem + wave * 32 * T::RSTR, a.dqkv + ((int64_t)b * S + q0) * a.ld + (int64_t)head * hd, a.ld, dq,
                         1.0f, lane, S - q0, hd);
    else if (q < S)
      store_rows<DP>(a.dqkv + ((int64_t)b * S + q) * a.ld + (int64_t)head * hd, dq, 1.0f, half, hd);
  }
}

// --------------------------------------------------------------------------- dK, dV
template <int DP, bool DROP>
__global__ __launch_bounds__(256, 2) void flash_bwd_dkv_k(FArgs a) {
  using T = AT<DP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *Qt = smem;
  char *Gt = smem + 64 * T::RSTR;                                   // dO tile
  float *lse_s = reinterpret_cast<float *>(smem + 128 * T::RSTR);   // [64] -lse*log2e (or -inf)
  float *dlt_s = lse_s + 64;                                         // [64] -delta
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, c = lane & 31;
  const Blk blk = decode_blk(a.S, a.H);
  const int b = blk.b, head = blk.head;
  const int S = a.S, hd = a.hd;
  const int64_t D = (int64_t)a.H * hd;
  const bf16_t *qb = a.qkv + (int64_t)b * S * a.ld + (int64_t)head * hd;
  const bf16_t *kb = qb + D, *vb = qb + 2 * D;
  const bf16_t *dob = a.dout + (int64_t)b * S * a.ld_o + (int64_t)head * hd;
  const int key = blk.blk * 128 + wave * 32 + c;
  bool key_ok = key < S;
  if (key_ok && a.key_valid) key_ok = a.key_valid[(int64_t)b * S + key] != 0;
  const int keyc = key < S ? key : S - 1;
  const unsigned drop_bh = DROP ? ldm_drop_bh(a.seed_lo, (unsigned)(b * a.H + head)) : 0u;

  bf8_t kf[T::NK], vf[T::NK];
#pragma unroll
  for (int kk = 0; kk < T::NK; ++kk) {
    kf[kk] = global_frag(kb, a.ld, keyc, key < S, kk, half, hd);
    vf[kk] = global_frag(vb, a.ld, keyc, key < S, kk, half, hd);
  }
  f16_t dk[T::ND], dv[T::ND];
#pragma unroll
  for (int i = 0; i < T::ND; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[i][r] = 0.f; dv[i][r] = 0.f; }

  const int64_t stat0 = ((int64_t)b * a.H + head) * S;
  const int nt = (S + 63) / 64;
  uint4 qr[T::NCH], gr[T::NCH];
  float lse_r = 0.f, dlt_r = 0.f;
  auto load_stats = [&](int q0) {
    if (tid < 64) {
      const int qq = q0 + tid;
      lse_r = qq < S ? -a.lse_in[stat0 + qq] * LOG2E : -INFINITY;
      dlt_r = qq < S ? -a.delta[stat0 + qq] : 0.f;
    }
  };
  auto store_stats = [&]() {
    if (tid < 64) { lse_s[tid] = lse_r; dlt_s[tid] = dlt_r; }
  };
  tile_load<DP>(qb, a.ld, 0, S, hd, tid, qr);
  tile_load<DP>(dob, a.ld_o, 0, S, hd, tid, gr);
  load_stats(0);
  tile_store<DP, T::RSTR>(Qt, tid, qr);
  tile_store<DP, T::RSTR>(Gt, tid, gr);
  store_stats();
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) {
      tile_load<DP>(qb, a.ld, (t + 1) * 64, S, hd, tid, qr);
      tile_load<DP>(dob, a.ld_o, (t + 1) * 64, S, hd, tid, gr);
      load_stats((t + 1) * 64);
    }
#pragma unroll
    for (int qbk = 0; qbk < 2; ++qbk) {
      f16_t sacc, dpacc;
      [[maybe_unused]] float nd[16];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float4 l4 = *reinterpret_cast<const float4 *>(lse_s + qbk * 32 + 8 * r4 + 4 * half);
        const float4 d4 = *reinterpret_cast<const float4 *>(dlt_s + qbk * 32 + 8 * r4 + 4 * half);
        sacc[4 * r4 + 0] = l4.x; sacc[4 * r4 + 1] = l4.y; sacc[4 * r4 + 2] = l4.z; sacc[4 * r4 + 3] = l4.w;
        if constexpr (DROP) {   // the mask applies to dP alone: -delta is added after the products
          nd[4 * r4 + 0] = d4.x; nd[4 * r4 + 1] = d4.y; nd[4 * r4 + 2] = d4.z; nd[4 * r4 + 3] = d4.w;
          dpacc[4 * r4 + 0] = 0.f; dpacc[4 * r4 + 1] = 0.f; dpacc[4 * r4 + 2] = 0.f; dpacc[4 * r4 + 3] = 0.f;
        } else {
          dpacc[4 * r4 + 0] = d4.x; dpacc[4 * r4 + 1] = d4.y; dpacc[4 * r4 + 2] = d4.z; dpacc[4 * r4 + 3] = d4.w;
        }
      }
      // sacc holds the row constants -lse*log2e; dpacc starts at -delta so that the
      // MFMA chain leaves dP - delta.
      f16_t s0;
#pragma unroll
      for (int r = 0; r < 16; ++r) s0[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < T::NK; ++kk) {
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<T::RSTR>(Qt, qbk * 32 + c, kk, half), kf[kk], s0, 0, 0, 0);
        dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<T::RSTR>(Gt, qbk * 32 + c, kk, half), vf[kk], dpacc, 0, 0, 0);
      }
      float p[16], ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pr = key_ok ? fast_exp2(fmaf(s0[r], LOG2E, sacc[r])) : 0.f;
        if constexpr (DROP) {
          const unsigned qi = (unsigned)(t * 64 + qbk * 32 + acc_row(r) + 4 * half);
          const bool keep = ldm_drop_keep(ldm_drop_row(drop_bh, a.seed_hi, qi), (unsigned)key, a.drop_thresh);
          p[r] = keep ? pr * a.drop_inv : 0.f;
          ds[r] = pr * ((keep ? dpacc[r] * a.drop_inv : 0.f) + nd[r]);
        } else {
          p[r] = pr;
          ds[r] = pr * dpacc[r];
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf8_t pf = pack8(p + 8 * s), dsf = pack8(ds + 8 * s);
#pragma unroll
        for (int db = 0; db < T::ND; ++db) {
          dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<T::RSTR>(Gt, qbk * 32 + 16 * s, db, lane), pf, dv[db], 0, 0, 0);
          dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<T::RSTR>(Qt, qbk * 32 + 16 * s, db, lane), dsf, dk[db], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (t + 1 < nt) {
      tile_store<DP, T::RSTR>(Qt, tid, qr);
      tile_store<DP, T::RSTR>(Gt, tid, gr);
      store_stats();
      __syncthreads();
    }
  }
  {
    const int k0w = blk.blk * 128 + wave * 32;
    if (a.rope_cos) unrope_acc<DP>(dk, 1.0f, half, a.rope_cos + (int64_t)keyc * (DP / 2), a.rope_sin + (int64_t)keyc * (DP / 2));
    if (aligned_to_dev(a.dqkv, 16)) {
      char *wl = smem + wave * 32 * T::RSTR;
      bf16_t *row0 = a.dqkv + ((int64_t)b * S + k0w) * a.ld + (int64_t)head * hd;
      store_tile_lds<DP>(wl, row0 + D, a.ld, dk, 1.0f, lane, S - k0w, hd);
      store_tile_lds<DP>(wl, row0 + 2 * D, a.ld, dv, 1.0f, lane, S - k0w, hd);
    } else if (key < S) {
      bf16_t *row = a.dqkv + ((int64_t)b * S + key) * a.ld + (int64_t)head * hd;
      store_rows<DP>(row + D, dk, 1.0f, half, hd);
      store_rows<DP>(row + 2 * D, dv, 1.0f, half, hd);
    }
  }
}

template <int DP, bool DROP>
int flash_fwd_launch_t(const FArgs &a, hipStream_t s) {
  size_t lds = 64 * AT<DP>::RSTR + 64 * AT<DP>::TSTR;
  if (lds < (size_t)128 * AT<DP>::RSTR) lds = (size_t)128 * AT<DP>::RSTR;   // the epilogue stages O through LDS
  dim3 grid((unsigned)(ceil_div(a.S, 128) * a.H * a.B));
  hipLaunchKernelGGL((flash_fwd_k<DP, DROP>), grid, dim3(256), lds, s, a);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

template <int DP>
int flash_fwd_launch(const FArgs &a, hipStream_t s) {
  return a.drop_thresh ? flash_fwd_launch_t<DP, true>(a, s) : flash_fwd_launch_t<DP, false>(a, s);
}

template <int DP, bool DROP>
int flash2_fwd_launch_t(const FArgs &a, hipStream_t s) {
  constexpr int lds = 4 * Lay<DP>::TILE;
  static std::atomic<uint64_t> attr_done{0};
  uint64_t dev_bit;
  if (ldm_attr_pending(attr_done, &dev_bit)) {
    (void)hipFuncSetAttribute((const void *)flash2_fwd_k<DP, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done.fetch_or(dev_bit, std::memory_order_release);
  }
  dim3 grid((unsigned)(ceil_div(a.S, 128) * a.H * a.B));
  hipLaunchKernelGGL((flash2_fwd_k<DP, DROP>), grid, dim3(256), lds, s, a);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

template <int DP>
int flash2_fwd_launch(const FArgs &a, hipStream_t s) {
  return a.drop_thresh ? flash2_fwd_launch_t<DP, true>(a, s) : flash2_fwd_launch_t<DP, false>(a, s);
}

// LDM_ATTN_DMA=0 pins the register-staged kernels (A/B measurements)
bool attn_dma_enabled() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("LDM_ATTN_DMA");
    v = (e && atoi(e) == 0) ? 0 : 1;
  }
  return v == 1;
}

template <int DP, bool DROP>
int flash2_bwd_launch_t(const FArgs &a, hipStream_t s) {
  constexpr int lds_dq = 4 * Lay<DP>::TILE, lds_dkv = 2 * (2 * Lay<DP>::TILE + 512) + 2 * Lay<DP>::TILE;
  static std::atomic<uint64_t> attr_done{0};
  uint64_t dev_bit;
  if (ldm_attr_pending(attr_done, &dev_bit)) {
    (void)hipFuncSetAttribute((const void *)flash2_bwd_dq_k<DP, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_dq);
    (void)hipFuncSetAttribute((const void *)flash2_bwd_dkv_k<DP, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_dkv);
    attr_done.fetch_or(dev_bit, std::memory_order_release);
  }
  dim3 grid((unsigned)(ceil_div(a.S, 128) * a.H * a.B));
  hipLaunchKernelGGL((flash2_bwd_dq_k<DP, DROP>), grid, dim3(256), lds_dq, s, a);
  LDM_CHECK_LAUNCH();
  hipLaunchKernelGGL((flash2_bwd_dkv_k<DP, DROP>), grid, dim3(256), lds_dkv, s, a);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

template <int DP>
int flash2_bwd_launch(const FArgs &a, hipStream_t s) {
  return a.drop_thresh ? flash2_bwd_launch_t<DP, true>(a, s) : flash2_bwd_launch_t<DP, false>(a, s);
}

template <int DP, bool DROP>
int flash_bwd_launch_t(const FArgs &a, hipStream_t s) {
  dim3 grid((unsigned)(ceil_div(a.S, 128) * a.H * a.B));
  hipLaunchKernelGGL((flash_bwd_dq_k<DP, DROP>), grid, dim3(256), (size_t)128 * AT<DP>::RSTR, s, a);
  LDM_CHECK_LAUNCH();
  hipLaunchKernelGGL((flash_bwd_dkv_k<DP, DROP>), grid, dim3(256), (size_t)128 * AT<DP>::RSTR + 512, s, a);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

template <int DP>
int flash_bwd_launch(const FArgs &a, hipStream_t s) {
  return a.drop_thresh ? flash_bwd_launch_t<DP, true>(a, s) : flash_bwd_launch_t<DP, false>(a, s);
}

int check(const FArgs &a) {
  if (a.B <= 0 || a.H <= 0 || a.S <= 0 || a.hd <= 0) return LDM_E_BADARG;
  if (a.hd % 8 || a.hd > 128) return LDM_E_SHAPE;
  if (a.ld % 8 || a.ld_o % 8) return LDM_E_ALIGN;
  if (ceil_div(a.S, 128) * (int64_t)a.H * a.B > 0x7fffffff) return LDM_E_SHAPE;   // one-dimensional grid
  return LDM_OK;
}

}  // namespace

int ldm_flash_fwd_bf16(const bf16_t *qkv, int64_t ld, const uint8_t *key_valid, int64_t B, int64_t H, int64_t S,
                       int64_t hd, bf16_t *o, int64_t ld_o, float *lse, bf16_t *o_lo, unsigned drop_thresh,
                       uint64_t seed, hipStream_t s) {
  FArgs a{};
  a.o_lo = o_lo;
  a.drop_thresh = drop_thresh; a.drop_inv = (float)(4294967296.0 / (4294967296.0 - (double)drop_thresh));
  a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32);
  a.qkv = qkv; a.ld = ld; a.key_valid = key_valid; a.B = (int)B; a.H = (int)H; a.S = (int)S; a.hd = (int)hd;
  a.o = o; a.ld_o = ld_o; a.lse = lse;
#ifdef LDM_ATTN_TIMING
  { const char *tb = getenv("LDM_ATTN_TIMING_BUF"); a.timing = tb ? (unsigned long long *)strtoull(tb, nullptr, 0) : nullptr; }
#endif
  int rc = check(a);
  if (rc) return rc;
  if (!aligned_to(qkv, 16) || !aligned_to(o, 8)) return LDM_E_ALIGN;
  if (hd <= 32) return flash_fwd_launch<32>(a, s);
  if (hd <= 64) return flash_fwd_launch<64>(a, s);
  if (attn_dma_enabled() && a.ld % 8 == 0 && aligned_to(a.qkv, 16)) {
    if (hd == 96) return flash2_fwd_launch<96>(a, s);
    if (hd == 128) return flash2_fwd_launch<128>(a, s);
  }
  if (hd <= 96) return flash_fwd_launch<96>(a, s);
  return flash_fwd_launch<128>(a, s);
}

int ldm_flash_bwd_bf16(const bf16_t *qkv, int64_t ld, const uint8_t *key_valid, int64_t B, int64_t H, int64_t S,
                       int64_t hd, const bf16_t *o, const bf16_t *dout, int64_t ld_o, const float *lse,
                       const bf16_t *o_lo, bf16_t *dqkv, float *delta, const float *rope_cos, const float *rope_sin,
                       float rope_scale, unsigned drop_thresh, uint64_t seed, hipStream_t s) {
  FArgs a{};
  a.o_lo_in = o_lo;
  a.drop_thresh = drop_thresh; a.drop_inv = (float)(4294967296.0 / (4294967296.0 - (double)drop_thresh));
  a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32);
  if (rope_cos) {   // fused un-rotation: head dim must fill its template and the tables must be 16-byte aligned
    if (!rope_sin || !(hd == 32 || hd == 64 || hd == 96 || hd == 128)) return LDM_E_SHAPE;
    if (!aligned_to(rope_cos, 16) || !aligned_to(rope_sin, 16)) return LDM_E_ALIGN;
  }
  a.rope_cos = rope_cos; a.rope_sin = rope_sin; a.rope_scale = rope_scale;
  a.qkv = qkv; a.ld = ld; a.key_valid = key_valid; a.B = (int)B; a.H = (int)H; a.S = (int)S; a.hd = (int)hd;
  a.o_in = o; a.dout = dout; a.ld_o = ld_o; a.lse_in = lse; a.dqkv = dqkv; a.delta = delta;
#ifdef LDM_ATTN_TIMING
  { const char *tb = getenv("LDM_ATTN_TIMING_BUF"); a.timing = tb ? (unsigned long long *)strtoull(tb, nullptr, 0) : nullptr; }
#endif
  int rc = check(a);
  if (rc) return rc;
  if (!aligned_to(qkv, 16) || !aligned_to(dout, 16) || !aligned_to(o, 8) || !aligned_to(dqkv, 8)) return LDM_E_ALIGN;
  // (the LDS-DMA kernels fetch q/k/v, dO, O and the stored rounding residual of O in 16-byte pieces: an O or a
  // residual that is only 8-byte aligned goes to the register-staged kernels)
  if (attn_dma_enabled() && a.ld % 8 == 0 && a.ld_o % 8 == 0 && aligned_to(a.qkv, 16) && aligned_to(a.dout, 16) &&
      aligned_to(a.o_in, 16) && (!a.o_lo_in || aligned_to(a.o_lo_in, 16))) {
    if (hd == 96) return flash2_bwd_launch<96>(a, s);
    if (hd == 128) return flash2_bwd_launch<128>(a, s);
  }
  if (hd <= 32) return flash_bwd_launch<32>(a, s);
  if (hd <= 64) return flash_bwd_launch<64>(a, s);
  if (hd <= 96) return flash_bwd_launch<96>(a, s);
  return flash_bwd_launch<128>(a, s);
}
