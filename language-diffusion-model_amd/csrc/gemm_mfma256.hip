// Large-tile bf16 MFMA GEMM for gfx950: 256 x BN output tile (BN = 256 or 128), 8 waves,
// one workgroup per CU, operands streamed global -> LDS by 16-byte LDS-DMA through a ring
// of FIVE 32-deep half-tiles (NSLOT) so that up to four half-tiles stay in flight across the
// per-phase barrier (counted vmcnt, raw s_barrier).
//
// Why the big tile: at 128x128 a workgroup streams 32 KiB from L2 per 2.1 MFLOP
// (64 FLOP/B); the measured L2 -> CU rate (~70 GB/s per CU) then caps the chip near
// 1.1 PFLOP/s.  256x256 doubles the intensity.
//
// One phase (= one 32-deep half-tile, 32 MFMAs per wave at 256x256), per wave:
//   segment R:  s_waitcnt vmcnt(P)      this wave's pieces of half-tile p+1 have landed
//               ds_read fragments of half-tile p
//               issue LDS-DMA for half-tile p+3 into the slot of p-1
//               s_waitcnt lgkmcnt(0); s_barrier
//   segment M:  MFMAs on the fragments; s_barrier
// P = LDS-DMA instructions per wave per half-tile.  Waves 4-7 run one barrier behind waves
// 0-3, so on every SIMD one wave's segment M overlaps the other's segment R.  A half-tile is
// read only after EVERY wave's vmcnt for it and a barrier all waves passed; a slot is refilled
// only after every wave's reads of it completed (lgkmcnt(0)) and a barrier.
//
// LDS images (per operand, 32 k per half-tile):
//   K-contiguous operand (A stored [M][K] / B stored [N][K]): [rows][32 k], 64-B rows;
//     16-B chunk c of row r sits at chunk c ^ (3 * ((r >> 2) & 1))   (ds_read_b128 conflict free)
//   strided operand (A stored [K][M] / B stored [K][N]): [32 k][cols], 256/512-B rows;
//     16-B chunk c of row k sits at c ^ (s(k) << 1), s(k) = (k & 3) | (((k >> 3) & 1) << 2)
//     (ds_read_b64_tr_b16 conflict free)
// The swizzle is applied to the per-lane SOURCE address of the LDS-DMA (its LDS
// destination is lane-linear) and again on the read.
#include "common.h"
#include <type_traits>
#include <vector>
#include <stdio.h>
#include "gemm.h"
#include <stdlib.h>

namespace {

typedef short s4_t __attribute__((ext_vector_type(4)));
typedef short s8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

constexpr int BM = 256;
constexpr int KH = 32;                 // k depth of a half-tile
constexpr int NSLOT = 5;               // ring slots (half-tiles)
constexpr int LDS_BYTES = 160 * 1024;  // ring of 5 half-tiles (160 or 120 KiB) + 32 KiB epilogue staging (inside slot 4 at BN = 256)

struct MArgs {
  int64_t M, N, K;
  float alpha;
  const bf16_t *A; int64_t lda;
  const bf16_t *B; int64_t ldb;
  void *C; int64_t ldc;
  const float *bias;
  const void *R; int64_t ldr;
  int64_t gate_off;   // RES 2: columns of the gate half in R and C (R = [a | g], C = [da | dg])
                      // RES 3: rows of the gate weights in B = [w1; w2] and columns of g in C = [a | g]
  void *C2; int64_t ldc2;   // RES 3: h = a * silu(g), [M, gate_off]
  // RES 5: dropout of the product (+ bias) before the residual is added; drop_thresh == 0: none
  unsigned drop_thresh, drop_lo, drop_hi; float drop_inv;
  // RES 4 (q/k/v projection with the rotary embedding in the epilogue): fp32 tables [rope_S, rope_hd/2],
  // rope_step = [cos(16 theta_d) | sin(16 theta_d)] (2 x rope_hd/2), rope_D = H * hd columns per third
  const float *rope_cos, *rope_sin, *rope_step;
  int rope_hd, rope_S;
  int64_t rope_D;
  float rope_qscale;
  int tiles_m, tiles_n, splits;
  int band_m;   // m-tiles per band of the tile walk (see decode_item)
  // optional device-side sizes (a row count known only on the GPU): rows >= *dyn_m of A/C do
  // not exist; K is *dyn_k rounded up to 32 (the caller zero-fills the rounding rows);
  // row_map sends output row m to C row row_map[m]
  const int *dyn_m, *dyn_k, *row_map;
  // EPI 1 only: rowsum[m] += alpha * sum_k opA[m, k] (the bias gradient when opA is dy^T),
  // computed by the workgroups of the first tile column with a ones fragment
  float *rowsum;
  unsigned long long *timing;   // experiment builds: segment cycle totals [8 waves][8]
};

// Experiment builds (-DLDM_GEMM_TIMING): cycle totals of the main-loop segments of workgroup 17
// are written to the device address given in the environment variable LDM_GEMM_TIMING_BUF.
#ifdef LDM_GEMM_TIMING
#define TSTAMP(var) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); var = __builtin_amdgcn_s_memtime(); \
                         asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#else
#define TSTAMP(var) do { } while (0)
#endif

__device__ __forceinline__ int swz_kc(int row) { return 3 * ((row >> 2) & 1); }
__device__ __forceinline__ int swz_st(int kr) { return ((kr & 3) | (((kr >> 3) & 1) << 2)) << 1; }

// Source pointers of the LDS-DMA pieces (1 KiB each) this wave issues per half-tile of an
// operand with ROWS rows (KC) or ROWS columns (strided).  Piece index = wave + 8*i.
template <bool KC, int ROWS>
__device__ __forceinline__ void stage_ptrs(const bf16_t *P, int64_t ld, int64_t row0, int64_t nrows, int wave,
                                           int lane, const bf16_t *(&src)[ROWS / 128], int64_t gate_rows = 0) {
#pragma unroll
  for (int i = 0; i < ROWS / 128; ++i) {
    const int piece = wave + 8 * i;
    if (KC) {
      const int r = piece * 16 + (lane >> 2);
      const int lc = (lane & 3) ^ swz_kc(r);
      // gate_rows != 0 (SwiGLU forward): tile rows come in blocks of 32, alternately from the first
      // operand half (rows row0 + 32 j ..) and from the same rows gate_rows further down
      int64_t gr = gate_rows ? row0 + ((r >> 6) << 5) + (r & 31) + ((r >> 5) & 1) * gate_rows : row0 + r;
      if (gr > nrows - 1) gr = nrows - 1;
      src[i] = P + gr * ld + lc * 8;
    } else {
      constexpr int RB = ROWS * 2;              // bytes per k row of the image
      constexpr int RPP = 1024 / RB;            // k rows per piece
      const int kr = piece * RPP + (lane * 16) / RB;
      const int pc = ((lane * 16) % RB) / 16;
      const int lc = pc ^ swz_st(kr);
      int64_t col = row0 + lc * 8;
      if (col > nrows - 8) col = nrows - 8;
      src[i] = P + (int64_t)kr * ld + col;
    }
  }
}

// RES 4: which output column does the 16-column group `grp` (0..15) of n-tile tn hold?  Tiles over the q
// and k thirds are fed so that accumulator fragments 0-1 of a wave hold two 16-column groups of FIRST
// half-heads and fragments 2-3 their rotation partners hd/2 further on (same lane, same register index:
// the rotation needs no exchange); the walk over first-half groups runs through q, then k.  Tiles over
// the v third are identity-mapped.
__device__ __forceinline__ int64_t rope_group_col(int tn, int grp, int64_t D, int hd) {
  const int rope_tiles = (int)(2 * D / 256);
  if (tn >= rope_tiles) return (int64_t)tn * 256 + grp * 16;
  const int f = grp & 3;
  const int gph = hd / 32;                       // 16-column groups per half head
  int L = 2 * (4 * tn + (grp >> 2)) + (f & 1);
  const int per_region = (int)(D / 32);
  const int region = L >= per_region ? 1 : 0;    // 0: q, 1: k
  L -= region * per_region;
  const int head = L / gph, sub = L - head * gph;
  return (int64_t)region * D + (int64_t)head * hd + sub * 16 + (f >> 1) * (hd / 2);
}
// first-half position (0 .. hd/2 - 16, step 16) of fragment ni (0/1) of wave-column wn in n-tile tn
__device__ __forceinline__ int rope_group_d(int tn, int wn, int ni, int64_t D, int hd) {
  const int gph = hd / 32;
  int L = 2 * (4 * tn + wn) + ni;
  const int per_region = (int)(D / 32);
  if (L >= per_region) L -= per_region;
  return (L % gph) * 16;
}
__device__ __forceinline__ void stage_ptrs_rope(const bf16_t *P, int64_t ld, int tn, int wave, int lane, int64_t D, int hd,
                                                const bf16_t *(&src)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int piece = wave + 8 * i;               // = 16-column group of the tile
    const int r = piece * 16 + (lane >> 2);
    const int lc = (lane & 3) ^ swz_kc(r);
    src[i] = P + (rope_group_col(tn, piece, D, hd) + (lane >> 2)) * ld + lc * 8;
  }
}

template <int NP>
__device__ __forceinline__ void stage_issue(const bf16_t *(&src)[NP], int64_t koff, char *img, int wave) {
#pragma unroll
  for (int i = 0; i < NP; ++i)
    __builtin_amdgcn_global_load_lds(GLB_PTR(src[i] + koff), LDS_PTR(img + (wave + 8 * i) * 1024), 16, 0, 0);
}

// Two consecutive half-tiles of one operand, piece by piece: for a K-contiguous operand the two 64-byte
// row segments are the halves of one 128-byte line, and requested back to back by the same wave the
// second is served from the vector L1 instead of by a second trip to L2 (measured L2->LDS rate of one CU:
// 72 GB/s for the halves a phase apart, 135 GB/s back to back).
template <int NP>
__device__ __forceinline__ void stage_issue_pair(const bf16_t *(&src)[NP], int64_t koff0, int64_t koff1, char *img0,
                                                 char *img1, int wave) {
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(src[i] + koff0), LDS_PTR(img0 + (wave + 8 * i) * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(GLB_PTR(src[i] + koff1), LDS_PTR(img1 + (wave + 8 * i) * 1024), 16, 0, 0);
  }
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wait_vm<n> for an n that is a constant only after unrolling
template <int LO, int HI>
__device__ __forceinline__ void wait_vm_sw(int n) {
  if constexpr (LO == HI) wait_vm<LO>();
  else {
    constexpr int MID = (LO + HI) / 2;
    if (n <= MID) wait_vm_sw<LO, MID>(n); else wait_vm_sw<MID + 1, HI>(n);
  }
}
// 16-byte global load whose wait the caller places (wait_vm + a launder of `v`): the reading epilogues
// count their own stores and the next item's prefetch into the wait.  The compiler's own wait for such a
// load is vmcnt(0) there -- for every store of the epilogue so far and the whole prefetch.
typedef unsigned u4g_t __attribute__((ext_vector_type(4)));
// (scalar base + 32-bit per-lane byte offset: one address register for all of a tile's loads)
__device__ __forceinline__ void gload16(u4g_t &v, const void *sbase, unsigned lane_off) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(lane_off), "s"(sbase) : "memory");
}

__device__ __forceinline__ void gload4(unsigned &v, const void *sbase, unsigned lane_off) {
  asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(lane_off), "s"(sbase) : "memory");
}

// 16 rows x 32 k fragment of a K-contiguous half-tile image (64-byte rows)
__device__ __forceinline__ bf8_t read_frag_kc(const char *img, int rbase, int lane) {
  const int r = rbase + (lane & 15);
  const int pc = (lane >> 4) ^ swz_kc(r);
  return *reinterpret_cast<const bf8_t *>(img + r * 64 + pc * 16);
}

// The strided-operand fragment as two transposing reads written as asm: behind the compiler's own
// transposing-read intrinsic every LDS-DMA in flight is drained first (s_waitcnt vmcnt(0): it cannot
// see that the slot being read and the slots being filled are disjoint), which costs the ring its depth.
// The values exist only after the s_waitcnt lgkmcnt that tr_done() is placed behind.
template <int ROWS>
__device__ __forceinline__ void tr_issue(unsigned img_lds, int rbase, int lane, s4_t &lo, s4_t &hi) {
  constexpr int RB = ROWS * 2;
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int kr0 = 8 * g + q;
  const int ch = ((rbase >> 3) + (p >> 1)) ^ swz_st(kr0);
  const unsigned a0 = img_lds + kr0 * RB + (ch << 4) + ((p & 1) << 3);
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a0), "n"(4 * RB));
}
template <int N>
__device__ __forceinline__ void tr_done(s4_t (&lo)[N], s4_t (&hi)[N], bf8_t (&f)[N]) {
  static_assert(N == 2 || N == 4 || N == 8, "fragment count");
  if constexpr (N == 2)
    asm volatile("" : "+v"(lo[0]), "+v"(lo[1]), "+v"(hi[0]), "+v"(hi[1]));
  else if constexpr (N == 4)
    asm volatile("" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
  else
    asm volatile("" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(lo[5]), "+v"(lo[6]), "+v"(lo[7]),
                      "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4]), "+v"(hi[5]), "+v"(hi[6]), "+v"(hi[7]));
#pragma unroll
  for (int i = 0; i < N; ++i) {
    s8_t v = {lo[i][0], lo[i][1], lo[i][2], lo[i][3], hi[i][0], hi[i][1], hi[i][2], hi[i][3]};
    f[i] = __builtin_bit_cast(bf8_t, v);
  }
}

// Work item -> (split, tile) with an XCD-aware order: work items w and w+8 land on the same
// XCD (shared L2), so each of the 8 groups gets a contiguous run of items; tiles are walked
// in bands of 4 m-tiles.
struct TileId { int split, tm, tn; };
__device__ __forceinline__ TileId decode_item(int w, int total, int tiles_m, int tiles_n, int GM) {
  {
    const int q = total >> 3, r = total & 7, x = w & 7;
    w = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (w >> 3);
  }
  const int ntiles = tiles_m * tiles_n;
  TileId t;
  t.split = w / ntiles;
  w -= t.split * ntiles;
  const int width = GM * tiles_n;
  const int grp = w / width;
  const int first = grp * GM;
  const int gsz = (tiles_m - first < GM) ? tiles_m - first : GM;
  const int in = w - grp * width;
  t.tm = first + in % gsz;
  t.tn = in / gsz;
  return t;
}

// BN: 256 (waves 2M x 4N, 128x64 per wave) or 128 (waves 4M x 2N, 64x64 per wave).
// PERSISTENT: gridDim.x workgroups (one per CU) walk the work items w = blockIdx.x,
// blockIdx.x + gridDim.x, ...; the LDS-DMA of the NEXT item's first four half-tiles is issued
// before the epilogue of the current one, so the store tail and the load latency overlap.
// Half-tiles of a K-contiguous operand travel in PAIRS (2q, 2q+1): the first two pairs before the item
// starts, pair (p+3, p+4) at the start of the MFMA segment of every odd phase p, into the ring slots of
// half-tiles p-2 and p-1.  Half-tiles of a strided operand (whole 128-byte lines per piece anyway) travel singly,
// half-tile p+3 in phase p.
template <bool A_KC, bool B_KC, int BN, typename TO, int EPI, int RES = 0>
__global__ __launch_bounds__(512, 2) void gemm256_k(MArgs g) {
  constexpr int WN_ = BN / 64;            // waves along N
  constexpr int WM_ = 8 / WN_;            // waves along M
  constexpr int MI = BM / WM_ / 16;       // 16-row fragments per wave
  constexpr int NI = 4;                   // 16-col fragments per wave (64 columns)
  constexpr int PA = BM / 128, PB = BN / 128;   // LDS-DMA pieces per wave per half-tile
  constexpr int HT_A = BM * 64, HT_B = BN * 64; // bytes of one half-tile image
  constexpr int HT = HT_A + HT_B;
  // epilogue staging (8 waves x 4 KiB): ring slot 4 when a slot is 32 KiB, else behind the ring
  constexpr int STAGE_OFF = (HT >= 32768) ? (NSLOT - 1) * HT : NSLOT * HT;
  static_assert(STAGE_OFF + 32768 <= LDS_BYTES, "ring + staging exceed the LDS allocation");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN_, wn = wave % WN_;
  // The split-K (weight-gradient) form takes its 16-row fragments of A in an order rotated by the wave's
  // column: the fragments whose row sums this wave owns (RS of them, for the bias gradient) are then its
  // first ones whatever the wave, and the row-sum MFMAs take them from the registers the products use.
  // (Read a second time at addresses of their own they cost two address registers, which spilled: a scratch
  // reload, and with it a drain of every LDS-DMA in flight, in every phase of the tiles that sum rows.)
  auto afrag = [&](int i) { return EPI == 1 ? ((i + (MI / WN_) * wn) & (MI - 1)) : i; };
  const bool late = wave >= 4;

  const int64_t Mdyn = g.dyn_m ? (int64_t)*g.dyn_m : g.M;
  // with a device-side row count only the m-tiles that hold rows are walked
  const int tiles_m = (int)((Mdyn + BM - 1) / BM) < g.tiles_m ? (int)((Mdyn + BM - 1) / BM) : g.tiles_m;
  const int total = tiles_m * g.tiles_n * g.splits;
  const int64_t Kdyn = g.dyn_k ? (((int64_t)*g.dyn_k + KH - 1) / KH) * KH : g.K;
  const int np_total = (int)(Kdyn / KH);
  const int base = np_total / g.splits, rem = np_total % g.splits;
  const int64_t kstepA = A_KC ? KH : (int64_t)KH * g.lda;
  const int64_t kstepB = B_KC ? KH : (int64_t)KH * g.ldb;

  int w = blockIdx.x;
  if (w >= total) return;   // workgroup-uniform
  TileId tid = decode_item(w, total, tiles_m, g.tiles_n, g.band_m);

  const bf16_t *srcA[PA], *srcB[PB];
  int p_begin = tid.split * base + (tid.split < rem ? tid.split : rem);
  int np = base + (tid.split < rem ? 1 : 0);
  stage_ptrs<A_KC, BM>(g.A, g.lda, (int64_t)tid.tm * BM, Mdyn, wave, lane, srcA);
  if constexpr (RES == 4) stage_ptrs_rope(g.B, g.ldb, tid.tn, wave, lane, g.rope_D, g.rope_hd, srcB);
  else stage_ptrs<B_KC, BN>(g.B, g.ldb, (int64_t)tid.tn * (RES == 3 ? BN / 2 : BN), g.N, wave, lane, srcB,
                            RES == 3 ? g.gate_off : 0);
  // Per half-tile a wave issues SI instructions for the operands that travel singly, and per pair PI
  // for those that travel in pairs.
  constexpr bool PAIR_A = A_KC, PAIR_B = B_KC;
  constexpr int SI = (PAIR_A ? 0 : PA) + (PAIR_B ? 0 : PB);
  constexpr int PI = (PAIR_A ? 2 * PA : 0) + (PAIR_B ? 2 * PB : 0);
  // Half of one operand's fragments is read in segment M (under the first MFMAs): half of A, or, when
  // both operands travel in pairs, half of B.  A pair's second slot is that of half-tile p-1, which the
  // other half of the workgroup (one barrier behind) may still be reading in ITS segment M while this
  // half is in segment R: the pair of the operand that is read in segment M goes out in segment M (once a
  // wave is there, every wave has left phase p-1), the other operand's pair in segment R.
  constexpr bool SPLIT_B = PAIR_A && PAIR_B, SPLIT_A = !SPLIT_B;
  constexpr bool A_PAIR_M = PAIR_A && SPLIT_A, B_PAIR_M = PAIR_B && SPLIT_B;
  constexpr int PR = (PAIR_A && !A_PAIR_M ? 2 * PA : 0) + (PAIR_B && !B_PAIR_M ? 2 * PB : 0);   // pair instructions sent in R
  typedef std::integral_constant<bool, !A_PAIR_M> a_in_r_t;
  typedef std::integral_constant<bool, !B_PAIR_M> b_in_r_t;
  typedef std::integral_constant<bool, A_PAIR_M> a_in_m_t;
  typedef std::integral_constant<bool, B_PAIR_M> b_in_m_t;
  // Element offsets into A and B of the half-tile that segment R of the current phase sends (p + 3):
  // advanced by one k step per phase (scalar adds instead of 64-bit multiplies in every phase).
  int64_t offA = 0, offB = 0;
  // pair of the half-tiles at offsets (oa0, ob0) and (oa1, ob1) into ring slots s0, s1
  auto issue_pair_at = [&](auto do_a, auto do_b, int64_t oa0, int64_t oa1, int64_t ob0, int64_t ob1, int s0, int s1) {
    if constexpr (PAIR_A && decltype(do_a)::value) stage_issue_pair<PA>(srcA, oa0, oa1, smem + s0 * HT, smem + s1 * HT, wave);
    if constexpr (PAIR_B && decltype(do_b)::value)
      stage_issue_pair<PB>(srcB, ob0, ob1, smem + s0 * HT + HT_A, smem + s1 * HT + HT_A, wave);
  };
  auto issue_single_at = [&](int64_t oa, int64_t ob, int s0) {
    if constexpr (!PAIR_A) stage_issue<PA>(srcA, oa, smem + s0 * HT, wave);
    if constexpr (!PAIR_B) stage_issue<PB>(srcB, ob, smem + s0 * HT + HT_A, wave);
  };
  // by half-tile index; a half-tile past the item's last one is a second request of the last one (same
  // instruction count for the counted waits; its slot is free)
  auto issue_pair = [&](int h, int s0, int s1) {
    const int h0 = h < np ? h : np - 1, h1 = h + 1 < np ? h + 1 : np - 1;
    issue_pair_at(std::true_type{}, std::true_type{}, (int64_t)(p_begin + h0) * kstepA, (int64_t)(p_begin + h1) * kstepA, (int64_t)(p_begin + h0) * kstepB,
                  (int64_t)(p_begin + h1) * kstepB, s0, s1);
  };
  auto issue_single = [&](int h, int s0) {
    const int h0 = h < np ? h : np - 1;
    issue_single_at((int64_t)(p_begin + h0) * kstepA, (int64_t)(p_begin + h0) * kstepB, s0);
  };
  // before an item starts: pairs (0,1) and (2,3), then the single half-tiles 0, 1, 2
  auto issue_first = [&]() {
    if (np <= 0) return;
    issue_pair(0, 0, 1);
    issue_pair(2, 2, 3);
    issue_single(0, 0);
    issue_single(1, 1);
    issue_single(2, 2);
    offA = (int64_t)(p_begin + 3) * kstepA;
    offB = (int64_t)(p_begin + 3) * kstepB;
  };
  issue_first();

  // stores per wave in the fast-path epilogue (16 bytes per lane each)
  constexpr int NST = RES == 3 ? MI * 3 : MI * (sizeof(TO) == 2 ? 2 : 4) * (RES == 2 ? 2 : 1);
  static_assert(2 * PI + 3 * SI + NST <= 63, "vmcnt is a 6-bit counter");
#ifdef LDM_GEMM_TIMING
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long te0 = 0, te1 = 0;
#endif
  bool carry_next = false;
  const unsigned stage_lds = (unsigned)(uintptr_t)LDS_PTR(smem) + STAGE_OFF + wave * 4096;

  while (true) {
    const int64_t m0 = (int64_t)tid.tm * BM, n0 = (int64_t)tid.tn * BN;
    const int64_t nw = n0 + wn * 64;
    // Output columns of a lane in the store phase of the epilogue: CPL consecutive ones
    // (16 bytes of TO), so a row of the wave's 64 columns is written by 64 / CPL lanes.
    constexpr int CPL = sizeof(TO) == 2 ? 8 : 4;
    constexpr int LPR = 64 / CPL;         // lanes per row
    constexpr int RPP = 64 / LPR;         // rows per store instruction
    f4_t acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
    // row sums of opA: wave wn of a first-column tile takes fragments mi = RS*wn .. RS*wn + RS-1
    constexpr int RS = MI / WN_;
    f4_t racc[RS];
#pragma unroll
    for (int j = 0; j < RS; ++j) racc[j] = f4_t{0.f, 0.f, 0.f, 0.f};
    const bool do_rowsum = EPI == 1 && g.rowsum != nullptr && tid.tn == 0;

    // half-tiles 0 and 1 landed for everybody (also: everybody has left the previous epilogue).
    // After a fast-path epilogue exactly NST stores of this wave are YOUNGER than the two
    // pairs prefetched before it (vmcnt counts loads and stores together, in issue
    // order): counting them in lets the first phases start while those stores drain.
    TSTAMP(te1);
#ifdef LDM_GEMM_TIMING
    if (te0) tacc[6] += te1 - te0;   // previous item's epilogue (+ next item's set-up and prefetch issue)
#endif
    const bool carry = carry_next;
    carry_next = false;
    // younger than the last piece of half-tile 0: single half-tiles 1 and 2, or else pair (2,3)
    constexpr int W0 = SI > 0 ? 2 * SI : PI;
    if (carry) wait_vm<W0 + NST>(); else wait_vm<W0>();
    __builtin_amdgcn_s_barrier();

    // The two halves of the workgroup (waves 0-3 / 4-7: one wave of each per SIMD) run ONE
    // barrier apart: while one half is in its MFMA segment the other reads LDS and issues
    // LDS-DMA.  The second half takes one extra barrier here and the first half one after
    // the loop, so every s_barrier below pairs segment R of one half with segment M of the other.
    if (late) __builtin_amdgcn_s_barrier();
#ifdef LDM_GEMM_TIMING
    { unsigned long long tp; TSTAMP(tp); tacc[7] += tp - te1; }   // wait for the first half-tile + barriers
#endif

#ifdef LDM_GEMM_TIMING
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
#endif
    int slot = 0;   // ring slot of half-tile p
    for (int p = 0; p < np; ++p) {
      TSTAMP(t0);
      // ---- segment R: read p; in odd phases send pair (p+3, p+4) into the slots of p-2 and p-1 and
      // retire this wave's pieces of pair (p+1, p+2), which the next two phases read
      // fragments: (see SPLIT_A / SPLIT_B) the second half of one operand is requested at the start of
      // segment M and arrives under the first half's MFMAs
      constexpr int MH = SPLIT_A ? MI / 2 : MI, NH = SPLIT_B ? NI / 2 : NI;
      bf8_t fa[MH], fa2[SPLIT_A ? MH : 1], fb[NH], fb2[SPLIT_B ? NH : 1];
      [[maybe_unused]] s4_t alo[MH], ahi[MH], alo2[SPLIT_A ? MH : 1], ahi2[SPLIT_A ? MH : 1], blo[NH], bhi[NH];
      [[maybe_unused]] const char *cur = smem + slot * HT;
      [[maybe_unused]] const unsigned cur_lds = (unsigned)(uintptr_t)LDS_PTR(smem) + slot * HT;
#pragma unroll
      for (int i = 0; i < NH; ++i) {
        if constexpr (B_KC) fb[i] = read_frag_kc(cur + HT_A, wn * 64 + i * 16, lane);
        else tr_issue<BN>(cur_lds + HT_A, wn * 64 + i * 16, lane, blo[i], bhi[i]);
      }
#pragma unroll
      for (int i = 0; i < MH; ++i) {
        if constexpr (A_KC) fa[i] = read_frag_kc(cur, wm * (MI * 16) + afrag(i) * 16, lane);
        else tr_issue<BM>(cur_lds, wm * (MI * 16) + afrag(i) * 16, lane, alo[i], ahi[i]);
      }
      {
        const int rem = np - p;             // half-tiles left, this one included
        const bool more3 = rem > 3;
        const int s3 = slot + 3 >= NSLOT ? slot + 3 - NSLOT : slot + 3;
        if (SI > 0 && more3) issue_single_at(offA, offB, s3);
        if (PR > 0 && (p & 1) && more3)
          issue_pair_at(a_in_r_t{}, b_in_r_t{}, offA, rem > 4 ? offA + kstepA : offA, offB, rem > 4 ? offB + kstepB : offB, s3,
                        s3 + 1 >= NSLOT ? 0 : s3 + 1);
        // Retire this wave's pieces of half-tile p+1.  Issue order: before the item P01 P23 S0 S1 S2 (then
        // the previous item's NST stores, if carried), then per phase q: S(q+3) in segment R and, q odd,
        // P(q+3, q+4), one operand's part in segment R and the other's in segment M.  Younger than the
        // last piece of half-tile p+1 are therefore: the singles of p+2 and p+3, in even phases the pairs
        // sent in the phase before, and in odd phases the part of the pair just sent in segment R.
        TSTAMP(t1);
        constexpr int PODD = PR;
        if (p >= 2) {
          if (rem > 3) {
            if (p & 1) wait_vm<2 * SI + PODD>(); else wait_vm<2 * SI + PI>();
          } else if (rem == 3) {
            if (p & 1) wait_vm<SI>(); else wait_vm<SI + PI>();
          } else {
            wait_vm<0>();
          }
        } else if (p == 1) {
          if (rem > 3) { if (carry) wait_vm<2 * SI + PODD + NST>(); else wait_vm<2 * SI + PODD>(); }
          else if (rem == 3) { if (carry) wait_vm<SI + NST>(); else wait_vm<SI>(); }
          else { if (carry) wait_vm<NST>(); else wait_vm<0>(); }
        } else {
          // p == 0: younger than S1 are S2, [stores], S3; without singles, younger than P01 is P23
          constexpr int Y0 = SI > 0 ? SI : PI;
          if (rem > 3) { if (carry) wait_vm<Y0 + SI + NST>(); else wait_vm<Y0 + SI>(); }
          else { if (carry) wait_vm<Y0 + NST>(); else wait_vm<Y0>(); }
        }
      }
      slot = slot + 1 >= NSLOT ? 0 : slot + 1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of half-tile p are complete
      if constexpr (!B_KC) tr_done<NH>(blo, bhi, fb);
      if constexpr (!A_KC) tr_done<MH>(alo, ahi, fa);
      TSTAMP(t2);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      TSTAMP(t3);
      // ---- segment M
      __builtin_amdgcn_s_setprio(1);
      if constexpr (SPLIT_A) {
#pragma unroll
        for (int i = 0; i < MH; ++i) {
          if constexpr (A_KC) fa2[i] = read_frag_kc(cur, wm * (MI * 16) + afrag(MH + i) * 16, lane);
          else tr_issue<BM>(cur_lds, wm * (MI * 16) + afrag(MH + i) * 16, lane, alo2[i], ahi2[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NH; ++i) fb2[i] = read_frag_kc(cur + HT_A, wn * 64 + (NH + i) * 16, lane);
      }
      // (Measured: sending this part of the pair behind the first eight MFMAs, or one instruction at a time
      // between groups of four MFMAs, or starting the segment with four MFMAs, is no faster than this.)
      if (PI - PR > 0 && (p & 1) && p + 3 < np) {
        const int s3 = slot + 2 >= NSLOT ? slot + 2 - NSLOT : slot + 2;   // (slot already names half-tile p+1)
        const bool two = p + 4 < np;
        issue_pair_at(a_in_m_t{}, b_in_m_t{}, offA, two ? offA + kstepA : offA, offB, two ? offB + kstepB : offB, s3,
                      s3 + 1 >= NSLOT ? 0 : s3 + 1);
      }
      __builtin_amdgcn_sched_barrier(0);   // the requests above stay ahead of the MFMAs
      if constexpr (SPLIT_A) {
#pragma unroll
        for (int mi = 0; mi < MH; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ni], fa[mi], acc[ni][mi], 0, 0, 0);
        if constexpr (!A_KC) {
          // the wait for the second half stays HERE, behind the sixteen MFMAs that cover the reads (unpinned
          // it moves up to the first MFMA of the segment).  (The K-contiguous reads are the compiler's own:
          // its wait for them lands in front of the first MFMA, behind the LDS-DMA requests that cover
          // them; as asm with the wait pinned here the forward GEMMs measured 1 % slower.)
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          tr_done<MH>(alo2, ahi2, fa2);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mi = 0; mi < MH; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[ni][MH + mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ni], fa2[mi], acc[ni][MH + mi], 0, 0, 0);
      } else {
#pragma unroll
        for (int ni = 0; ni < NH; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ni], fa[mi], acc[ni][mi], 0, 0, 0);
#pragma unroll
        for (int ni = 0; ni < NH; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[NH + ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb2[ni], fa[mi], acc[NH + ni][mi], 0, 0, 0);
      }
      offA += kstepA;
      offB += kstepB;
      asm volatile("" : "+s"(offA), "+s"(offB));   // stay scalar (else: one running VGPR pointer pair per piece)
      if (EPI == 1 && do_rowsum) {
        static_assert(RS <= MH, "the row-sum fragments are among the first half's");
        const bf8_t ones = {1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
        for (int j = 0; j < RS; ++j) racc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[j], racc[j], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      TSTAMP(t4);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#ifdef LDM_GEMM_TIMING
      TSTAMP(t5);
      tacc[0] += t1 - t0;   // LDS reads + LDS-DMA issue (the stamp waits for the reads)
      tacc[1] += t2 - t1;   // counted vmcnt wait
      tacc[2] += t3 - t2;   // barrier after R
      tacc[3] += t4 - t3;   // MFMA issue
      tacc[4] += t5 - t4;   // barrier after M
      tacc[5] += 1;         // phases
#endif
    }
    if (np < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the second prologue pair was never waited for)
    if (!late) __builtin_amdgcn_s_barrier();
    TSTAMP(te0);
    // All reads of the ring are complete for every wave (lgkmcnt(0) before each barrier) and
    // no LDS-DMA is outstanding: slots 0-3 can take the next item's first half-tiles now.

    // Bias of this lane's output columns (RES 3 / 4: of its accumulator columns, 4 consecutive ones per
    // fragment; RES 3: fragments 0-1 a, 2-3 g of the same hidden columns).  Requested HERE -- nothing else is
    // in flight -- and waited for, counted, behind the next item's prefetch.  (Loaded at the top of the item
    // with the compiler's waits, every item began with vmcnt(0) -- for the previous item's stores, which the
    // first phases are meant to overlap, and the prefetch -- once per load: five round trips in RES 3 / 4.)
    // Without a bias there are neither loads nor a wait; a column past N reads the last one (a fixed number
    // of loads for the counted waits).
    constexpr int NBIAS = EPI != 0 ? 0 : (RES == 3 || RES == 4) ? 4 : CPL;
    float bv[CPL];
    f4_t bag[(RES == 3 || RES == 4) ? 4 : 1];
    [[maybe_unused]] unsigned bvr[CPL] = {};
    [[maybe_unused]] u4g_t bagr[(RES == 3 || RES == 4) ? 4 : 1] = {};
#pragma unroll
    for (int j = 0; j < CPL; ++j) bv[j] = 0.f;
#pragma unroll
    for (int i = 0; i < ((RES == 3 || RES == 4) ? 4 : 1); ++i) bag[i] = f4_t{0.f, 0.f, 0.f, 0.f};
    if (NBIAS > 0 && g.bias) {
      const void *bsrc = g.bias;
      if constexpr (RES == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t c = rope_group_col(tid.tn, wn * 4 + i, g.rope_D, g.rope_hd) + (lane >> 4) * 4;
          gload16(bagr[i], bsrc, (unsigned)(c * 4));
        }
      } else if constexpr (RES == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t hc = (int64_t)tid.tn * (BN / 2) + wn * 32 + (i & 1) * 16 + (lane >> 4) * 4 + (i >> 1) * g.gate_off;
          gload16(bagr[i], bsrc, (unsigned)(hc * 4));
        }
      } else {
        const int64_t n = nw + (lane % LPR) * CPL;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          const int64_t c = n + j < g.N ? n + j : g.N - 1;
          gload4(bvr[j], bsrc, (unsigned)(c * 4));
        }
      }
    }

    // Fast-path epilogue with a residual (bf16 output): the residual rows of this wave's tile are
    // loaded 16 bytes per lane exactly where the stores will go, in batches of RQ loads held in two
    // register banks: batch 0 is issued HERE, before the next item's LDS-DMA prefetch (so its
    // latency runs under the prefetch set-up), batch k+2 as soon as batch k has been consumed.
    constexpr bool RES_FAST = (RES == 1 || RES == 5) && EPI == 0 && sizeof(TO) == 2 && MI == 8;   // 256-column tiles, bf16 output
    // RES 5: RES 1 for a problem whose every tile is full and aligned and has a residual (the host checks): no
    // other epilogue is compiled into the variant, which leaves the registers for residual loads written as
    // asm with counted waits
    constexpr bool RES_FULL = RES == 5;
    static_assert(!RES_FULL || RES_FAST, "RES 5 is a 256-column bf16 variant");
    // RES 2: C = [dh * silu(g) | dh * a * silu'(g)] with dh the product and [a | g] read from R (SwiGLU backward)
    constexpr bool SWI_FAST = RES == 2 && EPI == 0 && sizeof(TO) == 2 && MI == 8;
    typedef unsigned u4v_t __attribute__((ext_vector_type(4)));
    const bool fast_tile = RES_FULL || (EPI == 0 && !g.row_map && m0 + BM <= Mdyn && n0 + BN <= g.N && (g.ldc % CPL) == 0);
    const bool res_fast = RES_FULL || (RES_FAST && fast_tile && g.R != nullptr && (g.ldr % CPL) == 0 && g.ldr < (1 << 27));   // (32-bit lane offset)
    // RQ loads (a quarter of the tile's 2 MI store positions) per batch, two batches in registers
    constexpr int RQ = MI / 2;
    // RES 5 holds the residual of the whole tile (four batches, all requested before the first is used)
    u4v_t rr[RES_FULL ? 4 * RQ : (RES_FAST || SWI_FAST) ? 2 * RQ : 1];   // SWI: per bank (a0, a1, g0, g1) of two store positions
    const TO *rrow = nullptr;        // wave-uniform: the wave's first row of R at its first column
    int lane_e = lane;
    if constexpr (RES == 1) asm volatile("" : "+v"(lane_e));   // (not kept across the main loop either)
    [[maybe_unused]] const unsigned rlane = (unsigned)(((lane_e / LPR) * g.ldr + (lane_e % LPR) * CPL) * sizeof(TO));
    if (SWI_FAST) {
      rrow = (const TO *)g.R + (m0 + wm * (MI * 16)) * g.ldr + nw;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        gload16(rr[i], rrow + (int64_t)i * RPP * g.ldr, rlane);
        gload16(rr[2 + i], rrow + (int64_t)i * RPP * g.ldr + g.gate_off, rlane);
      }
    }
    if (RES_FAST && res_fast) {
      rrow = (const TO *)g.R + (m0 + wm * (MI * 16)) * g.ldr + nw;
#pragma unroll
      for (int i = 0; i < RQ; ++i) {
        if constexpr (RES_FULL) gload16(rr[i], rrow + (int64_t)i * RPP * g.ldr, rlane);
        else rr[i] = *reinterpret_cast<const u4v_t *>((const char *)(rrow + (int64_t)i * RPP * g.ldr) + rlane);
      }
    }

    [[maybe_unused]] const int tid_cur_tn = tid.tn;
    // RES 4: bias, then (on the q and k tiles) the rotary embedding of src/transformer.py:178-185 with
    // the query scale of :261 folded in: fragments ni and ni + 2 hold x[d] and x[d + hd/2] of the same row.
    // cos/sin of the row's position come from the tables for the first 16-row fragment and advance by
    // the angle-addition step (16 positions) from one fragment to the next; they are loaded HERE, before
    // the next item's prefetch goes out, so waiting for them does not drain that prefetch.  The v tiles
    // run the same code with cos = 1, sin = 0.
    f4_t rcs[RES == 4 ? 2 : 1], rsn[RES == 4 ? 2 : 1], rstc[RES == 4 ? 2 : 1], rsts[RES == 4 ? 2 : 1];
    if constexpr (RES == 4) {
      const bool is_rope = tid.tn < (int)(2 * g.rope_D / 256);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        rcs[ni] = f4_t{1.f, 1.f, 1.f, 1.f}; rsn[ni] = f4_t{0.f, 0.f, 0.f, 0.f};
        rstc[ni] = f4_t{1.f, 1.f, 1.f, 1.f}; rsts[ni] = f4_t{0.f, 0.f, 0.f, 0.f};
      }
      if (is_rope) {
        const int h2 = g.rope_hd / 2;
        const int pos = (int)((m0 + wm * (MI * 16) + (lane & 15)) % g.rope_S);
        const float sc = 2 * (4 * tid.tn + wn) < (int)(g.rope_D / 32) ? g.rope_qscale : 1.0f;   // this wave's groups are q's
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int d = rope_group_d(tid.tn, wn, ni, g.rope_D, g.rope_hd) + (lane >> 4) * 4;
          rcs[ni] = *reinterpret_cast<const f4_t *>(g.rope_cos + (int64_t)pos * h2 + d);
          rsn[ni] = *reinterpret_cast<const f4_t *>(g.rope_sin + (int64_t)pos * h2 + d);
          rstc[ni] = *reinterpret_cast<const f4_t *>(g.rope_step + d);
          rsts[ni] = *reinterpret_cast<const f4_t *>(g.rope_step + h2 + d);
        }
        // (all eight requested before the first is used: one round trip, not one per pair)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) { rcs[ni] = rcs[ni] * sc; rsn[ni] = rsn[ni] * sc; }
      }
    }
    w += gridDim.x;
    const bool more = w < total;
    if (more) {
      tid = decode_item(w, total, tiles_m, g.tiles_n, g.band_m);
      p_begin = tid.split * base + (tid.split < rem ? tid.split : rem);
      np = base + (tid.split < rem ? 1 : 0);
      // (the operand bases as opaque scalars: left visible, base + per-lane column offset is hoisted out of
      // the tile loop into a register pair per operand, which the full variants spill -- and the reload,
      // with its wait for everything in flight, sits between the residual loads above and the prefetch)
      // (and the lane number as an opaque value: nothing derived from it for this set-up stays in
      // registers across the main loop)
      // (RES 1 only, the variant at the register limit: the recomputation costs the plain input-gradient
      // GEMMs 1 %)
      const bf16_t *baseA = g.A, *baseB = g.B;
      int lane_t = lane;
#ifdef LDM_X_LAUNDER_ALL   // (experiment: every variant)
      asm volatile("" : "+s"(baseA), "+s"(baseB), "+v"(lane_t));
#else
      if constexpr (RES == 1) asm volatile("" : "+s"(baseA), "+s"(baseB), "+v"(lane_t));
#endif
      stage_ptrs<A_KC, BM>(baseA, g.lda, (int64_t)tid.tm * BM, Mdyn, wave, lane_t, srcA);
      if constexpr (RES == 4) stage_ptrs_rope(baseB, g.ldb, tid.tn, wave, lane_t, g.rope_D, g.rope_hd, srcB);
      else stage_ptrs<B_KC, BN>(baseB, g.ldb, (int64_t)tid.tn * (RES == 3 ? BN / 2 : BN), g.N, wave, lane_t, srcB,
                                RES == 3 ? g.gate_off : 0);
      issue_first();
    }
    // (without a bias nothing was requested and the wait passes at once; the plain and SwiGLU-backward
    // variants, whose callers mostly have none, skip it then: with it their input-gradient GEMMs are 1 % slower)
    if (NBIAS > 0 && ((RES != 0 && RES != 2) || g.bias)) {
      // younger than the bias loads (which go out first of all): the first residual batch of the
      // asm-reading variants and the prefetch (anything the compiler has put between them only makes the
      // wait stricter)
      constexpr int NPF0 = 2 * PI + 3 * SI;
      constexpr int NL0 = (RES == 2 || RES == 5) ? 4 : 0;
      if (more && np > 0) wait_vm<NPF0 + NL0>(); else wait_vm<NL0>();
      if constexpr (RES == 3 || RES == 4) {
        asm volatile("" : "+v"(bagr[0]), "+v"(bagr[1]), "+v"(bagr[2]), "+v"(bagr[3]));
#pragma unroll
        for (int i = 0; i < 4; ++i) bag[i] = g.bias ? __builtin_bit_cast(f4_t, bagr[i]) : f4_t{0.f, 0.f, 0.f, 0.f};
      } else {
        const int64_t n = nw + (lane % LPR) * CPL;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          asm volatile("" : "+v"(bvr[j]));
          bv[j] = (g.bias && n + j < g.N) ? __uint_as_float(bvr[j]) : 0.f;
        }
      }
    }

    if (EPI == 1 && do_rowsum && lane < 16) {
      // every register of lane l holds the sum for row (l & 15) of the fragment
#pragma unroll
      for (int j = 0; j < RS; ++j) {
        const int64_t m = m0 + wm * (MI * 16) + (wn * RS + j) * 16 + lane;
        if (m < Mdyn) atomicAdd(g.rowsum + m, racc[j][0] * g.alpha);
      }
    }
    // Fast-path epilogue (tile fully inside C, no residual, no row map, 16-byte-aligned rows):
    // the same LDS re-shaping as below, but with the LDS accesses written as asm so that the
    // compiler does not order them behind the LDS-DMA prefetch just issued (it would drain it
    // with vmcnt(0): it cannot see that the staging area and the ring are disjoint), and with a
    // fixed number of unconditional 16-byte stores (NST), which the next item's waits count on.
    // RES 3 (SwiGLU forward; all tiles full): h = a * silu(g) is formed in registers, where fragments
    // ni and ni + 2 hold a and g of the same hidden columns; per 16-row fragment the wave stages
    // [a(32) | g(32)] and then h(32) through its LDS area and writes three row-contiguous outputs.
    if constexpr (RES == 3) {
      TO *AG = (TO *)g.C, *HH = (TO *)g.C2;
      const int cl = lane & 7, wrow = lane & 15;
      const unsigned waddr = stage_lds + wrow * 256;
      const int64_t hc0 = (int64_t)tid_cur_tn * (BN / 2) + wn * 32;
      TO *crow = AG + (m0 + wm * (MI * 16) + (lane >> 3)) * g.ldc + (cl < 4 ? hc0 + cl * 8 : g.gate_off + hc0 + (cl - 4) * 8);
      TO *hrow = HH + (m0 + wm * (MI * 16) + (lane >> 2)) * g.ldc2 + hc0 + (lane & 3) * 8;
      const int64_t rstep = (int64_t)8 * g.ldc, hstep = (int64_t)16 * g.ldc2;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        f4_t hv[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const f4_t av = acc[ni][mi] + bag[ni], gv = acc[ni + 2][mi] + bag[2 + ni];
#pragma unroll
          for (int j = 0; j < 4; ++j) hv[ni][j] = av[j] * gv[j] * __builtin_amdgcn_rcpf(1.0f + __expf(-gv[j]));
          asm volatile("ds_write_b128 %0, %1" ::"v"(waddr + (((ni * 4 + (lane >> 4)) ^ wrow) << 4)), "v"(av) : "memory");
          asm volatile("ds_write_b128 %0, %1" ::"v"(waddr + ((((ni + 2) * 4 + (lane >> 4)) ^ wrow) << 4)), "v"(gv) : "memory");
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 8 + (lane >> 3);
          f4_t t[2];
#pragma unroll
          for (int q = 0; q < 2; ++q)
            asm volatile("ds_read_b128 %0, %1" : "=v"(t[q]) : "v"(stage_lds + row * 256 + (((cl * 2 + q) ^ row) << 4)) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1])::"memory");
          u4v_t pk;
          pk[0] = pack_bf16x2(t[0][0], t[0][1]);
          pk[1] = pack_bf16x2(t[0][2], t[0][3]);
          pk[2] = pack_bf16x2(t[1][0], t[1][1]);
          pk[3] = pack_bf16x2(t[1][2], t[1][3]);
          __builtin_nontemporal_store(pk, reinterpret_cast<u4v_t *>(crow));
          crow += rstep;
        }
        // the LDS unit executes a wave's accesses in order: h may overwrite the area just read
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          asm volatile("ds_write_b128 %0, %1" ::"v"(waddr + (((ni * 4 + (lane >> 4)) ^ wrow) << 4)), "v"(hv[ni]) : "memory");
        {
          const int row = lane >> 2;
          f4_t t[2];
#pragma unroll
          for (int q = 0; q < 2; ++q)
            asm volatile("ds_read_b128 %0, %1" : "=v"(t[q]) : "v"(stage_lds + row * 256 + ((((lane & 3) * 2 + q) ^ row) << 4)) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1])::"memory");
          u4v_t pk;
          pk[0] = pack_bf16x2(t[0][0], t[0][1]);
          pk[1] = pack_bf16x2(t[0][2], t[0][3]);
          pk[2] = pack_bf16x2(t[1][0], t[1][1]);
          pk[3] = pack_bf16x2(t[1][2], t[1][3]);
          *reinterpret_cast<u4v_t *>(hrow) = pk;   // plain store: the down projection reads h next (-0.7 ms/step against non-temporal)
          hrow += hstep;
        }
      }
      carry_next = more;
      if (!more) break;
      continue;
    }
    if (EPI == 0 && (SWI_FAST || RES_FULL || (fast_tile && (!g.R || res_fast)))) {
      TO *C = (TO *)g.C;
      if (SWI_FAST) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          gload16(rr[RQ + i], rrow + (int64_t)(2 + i) * RPP * g.ldr, rlane);
          gload16(rr[RQ + 2 + i], rrow + (int64_t)(2 + i) * RPP * g.ldr + g.gate_off, rlane);
        }
      }
      if (RES_FAST && res_fast) {
        // batch 1 goes out before batch 0 is used.  (In RES 1 these loads and their waits are the compiler's
        // -- vmcnt(0) in front of every store position: with the other epilogues compiled in, the asm form
        // spills 35 registers.  RES 5 has the asm form.)
#pragma unroll
        for (int i = 0; i < (RES_FULL ? 3 * RQ : RQ); ++i) {
          if constexpr (RES_FULL) gload16(rr[RQ + i], rrow + (int64_t)(RQ + i) * RPP * g.ldr, rlane);
          else rr[RQ + i] = *reinterpret_cast<const u4v_t *>((const char *)(rrow + (int64_t)(RQ + i) * RPP * g.ldr) + rlane);
        }
      }
      // Waits of the reading epilogues (RES 2, RES 5), counted (vmcnt counts loads, LDS-DMA and stores alike, in issue
      // order).  Issue order of a wave: batch 0 | the next item's prefetch (NPF requests, if there is a next
      // item) | batch 1 (RES 5: batches 1-3) | then per store position its stores, (RES 2) with a new batch
      // in front of them at the positions that free a bank.  `young` = requests issued after the batch that
      // is waited for.
      constexpr int NPF = 2 * PI + 3 * SI;
      [[maybe_unused]] const bool prefetched = more && np > 0;
      // (the prefetch is younger than batch 0 only)
      [[maybe_unused]] auto batch_wait = [&](int young, bool first) {
        if (first && prefetched) wait_vm_sw<0, 63>(young + NPF); else wait_vm_sw<0, 63>(young);
      };
      const int cl = lane % LPR;
      const int wrow = lane & 15;
      const unsigned waddr = stage_lds + wrow * 256;
      TO *crow = C + (m0 + wm * (MI * 16) + lane / LPR) * g.ldc + nw + cl * CPL;
      if constexpr (RES == 4)   // the tile's 16-column groups are scattered over the q/k/v thirds (32-byte runs)
        crow = C + (m0 + wm * (MI * 16) + lane / LPR) * g.ldc + rope_group_col(tid_cur_tn, wn * 4 + (cl >> 1), g.rope_D, g.rope_hd) + (cl & 1) * 8;
      const int64_t rstep = (int64_t)RPP * g.ldc;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        if constexpr (RES == 4) {
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const f4_t x1 = acc[ni][mi] + bag[ni], x2 = acc[ni + 2][mi] + bag[ni + 2];
            const f4_t y1 = x1 * rcs[ni] - x2 * rsn[ni], y2 = x2 * rcs[ni] + x1 * rsn[ni];
            asm volatile("ds_write_b128 %0, %1" ::"v"(waddr + (((ni * 4 + (lane >> 4)) ^ wrow) << 4)), "v"(y1) : "memory");
            asm volatile("ds_write_b128 %0, %1" ::"v"(waddr + ((((ni + 2) * 4 + (lane >> 4)) ^ wrow) << 4)), "v"(y2) : "memory");
            const f4_t c2 = rcs[ni] * rstc[ni] - rsn[ni] * rsts[ni];
            rsn[ni] = rsn[ni] * rstc[ni] + rcs[ni] * rsts[ni];
            rcs[ni] = c2;
          }
        } else {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int c4 = ni * 4 + (lane >> 4);
          f4_t a = acc[ni][mi] * g.alpha;
          asm volatile("ds_write_b128 %0, %1" ::"v"(waddr + ((c4 ^ wrow) << 4)), "v"(a) : "memory");
        }
        }
#pragma unroll
        for (int it = 0; it < 16 / RPP; ++it) {
          const int row = it * RPP + lane / LPR;
          f4_t t[CPL / 4];
#pragma unroll
          for (int q = 0; q < CPL / 4; ++q) {
            const int c4 = cl * (CPL / 4) + q;
            asm volatile("ds_read_b128 %0, %1" : "=v"(t[q]) : "v"(stage_lds + row * 256 + ((c4 ^ row) << 4)) : "memory");
          }
          if (CPL == 8) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[CPL / 4 - 1])::"memory");
            float o8[8] = {t[0][0] + bv[0], t[0][1] + bv[1], t[0][2] + bv[2], t[0][3] + bv[3],
                           t[CPL / 4 - 1][0] + bv[CPL - 4], t[CPL / 4 - 1][1] + bv[CPL - 3],
                           t[CPL / 4 - 1][2] + bv[CPL - 2], t[CPL / 4 - 1][3] + bv[CPL - 1]};
            if (SWI_FAST) {
              constexpr int IPM = 16 / RPP;
              const int ri = mi * IPM + it;                 // store position; batches of two positions
              const int bank = (ri >> 1) & 1;
              if ((ri & 1) == 0) {
                // batch ri/2: behind it went (batch 0) the prefetch and batch 1; (batch 1) the stores of two
                // positions and batch 2; (later ones) the stores of three positions and a batch; (the last)
                // the stores of three positions
                batch_wait(ri == 0 ? 4 : ri == 2 ? 8 : ri == 2 * MI - 2 ? 6 : 10, ri == 0);
                asm volatile("" : "+v"(rr[bank * RQ]), "+v"(rr[bank * RQ + 1]), "+v"(rr[bank * RQ + 2]), "+v"(rr[bank * RQ + 3]));
              }
              const u4v_t av = rr[bank * RQ + (ri & 1)], gv = rr[bank * RQ + 2 + (ri & 1)];
              float dg8[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const unsigned aw = av[j >> 1], gw = gv[j >> 1];
                const float a_ = (j & 1) ? __uint_as_float(aw & 0xffff0000u) : __uint_as_float(aw << 16);
                const float g_ = (j & 1) ? __uint_as_float(gw & 0xffff0000u) : __uint_as_float(gw << 16);
                const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-g_));   // 1 ulp; the output is bf16
                const float dh = o8[j];
                o8[j] = dh * (g_ * sg);
                dg8[j] = dh * a_ * sg * (1.0f + g_ * (1.0f - sg));
              }
              if ((ri & 1) && ri + 3 < 2 * MI) {             // this bank is free: the batch after next
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                  gload16(rr[bank * RQ + i], rrow + (int64_t)(ri + 3 + i) * RPP * g.ldr, rlane);
                  gload16(rr[bank * RQ + 2 + i], rrow + (int64_t)(ri + 3 + i) * RPP * g.ldr + g.gate_off, rlane);
                }
              }
              u4v_t pg;
              pg[0] = pack_bf16x2(dg8[0], dg8[1]);
              pg[1] = pack_bf16x2(dg8[2], dg8[3]);
              pg[2] = pack_bf16x2(dg8[4], dg8[5]);
              pg[3] = pack_bf16x2(dg8[6], dg8[7]);
              __builtin_nontemporal_store(pg, reinterpret_cast<u4v_t *>(crow + g.gate_off));
            }
            if constexpr (RES_FULL) {
              // residual dropout (src/transformer.py:352,361: src + self.dropout(sublayer)): the mask of
              // ldm_dropout_add -- keep(row, column) of common.h in the 0xffffffff stream -- on the product
              // + bias, before the residual joins
              if (g.drop_thresh) {
                const unsigned row = (unsigned)(m0 + wm * (MI * 16) + mi * 16 + it * RPP + lane / LPR);
                const unsigned rk = ldm_drop_row(ldm_drop_bh(g.drop_lo, 0xffffffffu), g.drop_hi, row);
                const unsigned c0 = (unsigned)(nw + cl * CPL);
#pragma unroll
                for (int j = 0; j < 8; ++j) o8[j] = ldm_drop_keep(rk, c0 + j, g.drop_thresh) ? o8[j] * g.drop_inv : 0.f;
              }
            }
            if (RES_FAST && res_fast) {
              constexpr int IPM = 16 / RPP;                 // store instructions per 16-row fragment
              const int ri = mi * IPM + it;                 // index of this store = index of its residual load
              const int bank = RES_FULL ? ri / RQ : (ri / RQ) & 1;
              if constexpr (RES_FULL) {
                if (ri % RQ == 0) {
                  // batch b = ri/RQ: behind it went the later batches and the stores so far (and, behind batch
                  // 0, the prefetch): 3 RQ - b RQ + ri = 3 RQ requests
                  static_assert(!RES_FULL || RQ == 4, "four batches of four");
                  batch_wait(3 * RQ, ri == 0);
                  asm volatile("" : "+v"(rr[bank * RQ]), "+v"(rr[bank * RQ + 1]), "+v"(rr[bank * RQ + 2]), "+v"(rr[bank * RQ + 3]));
                }
              }
              const u4v_t rv = rr[bank * RQ + ri % RQ];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                o8[2 * j] += __uint_as_float(rv[j] << 16);
                o8[2 * j + 1] += __uint_as_float(rv[j] & 0xffff0000u);
              }
              // the bank just emptied takes the batch after next (issued after this batch's last use)
              if (!RES_FULL && ri % RQ == RQ - 1 && ri + 1 + RQ < 2 * MI) {
#pragma unroll
                for (int i = 0; i < RQ; ++i)
                  rr[bank * RQ + i] = *reinterpret_cast<const u4v_t *>((const char *)(rrow + (int64_t)(ri + 1 + RQ + i) * RPP * g.ldr) + rlane);
              }
            }
            // non-temporal: the tile is not read again by this kernel, and keeping it out of the
            // way of the operand panels in L2 is worth +17 % on the LM-head GEMM (2 GB of logits)
            u4v_t pk;
            pk[0] = pack_bf16x2(o8[0], o8[1]);
            pk[1] = pack_bf16x2(o8[2], o8[3]);
            pk[2] = pack_bf16x2(o8[4], o8[5]);
            pk[3] = pack_bf16x2(o8[6], o8[7]);
            // (RES 4 writes 32- and 64-byte runs: plain stores, so that L2 can merge them into full lines)
            if constexpr (RES == 4) *reinterpret_cast<u4v_t *>(crow) = pk;
            else __builtin_nontemporal_store(pk, reinterpret_cast<u4v_t *>(crow));
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0])::"memory");
            st4(crow, f4{{t[0][0] + bv[0], t[0][1] + bv[1], t[0][2] + bv[2], t[0][3] + bv[3]}});
          }
          crow += rstep;
        }
      }
      carry_next = more;
#ifdef LDM_GEMM_TIMING
      if (!more && g.timing && lane == 0 && blockIdx.x == 17)
        for (int i = 0; i < 8; ++i) g.timing[wave * 8 + i] = tacc[i];
#endif
      if (!more) break;
      continue;
    }
    // Epilogue through LDS, 16 rows x 64 columns of fp32 (4 KiB, private to the wave) at a
    // time: fragments go in as 16-byte pieces at chunk c4 ^ row, rows come out contiguous.
    float *cs = reinterpret_cast<float *>(smem + STAGE_OFF + wave * 4096);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int row = lane & 15;
        const int c4 = ni * 4 + (lane >> 4);
        const f4_t a = acc[ni][mi];
        *reinterpret_cast<float4 *>(cs + row * 64 + ((c4 ^ row) << 2)) =
            make_float4(a[0] * g.alpha, a[1] * g.alpha, a[2] * g.alpha, a[3] * g.alpha);
      }
      const int64_t mw = m0 + wm * (MI * 16) + afrag(mi) * 16;
      if (EPI == 1) {
        float *C = (float *)g.C;
        const int64_t n = nw + lane;
#pragma unroll 4
        for (int row = 0; row < 16; ++row) {
          const float v = cs[row * 64 + ((((lane >> 2) ^ row) << 2) | (lane & 3))];
          if (mw + row < Mdyn && n < g.N) atomicAdd(C + (mw + row) * g.ldc + n, v);   // 256 contiguous bytes
        }
      } else {
        TO *C = (TO *)g.C;
        const TO *R = (const TO *)g.R;
        const bool vec_ok = (g.ldc % CPL == 0) && (!R || g.ldr % CPL == 0);
        const int cl = lane % LPR;                 // this lane's column group
        const int64_t n = nw + cl * CPL;
#pragma unroll
        for (int it = 0; it < 16 / RPP; ++it) {
          const int row = it * RPP + lane / LPR;
          float v[CPL];
#pragma unroll
          for (int q = 0; q < CPL / 4; ++q) {
            const int c4 = cl * (CPL / 4) + q;
            const float4 t = *reinterpret_cast<const float4 *>(cs + row * 64 + ((c4 ^ row) << 2));
            v[q * 4 + 0] = t.x + bv[q * 4 + 0];
            v[q * 4 + 1] = t.y + bv[q * 4 + 1];
            v[q * 4 + 2] = t.z + bv[q * 4 + 2];
            v[q * 4 + 3] = t.w + bv[q * 4 + 3];
          }
          int64_t m = mw + row;
          if (m >= Mdyn || n >= g.N) continue;
          if (g.row_map) m = g.row_map[m];
          if (n + CPL - 1 < g.N && vec_ok) {
            if (CPL == 8) {
              if (R) {
                Raw8<TO> r;
                r.load(R + m * g.ldr + n);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += r.get(j);
              }
              float o[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) o[j] = v[j % CPL];
              Raw8<TO>::store(C + m * g.ldc + n, o);
            } else {
              f4 o{{v[0], v[1], v[2], v[3]}};
              if (R) {
                const f4 r = ld4(R + m * g.ldr + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) o.v[j] += r.v[j];
              }
              st4(C + m * g.ldc + n, o);
            }
          } else {
#pragma unroll
            for (int j = 0; j < CPL; ++j)
              if (n + j < g.N) {
                float x = v[j];
                if (R) x += Elt<TO>::ld(R + m * g.ldr + n + j);
                Elt<TO>::st(C + m * g.ldc + n + j, x);
              }
          }
        }
      }
    }
#ifdef LDM_GEMM_TIMING
      if (!more && g.timing && lane == 0 && blockIdx.x == 17)
        for (int i = 0; i < 8; ++i) g.timing[wave * 8 + i] = tacc[i];
#endif
    if (!more) break;
  }
}

int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// Experiments: LDM_GEMM_BAND_MAP="res:epi:K:N=band,..." (N = 0: any) overrides the band height per call site.
int band_override(int res, int epi, int64_t K, int64_t N) {
  struct Ent { int res, epi; long long k, n; int band; };
  static std::vector<Ent> *tab = nullptr;
  static bool parsed = false;
  if (!parsed) {
    parsed = true;
    if (const char *e = getenv("LDM_GEMM_BAND_MAP")) {
      tab = new std::vector<Ent>();
      const char *p = e;
      while (*p) {
        Ent t;
        int used = 0;
        if (sscanf(p, "%d:%d:%lld:%lld=%d%n", &t.res, &t.epi, &t.k, &t.n, &t.band, &used) == 5) tab->push_back(t);
        else break;
        p += used;
        if (*p == ',') ++p;
      }
    }
  }
  if (!tab) return 0;
  for (const Ent &t : *tab)
    if (t.res == res && t.epi == epi && t.k == K && (t.n == 0 || t.n == N)) return t.band;
  return 0;
}

template <bool A_KC, bool B_KC, int BN, typename TO, int EPI, int RES = 0>
int launch256(const MArgs &g_in, hipStream_t s) {
  MArgs g = g_in;
  if (const int b = band_override(RES, EPI, g.K, g.N)) g.band_m = b;
  static std::atomic<uint64_t> attr_done{0};
  uint64_t dev_bit;
  if (ldm_attr_pending(attr_done, &dev_bit)) {
    (void)hipFuncSetAttribute((const void *)gemm256_k<A_KC, B_KC, BN, TO, EPI, RES>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done.fetch_or(dev_bit, std::memory_order_release);
  }
  const int64_t items = (int64_t)g.tiles_m * g.tiles_n * g.splits;
  const int64_t nwg = items < num_cus() ? items : num_cus();   // persistent: one workgroup per CU
  hipLaunchKernelGGL((gemm256_k<A_KC, B_KC, BN, TO, EPI, RES>), dim3((unsigned)nwg), dim3(512), LDS_BYTES, s, g);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

template <int BN, typename TO, int EPI>
int dispatch_layout(const MArgs &g, bool akc, bool bkc, hipStream_t s) {
  if (akc && bkc) return launch256<true, true, BN, TO, EPI>(g, s);
  if (akc && !bkc) return launch256<true, false, BN, TO, EPI>(g, s);
  if (!akc && bkc) return launch256<false, true, BN, TO, EPI>(g, s);
  return launch256<false, false, BN, TO, EPI>(g, s);
}

int gm_band_env() {
  static const int v = getenv("LDM_GEMM_BAND") ? atoi(getenv("LDM_GEMM_BAND")) : 0;
  return v;
}

int fill(MArgs &g, int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const void *A,
         int64_t lda, const void *B, int64_t ldb, void *C, int64_t ldc, const float *bias, const void *R,
         int64_t ldr) {
  if (K % KH || K <= 0) return LDM_E_SHAPE;
  if (lda % 8 || ldb % 8) return LDM_E_ALIGN;
  if (!aligned_to(A, 16) || !aligned_to(B, 16) || !aligned_to(C, 16)) return LDM_E_ALIGN;
  if (bias && !aligned_to(bias, 16)) return LDM_E_ALIGN;
  if (R && !aligned_to(R, 16)) return LDM_E_ALIGN;
  if (transA && (M < 8 || M % 8)) return LDM_E_SHAPE;
  if (!transB && (N < 8 || N % 8)) return LDM_E_SHAPE;
  g.M = M; g.N = N; g.K = K; g.alpha = alpha;
  g.A = (const bf16_t *)A; g.lda = lda; g.B = (const bf16_t *)B; g.ldb = ldb;
  g.C = C; g.ldc = ldc; g.bias = bias; g.R = R; g.ldr = ldr;
  g.gate_off = 0;
  g.drop_thresh = g.drop_lo = g.drop_hi = 0; g.drop_inv = 1.0f;
  g.C2 = nullptr; g.ldc2 = 0;
  g.rope_cos = g.rope_sin = g.rope_step = nullptr; g.rope_hd = 0; g.rope_S = 0; g.rope_D = 0; g.rope_qscale = 1.0f;
  g.splits = 1;
  // m-tiles per band of the tile walk: the band's A panel (band x 256 rows x K) is sized to about 6 MB,
  // within 2..16 (measured on one box at M = 262144: K = 768 prefers 8-16 m-tiles, -11 % at N = 2048 and
  // -6 % at N = 1024 against 4; K >= 2048 prefers 2-4; experiments: LDM_GEMM_BAND)
  const int gm_env = gm_band_env();
  {
    int b = 16;
    while (b > 2 && (int64_t)b * 512 * K > (int64_t)8 << 20) b >>= 1;
    g.band_m = gm_env > 0 ? gm_env : b;
  }
  g.dyn_m = g.dyn_k = g.row_map = nullptr;
  g.rowsum = nullptr;
  static const char *tb = getenv("LDM_GEMM_TIMING_BUF");
  g.timing = tb ? (unsigned long long *)strtoull(tb, nullptr, 0) : nullptr;
  return LDM_OK;
}

// wave-quantisation efficiency of `tiles` workgroups on 256 CUs
double quant_eff(int64_t tiles) { return (double)tiles / (double)(ceil_div(tiles, 256) * 256); }

}  // namespace

// Picks the N tile: 256 unless 128 fills the 256 CUs distinctly better.  Returns
// LDM_E_SHAPE for problems too small to be worth a 256-row tile (caller falls back).
int ldm_gemm_mfma256_bf16(int out_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                          const void *A, int64_t lda, const void *B, int64_t ldb, void *C, int64_t ldc,
                          const float *bias, const void *R, int64_t ldr, hipStream_t s) {
  // the four-wave kernel takes what it can (K-contiguous operands, full tiles; a bf16 residual with C's row stride)
  if (ldm_gemm_4w_enabled() && out_dtype == LDM_BF16 && !transA && transB && (!R || ldr == ldc) && alpha == 1.0f) {
    const int rc4 = ldm_gemm_mfma4w_bf16(M, N, K, A, lda, B, ldb, C, ldc, bias, R, nullptr, s);
    if (rc4 != LDM_E_SHAPE && rc4 != LDM_E_ALIGN) return rc4;
  }
  MArgs g;
  const int rc = fill(g, transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, R, ldr);
  if (rc) return rc;
  if (M < 256 || N < 128) return LDM_E_SHAPE;
  const int64_t tm = ceil_div(M, BM);
  const int64_t t256 = tm * ceil_div(N, 256), t128 = tm * ceil_div(N, 128);
  if (t128 > 0x3fffffff) return LDM_E_SHAPE;
  // 256x128 tiles stream 1.5x the bytes per flop of 256x256: only when they fill the chip better
  const bool use128 = N < 256 || quant_eff(t128) * 0.85 > quant_eff(t256) * ((double)N / (ceil_div(N, 256) * 256.0));
  g.tiles_m = (int)tm;
  g.tiles_n = (int)ceil_div(N, use128 ? 128 : 256);
  const bool akc = !transA, bkc = transB != 0;
  if (out_dtype == LDM_BF16) {
    // projection + bias + residual (out-projection, MLP down-projection): the kernel variant whose
    // epilogue streams the residual rows in around the next tile's prefetch
    if (!use128 && R && akc && bkc && ldr % 8 == 0 && ldc % 8 == 0) {
      // every tile full and aligned: the variant with nothing but the streaming epilogue
      if (M % 256 == 0 && N % 256 == 0 && ldr < (1 << 27) && aligned_to(C, 16) && aligned_to(R, 16) && !g.row_map)
        return launch256<true, true, 256, bf16_t, 0, 5>(g, s);
      return launch256<true, true, 256, bf16_t, 0, 1>(g, s);
    }
    return use128 ? dispatch_layout<128, bf16_t, 0>(g, akc, bkc, s) : dispatch_layout<256, bf16_t, 0>(g, akc, bkc, s);
  }
  if (out_dtype == LDM_F32)
    return use128 ? dispatch_layout<128, float, 0>(g, akc, bkc, s) : dispatch_layout<256, float, 0>(g, akc, bkc, s);
  return LDM_E_DTYPE;
}

// C[M, N] = R + dropout(A[M, K] * W[N, K]^T + bias) (src/transformer.py:352,361 around the out- and the
// down-projection): the residual-dropout mask of ldm_dropout_add applied in the epilogue of the RES 5
// variant.  Only what that variant takes (bf16, every tile full and aligned); otherwise LDM_E_SHAPE /
// LDM_E_ALIGN and the caller runs ldm_gemm and ldm_dropout_add.
int ldm_gemm_mfma256_bf16_drop_res(int64_t M, int64_t N, int64_t K, const void *A, int64_t lda, const void *W,
                                   int64_t ldw, void *C, int64_t ldc, const float *bias, const void *R, int64_t ldr,
                                   unsigned thresh, float inv, uint64_t seed, hipStream_t s) {
  MArgs g;
  const int rc = fill(g, 0, 1, M, N, K, 1.0f, A, lda, W, ldw, C, ldc, bias, R, ldr);
  if (rc) return rc;
  if (M < 256 || M % 256 || N < 256 || N % 256 || M > 0xffffffffll || ldr >= (1 << 27)) return LDM_E_SHAPE;
  if (!R || ldr % 8 || ldc % 8 || !aligned_to(C, 16) || !aligned_to(R, 16)) return LDM_E_ALIGN;
  const int64_t tm = M / 256, tn = N / 256;
  if (tm * tn > 0x3fffffff) return LDM_E_SHAPE;
  g.tiles_m = (int)tm;
  g.tiles_n = (int)tn;
  g.drop_thresh = thresh; g.drop_inv = inv;
  g.drop_lo = (unsigned)seed; g.drop_hi = (unsigned)(seed >> 32);
  return launch256<true, true, 256, bf16_t, 0, 5>(g, s);
}

// dag[T, 2F] = [ dh * silu(g) | dh * a * silu'(g) ] with dh = dy[T, K] * W[K, F] never written to memory:
// the SwiGLU backward (src/transformer.py:93 under autograd) as the epilogue of the down-projection's
// input-gradient GEMM.  ag = [a | g] and dag are [T, 2F] (row strides ld_ag, ld_dag).  Full 256 x 256
// tiles only (T % 256 == 0, F % 256 == 0); anything else returns LDM_E_SHAPE and the caller runs the
// GEMM and ldm_swiglu_bwd separately.
int ldm_gemm_mfma256_bf16_swiglu_bwd(int64_t T, int64_t F, int64_t K, const void *dy, int64_t ld_dy, const void *W,
                                     int64_t ldw, const void *ag, int64_t ld_ag, void *dag, int64_t ld_dag,
                                     hipStream_t s) {
  MArgs g;
  const int rc = fill(g, 0, 0, T, F, K, 1.0f, dy, ld_dy, W, ldw, dag, ld_dag, nullptr, ag, ld_ag);
  if (rc) return rc;
  if (T < 256 || T % 256 || F < 256 || F % 256) return LDM_E_SHAPE;
  if (ld_ag >= (1 << 27)) return LDM_E_SHAPE;   // (the epilogue's per-lane byte offsets into ag are 32-bit)
  if (ld_ag % 8 || ld_dag % 8 || ld_ag < 2 * F || ld_dag < 2 * F || !aligned_to(ag, 16) || !aligned_to(dag, 16)) return LDM_E_ALIGN;
  const int64_t tm = T / 256, tn = F / 256;
  if (tm * tn > 0x3fffffff) return LDM_E_SHAPE;
  g.gate_off = F;
  g.tiles_m = (int)tm;
  g.tiles_n = (int)tn;
  return launch256<true, false, 256, bf16_t, 0, 2>(g, s);
}

// SwiGLU forward inside the gate/up projection: ag = x W12^T + b12 ([a | g], 2F columns) AND
// h = a * silu(g) ([T, F]) from one GEMM.  W12 = [w1; w2] ([2F, K], K contiguous), bias fp32 [2F] or null.
int ldm_gemm_mfma256_bf16_swiglu_fwd(int64_t T, int64_t F, int64_t K, const void *x, int64_t ldx, const void *W12,
                                     int64_t ldw, const float *bias, void *ag, int64_t ld_ag, void *h, int64_t ld_h,
                                     hipStream_t s) {
  MArgs g;
  const int rc = fill(g, 0, 1, T, 2 * F, K, 1.0f, x, ldx, W12, ldw, ag, ld_ag, bias, nullptr, 0);
  if (rc) return rc;
  if (T < 256 || T % 256 || F < 128 || F % 128) return LDM_E_SHAPE;
  if (ld_ag % 8 || ld_h % 8 || ld_ag < 2 * F || ld_h < F || !aligned_to(h, 16)) return LDM_E_ALIGN;
  const int64_t tm = T / 256, tn = F / 128;
  if (tm * tn > 0x3fffffff) return LDM_E_SHAPE;
  g.gate_off = F;
  g.C2 = h; g.ldc2 = ld_h;
  g.tiles_m = (int)tm;
  g.tiles_n = (int)tn;
  return launch256<true, true, 256, bf16_t, 0, 3>(g, s);
}

// q/k/v projection with the rotary embedding and the query scale applied in the epilogue:
// qkv[T, 3D] = x W^T + b, then rotate-half RoPE on every head of the q and k thirds (position = row % S)
// and q *= q_scale.  Full tiles only.
int ldm_gemm_mfma256_bf16_qkv_rope(int64_t T, int64_t D, int64_t K, int64_t hd, int64_t S, const void *x, int64_t ldx,
                                   const void *W, int64_t ldw, const float *bias, const float *cos_t, const float *sin_t,
                                   const float *step, float q_scale, void *qkv, int64_t ld_qkv, hipStream_t s) {
  MArgs g;
  const int rc = fill(g, 0, 1, T, 3 * D, K, 1.0f, x, ldx, W, ldw, qkv, ld_qkv, bias, nullptr, 0);
  if (rc) return rc;
  if (T < 256 || T % 256 || D < 256 || D % 256) return LDM_E_SHAPE;
  if (!(hd == 32 || hd == 64 || hd == 96 || hd == 128) || D % hd) return LDM_E_SHAPE;
  // a wave's 128 rows must lie inside one sequence (the angle-addition walk does not wrap)
  if (S < 128 || S % 128 || T % S) return LDM_E_SHAPE;
  if (ld_qkv % 8 || ld_qkv < 3 * D) return LDM_E_ALIGN;
  if (!cos_t || !sin_t || !step || !aligned_to(cos_t, 16) || !aligned_to(sin_t, 16) || !aligned_to(step, 16)) return LDM_E_ALIGN;
  const int64_t tm = T / 256, tn = 3 * D / 256;
  if (tm * tn > 0x3fffffff) return LDM_E_SHAPE;
  g.rope_cos = cos_t; g.rope_sin = sin_t; g.rope_step = step;
  g.rope_hd = (int)hd; g.rope_S = (int)S; g.rope_D = D; g.rope_qscale = q_scale;
  // narrow bands: the attention kernel that follows reads q, k, v of whole sequences, and finds more of
  // them in cache when the rows were written band by band (measured in the step: 5.45 against 5.56 ms)
  if (gm_band_env() <= 0) g.band_m = 4;
  g.tiles_m = (int)tm;
  g.tiles_n = (int)tn;
  return launch256<true, true, 256, bf16_t, 0, 4>(g, s);
}

// Forms with device-side sizes (see MArgs): used for the LM-head backward over the compacted
// masked rows.  dyn_m/row_map apply to the store form, dyn_k to the accumulate form.
int ldm_gemm_mfma256_bf16_dyn(int out_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                              const void *A, int64_t lda, const void *B, int64_t ldb, void *C, int64_t ldc,
                              const float *bias, const int *dyn_m, const int *row_map, hipStream_t s) {
  // the four-wave kernel: K-contiguous operands, full tiles (A holds all M rows), rows in place.  Long
  // reductions only: at the vocabulary projection's forward shape (N = 30720, K = 768, output rows 60 KiB
  // apart) it measured 7 % SLOWER than the eight-wave kernel (6.35 against 5.91 ms mean over the same seven
  // steps), at its input-gradient shape (N = 768, K = 30720) 11 % faster (7.8 against 8.8 ms at the same count)
  if (ldm_gemm_4w_enabled() && out_dtype == LDM_BF16 && !transA && transB && !row_map && alpha == 1.0f && K >= 2048) {
    const int rc4 = ldm_gemm_mfma4w_bf16(M, N, K, A, lda, B, ldb, C, ldc, bias, nullptr, dyn_m, s);
    if (rc4 != LDM_E_SHAPE && rc4 != LDM_E_ALIGN) return rc4;
  }
  MArgs g;
  const int rc = fill(g, transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, nullptr, 0);
  if (rc) return rc;
  if (transA || M < 256 || N < 128) return LDM_E_SHAPE;
  g.dyn_m = dyn_m; g.row_map = row_map;
  // 128-wide tiles only where a 256-wide last column would waste a visible share of the work
  const bool use128 = N < 256 || (N % 256 != 0 && N % 256 <= 128 && N < 4096) ||
                      ceil_div(M, BM) * ceil_div(N, 256) < 512;
  g.tiles_m = (int)ceil_div(M, BM);
  g.tiles_n = (int)ceil_div(N, use128 ? 128 : 256);
  const bool bkc = transB != 0;
  if (out_dtype == LDM_BF16)
    return use128 ? dispatch_layout<128, bf16_t, 0>(g, true, bkc, s) : dispatch_layout<256, bf16_t, 0>(g, true, bkc, s);
  if (out_dtype == LDM_F32)
    return use128 ? dispatch_layout<128, float, 0>(g, true, bkc, s) : dispatch_layout<256, float, 0>(g, true, bkc, s);
  return LDM_E_DTYPE;
}

int ldm_gemm_mfma256_bf16_acc_dynk(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                                   const void *A, int64_t lda, const void *B, int64_t ldb, float *C, int64_t ldc,
                                   const int *dyn_k, float *rowsum, hipStream_t s) {
  MArgs g;
  const int rc = fill(g, transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0);
  if (rc) return rc;
  if (M < 256 || N < 128) return LDM_E_SHAPE;
  const bool use128 = N < 256 || (N % 256 != 0 && N % 256 <= 128);
  g.tiles_m = (int)ceil_div(M, BM);
  g.tiles_n = (int)ceil_div(N, use128 ? 128 : 256);
  const int64_t nt = (int64_t)g.tiles_m * g.tiles_n;
  if (nt < 200) return LDM_E_SHAPE;
  // K splits (the kernel divides the device-side K among them) only to even out the rounds of the
  // 256 persistent workgroups: 360 tiles are 1.4 rounds, 720 half-K items are 2.8
  int best = 1;
  double best_eff = quant_eff(nt);
  for (int sp = 2; sp <= 3; ++sp) {
    const double e = quant_eff(nt * sp) - 0.03 * (sp - 1);   // each split adds one pass of float atomics
    if (e > best_eff + 0.05) { best = sp; best_eff = e; }
  }
  g.splits = best;
  g.dyn_k = dyn_k;
  g.rowsum = rowsum;
  if (gm_band_env() <= 0) g.band_m = 4;   // few, long work items: the band height does not matter (measured)
  const bool akc = !transA, bkc = transB != 0;
  return use128 ? dispatch_layout<128, float, 1>(g, akc, bkc, s) : dispatch_layout<256, float, 1>(g, akc, bkc, s);
}

int ldm_gemm_mfma256_bf16_acc(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const void *A,
                              int64_t lda, const void *B, int64_t ldb, float *C, int64_t ldc, float *rowsum,
                              hipStream_t s) {
  // (A four-wave form of this product -- transposing reads, 32 per step and wave -- was built and measured:
  // equal to this kernel at the q/k/v and gate/up shapes, 20-40 % slower at the 768-wide ones and at the
  // shapes of the larger models; scripts/micro/gemm4w_st_probe.hip keeps it.)
  MArgs g;
  const int rc = fill(g, transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0);
  if (rc) return rc;
  if (M < 256 || N < 128) return LDM_E_SHAPE;
  const int64_t tm = ceil_div(M, BM);
  const bool use128 = N < 256 || (N % 256 != 0 && N % 256 <= 128);
  const int64_t tn = ceil_div(N, use128 ? 128 : 256);
  const int64_t ntiles = tm * tn;
  if (ntiles > 0x3fffffff) return LDM_E_SHAPE;
  const int64_t np_total = K / KH;
  // Small outputs go to the 128x128 kernel: filling 256 CUs with 256x256 tiles needs so many
  // K splits that the float-atomic epilogue (~1.3 TB/s chip-wide) costs more than it saves.
  static const bool force = getenv("LDM_GEMM_TILE") && atoi(getenv("LDM_GEMM_TILE")) == 256;
  // K splits: each work item costs its half-tiles (~1.6k cycles each) plus an epilogue of float
  // atomics (~16k cycles); the persistent workgroups run ceil(items / CUs) rounds of them.  The
  // split count that minimises rounds x item cost evens out the last round (192 tiles on 256 CUs:
  // 4 splits = exactly 3 rounds); a larger count must win by 3 % to be taken.
  int64_t splits = 1;
  {
    const int64_t cus = num_cus();
    const int64_t smax = np_total / 8 < 64 ? np_total / 8 : 64;   // >= 8 half-tiles per work item
    double best = 0.0;
    for (int64_t sp = 1; sp <= smax; ++sp) {
      const double cost = (double)ceil_div(ntiles * sp, cus) * ((double)ceil_div(np_total, sp) * 1600.0 + 16000.0);
      if (sp == 1 || cost < best * 0.97) { best = cost; splits = sp; }
    }
  }
  if (const char *e = getenv("LDM_GEMM_SPLITS")) { if (atoi(e) > 0) splits = atoi(e); }   // experiments only
  // measured: with >= 9 tiles and >= 64 half-tiles per work item the persistent 256-row kernel
  // wins (861 vs 625 TFLOP/s at 2304x768x65536; equal to the 128x128 kernel at 768x768x65536 on
  // cache-warm operands, ~15 % faster inside the training step, where the operands come from HBM
  // and the smaller tile streams twice the bytes per flop); below that the atomics dominate
  const bool worth = ntiles >= 200 || (ntiles >= 9 && np_total / splits >= 64);
  if (!worth && !force) return LDM_E_SHAPE;
  g.tiles_m = (int)tm; g.tiles_n = (int)tn; g.splits = (int)splits;
  g.rowsum = rowsum;
  if (gm_band_env() <= 0) g.band_m = 4;   // few, long work items: the band height does not matter (measured)
  const bool akc = !transA, bkc = transB != 0;
  return use128 ? dispatch_layout<128, float, 1>(g, akc, bkc, s) : dispatch_layout<256, float, 1>(g, akc, bkc, s);
}
