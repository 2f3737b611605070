// bf16 MFMA GEMM for gfx950: 128x128x64 tiles, 4 waves (2x2), each wave a 64x64
// output as 4x4 fragments of v_mfma_f32_16x16x32_bf16; both operand tiles are
// streamed global -> LDS with 16-byte LDS-DMA loads (double buffered) and the
// LDS images are XOR-swizzled on the SOURCE address so that fragment reads are
// bank-conflict free.
//
//   C[M,N] = alpha * opA(A) * opB(B) (+ bias[N]) (+ R[M,N])
//
// Operand storage (per operand, chosen at compile time):
//   K-contiguous ("KC"): rows of the operand tile hold consecutive k
//       A stored [M][K] (transA = 0),  B stored [N][K] (transB = 1)
//       LDS image [128 rows][64 k] (128-B rows), fragments by ds_read_b128
//   strided: consecutive m (or n) are contiguous, k is the row index
//       A stored [K][M] (transA = 1),  B stored [K][N] (transB = 0)
//       LDS image [64 k][128 cols] (256-B rows), fragments by two
//       ds_read_b64_tr_b16 (hardware transposing read)
//
// The MFMA is issued with the operands swapped (B-fragment as the A operand) so
// that each lane ends up with 4 CONSECUTIVE n of one m: the epilogue then moves
// 8 B (bf16) or 16 B (fp32) per lane.
//
// Requirements (checked by the host wrapper): K % 64 == 0, lda/ldb % 8 == 0,
// A/B 16-byte aligned, M >= 1, N >= 8 for strided B, M >= 8 for strided A.
#include "common.h"
#include "gemm.h"
#include <stdlib.h>

namespace {

typedef short s4_t __attribute__((ext_vector_type(4)));
typedef short s8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;   // one operand tile image (either layout)

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

struct MArgs {
  int64_t M, N, K;
  float alpha;
  const bf16_t *A; int64_t lda;
  const bf16_t *B; int64_t ldb;
  void *C; int64_t ldc;
  const float *bias;
  const void *R; int64_t ldr;
  int tiles_m, tiles_n, splits;
  float *rowsum;   // EPI 1 only: rowsum[m] += alpha * sum_k opA[m, k], by the first tile column
};

__device__ __forceinline__ int swz_tr(int kr) { return ((kr & 3) << 2) | ((kr >> 2) & 3); }

// Per-lane global source pointers for the 4 LDS-DMA pieces this wave issues per tile.
// KC: piece i covers tile rows wave*32 + i*8 .. +7 (8 rows x 128 B = 1 KiB).
// strided: piece i covers k rows wave*16 + i*4 .. +3 (4 rows x 256 B = 1 KiB).
template <bool KC>
__device__ __forceinline__ void stage_ptrs(const bf16_t *P, int64_t ld, int64_t row0, int64_t nrows,
                                           int wave, int lane, const bf16_t *(&src)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (KC) {
      const int r = wave * 32 + i * 8 + (lane >> 3);
      const int lc = (lane & 7) ^ (r & 7);
      int64_t gr = row0 + r;
      if (gr > nrows - 1) gr = nrows - 1;
      src[i] = P + gr * ld + lc * 8;
    } else {
      const int kr = wave * 16 + i * 4 + (lane >> 4);
      const int lc = (lane & 15) ^ swz_tr(kr);
      int64_t col = row0 + lc * 8;
      if (col > nrows - 8) col = nrows - 8;
      src[i] = P + (int64_t)kr * ld + col;
    }
  }
}

template <bool KC>
__device__ __forceinline__ void stage_issue(const bf16_t *(&src)[4], int64_t koff, char *tile, int wave) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    char *dst = tile + (KC ? (wave * 32 + i * 8) * 128 : (wave * 16 + i * 4) * 256);
    __builtin_amdgcn_global_load_lds(GLB_PTR(src[i] + koff), LDS_PTR(dst), 16, 0, 0);
  }
}

// Fragment (16 rows/cols x 32 k) for k-substep ks of the tile; rbase = first row/col.
template <bool KC>
__device__ __forceinline__ bf8_t read_frag(const char *tile, int rbase, int ks, int lane) {
  if (KC) {
    const int r = rbase + (lane & 15);
    const int pc = (ks * 4 + (lane >> 4)) ^ (r & 7);
    return *reinterpret_cast<const bf8_t *>(tile + r * 128 + pc * 16);
  } else {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int lc = (rbase >> 3) + (p >> 1);
    const int kr0 = ks * 32 + 8 * g + q;
    const int kr1 = kr0 + 4;
    const char *a0 = tile + kr0 * 256 + ((lc ^ swz_tr(kr0)) << 4) + ((p & 1) << 3);
    const char *a1 = tile + kr1 * 256 + ((lc ^ swz_tr(kr1)) << 4) + ((p & 1) << 3);
    s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(a0));
    s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t *)(a1));
    s8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf8_t, v);
  }
}

// EPI 0: C = alpha*acc (+bias) (+R), stored as TO.   EPI 1 (split-K): C(fp32) += alpha*acc by
// float atomics, blocks with the same tile and different `split` each take a K range.
template <bool A_KC, bool B_KC, typename TO, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_mfma_k(MArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A tile | B tile]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: branches on it stay uniform (MFMA ignores EXEC)
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware remap (blocks b and b+8 share an XCD's L2): give each of the 8 groups
  // a contiguous run of work items, then walk tiles in bands of 8 m-tiles.
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int ntiles = g.tiles_m * g.tiles_n;
  const int split = bid / ntiles;
  bid -= split * ntiles;
  int tm, tn;
  {
    const int GM = 8;
    const int width = GM * g.tiles_n;
    const int grp = bid / width;
    const int first = grp * GM;
    const int gsz = (g.tiles_m - first < GM) ? g.tiles_m - first : GM;
    const int in = bid - grp * width;
    tm = first + in % gsz;
    tn = in / gsz;
  }
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  // K range of this split, in 64-deep tiles
  const int nt_total = (int)(g.K / BK);
  const int base = nt_total / g.splits, rem = nt_total % g.splits;
  const int t_begin = split * base + (split < rem ? split : rem);
  const int nt = base + (split < rem ? 1 : 0);

  const bf16_t *srcA[4], *srcB[4];
  stage_ptrs<A_KC>(g.A, g.lda, m0, g.M, wave, lane, srcA);
  stage_ptrs<B_KC>(g.B, g.ldb, n0, g.N, wave, lane, srcB);
  const int64_t kstepA = A_KC ? BK : (int64_t)BK * g.lda;
  const int64_t kstepB = B_KC ? BK : (int64_t)BK * g.ldb;

  f4_t acc[4][4];   // [ni][mi]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

  // row sums of opA: wave wn of a first-column tile takes fragments mi = 2*wn, 2*wn + 1
  f4_t racc[2] = {f4_t{0.f, 0.f, 0.f, 0.f}, f4_t{0.f, 0.f, 0.f, 0.f}};
  const bool do_rowsum = EPI == 1 && g.rowsum != nullptr && tn == 0;

  if (nt > 0) {
    stage_issue<A_KC>(srcA, (int64_t)t_begin * kstepA, smem, wave);
    stage_issue<B_KC>(srcB, (int64_t)t_begin * kstepB, smem + TILE_BYTES, wave);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const char *cur = smem + (t & 1) * 2 * TILE_BYTES;
    if (t + 1 < nt) {
      char *nxt = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
      stage_issue<A_KC>(srcA, (int64_t)(t_begin + t + 1) * kstepA, nxt, wave);
      stage_issue<B_KC>(srcB, (int64_t)(t_begin + t + 1) * kstepB, nxt + TILE_BYTES, wave);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf8_t fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = read_frag<A_KC>(cur, wm * 64 + i * 16, ks, lane);
        fb[i] = read_frag<B_KC>(cur + TILE_BYTES, wn * 64 + i * 16, ks, lane);
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ni], fa[mi], acc[ni][mi], 0, 0, 0);
      if (EPI == 1 && do_rowsum) {
        const bf8_t ones = {1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          if ((mi >> 1) == wn)
            racc[mi & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[mi], racc[mi & 1], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next tile landed before anyone reads it
    __syncthreads();
  }

  // Epilogue through LDS.  Each wave re-shapes its own 64x64 fp32 sub-tile in a private
  // 16 KiB region (the staging buffers are free now): fragments go in as 16-byte pieces
  // (lane: row m = mi*16 + (lane&15), 4 columns at ni*16 + (lane>>4)*4), rows come out
  // contiguous.  16-byte column c4 of row r lives at slot c4 ^ (r & 15): conflict free both ways.
  float *cs = reinterpret_cast<float *>(smem + wave * 16384);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int row = mi * 16 + (lane & 15);
      const int c4 = ni * 4 + (lane >> 4);
      float4 v = make_float4(acc[ni][mi][0] * g.alpha, acc[ni][mi][1] * g.alpha, acc[ni][mi][2] * g.alpha,
                             acc[ni][mi][3] * g.alpha);
      *reinterpret_cast<float4 *>(cs + row * 64 + ((c4 ^ (row & 15)) << 2)) = v;
    }
  const int64_t mw = m0 + wm * 64, nw = n0 + wn * 64;
  if (EPI == 1 && do_rowsum && lane < 16) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t m = mw + (wn * 2 + j) * 16 + lane;
      if (m < g.M) atomicAdd(g.rowsum + m, racc[j][0] * g.alpha);
    }
  }
  if (EPI == 1) {
    float *C = (float *)g.C;
    const int64_t n = nw + lane;
#pragma unroll 4
    for (int row = 0; row < 64; ++row) {
      const float v = cs[row * 64 + ((((lane >> 2) ^ (row & 15)) << 2) | (lane & 3))];
      if (mw + row < g.M && n < g.N) atomicAdd(C + (mw + row) * g.ldc + n, v);   // 256 contiguous bytes per wave
    }
  } else {
    TO *C = (TO *)g.C;
    const TO *R = (const TO *)g.R;
    const bool vec_ok = (g.ldc % 4 == 0) && (!R || g.ldr % 4 == 0);
    const int c4 = lane & 15;
    const int64_t n = nw + c4 * 4;
    f4 bv{{0.f, 0.f, 0.f, 0.f}};
    if (g.bias) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n + j < g.N) bv.v[j] = g.bias[n + j];
    }
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int row = it * 4 + (lane >> 4);
      const float4 t = *reinterpret_cast<const float4 *>(cs + row * 64 + ((c4 ^ (row & 15)) << 2));
      const int64_t m = mw + row;
      if (m >= g.M || n >= g.N) continue;
      f4 v{{t.x + bv.v[0], t.y + bv.v[1], t.z + bv.v[2], t.w + bv.v[3]}};
      if (n + 3 < g.N && vec_ok) {
        if (R) {
          const f4 r = ld4(R + m * g.ldr + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) v.v[j] += r.v[j];
        }
        st4(C + m * g.ldc + n, v);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n + j < g.N) {
            float x = v.v[j];
            if (R) x += Elt<TO>::ld(R + m * g.ldr + n + j);
            Elt<TO>::st(C + m * g.ldc + n + j, x);
          }
      }
    }
  }
}

template <bool A_KC, bool B_KC, typename TO, int EPI>
int launch_mfma(const MArgs &g, hipStream_t s) {
  static std::atomic<uint64_t> attr_done{0};
  const size_t lds = 4 * TILE_BYTES;
  uint64_t dev_bit;
  if (ldm_attr_pending(attr_done, &dev_bit)) {
    (void)hipFuncSetAttribute((const void *)gemm_mfma_k<A_KC, B_KC, TO, EPI>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done.fetch_or(dev_bit, std::memory_order_release);
  }
  const int64_t nwg = (int64_t)g.tiles_m * g.tiles_n * g.splits;
  hipLaunchKernelGGL((gemm_mfma_k<A_KC, B_KC, TO, EPI>), dim3((unsigned)nwg), dim3(256), lds, s, g);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

int check_and_fill(MArgs &g, int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const void *A,
                   int64_t lda, const void *B, int64_t ldb, void *C, int64_t ldc, const float *bias,
                   const void *R, int64_t ldr) {
  if (K % BK || K <= 0) return LDM_E_SHAPE;
  if (lda % 8 || ldb % 8) return LDM_E_ALIGN;
  if (!aligned_to(A, 16) || !aligned_to(B, 16) || !aligned_to(C, 16)) return LDM_E_ALIGN;
  if (bias && !aligned_to(bias, 16)) return LDM_E_ALIGN;
  if (R && !aligned_to(R, 16)) return LDM_E_ALIGN;
  if (transA && (M < 8 || M % 8)) return LDM_E_SHAPE;
  if (!transB && (N < 8 || N % 8)) return LDM_E_SHAPE;
  const int64_t tm = ceil_div(M, BM), tn = ceil_div(N, BN);
  if (tm * tn > 0x3fffffff) return LDM_E_SHAPE;
  g.M = M; g.N = N; g.K = K; g.alpha = alpha;
  g.A = (const bf16_t *)A; g.lda = lda; g.B = (const bf16_t *)B; g.ldb = ldb;
  g.C = C; g.ldc = ldc; g.bias = bias; g.R = R; g.ldr = ldr;
  g.tiles_m = (int)tm; g.tiles_n = (int)tn; g.splits = 1;
  g.rowsum = nullptr;
  return LDM_OK;
}

}  // namespace

int ldm_gemm_mfma_bf16(int out_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                       const void *A, int64_t lda, const void *B, int64_t ldb, void *C, int64_t ldc,
                       const float *bias, const void *R, int64_t ldr, hipStream_t s) {
  MArgs g;
  const int rc = check_and_fill(g, transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, R, ldr);
  if (rc) return rc;
  const bool akc = !transA, bkc = transB != 0;
#define GO(AK, BKC)                                                                   \
  return out_dtype == LDM_BF16 ? launch_mfma<AK, BKC, bf16_t, 0>(g, s)                \
         : out_dtype == LDM_F32 ? launch_mfma<AK, BKC, float, 0>(g, s) : (int)LDM_E_DTYPE
  if (akc && bkc) { GO(true, true); }
  if (akc && !bkc) { GO(true, false); }
  if (!akc && bkc) { GO(false, true); }
  GO(false, false);
#undef GO
}

// C(fp32) += alpha * opA(A) * opB(B), K split over workgroups so that small outputs with a
// long reduction (weight gradients: K = number of tokens) still fill the chip.
int ldm_gemm_mfma_bf16_acc(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const void *A,
                           int64_t lda, const void *B, int64_t ldb, float *C, int64_t ldc, float *rowsum,
                           hipStream_t s) {
  MArgs g;
  const int rc = check_and_fill(g, transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0);
  if (rc) return rc;
  const int64_t ntiles = (int64_t)g.tiles_m * g.tiles_n;
  const int64_t nt_total = K / BK;
  // 2 workgroups per CU are resident: take the largest split count that still fits in ONE
  // round of 512 workgroups (equal-length workgroups: 540 of them would take two rounds).
  int64_t splits = 512 / ntiles;
  if (splits > nt_total / 4) splits = nt_total / 4; // keep >= 4 K tiles per workgroup
  if (splits > 64) splits = 64;
  if (splits < 1) splits = 1;
  g.splits = (int)splits;
  g.rowsum = rowsum;
  const bool akc = !transA, bkc = transB != 0;
  if (akc && bkc) return launch_mfma<true, true, float, 1>(g, s);
  if (akc && !bkc) return launch_mfma<true, false, float, 1>(g, s);
  if (!akc && bkc) return launch_mfma<false, true, float, 1>(g, s);
  return launch_mfma<false, false, float, 1>(g, s);
}
