// NOT PART OF THE LIBRARY: a measured dead end, kept for the record.  Built as csrc/gemm_mfma4w_st.hip and
// dispatched from ldm_gemm_mfma256_bf16_acc it passed tests/test_gpu_kernels.py -k gemm_acc, and ran (one box,
// K = 262144, with row sums, four-wave against eight-wave): M x N = 2304 x 768 885 against 911-965 us,
// 2048 x 768 740 against 742, 768 x 768 412 against 298, 768 x 1024 453 against 368; 11264 x 2048 x 16384 880
// against 690, 12288 x 4096 x 16384 1500 against 1295.  With 32 transposing reads, 8 LDS-DMA, 8 writes of M0 and
// 8 address operations behind the 64 MFMAs of a step almost every place is taken, and the step runs at ~1850
// cycles against 1024.
//
// Weight-gradient form of the four-wave bf16 MFMA GEMM (see gemm_mfma4w.hip for the wave layout and the
// one-filler-per-gap rule): C[M, N] (fp32) += A^T B with BOTH operands stored along the reduction
// -- A [K, M], B [K, N], i.e. dW = dY^T X of a projection --, the reduction split over work items whose
// partial products are added with float atomics, and optionally the column sums of A (the bias gradient)
// from four extra MFMAs per step on the tiles of the first output column.
//
// What differs from the K-contiguous kernel:
//   * a half-tile image is [32 k][256 columns] (512-byte rows); 16-byte chunk c of row k sits at chunk
//     c ^ (2 s(k)), s(k) = (k & 3) | (((k >> 3) & 1) << 2): the layout and the swizzle of gemm_mfma256.hip's
//     strided operands (ds_read_b64_tr_b16 conflict free).  An LDS-DMA piece is two k-rows;
//   * a fragment is two transposing reads (k rows 0-3 and 4-7 of the lane's group), so a step has 32 reads
//     per wave instead of 16, and their address is lane_part ^ (32 * block): one v_xor per fragment pair
//     (A and B fragments of the same index share it);
//   * no stores in the main loop: an item is a long run of steps (hundreds), its epilogue 256 float atomics
//     per lane.
// A step therefore carries 8 LDS-DMA (+ 8 writes of M0), 32 reads and 8 address operations in the 64 places
// behind its MFMAs -- almost every place is taken, one instruction each.
#include "common.h"
#include <type_traits>
#include <stdlib.h>
#include "gemm.h"

namespace {

typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef short s4_t __attribute__((ext_vector_type(4)));
typedef short s8_t __attribute__((ext_vector_type(8)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

constexpr int BM = 256, BN = 256, KH = 32, NSLOT = 5;
constexpr int ROWB = 512;                         // bytes of one k-row of an image
constexpr int HT_A = KH * ROWB, HT_B = KH * ROWB, HT = HT_A + HT_B;
constexpr int LDS_BYTES = NSLOT * HT;

struct SArgs {
  int64_t M, N, K;
  const bf16_t *A; int64_t lda;
  const bf16_t *B; int64_t ldb;
  float *C; int64_t ldc;
  float *rowsum;      // nullable: rowsum[m] += sum_k A[k, m]
  int tiles_m, tiles_n, splits, band_m;
};

__device__ __forceinline__ int swz_st(int kr) { return ((kr & 3) | (((kr >> 3) & 1) << 2)) << 1; }

template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void mfma_acc(f4_t &acc, const bf8_t &a, const bf8_t &b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_new(f4_t &acc, const bf8_t &a, const bf8_t &b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
}
// the row-sum accumulators live in vector registers (all 256 accumulation registers hold the products)
__device__ __forceinline__ void mfma_vacc(f4_t &acc, const bf8_t &a, const bf8_t &b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// (first step: C = 0 -- a register cleared by a move right in front of an MFMA written as asm would be read
// before the move has landed: the compiler does not see the MFMA's operands and inserts no wait state)
__device__ __forceinline__ void mfma_vnew(f4_t &acc, const bf8_t &a, const bf8_t &b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b));
}

struct ItemId { int split, tm, tn; };
__device__ __forceinline__ ItemId decode_item(int w, int total, int tiles_m, int tiles_n, int GM) {
  {
    const int q = total >> 3, r = total & 7, x = w & 7;
    w = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (w >> 3);
  }
  const int ntiles = tiles_m * tiles_n;
  ItemId t;
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

template <int V> using IC = std::integral_constant<int, V>;
typedef std::true_type T_;
typedef std::false_type F_;

// WR: the wave (0..3); wave (wm, wn) = (WR >> 1, WR & 1) owns rows wm * 128.., columns wn * 128.. of the tile
template <int WR>
__device__ __forceinline__ void gemm4ws_body(const SArgs &g, char *smem) {
  constexpr int wave = WR, wm = WR >> 1, wn = WR & 1;
  const int total = g.tiles_m * g.tiles_n * g.splits;
  int w = blockIdx.x;
  if (w >= total) return;
  const unsigned lds_base = (unsigned)(uintptr_t)LDS_PTR(smem);
  if (lds_base & 255u) return;   // (the read addresses below are formed by XOR on bits 5..7: the ring starts 256-byte aligned)
  const int lane = threadIdx.x & 63;

  // ---- fragment reads.  Lane (g4, q, p) = (lane >> 4, (lane >> 2) & 3, lane & 3) reads, for 16-column block b of
  // the tile, 8 bytes of k-row kr0 = 8 g4 + q (and of kr0 + 4): chunk 2 (b ^ s(kr0)) + (p >> 1), half p & 1.
  // With b = 8 * (wave's half) + i: address = [kr0 * 512 + 16 (p >> 1) + 8 (p & 1) + 32 s] ^ (32 i) + 256 * half.
  const int g4 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  const int kr0 = 8 * g4 + q4;
  const unsigned lane_rd = (unsigned)(kr0 * ROWB + 16 * (p4 >> 1) + 8 * (p4 & 1) + 32 * (swz_st(kr0) >> 1));

  // ---- LDS-DMA.  Piece P = 4 * wave + i (i = 0..3) of a half-tile is k-rows 2P, 2P + 1: lane l brings the 16
  // bytes that belong at chunk l & 31 of row 2P + (l >> 5), i.e. logical chunk (l & 31) ^ swz(row).  The row's
  // swizzle depends on i only through i & 1 (rows 8 wave + 2 i + (l >> 5)): two lane offsets per operand.
  const int lh = lane >> 5;
  unsigned laneA[2], laneB[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int sw = swz_st(8 * wave + 2 * e + lh);
    const unsigned ch = (unsigned)(((lane & 31) ^ sw) << 4);
    laneA[e] = (unsigned)(lh * (int)g.lda * 2) + ch;
    laneB[e] = (unsigned)(lh * (int)g.ldb * 2) + ch;
  }
  const unsigned kstepA = (unsigned)(KH * g.lda * 2), kstepB = (unsigned)(KH * g.ldb * 2);   // bytes per half-tile
  // piece i is addressed by base_i + lane offset + k offset + immediate i * 1024; the immediate also advances
  // the LDS address, so base_i is 1024 i lower than the piece
  const char *bA[4], *bB[4];
  auto dma = [&](auto I, bool opB, unsigned voff) {
    constexpr int i = decltype(I)::value;
    const char *base = opB ? bB[i] : bA[i];
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(base), "n"(i * 1024) : "memory");
  };
  auto set_m0 = [&](unsigned v) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(v) : "memory"); };
  auto set_m0_early = [&](unsigned v) { asm volatile("s_mov_b32 m0, %0" ::"s"(v) : "memory"); };
  auto m0_of = [&](int h, bool opB) { return lds_base + (h % NSLOT) * HT + (opB ? HT_A : 0) + wave * 4096; };
  auto tr_read = [&](s4_t &dst, unsigned va, auto OFF) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(va), "n"(decltype(OFF)::value));
  };

  const int np_total = (int)(g.K / KH);
  const int pairs_total = np_total >> 1, basep = pairs_total / g.splits, remp = pairs_total % g.splits;

  while (true) {
    const ItemId tid = decode_item(w, total, g.tiles_m, g.tiles_n, g.band_m);
    const int64_t m0 = (int64_t)tid.tm * BM, n0 = (int64_t)tid.tn * BN;
    const int p_begin = 2 * (tid.split * basep + (tid.split < remp ? tid.split : remp));
    const int np = 2 * (basep + (tid.split < remp ? 1 : 0));
    const bool do_rs = g.rowsum != nullptr && tid.tn == 0;

    {
      const char *tA = (const char *)(g.A + ((int64_t)p_begin * KH + 8 * wave) * g.lda + m0);
      const char *tB = (const char *)(g.B + ((int64_t)p_begin * KH + 8 * wave) * g.ldb + n0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bA[i] = tA + (int64_t)(2 * i) * g.lda * 2 - i * 1024;
        bB[i] = tB + (int64_t)(2 * i) * g.ldb * 2 - i * 1024;
      }
    }
    // before the item: pairs (0, 1) and (2, 3), 32 requests
    auto pair = [&](int h) {
      const unsigned a0 = m0_of(h, false), a1 = m0_of(h + 1, false), b0 = m0_of(h, true), b1 = m0_of(h + 1, true);
      const unsigned va0 = laneA[0] + h * kstepA, va1 = laneA[1] + h * kstepA;
      const unsigned vb0 = laneB[0] + h * kstepB, vb1 = laneB[1] + h * kstepB;
      set_m0(a0); dma(IC<0>{}, false, va0); set_m0(a1); dma(IC<0>{}, false, va0 + kstepA);
      set_m0(a0); dma(IC<1>{}, false, va1); set_m0(a1); dma(IC<1>{}, false, va1 + kstepA);
      set_m0(a0); dma(IC<2>{}, false, va0); set_m0(a1); dma(IC<2>{}, false, va0 + kstepA);
      set_m0(a0); dma(IC<3>{}, false, va1); set_m0(a1); dma(IC<3>{}, false, va1 + kstepA);
      set_m0(b0); dma(IC<0>{}, true, vb0); set_m0(b1); dma(IC<0>{}, true, vb0 + kstepB);
      set_m0(b0); dma(IC<1>{}, true, vb1); set_m0(b1); dma(IC<1>{}, true, vb1 + kstepB);
      set_m0(b0); dma(IC<2>{}, true, vb0); set_m0(b1); dma(IC<2>{}, true, vb0 + kstepB);
      set_m0(b0); dma(IC<3>{}, true, vb1); set_m0(b1); dma(IC<3>{}, true, vb1 + kstepB);
    };
    // (every wave has left the epilogue of the item before -- and with it the last reads of the ring -- before
    // the ring is refilled)
    __builtin_amdgcn_s_barrier();
    pair(0);
    pair(2);

    f4_t acc[8][8];
    f4_t racc[4];
    s4_t alo[2][8], ahi[2][8], blo[2][8], bhi[2][8];
    bf8_t ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
    asm volatile("" : "+v"(ones));   // in registers from here on (not re-created in front of the MFMA that reads it)
#pragma unroll
    for (int i = 0; i < 4; ++i) racc[i] = f4_t{0.f, 0.f, 0.f, 0.f};
    auto fragA = [&](int set, int i) {
      const s8_t v = {alo[set][i][0], alo[set][i][1], alo[set][i][2], alo[set][i][3],
                      ahi[set][i][0], ahi[set][i][1], ahi[set][i][2], ahi[set][i][3]};
      return __builtin_bit_cast(bf8_t, v);
    };
    auto fragB = [&](int set, int i) {
      const s8_t v = {blo[set][i][0], blo[set][i][1], blo[set][i][2], blo[set][i][3],
                      bhi[set][i][0], bhi[set][i][1], bhi[set][i][2], bhi[set][i][3]};
      return __builtin_bit_cast(bf8_t, v);
    };

    // half-tiles 0, 1 landed (younger: the 16 requests of pair (2, 3))
    wait_vm<16>();
    __builtin_amdgcn_s_barrier();
    {
      unsigned va = lane_rd + lds_base;
      asm volatile("" : "+v"(va));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        unsigned t = va ^ (unsigned)(i << 5);
        asm volatile("" : "+v"(t));
        tr_read(alo[0][i], t, IC<wm * 256>{});
        tr_read(ahi[0][i], t, IC<wm * 256 + 4 * ROWB>{});
        tr_read(blo[0][i], t, IC<HT_A + wn * 256>{});
        tr_read(bhi[0][i], t, IC<HT_A + wn * 256 + 4 * ROWB>{});
      }
    }

    // One 32-deep step: MFMAs on the fragments of half-tile p (set cur) with, in the places behind them, the reads
    // of half-tile p + 1 (set nxt) and this step's LDS-DMA.  RS: the wave also sums four fragments of A.
    auto step = [&](auto PAR, auto FIRST, auto ISSUE, auto RS, int p) {
      constexpr int cur = decltype(PAR)::value, nxt = cur ^ 1;
      constexpr bool first = decltype(FIRST)::value;
      constexpr bool do_issue = decltype(ISSUE)::value;
      constexpr bool rs = decltype(RS)::value;
      // an odd step needs the pair (p + 1, p + 2) complete: its last request went out in step p - 2 (or before the
      // item); younger are the eight requests of step p - 1; the last four steps have sent nothing
      if constexpr (cur == 1) {
        if constexpr (do_issue) wait_vm<8>(); else wait_vm<0>();
      }
      // even step p: A pieces of pair (p + 4, p + 5); odd step p: B pieces of pair (p + 3, p + 4)
      const int hpair = cur == 0 ? p + 4 : p + 3;
      const unsigned kst = cur == 1 ? kstepB : kstepA;
      unsigned vo0 = (cur == 1 ? laneB[0] : laneA[0]) + hpair * kst;
      unsigned vo1 = (cur == 1 ? laneB[1] : laneA[1]) + hpair * kst;
      unsigned vo0b = vo0 + kst, vo1b = vo1 + kst;
      unsigned mA = m0_of(hpair, cur == 1), mB = m0_of(hpair + 1, cur == 1);
      unsigned va = lane_rd + lds_base + ((p + 1) % NSLOT) * HT;
      unsigned t0 = va, t1 = va;
      asm volatile("" : "+v"(va), "+v"(vo0), "+v"(vo1), "+v"(vo0b), "+v"(vo1b), "+s"(mA), "+s"(mB));
      // this wave's reads of half-tile p are complete: the values exist from here on
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("" : "+v"(alo[cur][i]), "+v"(ahi[cur][i]), "+v"(blo[cur][i]), "+v"(bhi[cur][i]));
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // Place s (behind MFMA s), period of 16 places, q = s % 16, j = s / 16:
      //   LDS-DMA of piece j: M0 at q = 4 WR, first half-tile at 4 WR + 1, M0 at 4 WR + 2, second at 4 WR + 3;
      //   the other twelve places, in order: v_xor for fragment pair 2j, v_xor for 2j + 1, then the eight reads
      //   (A lo, A hi, B lo, B hi of 2j, then of 2j + 1); two places stay free.
      auto place = [&](auto S) {
        constexpr int s_ = decltype(S)::value, q = s_ % 16, j = s_ / 16;
        if constexpr (q >= 4 * WR && q < 4 * WR + 4) {
          constexpr int d = q - 4 * WR;
          if constexpr (do_issue) {
            if constexpr (d == 0) set_m0_early(mA);
            else if constexpr (d == 2) set_m0_early(mB);
            else if constexpr (d == 1) dma(IC<j>{}, cur == 1, (j & 1) ? vo1 : vo0);
            else dma(IC<j>{}, cur == 1, (j & 1) ? vo1b : vo0b);
          }
        } else {
          constexpr int r = q - (q > 4 * WR ? 4 : 0);   // rank among the twelve other places
          if constexpr (r == 0) { t0 = va ^ (unsigned)((2 * j) << 5); asm volatile("" : "+v"(t0)); }
          else if constexpr (r == 1) { t1 = va ^ (unsigned)((2 * j + 1) << 5); asm volatile("" : "+v"(t1)); }
          else if constexpr (r < 10) {
            constexpr int k = r - 2, f = 2 * j + k / 4, part = k % 4;
            const unsigned t = (k / 4) ? t1 : t0;
            if constexpr (part == 0) tr_read(alo[nxt][f], t, IC<wm * 256>{});
            else if constexpr (part == 1) tr_read(ahi[nxt][f], t, IC<wm * 256 + 4 * ROWB>{});
            else if constexpr (part == 2) tr_read(blo[nxt][f], t, IC<HT_A + wn * 256>{});
            else tr_read(bhi[nxt][f], t, IC<HT_A + wn * 256 + 4 * ROWB>{});
          }
        }
      };
      auto run = [&](auto self, auto K) {
        constexpr int k = decltype(K)::value;
        if constexpr (k < 64) {
          constexpr int gi = k / 8, ni = k % 8;
          if constexpr (first) mfma_new(acc[ni][gi], fragB(cur, ni), fragA(cur, gi));
          else mfma_acc(acc[ni][gi], fragB(cur, ni), fragA(cur, gi));
          __builtin_amdgcn_sched_barrier(0);
          place(IC<k>{});
          __builtin_amdgcn_sched_barrier(0);
          // the wave's share of the row sums: fragments 4 wn .. 4 wn + 3 of A, once each, behind the
          // products that last use them
          if constexpr (rs && ni == 7 && gi / 4 == wn) {
            if constexpr (first) mfma_vnew(racc[gi % 4], ones, fragA(cur, gi));
            else mfma_vacc(racc[gi % 4], ones, fragA(cur, gi));
            __builtin_amdgcn_sched_barrier(0);
          }
          self(self, IC<k + 1>{});
        }
      };
      run(run, IC<0>{});
    };

    auto run_item = [&](auto RS) {
      step(IC<0>{}, T_{}, T_{}, RS, 0);
      step(IC<1>{}, F_{}, T_{}, RS, 1);
      int p = 2;
      // steps up to np - 5 send (the pairs up to (np - 2, np - 1)); the last four send nothing
      for (; p < np - 4; p += 2) {
        step(IC<0>{}, F_{}, T_{}, RS, p);
        step(IC<1>{}, F_{}, T_{}, RS, p + 1);
      }
      for (; p < np; p += 2) {
        step(IC<0>{}, F_{}, F_{}, RS, p);
        step(IC<1>{}, F_{}, F_{}, RS, p + 1);
      }
    };
    if (do_rs) run_item(T_{}); else run_item(F_{});
    // (the asm MFMAs' results: wait states the compiler does not count; the last step's reads of the slot
    // behind the item are complete before the registers are reused)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);

    // ---- epilogue: C += acc.  Lane (row = lane & 15, group = lane >> 4) holds, per fragment (ni, gi), the four
    // consecutive columns n0 + wn * 128 + ni * 16 + 4 group .. + 3 of row m0 + wm * 128 + gi * 16 + row.
    {
      float *cb = g.C + (m0 + wm * 128 + (lane & 15)) * g.ldc + n0 + wn * 128 + (lane >> 4) * 4;
#pragma unroll
      for (int gi = 0; gi < 8; ++gi) {
        float *cr = cb + (int64_t)gi * 16 * g.ldc;
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
          f4_t v;
          asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                       : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3])
                       : "a"(acc[ni][gi][0]), "a"(acc[ni][gi][1]), "a"(acc[ni][gi][2]), "a"(acc[ni][gi][3]));
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(cr + ni * 16 + r, v[r]);
        }
      }
      if (do_rs && lane < 16) {
        // every register of lane l holds the sum for row (l & 15) of the fragment
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(g.rowsum + m0 + wm * 128 + (4 * wn + j) * 16 + lane, racc[j][0]);
      }
    }
    w += gridDim.x;
    if (w >= total) break;
  }
}

__global__ __launch_bounds__(256, 1) void gemm4ws_k(SArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave == 0) gemm4ws_body<0>(g, smem);
  else if (wave == 1) gemm4ws_body<1>(g, smem);
  else if (wave == 2) gemm4ws_body<2>(g, smem);
  else gemm4ws_body<3>(g, smem);
}

int num_cus4s() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

}  // namespace

// C[M, N] (fp32) += A^T B, A [K, M] and B [K, N] bf16 (row strides lda, ldb), rowsum[m] += sum_k A[k, m] when
// given.  LDM_E_SHAPE / LDM_E_ALIGN for what the kernel does not take (the caller then runs the eight-wave
// kernel): M and N multiples of 256, K a multiple of 64, at least eight 32-deep steps per work item, and the
// byte offset of a work item's last k-row below 4 GiB (32-bit lane offsets).
int ldm_gemm_mfma4w_st_bf16(int64_t M, int64_t N, int64_t K, const void *A, int64_t lda, const void *B, int64_t ldb,
                            float *C, int64_t ldc, float *rowsum, hipStream_t s) {
  if (M < 256 || M % 256 || N < 256 || N % 256) return LDM_E_SHAPE;
  if (K % 64 || K < 512) return LDM_E_SHAPE;
  if (lda % 8 || ldb % 8 || ldc % 4 || lda < M || ldb < N || ldc < N) return LDM_E_ALIGN;
  if (!aligned_to(A, 16) || !aligned_to(B, 16) || !aligned_to(C, 16)) return LDM_E_ALIGN;
  const int64_t tm = M / 256, tn = N / 256, ntiles = tm * tn;
  if (ntiles > 4096) return LDM_E_SHAPE;
  const int64_t pairs = K / 64;
  // K splits: as many work items as CUs (one round), each at least 16 steps long; more rounds only when the
  // tiles alone exceed the CUs
  const int64_t cus = num_cus4s();
  int64_t splits = ntiles >= cus ? 1 : cus / ntiles;
  if (splits > pairs / 8) splits = pairs / 8;
  if (splits < 1) splits = 1;
  if (const char *e = getenv("LDM_GEMM_SPLITS")) { if (atoi(e) > 0) splits = atoi(e); }   // experiments only
  if (pairs / splits < 4) return LDM_E_SHAPE;
  // 32-bit lane offsets: (rows of an item + 2) * row bytes
  const int64_t rows_item = (pairs / splits + 1) * 64 + 64;
  if (rows_item * lda * 2 >= ((int64_t)1 << 32) || rows_item * ldb * 2 >= ((int64_t)1 << 32)) return LDM_E_SHAPE;
  SArgs g;
  g.M = M; g.N = N; g.K = K;
  g.A = (const bf16_t *)A; g.lda = lda; g.B = (const bf16_t *)B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.rowsum = rowsum;
  g.tiles_m = (int)tm; g.tiles_n = (int)tn; g.splits = (int)splits; g.band_m = 4;
  static std::atomic<uint64_t> attr_done{0};
  uint64_t dev_bit;
  if (ldm_attr_pending(attr_done, &dev_bit)) {
    (void)hipFuncSetAttribute((const void *)gemm4ws_k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done.fetch_or(dev_bit, std::memory_order_release);
  }
  const int64_t items = ntiles * splits;
  const int64_t nwg = items < cus ? items : cus;   // persistent: one workgroup per CU
  hipLaunchKernelGGL(gemm4ws_k, dim3((unsigned)nwg), dim3(256), LDS_BYTES, s, g);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}
