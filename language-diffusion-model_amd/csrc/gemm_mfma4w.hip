// bf16 MFMA GEMM for gfx950 with FOUR waves per workgroup, one per SIMD, each owning 128 x 128 outputs of a
// 256 x 256 tile (64 accumulator fragments = 256 accumulation registers per lane; the 256 vector registers
// hold the operand fragments and, behind the main loop, the packed output).  One persistent workgroup per
// CU walks the tiles in the XCD-aware order of gemm_mfma256.hip.
//
// Why four waves: with one wave per SIMD a 32-deep step is 64 MFMAs (1024 cycles) against 16 fragment reads
// and 8 LDS-DMA requests per wave -- half the LDS reads per MFMA of the eight-wave kernel -- and everything
// that is not an MFMA is placed, one instruction at a time, in the gaps behind every second MFMA:
//   * a gap carries at most ONE other instruction (a 16x16x32 MFMA occupies its SIMD for 16 cycles and the
//     wave's issue for 8 of them: one more instruction per MFMA pair is free, a burst of four is not);
//   * the four waves send their LDS-DMA in DIFFERENT gaps (the CU's address unit takes ~16 cycles per 1-KiB
//     request: four waves that reach the same request together queue behind each other, and a wave that
//     waits there issues no MFMA).  Each wave therefore runs its own copy of the code, which differs only in
//     which gap holds what;
//   * the steps that send nothing (the last four of an item) are compiled separately: no branch per gap;
//   * per LDS-DMA the MFMA stream holds two instructions (s_mov m0 + the request): the piece bases are
//     scalar registers set once per item, the k offset of a half-tile is added to the lane offset once per step.
// Stores: a CU's store path takes ~60 cycles per 1-KiB store instruction whatever its destination, and the
// issuing wave waits for its own; 128 of them back to back behind the main loop are ~8k cycles per tile
// (a third of a K = 768 tile).  Here the accumulators are packed to bf16 right behind the main loop and
// exchanged between 16-lane rows (v_permlane16_swap) into 16-byte runs -- 32 quads = 128 registers per
// lane, no LDS round trip --, the first eight quads are stored while the rest is being packed and the
// other 24 go out as fillers of the NEXT item's first six steps, one store per wave every eight gaps and
// the waves two gaps apart.  They are buffer stores: before the first item the descriptor has zero records,
// so that item's (empty) deferred stores are dropped by the range check and no step needs a branch.
//
// Ring: five 32-deep half-tiles of A and B (160 KiB), images and swizzle as in gemm_mfma256.hip (K-contiguous
// operands: [rows][32 k], 64-byte rows, chunk c of row r at c ^ (3 * ((r >> 2) & 1))).  Half-tiles travel in
// pairs (2q, 2q+1) -- the two 64-byte halves of a 128-byte line requested by consecutive instructions of one
// wave --, the A pieces of pair (p+4, p+5) in even step p, its B pieces in step p+1.  One barrier per step;
// an odd step first waits (counted vmcnt) for the pair that the next two steps read.
//
// Takes: A [M, K] and B [N, K] both K-contiguous, bf16 output, M and N multiples of 256, K a multiple of 64
// and at least 320, optional fp32 bias, optionally a device-side row count (whole 256-row tiles are computed
// from whatever A holds; the rows of C at and beyond the count are not written).  Everything else stays with
// gemm_mfma256.hip.
#include "common.h"
#include <type_traits>
#include <stdlib.h>
#include "gemm.h"

namespace {

typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef unsigned u4v_t __attribute__((ext_vector_type(4)));
typedef int i4s_t __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

constexpr int BM = 256, BN = 256, KH = 32, NSLOT = 5;
constexpr int HT_A = BM * 64, HT_B = BN * 64, HT = HT_A + HT_B;
constexpr int LDS_BYTES = NSLOT * HT;
#ifndef LDM4W_NOW
#define LDM4W_NOW 8
#endif
#ifndef LDM4W_SPS
#define LDM4W_SPS 4
#endif
constexpr int NOW = LDM4W_NOW;      // quads stored while the others are packed (one behind every 32 / NOW packed quads)
constexpr int SPS = LDM4W_SPS;      // deferred stores per step
constexpr int NSS = (32 - NOW) / SPS;   // steps that carry deferred stores
static_assert((32 - NOW) % SPS == 0 && NSS % 2 == 0, "whole pairs of store steps");
static_assert(32 % NOW == 0 && 4 % SPS == 0, "stores at a fixed stride");

struct WArgs {
  int64_t M, N, K;
  const bf16_t *A; int64_t lda;
  const bf16_t *B; int64_t ldb;
  bf16_t *C; int64_t ldc;
  const float *bias;
  const bf16_t *R;    // nullable: residual [M, N] with the row stride of C, added before the rounding to bf16
  const int *dyn_m;   // nullable: device-side row count; only the m-tiles that hold rows < *dyn_m are walked
  int tiles_m, tiles_n, band_m;
#ifdef LDM_GEMM_TIMING
  unsigned long long *timing;   // [4 waves][8]: cycle totals of workgroup 17's per-item segments
#endif
};

// Experiment builds (-DLDM_GEMM_TIMING): cycle totals of the per-item segments of workgroup 17 are added to the
// device address given in the environment variable LDM_GEMM4W_TIMING_BUF.
#ifdef LDM_GEMM_TIMING
#define TSTAMP4(var) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); var = __builtin_amdgcn_s_memtime(); \
                          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#else
#define TSTAMP4(var) do { } while (0)
#endif

__device__ __forceinline__ int swz_kc(int row) { return 3 * ((row >> 2) & 1); }

template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// accumulator pinned to the accumulation registers; volatile: the MFMAs and the fillers between them stay
// in the order written
__device__ __forceinline__ void mfma_acc(f4_t &acc, const bf8_t &a, const bf8_t &b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// first step of an item: C = 0 (no pass over the 256 accumulation registers to clear them)
__device__ __forceinline__ void mfma_new(f4_t &acc, const bf8_t &a, const bf8_t &b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
}

struct TileId { int tm, tn; };
__device__ __forceinline__ TileId decode_item(int w, int total, int tiles_m, int tiles_n, int GM) {
  {
    const int q = total >> 3, r = total & 7, x = w & 7;
    w = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (w >> 3);
  }
  const int width = GM * tiles_n;
  const int grp = w / width;
  const int first = grp * GM;
  const int gsz = (tiles_m - first < GM) ? tiles_m - first : GM;
  const int in = w - grp * width;
  TileId t;
  t.tm = first + in % gsz;
  t.tn = in / gsz;
  return t;
}

template <int V> using IC = std::integral_constant<int, V>;

// WR: the wave (0..3); wave (wm, wn) = (WR >> 1, WR & 1) owns rows wm * 128.., columns wn * 128.. of the tile
// BIAS = false: no bias loads and no bias additions in the packing (the input-gradient products)
template <int WR, bool RESID, bool BIAS>
__device__ __forceinline__ void gemm4w_body(const WArgs &g, char *smem) {
  constexpr int wave = WR, wm = WR >> 1, wn = WR & 1;
  int tiles_m = g.tiles_m;
  int mdyn = 0x7fffffff;   // rows of C that exist (device-side count)
  if (g.dyn_m) {
    // (a scalar load, written out: the count and everything derived from it stay in scalar registers)
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(mdyn) : "s"(g.dyn_m) : "memory");
    const int td = (mdyn + BM - 1) / BM;
    tiles_m = td < tiles_m ? (td > 0 ? td : 0) : tiles_m;
  }

  const int total = tiles_m * g.tiles_n;
  const int np = (int)(g.K / KH);
  int w = blockIdx.x;
  if (w >= total) return;
  const unsigned lds_base = (unsigned)(uintptr_t)LDS_PTR(smem);
  const int lane = threadIdx.x & 63;
  // fragment reads: row lane & 15 of a 16-row block, chunk (lane >> 4) ^ swizzle; block and slot are added as
  // an immediate and once per step
  const unsigned lane_rd = (unsigned)((lane & 15) * 64 + (((lane >> 4) ^ swz_kc(lane & 15)) << 4));
  // LDS-DMA: lane part of the source offset inside a 16-row piece (the swizzle on the SOURCE: the LDS
  // destination of an LDS-DMA is lane-linear)
  const unsigned laneA = (unsigned)((lane >> 2) * (int)g.lda * 2 + (((lane & 3) ^ swz_kc(lane >> 2)) << 4));
  const unsigned laneB = (unsigned)((lane >> 2) * (int)g.ldb * 2 + (((lane & 3) ^ swz_kc(lane >> 2)) << 4));
  // stores: lane part of the byte offset inside the wave's 128 x 128 block of C: row lane & 15 of a 16-row
  // block, 16-byte run ((q & 1) * 32 + (q >> 1) * 16 of a 64-byte group, q = lane >> 4: see pack below)
  const unsigned lane_st = (unsigned)((lane & 15) * (int)g.ldc * 2 + ((lane >> 4) & 1) * 32 + (lane >> 5) * 16);
  const unsigned lane_bias = (unsigned)((lane >> 4) * 16);
  const unsigned ldc16 = (unsigned)(g.ldc * 32);   // bytes per 16 rows of C
  // piece i of a wave's four (16 rows each) is addressed by base_i + lane part + k offset + immediate i * 1024;
  // the immediate also advances the LDS address, so base_i is 1024 i lower than the piece
  const int64_t pieceA = (int64_t)32 * g.lda - 1024, pieceB = (int64_t)32 * g.ldb - 1024;
  const char *bA[4], *bB[4];
  auto dma = [&](auto I, bool opB, unsigned voff) {
    constexpr int i = decltype(I)::value;
    const char *base = opB ? bB[i] : bA[i];   // (named outside the asm: a generic lambda does not capture through an asm operand)
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(base), "n"(i * 1024) : "memory");
  };
  // (an LDS-DMA right behind the write of M0 needs one wait state; where an MFMA lies between them it has it)
  auto set_m0 = [&](unsigned v) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(v) : "memory"); };
  auto set_m0_early = [&](unsigned v) { asm volatile("s_mov_b32 m0, %0" ::"s"(v) : "memory"); };
  auto m0_of = [&](int h, bool opB) { return lds_base + (h % NSLOT) * HT + (opB ? HT_A : 0) + wave * 4096; };
  // before an item: pairs (0, 1) and (2, 3), 32 requests
  auto prologue = [&](TileId t) {
    const char *tA = (const char *)(g.A + ((int64_t)t.tm * BM + wave * 64) * g.lda);
    const char *tB = (const char *)(g.B + ((int64_t)t.tn * BN + wave * 64) * g.ldb);
#pragma unroll
    for (int i = 0; i < 4; ++i) { bA[i] = tA + i * pieceA; bB[i] = tB + i * pieceB; }
    auto pair = [&](int h) {
      const unsigned va = laneA + h * 64, vb = laneB + h * 64;
      const unsigned a0 = m0_of(h, false), a1 = m0_of(h + 1, false), b0 = m0_of(h, true), b1 = m0_of(h + 1, true);
      set_m0(a0); dma(IC<0>{}, false, va); set_m0(a1); dma(IC<0>{}, false, va + 64);
      set_m0(a0); dma(IC<1>{}, false, va); set_m0(a1); dma(IC<1>{}, false, va + 64);
      set_m0(a0); dma(IC<2>{}, false, va); set_m0(a1); dma(IC<2>{}, false, va + 64);
      set_m0(a0); dma(IC<3>{}, false, va); set_m0(a1); dma(IC<3>{}, false, va + 64);
      set_m0(b0); dma(IC<0>{}, true, vb); set_m0(b1); dma(IC<0>{}, true, vb + 64);
      set_m0(b0); dma(IC<1>{}, true, vb); set_m0(b1); dma(IC<1>{}, true, vb + 64);
      set_m0(b0); dma(IC<2>{}, true, vb); set_m0(b1); dma(IC<2>{}, true, vb + 64);
      set_m0(b0); dma(IC<3>{}, true, vb); set_m0(b1); dma(IC<3>{}, true, vb + 64);
    };
    pair(0);
    pair(2);
  };
  auto frag = [&](unsigned va, int off) {
    return *reinterpret_cast<const __attribute__((address_space(3))) bf8_t *>((uintptr_t)(va + off));
  };
  // descriptor of the wave's block of C in the PREVIOUS item (zero records before the first item: its
  // deferred stores do not exist and the range check drops them)
  i4s_t srd_prev = {0, 0, 0, 0x00020000};
  u4v_t pk[32];
  auto store_q = [&](const u4v_t &q, const i4s_t &srd, unsigned soff, auto T) {
    constexpr int t = decltype(T)::value;
    const unsigned vo = lane_st;
    // (plain, not non-temporal: a non-temporal store of 64-byte runs costs the issuing wave four times as long)
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:%4\n\ts_nop 0" ::"v"(q), "v"(vo), "s"(srd), "s"(soff), "n"(t * 64) : "memory");
  };

  // residual quad j of the current item: the 16 rows x 64 bytes that store j of this item will cover, requested
  // with the store's own lane offset from a scalar pointer that walks down the wave's block 16 rows at a time.
  // The data is in pk[j] only behind a counted wait; until then nothing touches those registers.
  [[maybe_unused]] const char *rptr = nullptr;
  [[maybe_unused]] auto res_load = [&](auto J) {
    constexpr int j = decltype(J)::value;
    const unsigned vo = lane_st;
    const char *base = rptr;
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(pk[j]) : "v"(vo), "s"(base), "n"((j % 4) * 64) : "memory");
    if constexpr (j % 4 == 3) rptr += ldc16;
  };

  TileId tid = decode_item(w, total, tiles_m, g.tiles_n, g.band_m);
  prologue(tid);
  bool first_item = true;

  [[maybe_unused]] unsigned long long ts[6], tsum[6] = {0, 0, 0, 0, 0, 0}, tsum6 = 0, tsum7 = 0;
  while (true) {
    const int64_t m0 = (int64_t)tid.tm * BM, n0 = (int64_t)tid.tn * BN;
    f4_t acc[8][8];
    bf8_t fa[2][8], fb[2][8];
    TSTAMP4(ts[0]);

    // half-tiles 0, 1 landed.  Issue order of this wave's vector-memory instructions: [bias loads] |
    // prologue (32) | stores of the packing phase (NOW, not before the first item) | step 0 ...: younger than
    // the pair (0, 1) are the pair (2, 3) and those stores.
    if (first_item) wait_vm<16>(); else wait_vm<16 + NOW>();
    first_item = false;
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    {
      unsigned va = lane_rd + lds_base;
      asm volatile("" : "+v"(va));
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        fa[0][i] = frag(va, wm * 8192 + i * 1024);
        fb[0][i] = frag(va, HT_A + wn * 8192 + i * 1024);
      }
    }
    TSTAMP4(ts[1]);

    // One 32-deep step: MFMAs on the fragments of half-tile p (set cur) with, in the gaps, the reads of
    // half-tile p+1 (set nxt), this step's LDS-DMA and deferred stores.
    //   PAR = p & 1; FIRST: step 0 (C = 0); ISSUE: the step sends LDS-DMA; SQ: first quad it stores (-1: none)
    auto step = [&](auto PAR, auto FIRST, auto ISSUE, auto SQ, int p, auto RQ) {
      constexpr int cur = decltype(PAR)::value, nxt = cur ^ 1;
      constexpr bool first = decltype(FIRST)::value;
      constexpr bool do_issue = decltype(ISSUE)::value;
      constexpr int sq = decltype(SQ)::value;
      constexpr int rq = decltype(RQ)::value;   // first of the eight residual quads this step requests (-1: none)
      // An odd step needs the pair (p+1, p+2) complete.  Its last request went out in step p-2 (or before the
      // item); younger are at least the eight requests of step p-1 and that step's deferred stores (counting
      // only those is the stricter wait); the last four steps have sent nothing.
      if constexpr (cur == 1) {
        if constexpr (do_issue) wait_vm<8 + (sq >= 0 ? SPS : 0)>();
        else if constexpr (rq < 0) wait_vm<0>();
        // with residual requests in the last four steps: the second of them still needs the last pair, behind
        // which only the eight requests of the step before lie; the last step needs nothing new
        else if constexpr (rq == 8) wait_vm<8>();
      }
      // even step p: A pieces of pair (p+4, p+5); odd step p: B pieces of pair (p+3, p+4)
      const int hpair = cur == 0 ? p + 4 : p + 3;
      unsigned voff = (cur == 1 ? laneB : laneA) + hpair * 64;
      unsigned voff2 = voff + 64;
      unsigned mA = m0_of(hpair, cur == 1), mB = m0_of(hpair + 1, cur == 1);
      unsigned va = lane_rd + lds_base + ((p + 1) % NSLOT) * HT;
      unsigned soff = (unsigned)(sq >= 0 ? sq / 4 : 0) * ldc16;
      asm volatile("" : "+v"(va), "+v"(voff), "+v"(voff2), "+s"(mA), "+s"(mB), "+s"(soff));
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's reads of half-tile p are complete
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // Gap s (behind MFMA 2s+1), period of eight gaps, gq = s % 8, j = s / 8, at most one instruction per gap:
      //   LDS-DMA of piece j: first half-tile in gap gq = 2 WR, second in gq = 2 WR + 1 (M0 between the gap's
      //   two MFMAs); deferred store j in gap gq = (2 WR + 5) % 8; fragment reads in the first four of the
      //   other five gaps.  Over the four waves every gap carries one LDS-DMA and every second gap one store.
      auto mid = [&](auto S) {
        constexpr int s_ = decltype(S)::value, gq = s_ % 8;
        if constexpr (do_issue && (gq == 2 * WR || gq == 2 * WR + 1)) set_m0_early(gq == 2 * WR ? mA : mB);
      };
      auto filler = [&](auto S) {
        constexpr int s_ = decltype(S)::value, gq = s_ % 8, j = s_ / 8;
        constexpr int gst = (2 * WR + 5) % 8;
        if constexpr (gq == 2 * WR || gq == 2 * WR + 1) {
          if constexpr (do_issue) dma(IC<j>{}, cur == 1, gq == 2 * WR ? voff : voff2);
          else if constexpr (rq >= 0) res_load(IC<rq + 2 * j + (gq - 2 * WR)>{});
        } else if constexpr (gq == gst) {
          // (SPS stores per step, evenly over its four periods; the quads of one step lie in one 16-row block)
          if constexpr (sq >= 0 && j % (4 / SPS) == 0) store_q(pk[sq + j / (4 / SPS)], srd_prev, soff, IC<(sq + j / (4 / SPS)) % 4>{});
        } else {
          // rank of this gap among the period's free gaps
          constexpr int rank = gq - (gq > 2 * WR ? 1 : 0) - (gq > 2 * WR + 1 ? 1 : 0) - (gq > gst ? 1 : 0);
          if constexpr (rank < 4) {
            constexpr int r = 4 * j + rank;
            if constexpr (r % 2 == 0) fa[nxt][r / 2] = frag(va, wm * 8192 + (r / 2) * 1024);
            else fb[nxt][r / 2] = frag(va, HT_A + wn * 8192 + (r / 2) * 1024);
          }
        }
      };
      auto run = [&](auto self, auto K) {
        constexpr int k = decltype(K)::value;
        if constexpr (k < 64) {
          constexpr int gi = k / 8, ni = k % 8;
          if constexpr (first) mfma_new(acc[ni][gi], fb[cur][ni], fa[cur][gi]);
          else mfma_acc(acc[ni][gi], fb[cur][ni], fa[cur][gi]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (k % 2 == 0) mid(IC<k / 2>{});
          else filler(IC<k / 2>{});
          __builtin_amdgcn_sched_barrier(0);
          self(self, IC<k + 1>{});
        }
      };
      run(run, IC<0>{});
    };
    typedef std::true_type T_;
    typedef std::false_type F_;
    // steps 0 .. NSS-1 carry the previous item's deferred stores (four quads = one 16-row block per step)
    auto store_steps = [&](auto self, auto P) {
      constexpr int p = decltype(P)::value;
      if constexpr (p < NSS) {
        if constexpr (p == 0) step(IC<0>{}, T_{}, T_{}, IC<NOW>{}, 0, IC<-1>{});
        else step(IC<(p & 1)>{}, F_{}, T_{}, IC<NOW + SPS * p>{}, p, IC<-1>{});
        self(self, IC<p + 1>{});
      }
    };
    store_steps(store_steps, IC<0>{});
    [[maybe_unused]] unsigned long long tm1, tm2;
    TSTAMP4(tm1);
    int p = NSS;
    // steps up to np-5 send (the pairs up to (np-2, np-1)); the last four send nothing
    for (; p < np - 4; p += 2) {
      step(IC<0>{}, F_{}, T_{}, IC<-1>{}, p, IC<-1>{});
      step(IC<1>{}, F_{}, T_{}, IC<-1>{}, p + 1, IC<-1>{});
    }
    TSTAMP4(tm2);
    // the last four steps send no LDS-DMA; with a residual their request gaps carry its 32 quads, which land in
    // the registers of the packed output (free since the deferred stores of the item before went out)
    if constexpr (RESID) {
      rptr = (const char *)(g.R + (m0 + wm * 128) * g.ldc + n0 + wn * 128);
      step(IC<0>{}, F_{}, F_{}, IC<-1>{}, p, IC<0>{});
      step(IC<1>{}, F_{}, F_{}, IC<-1>{}, p + 1, IC<8>{});
      step(IC<0>{}, F_{}, F_{}, IC<-1>{}, p + 2, IC<16>{});
      step(IC<1>{}, F_{}, F_{}, IC<-1>{}, p + 3, IC<24>{});
    } else {
      for (; p < np; p += 2) {
        step(IC<0>{}, F_{}, F_{}, IC<-1>{}, p, IC<-1>{});
        step(IC<1>{}, F_{}, F_{}, IC<-1>{}, p + 1, IC<-1>{});
      }
    }
    // (the asm MFMAs' results: wait states the compiler does not count)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    TSTAMP4(ts[2]);

    // bias of this lane's accumulator columns (four consecutive ones per fragment), requested before the
    // next item's prologue and waited for, counted, behind it
    [[maybe_unused]] f4_t bv[8];
    const bool has_bias = BIAS && g.bias != nullptr;
    if (has_bias) {
      const float *bb = g.bias + n0 + wn * 128;
#pragma unroll
      for (int ni = 0; ni < 8; ++ni)
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(bv[ni]) : "v"(lane_bias), "s"(bb), "n"(ni * 64) : "memory");
    }

    w += gridDim.x;
    const bool more = w < total;
    if (more) {
      tid = decode_item(w, total, tiles_m, g.tiles_n, g.band_m);
      prologue(tid);
    }
    TSTAMP4(ts[3]);
    if (has_bias) {
      if (more) wait_vm<32>(); else wait_vm<0>();
      asm volatile("" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]));
    } else {
      if constexpr (RESID) { if (more) wait_vm<32>(); else wait_vm<0>(); }
      if constexpr (BIAS) {
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) bv[ni] = f4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
    TSTAMP4(ts[4]);
    // (the residual quads are older than the bias requests and the next item's prologue: they have landed)
    // Pack: quad j = (16-row block j / 4, 64-byte group j % 4) = fragments 2t and 2t+1 (t = j % 4) of block
    // mi = j / 4.  A lane holds, per fragment, four consecutive columns of row lane & 15 at column group
    // q = lane >> 4; exchanging 16-lane rows between the packed fragments (odd rows of one with even rows of
    // the other) leaves every lane with EIGHT consecutive columns of one fragment: lane (row, q) holds
    // columns (q >> 1) * 8 .. + 7 of fragment 2t + (q & 1).  A store then covers 16 rows x 64 bytes.
    {
      const char *cb = (const char *)(g.C + (m0 + wm * 128) * g.ldc + n0 + wn * 128);
      i4s_t srd;
      srd[0] = (int)(uintptr_t)cb;
      srd[1] = (int)(((uintptr_t)cb >> 32) & 0xffff);
      // rows at and beyond the device-side count are not written: the descriptor ends with the last valid row
      // of this wave's block (a row is at least 512 bytes apart from the next, a block row 256 bytes long).
      // (The tile coordinates come out of an integer division, i.e. of the vector unit: the result is pinned
      // to a scalar register.)
      int left = mdyn - (int)(m0 + wm * 128);
      left = left < 0 ? 0 : (left > 128 ? 128 : left);
      srd[2] = __builtin_amdgcn_readfirstlane(mdyn == 0x7fffffff ? -1 : left * (int)(g.ldc * 2));
      srd[3] = 0x00020000;
      // out of the accumulation registers HERE (left to the compiler, all 256 are copied out at once, right
      // behind the loop, and the vector registers spill)
      auto rd = [&](const f4_t &x) {
        f4_t r;
        asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                     : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : "a"(x[0]), "a"(x[1]), "a"(x[2]), "a"(x[3]));
        return r;
      };
      auto packq = [&](auto self, auto J) {
        constexpr int j = decltype(J)::value;
        if constexpr (j < 32) {
          constexpr int mi = j / 4, t = j % 4;
          f4_t a = rd(acc[2 * t][mi]), b = rd(acc[2 * t + 1][mi]);
          if constexpr (BIAS) { a += bv[2 * t]; b += bv[2 * t + 1]; }
          if constexpr (RESID) {
            // the residual quad has the layout of the packed output; the same exchange of 16-lane rows takes
            // it back to the accumulators' layout (the exchange is its own inverse)
            asm volatile("" : "+v"(pk[j]));
            const auto s0 = __builtin_amdgcn_permlane16_swap(pk[j][0], pk[j][2], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(pk[j][1], pk[j][3], false, false);
            a[0] += __uint_as_float(s0[0] << 16); a[1] += __uint_as_float(s0[0] & 0xffff0000u);
            a[2] += __uint_as_float(s1[0] << 16); a[3] += __uint_as_float(s1[0] & 0xffff0000u);
            b[0] += __uint_as_float(s0[1] << 16); b[1] += __uint_as_float(s0[1] & 0xffff0000u);
            b[2] += __uint_as_float(s1[1] << 16); b[3] += __uint_as_float(s1[1] & 0xffff0000u);
          }
          const unsigned a0 = pack_bf16x2(a[0], a[1]), a1 = pack_bf16x2(a[2], a[3]);
          const unsigned b0 = pack_bf16x2(b[0], b[1]), b1 = pack_bf16x2(b[2], b[3]);
          const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
          const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
          pk[j][0] = r0[0]; pk[j][1] = r1[0]; pk[j][2] = r0[1]; pk[j][3] = r1[1];
          // the first NOW quads leave while the others are packed: one store behind every fourth packed quad
          constexpr int every = 32 / NOW;
          if constexpr (j % every == every - 1) store_q(pk[j / every], srd, (unsigned)((j / every) / 4) * ldc16, IC<(j / every) % 4>{});
          __builtin_amdgcn_sched_barrier(0);
          self(self, IC<j + 1>{});
        }
      };
      packq(packq, IC<0>{});
      srd_prev = srd;
      if (!more) {
        // last item: the rest at once
        auto flush = [&](auto self, auto J) {
          constexpr int j = decltype(J)::value;
          if constexpr (j < 32) {
            store_q(pk[j], srd, (unsigned)(j / 4) * ldc16, IC<j % 4>{});
            self(self, IC<j + 1>{});
          }
        };
        flush(flush, IC<NOW>{});
      }
    }
#ifdef LDM_GEMM_TIMING
    TSTAMP4(ts[5]);
    for (int i = 0; i < 5; ++i) tsum[i] += ts[i + 1] - ts[i];
    tsum[5] += 1;
    tsum6 += tm1 - ts[1];
    tsum7 += ts[2] - tm2;
#endif
    if (!more) break;
  }
#ifdef LDM_GEMM_TIMING
  if (g.timing && blockIdx.x == 17 && lane == 0)
  {
    for (int i = 0; i < 6; ++i) atomicAdd(g.timing + wave * 8 + i, tsum[i]);
    atomicAdd(g.timing + wave * 8 + 6, tsum6);     // of the main loop: the steps that carry deferred stores
    atomicAdd(g.timing + wave * 8 + 7, tsum7);     // and the last four (no LDS-DMA)
  }
#endif
}

template <bool RESID, bool BIAS>
__global__ __launch_bounds__(256, 1) void gemm4w_k(WArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // one copy of the code per wave: the copies differ in which gap of a step holds what
  if (wave == 0) gemm4w_body<0, RESID, BIAS>(g, smem);
  else if (wave == 1) gemm4w_body<1, RESID, BIAS>(g, smem);
  else if (wave == 2) gemm4w_body<2, RESID, BIAS>(g, smem);
  else gemm4w_body<3, RESID, BIAS>(g, smem);
}

int num_cus4() {
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

// LDM_GEMM_4W=0 keeps every problem on the eight-wave kernel (A/B measurements)
bool ldm_gemm_4w_enabled() {
  static const bool on = !(getenv("LDM_GEMM_4W") && atoi(getenv("LDM_GEMM_4W")) == 0);
  return on;
}

// C[M, N] (bf16) = A[M, K] * B[N, K]^T (+ bias[N]) (+ R[M, N], bf16, row stride ldc).  LDM_E_SHAPE /
// LDM_E_ALIGN for what the kernel does not take (the caller then runs the eight-wave kernel).  dyn_m
// (nullable): device-side row count <= M; A must hold all M rows.
int ldm_gemm_mfma4w_bf16(int64_t M, int64_t N, int64_t K, const void *A, int64_t lda, const void *B, int64_t ldb,
                         void *C, int64_t ldc, const float *bias, const void *R, const int *dyn_m, hipStream_t s) {
  if (M < 256 || M % 256 || N < 256 || N % 256) return LDM_E_SHAPE;
  if (K % 64 || K < (int64_t)(NSS + 4) * KH) return LDM_E_SHAPE;
  if (lda % 8 || ldb % 8 || ldc % 8 || lda < K || ldb < K || ldc < N) return LDM_E_ALIGN;
  if (!aligned_to(A, 16) || !aligned_to(B, 16) || !aligned_to(C, 16) || (bias && !aligned_to(bias, 16)) ||
      (R && !aligned_to(R, 16)))
    return LDM_E_ALIGN;
  // 32-bit lane offsets: 16 rows of an operand plus the k offset, 128 rows of C
  if (lda >= (1 << 26) || ldb >= (1 << 26) || ldc >= (1 << 23)) return LDM_E_SHAPE;
  const int64_t tm = M / 256, tn = N / 256;
  if (tm * tn > 0x3fffffff) return LDM_E_SHAPE;
  WArgs g;
  g.M = M; g.N = N; g.K = K;
  g.A = (const bf16_t *)A; g.lda = lda; g.B = (const bf16_t *)B; g.ldb = ldb; g.C = (bf16_t *)C; g.ldc = ldc;
  g.bias = bias;
  g.R = (const bf16_t *)R;
  g.dyn_m = dyn_m;
  g.tiles_m = (int)tm; g.tiles_n = (int)tn;
#ifdef LDM_GEMM_TIMING
  { const char *tb = getenv("LDM_GEMM4W_TIMING_BUF"); g.timing = tb ? (unsigned long long *)strtoull(tb, nullptr, 0) : nullptr; }
#endif
  {
    // m-tiles per band of the tile walk: at most 8, fewer while the band's A panel exceeds 8 MB (measured at
    // M = 262144, one box: K = 768 -- N = 768: 290 us at 8 against 302 at 16; N = 2304: 829 against 998;
    // K = 2304, N = 768: 691 at 4 or 8 against 749 at 16)
    static const int env = getenv("LDM_GEMM_BAND") ? atoi(getenv("LDM_GEMM_BAND")) : 0;
    int b = 8;
    while (b > 2 && (int64_t)b * 512 * K > (int64_t)8 << 20) b >>= 1;
    g.band_m = env > 0 ? env : b;
  }
  static std::atomic<uint64_t> attr_done{0};
  uint64_t dev_bit;
  if (ldm_attr_pending(attr_done, &dev_bit)) {
    (void)hipFuncSetAttribute((const void *)gemm4w_k<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void *)gemm4w_k<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void *)gemm4w_k<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done.fetch_or(dev_bit, std::memory_order_release);
  }
  const int64_t items = tm * tn;
  // persistent: one workgroup per CU (LDM_GEMM4W_WGS: fewer, for experiments on what bounds the kernel)
  static const int wgs_env = getenv("LDM_GEMM4W_WGS") ? atoi(getenv("LDM_GEMM4W_WGS")) : 0;
  const int64_t cus = wgs_env > 0 && wgs_env < num_cus4() ? wgs_env : num_cus4();
  const int64_t nwg = items < cus ? items : cus;
  if (R) hipLaunchKernelGGL((gemm4w_k<true, true>), dim3((unsigned)nwg), dim3(256), LDS_BYTES, s, g);
  else if (bias) hipLaunchKernelGGL((gemm4w_k<false, true>), dim3((unsigned)nwg), dim3(256), LDS_BYTES, s, g);
  else hipLaunchKernelGGL((gemm4w_k<false, false>), dim3((unsigned)nwg), dim3(256), LDS_BYTES, s, g);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}
