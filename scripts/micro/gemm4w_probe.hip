// Probe: bf16 GEMM C[M,N] = A[M,K] * B[N,K]^T with FOUR waves per workgroup (one per SIMD), each owning
// 128 x 128 outputs of a 256 x 256 tile, against the library's eight-wave kernel on the same box.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/micro/gemm4w_probe.hip -o gpurun_out/gemm4w_probe -ldl
//   gpurun_out/gemm4w_probe [M N K]...
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

typedef unsigned short bf16_t;
typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef unsigned u4v_t __attribute__((ext_vector_type(4)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int BM = 256, BN = 256, KH = 32, NSLOT = 5;
constexpr int HT_A = BM * 64, HT_B = BN * 64, HT = HT_A + HT_B;
constexpr int LDS_BYTES = 160 * 1024;
constexpr int STAGE_OFF = 4 * HT;

struct PArgs {
  int64_t M, N, K;
  const bf16_t *A; int64_t lda;
  const bf16_t *B; int64_t ldb;
  bf16_t *C; int64_t ldc;
  int tiles_m, tiles_n, band_m;
};

__device__ __forceinline__ int swz_kc(int row) { return 3 * ((row >> 2) & 1); }

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
  typedef float fl2_t __attribute__((ext_vector_type(2)));
  const fl2_t f{lo, hi};
  const bf2_t h = __builtin_convertvector(f, bf2_t);
  return __builtin_bit_cast(unsigned, h);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int LO, int HI>
__device__ __forceinline__ void wait_vm_sw(int n) {
  if constexpr (LO == HI) wait_vm<LO>();
  else {
    constexpr int MID = (LO + HI) / 2;
    if (n <= MID) wait_vm_sw<LO, MID>(n); else wait_vm_sw<MID + 1, HI>(n);
  }
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

__device__ __forceinline__ bf8_t read_frag_kc(const char *img, int rbase, int lane) {
  const int r = rbase + (lane & 15);
  const int pc = (lane >> 4) ^ swz_kc(r);
  return *reinterpret_cast<const bf8_t *>(img + r * 64 + pc * 16);
}

// PAIR: half-tiles (2q, 2q+1) requested together in even steps (both 64-byte halves of a 128-byte line
// back to back); else one half-tile per step.
// RPG: LDS fragment reads per group of eight MFMAs (2: spread over all eight groups; 4: first four groups)
template <int PAIR, int RPG, int NOEPI, int ABL = 0>
__global__ __launch_bounds__(256, 1) void g4w_k(PArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int total = g.tiles_m * g.tiles_n;
  const int np = (int)(g.K / KH);
  int w = blockIdx.x;
  if (w >= total) return;

  // per-lane byte offsets of this wave's four pieces (16 rows x 64 B each) per operand, relative to the
  // tile's first row at the half-tile's first k
  unsigned offA[4], offB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave + 4 * i) * 16 + (lane >> 2);
    const int lc = (lane & 3) ^ swz_kc(r);
    offA[i] = (unsigned)(r * (int)g.lda * 2 + lc * 16);
    offB[i] = (unsigned)(r * (int)g.ldb * 2 + lc * 16);
  }
  const char *tA = nullptr, *tB = nullptr;   // tile bases (scalar)
  auto dma = [&](const char *sbase, unsigned off, char *dst) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(sbase + off), LDS_PTR(dst), 16, 0, 0);
  };
  // vmcnt bookkeeping: `issued` counts this wave's vector-memory instructions; mark[] the count right after
  // the request of a (pair of) half-tile(s)
  int issued = 0;
  // one piece (both operands' i-th piece) of half-tile h / of the pair (h, h+1)
  auto issue_piece_single = [&](int h, int i, bool opB) {
    const int hh = h < np ? h : np - 1;
    char *slot = smem + (h % NSLOT) * HT;
    if (!opB) dma(tA + (int64_t)hh * 64, offA[i], slot + (wave + 4 * i) * 1024);
    else dma(tB + (int64_t)hh * 64, offB[i], slot + HT_A + (wave + 4 * i) * 1024);
    issued += 1;
  };
  auto issue_piece_pair = [&](int h, int i, bool opB) {
    const int h0 = h < np ? h : np - 1, h1 = h + 1 < np ? h + 1 : np - 1;
    char *s0 = smem + (h % NSLOT) * HT, *s1 = smem + ((h + 1) % NSLOT) * HT;
    if (!opB) {
      dma(tA + (int64_t)h0 * 64, offA[i], s0 + (wave + 4 * i) * 1024);
      dma(tA + (int64_t)h1 * 64, offA[i], s1 + (wave + 4 * i) * 1024);
    } else {
      dma(tB + (int64_t)h0 * 64, offB[i], s0 + HT_A + (wave + 4 * i) * 1024);
      dma(tB + (int64_t)h1 * 64, offB[i], s1 + HT_A + (wave + 4 * i) * 1024);
    }
    issued += 2;
  };
  int mark01 = 0, mark23 = 0;   // after the prologue requests of half-tiles (0,1) and (2,3)
  auto prologue = [&](TileId t) {
    tA = (const char *)(g.A + (int64_t)t.tm * BM * g.lda);
    tB = (const char *)(g.B + (int64_t)t.tn * BN * g.ldb);
#pragma unroll
    for (int i = 0; i < 4; ++i) { issue_piece_pair(0, i, false); issue_piece_pair(0, i, true); }
    mark01 = issued;
#pragma unroll
    for (int i = 0; i < 4; ++i) { issue_piece_pair(2, i, false); issue_piece_pair(2, i, true); }
    mark23 = issued;
  };
  auto wait_mark = [&](int mark) {
    int young = issued - mark;
    if (young > 63) young = 63;
    wait_vm_sw<0, 63>(young);
  };

  TileId tid = decode_item(w, total, g.tiles_m, g.tiles_n, g.band_m);
  prologue(tid);
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);
  const unsigned stage_lds = lds0 + STAGE_OFF + wave * 8192;

  while (true) {
    const int64_t m0 = (int64_t)tid.tm * BM, n0 = (int64_t)tid.tn * BN;
    f4_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
    bf8_t fa[2][8], fb[2][8];

    // half-tiles 0, 1 landed for every wave
    wait_mark(mark01);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      fa[0][i] = read_frag_kc(smem, wm * 128 + i * 16, lane);
      fb[0][i] = read_frag_kc(smem + HT_A, wn * 128 + i * 16, lane);
    }
    // request marks of the half-tiles the next three steps read (p+1, p+2, p+3)
    int m1 = mark01, m2 = mark23, m3 = mark23;

    auto step = [&](auto PAR, int p) {
      constexpr int cur = decltype(PAR)::value, nxt = cur ^ 1;
      if (!(ABL & 4)) {
      if (!PAIR || cur == 1) wait_mark(m1);
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's reads of half-tile p are complete
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      }
      // (the last step reads a slot that holds nothing: the values are not used)
      const char *rd = smem + ((p + 1) % NSLOT) * HT;
      const bool do_issue = !(ABL & 1) && p + 4 < np;
#pragma unroll
      for (int gi = 0; gi < 8; ++gi) {
        if constexpr (PAIR) {
          if (cur == 0 && do_issue) issue_piece_pair(p + 4, gi >> 1, gi & 1);
        } else {
          if (do_issue) issue_piece_single(p + 4, gi >> 1, gi & 1);
        }
        if constexpr (ABL & 2) {
          if (p == 0) {   // (ablation: the fragments are read once per item)
            fa[nxt][gi] = read_frag_kc(rd, wm * 128 + gi * 16, lane);
            fb[nxt][gi] = read_frag_kc(rd + HT_A, wn * 128 + gi * 16, lane);
            fa[cur][gi] = fa[nxt][gi]; fb[cur][gi] = fb[nxt][gi];
          }
        } else if constexpr (RPG == 2) {
          fa[nxt][gi] = read_frag_kc(rd, wm * 128 + gi * 16, lane);
          fb[nxt][gi] = read_frag_kc(rd + HT_A, wn * 128 + gi * 16, lane);
        } else {
          if (gi < 4) {
            fa[nxt][2 * gi] = read_frag_kc(rd, wm * 128 + (2 * gi) * 16, lane);
            fb[nxt][2 * gi] = read_frag_kc(rd + HT_A, wn * 128 + (2 * gi) * 16, lane);
            fa[nxt][2 * gi + 1] = read_frag_kc(rd, wm * 128 + (2 * gi + 1) * 16, lane);
            fb[nxt][2 * gi + 1] = read_frag_kc(rd + HT_A, wn * 128 + (2 * gi + 1) * 16, lane);
          }
        }
#pragma unroll
        for (int ni = 0; ni < 8; ++ni)
          acc[ni][gi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[cur][ni], fa[cur][gi], acc[ni][gi], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      m1 = m2;
      m2 = m3;
      if ((!PAIR || cur == 0) && do_issue) m3 = issued;
    };
    for (int p = 0; p < np; p += 2) {
      step(std::integral_constant<int, 0>{}, p);
      step(std::integral_constant<int, 1>{}, p + 1);
    }

    // next item: its first four half-tiles go out before this item's stores
    w += gridDim.x;
    const bool more = w < total;
    if (more) {
      tid = decode_item(w, total, g.tiles_m, g.tiles_n, g.band_m);
      prologue(tid);
    }
    if (!NOEPI) {
      const int cl = lane & 15, wrow = lane & 15;
      const unsigned waddr = stage_lds + wrow * 512;
      bf16_t *crow = g.C + (m0 + wm * 128 + (lane >> 4)) * g.ldc + n0 + wn * 128 + cl * 8;
      const int64_t rstep = 4 * g.ldc;
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
          const int c4 = ni * 4 + (lane >> 4);
          asm volatile("ds_write_b128 %0, %1" ::"v"(waddr + ((c4 ^ wrow) << 4)), "v"(acc[ni][mi]) : "memory");
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = it * 4 + (lane >> 4);
          f4_t t0, t1;
          asm volatile("ds_read_b128 %0, %1" : "=v"(t0) : "v"(stage_lds + row * 512 + (((cl * 2) ^ row) << 4)) : "memory");
          asm volatile("ds_read_b128 %0, %1" : "=v"(t1) : "v"(stage_lds + row * 512 + (((cl * 2 + 1) ^ row) << 4)) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0), "+v"(t1)::"memory");
          u4v_t pk;
          pk[0] = pack_bf16x2(t0[0], t0[1]);
          pk[1] = pack_bf16x2(t0[2], t0[3]);
          pk[2] = pack_bf16x2(t1[0], t1[1]);
          pk[3] = pack_bf16x2(t1[2], t1[3]);
          __builtin_nontemporal_store(pk, reinterpret_cast<u4v_t *>(crow));
          issued += 1;
          crow += rstep;
        }
      }
    } else {
      // keep the accumulators alive
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (s == 12345.678f) g.C[0] = 1;
    }
    if (!more) break;
  }
}


// accumulator pinned to the accumulation registers, program order kept (volatile)
__device__ __forceinline__ void mfma_a(f4_t &acc, const bf8_t &a, const bf8_t &b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

__device__ __forceinline__ void mfma_a0(f4_t &acc, const bf8_t &a, const bf8_t &b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
}

// Second structure: per step 64 MFMAs with ONE other instruction (an LDS fragment read or an LDS-DMA request)
// behind every second MFMA; the four pieces a wave sends per operand and half-tile are adjacent in the image
// (piece = 4 wave + i) and share one M0 value, the piece offset travels in the instruction's immediate.
// Pairs (h, h+1): the A operand's pieces go out in even steps, the B operand's in odd steps.
//   FPM: MFMAs per filler (2: fillers behind every second MFMA; 1: behind every MFMA)
template <int FPM, int NOEPI, int ABL, int WR>
__device__ __forceinline__ void g4w2_body(const PArgs &g, char *smem) {
  const int lane = threadIdx.x & 63;
  constexpr int wave = WR;
  const int wm = wave >> 1, wn = wave & 1;
  const int total = g.tiles_m * g.tiles_n;
  const int np = (int)(g.K / KH);
  int w = blockIdx.x;
  if (w >= total) return;

  unsigned offA[4], offB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 16 + (lane >> 2);
    const int lc = (lane & 3) ^ swz_kc(r);
    offA[i] = (unsigned)(r * (int)g.lda * 2 + lc * 16 - i * 1024);
    offB[i] = (unsigned)(r * (int)g.ldb * 2 + lc * 16 - i * 1024);
  }
  const char *tA = nullptr, *tB = nullptr;
  int issued = 0;
  const unsigned lds_base = (unsigned)(uintptr_t)LDS_PTR(smem);
  // piece i of half-tile h (and of h + 1) of one operand
  auto issue_pair_piece = [&](auto I, int h, bool opB) {
    constexpr int i = decltype(I)::value;
    const int h0 = h < np ? h : np - 1, h1 = h + 1 < np ? h + 1 : np - 1;
    char *d0 = smem + (h % NSLOT) * HT + (opB ? HT_A : 0) + wave * 4096;
    char *d1 = smem + ((h + 1) % NSLOT) * HT + (opB ? HT_A : 0) + wave * 4096;
    const char *sb = opB ? tB : tA;
    const unsigned off = opB ? offB[i] : offA[i];
    __builtin_amdgcn_global_load_lds(GLB_PTR(sb + (int64_t)h0 * 64 + off), LDS_PTR(d0), 16, i * 1024, 0);
    __builtin_amdgcn_global_load_lds(GLB_PTR(sb + (int64_t)h1 * 64 + off), LDS_PTR(d1), 16, i * 1024, 0);
    issued += 2;
  };
  auto issue_one = [&](auto I, int h, bool opB) {
    constexpr int i = decltype(I)::value;
    const int h0 = h < np ? h : np - 1;
    const unsigned d0 = lds_base + (h % NSLOT) * HT + (opB ? HT_A : 0) + wave * 4096;
    const char *sb = (opB ? tB : tA) + (int64_t)h0 * 64;
    const unsigned off = opB ? offB[i] : offA[i];
    // scalar base + 32-bit lane offset; M0 = LDS base of the wave's four pieces, piece i by the immediate
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3" ::"s"(d0), "v"(off), "s"(sb), "n"(i * 1024) : "memory");
    issued += 1;
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, 2> I2;
  typedef std::integral_constant<int, 3> I3;
  int mark01 = 0, mark23 = 0;
  auto prologue = [&](TileId t) {
    tA = (const char *)(g.A + (int64_t)t.tm * BM * g.lda);
    tB = (const char *)(g.B + (int64_t)t.tn * BN * g.ldb);
    issue_pair_piece(I0{}, 0, false); issue_pair_piece(I1{}, 0, false); issue_pair_piece(I2{}, 0, false); issue_pair_piece(I3{}, 0, false);
    issue_pair_piece(I0{}, 0, true); issue_pair_piece(I1{}, 0, true); issue_pair_piece(I2{}, 0, true); issue_pair_piece(I3{}, 0, true);
    mark01 = issued;
    issue_pair_piece(I0{}, 2, false); issue_pair_piece(I1{}, 2, false); issue_pair_piece(I2{}, 2, false); issue_pair_piece(I3{}, 2, false);
    issue_pair_piece(I0{}, 2, true); issue_pair_piece(I1{}, 2, true); issue_pair_piece(I2{}, 2, true); issue_pair_piece(I3{}, 2, true);
    mark23 = issued;
  };
  auto wait_mark = [&](int mark) {
    int young = issued - mark;
    if (young > 63) young = 63;
    wait_vm_sw<0, 63>(young);
  };

  TileId tid = decode_item(w, total, g.tiles_m, g.tiles_n, g.band_m);
  if constexpr (ABL & 16) {
    // start stagger: workgroup b waits (b / 8 % 32) x ~0.6 us, so that the workgroups' store phases do not coincide
    const int n = ((blockIdx.x >> 3) & 31) * 16;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
  }
  prologue(tid);
  const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);
  const unsigned stage_lds = lds0 + STAGE_OFF + wave * 8192;

  while (true) {
    const int64_t m0 = (int64_t)tid.tm * BM, n0 = (int64_t)tid.tn * BN;
    f4_t acc[8][8];
    if constexpr (NOEPI != 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};
    }
    bf8_t fa[2][8], fb[2][8];

    wait_mark(mark01);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      fa[0][i] = read_frag_kc(smem, wm * 128 + i * 16, lane);
      fb[0][i] = read_frag_kc(smem + HT_A, wn * 128 + i * 16, lane);
    }
    int mneed = mark23;   // the request an odd step waits for: pair (p+1, p+2), complete

    auto step = [&](auto PAR, auto FIRST, int p) {
      constexpr int cur = decltype(PAR)::value, nxt = cur ^ 1;
      constexpr bool first = decltype(FIRST)::value;
      if (!(ABL & 4)) {
        if (cur == 1) wait_mark(mneed);
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's reads of half-tile p are complete
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      const char *rd = smem + ((p + 1) % NSLOT) * HT;
      // even step p: A pieces of pair (p+4, p+5); odd step p: B pieces of pair (p+3, p+4)
      const int hpair = cur == 0 ? p + 4 : p + 3;
      const bool do_issue = !(ABL & 1);   // (a half-tile past the last one is a second request of the last one, into a free slot)
      // fillers: 16 fragment reads and 4 pair-piece requests (two LDS-DMA each), one behind every FPM-th MFMA:
      // slot s = 0..19 -> s % 5 == 2: request s / 5, else the next read (A and B fragments alternate)
      auto filler = [&](auto S) {
        constexpr int s_ = decltype(S)::value;
        if constexpr (FPM == 2) {
          // 32 gaps: a fragment read in every even gap; in gap s wave (s % 4) sends one LDS-DMA (its j-th of
          // this step, j = s / 4: piece j / 2, half-tile j % 2 of the pair), so that the CU's address unit sees
          // one request per gap instead of four waves' requests at once
          if constexpr (s_ % 2 == 0 && !(ABL & 2)) {
            constexpr int r = s_ / 2;
            if constexpr (r % 2 == 0) fa[nxt][r / 2] = read_frag_kc(rd, wm * 128 + (r / 2) * 16, lane);
            else fb[nxt][r / 2] = read_frag_kc(rd + HT_A, wn * 128 + (r / 2) * 16, lane);
          }
          if constexpr (!(ABL & 1) && (s_ & 3) == WR) issue_one(std::integral_constant<int, (s_ / 4) / 2>{}, hpair + ((s_ / 4) & 1), cur == 1);
        } else if constexpr (s_ < 20) {
          if constexpr (s_ % 5 == 2) {
            if (do_issue) issue_pair_piece(std::integral_constant<int, s_ / 5>{}, hpair, cur == 1);
          } else if constexpr (!(ABL & 2)) {
            constexpr int r = s_ - (s_ + 2) / 5;   // read index 0..15
            if constexpr (r % 2 == 0) fa[nxt][r / 2] = read_frag_kc(rd, wm * 128 + (r / 2) * 16, lane);
            else fb[nxt][r / 2] = read_frag_kc(rd + HT_A, wn * 128 + (r / 2) * 16, lane);
          }
        }
      };
      auto mf = [&](auto K) {
        constexpr int k = decltype(K)::value;
        constexpr int gi = k / 8, ni = k % 8;
        if constexpr (first) mfma_a0(acc[ni][gi], fb[cur][ni], fa[cur][gi]);
        else mfma_a(acc[ni][gi], fb[cur][ni], fa[cur][gi]);
      };
      // (a compile-time loop over the 64 MFMAs)
      auto run = [&](auto self, auto K) {
        constexpr int k = decltype(K)::value;
        if constexpr (k < 64) {
          mf(K);
          if constexpr ((k + 1) % FPM == 0) {
            __builtin_amdgcn_sched_barrier(0);
            filler(std::integral_constant<int, (k + 1) / FPM - 1>{});
            __builtin_amdgcn_sched_barrier(0);
          }
          self(self, std::integral_constant<int, k + 1>{});
        }
      };
      run(run, std::integral_constant<int, 0>{});
      if constexpr (ABL & 2) {
        if (p == 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { fa[nxt][i] = fa[cur][i]; fb[nxt][i] = fb[cur][i]; }
        }
      }
      if (cur == 1 && do_issue && hpair < np) mneed = issued;
    };
    if constexpr (NOEPI == 2) {
      step(std::integral_constant<int, 0>{}, std::true_type{}, 0);
      step(std::integral_constant<int, 1>{}, std::false_type{}, 1);
      for (int p = 2; p < np; p += 2) {
        step(std::integral_constant<int, 0>{}, std::false_type{}, p);
        step(std::integral_constant<int, 1>{}, std::false_type{}, p + 1);
      }
    } else {
      for (int p = 0; p < np; p += 2) {
        step(std::integral_constant<int, 0>{}, std::false_type{}, p);
        step(std::integral_constant<int, 1>{}, std::false_type{}, p + 1);
      }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (the asm MFMAs' results: wait states the compiler does not count)

    w += gridDim.x;
    const bool more = w < total;
    if (more) {
      tid = decode_item(w, total, g.tiles_m, g.tiles_n, g.band_m);
      prologue(tid);
    }
    if constexpr (NOEPI == 2) {
      // straight from the accumulator layout: fragments 2t and 2t+1 exchange 16-lane rows so that every lane
      // holds eight consecutive columns (16 bytes) of one of them; a store instruction covers 16 rows x 64 bytes
      // (the lane number as an opaque value: nothing derived from it for the stores is kept in registers
      // across the main loop)
      int lane_e = threadIdx.x & 63;
      asm volatile("" : "+v"(lane_e));
      const int q = lane_e >> 4;
      bf16_t *crow = g.C + (m0 + wm * 128 + (lane_e & 15)) * g.ldc + n0 + wn * 128 + (q & 1) * 16 + (q >> 1) * 8;
      if constexpr (ABL & 8)   // (ablation: every item of a workgroup stores to the same 128 KiB)
        crow = g.C + ((int64_t)blockIdx.x * 256 + wm * 128 + (lane_e & 15)) * g.ldc + wn * 128 + (q & 1) * 16 + (q >> 1) * 8;
      const int64_t rstep = 16 * g.ldc;
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f4_t a = acc[2 * t][mi], b = acc[2 * t + 1][mi];
          const unsigned a0 = pack_bf16x2(a[0], a[1]), a1 = pack_bf16x2(a[2], a[3]);
          const unsigned b0 = pack_bf16x2(b[0], b[1]), b1 = pack_bf16x2(b[2], b[3]);
          const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
          const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
          u4v_t pk;
          pk[0] = r0[0]; pk[1] = r1[0]; pk[2] = r0[1]; pk[3] = r1[1];
          __builtin_nontemporal_store(pk, reinterpret_cast<u4v_t *>(crow + t * 32));
          issued += 1;
          __builtin_amdgcn_sched_barrier(0);   // (one fragment pair at a time: 16 registers live, not 256)
        }
        crow += rstep;
      }
    } else if (!NOEPI) {
      const int cl = lane & 15, wrow = lane & 15;
      const unsigned waddr = stage_lds + wrow * 512;
      bf16_t *crow = g.C + (m0 + wm * 128 + (lane >> 4)) * g.ldc + n0 + wn * 128 + cl * 8;
      const int64_t rstep = 4 * g.ldc;
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
          const int c4 = ni * 4 + (lane >> 4);
          asm volatile("ds_write_b128 %0, %1" ::"v"(waddr + ((c4 ^ wrow) << 4)), "v"(acc[ni][mi]) : "memory");
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = it * 4 + (lane >> 4);
          f4_t t0, t1;
          asm volatile("ds_read_b128 %0, %1" : "=v"(t0) : "v"(stage_lds + row * 512 + (((cl * 2) ^ row) << 4)) : "memory");
          asm volatile("ds_read_b128 %0, %1" : "=v"(t1) : "v"(stage_lds + row * 512 + (((cl * 2 + 1) ^ row) << 4)) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0), "+v"(t1)::"memory");
          u4v_t pk;
          pk[0] = pack_bf16x2(t0[0], t0[1]);
          pk[1] = pack_bf16x2(t0[2], t0[3]);
          pk[2] = pack_bf16x2(t1[0], t1[1]);
          pk[3] = pack_bf16x2(t1[2], t1[3]);
          __builtin_nontemporal_store(pk, reinterpret_cast<u4v_t *>(crow));
          issued += 1;
          crow += rstep;
        }
      }
    } else {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (s == 12345.678f) g.C[0] = 1;
    }
    if (!more) break;
  }
}

template <int FPM, int NOEPI, int ABL>
__global__ __launch_bounds__(256, 1) void g4w2_k(PArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // one copy of the whole kernel per wave: the waves differ in WHERE in a step they send their LDS-DMA
  if (wave == 0) g4w2_body<FPM, NOEPI, ABL, 0>(g, smem);
  else if (wave == 1) g4w2_body<FPM, NOEPI, ABL, 1>(g, smem);
  else if (wave == 2) g4w2_body<FPM, NOEPI, ABL, 2>(g, smem);
  else g4w2_body<FPM, NOEPI, ABL, 3>(g, smem);
}


// Third structure: as the second, with every per-lane address reduced to ONE register per operand plus scalar
// terms (no address is kept per slot or per piece, nothing spills), reads placed by a table.
//   EPI: 1 = none (timing), 2 = straight from the accumulator layout
//   ABL: 1 no LDS-DMA in the loop, 2 no fragment reads, 32 = reads in the second half of the step
template <int EPI, int ABL, int WR>
__device__ __forceinline__ void g4w3_body(const PArgs &g, char *smem) {
  constexpr int wave = WR, wm = WR >> 1, wn = WR & 1;
  const int total = g.tiles_m * g.tiles_n;
  const int np = (int)(g.K / KH);
  int w = blockIdx.x;
  if (w >= total) return;
  const unsigned lds_base = (unsigned)(uintptr_t)LDS_PTR(smem);
  int lane = threadIdx.x & 63;
  // fragment reads: lane part of the address (row lane & 15 of a 16-row block, chunk (lane >> 4) ^ swizzle)
  const unsigned lane_rd = (unsigned)((lane & 15) * 64 + (((lane >> 4) ^ swz_kc(lane & 15)) << 4));
  // LDS-DMA: lane part of the source offset within a 16-row piece; the piece's first row, the k offset of the
  // half-tile and the immediate's bias are scalar
  const unsigned laneA = (unsigned)((lane >> 2) * (int)g.lda * 2 + (((lane & 3) ^ swz_kc(lane >> 2)) << 4));
  const unsigned laneB = (unsigned)((lane >> 2) * (int)g.ldb * 2 + (((lane & 3) ^ swz_kc(lane >> 2)) << 4));
  const int64_t pieceA = (int64_t)32 * g.lda - 1024, pieceB = (int64_t)32 * g.ldb - 1024;   // bytes from piece i to i+1, less the immediate
  const char *tA = nullptr, *tB = nullptr;   // first byte of piece 0 of this wave in the item's A / B panel
  int issued = 0;
  auto issue_one = [&](auto I, int h, bool opB) {
    constexpr int i = decltype(I)::value;
    const int h0 = h < np ? h : np - 1;
    const unsigned d0 = lds_base + (h % NSLOT) * HT + (opB ? HT_A : 0) + wave * 4096;
    const char *sb = (opB ? tB + i * pieceB : tA + i * pieceA) + (int64_t)h0 * 64;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3" ::"s"(d0), "v"(opB ? laneB : laneA), "s"(sb), "n"(i * 1024) : "memory");
    issued += 1;
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, 2> I2;
  typedef std::integral_constant<int, 3> I3;
  int mark01 = 0, mark23 = 0;
  auto prologue = [&](TileId t) {
    tA = (const char *)(g.A + ((int64_t)t.tm * BM + wave * 64) * g.lda);
    tB = (const char *)(g.B + ((int64_t)t.tn * BN + wave * 64) * g.ldb);
    auto pair = [&](int h) {
      issue_one(I0{}, h, false); issue_one(I0{}, h + 1, false); issue_one(I1{}, h, false); issue_one(I1{}, h + 1, false);
      issue_one(I2{}, h, false); issue_one(I2{}, h + 1, false); issue_one(I3{}, h, false); issue_one(I3{}, h + 1, false);
      issue_one(I0{}, h, true); issue_one(I0{}, h + 1, true); issue_one(I1{}, h, true); issue_one(I1{}, h + 1, true);
      issue_one(I2{}, h, true); issue_one(I2{}, h + 1, true); issue_one(I3{}, h, true); issue_one(I3{}, h + 1, true);
    };
    pair(0);
    mark01 = issued;
    pair(2);
    mark23 = issued;
  };
  auto wait_mark = [&](int mark) {
    int young = issued - mark;
    if (young > 63) young = 63;
    wait_vm_sw<0, 63>(young);
  };
  auto frag = [&](unsigned va, int off) {
    return *reinterpret_cast<const __attribute__((address_space(3))) bf8_t *>((uintptr_t)(va + off));
  };

  TileId tid = decode_item(w, total, g.tiles_m, g.tiles_n, g.band_m);
  prologue(tid);

  while (true) {
    const int64_t m0 = (int64_t)tid.tm * BM, n0 = (int64_t)tid.tn * BN;
    f4_t acc[8][8];
    bf8_t fa[2][8], fb[2][8];

    wait_mark(mark01);
    __builtin_amdgcn_s_waitcnt(0xC07F);
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
    int mneed = mark23;

    auto step = [&](auto PAR, auto FIRST, int p) {
      constexpr int cur = decltype(PAR)::value, nxt = cur ^ 1;
      constexpr bool first = decltype(FIRST)::value;
      if (cur == 1) wait_mark(mneed);
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's reads of half-tile p are complete
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      unsigned va = lane_rd + lds_base + ((p + 1) % NSLOT) * HT;
      asm volatile("" : "+v"(va));
      const int hpair = cur == 0 ? p + 4 : p + 3;
      auto filler = [&](auto S) {
        constexpr int s_ = decltype(S)::value;
        // 32 gaps (one behind every second MFMA).  Reads: 16, in the even gaps (ABL 32: in gaps 16..31).
        // LDS-DMA: in gap s by wave s % 4 (its j-th of this step, j = s / 4: piece j / 2, half-tile j % 2).
        constexpr int r = (ABL & 32) ? s_ - 16 : ((s_ % 2 == 0) ? s_ / 2 : -1);
        if constexpr (r >= 0 && r < 16 && !(ABL & 2)) {
          if constexpr (r % 2 == 0) fa[nxt][r / 2] = frag(va, wm * 8192 + (r / 2) * 1024);
          else fb[nxt][r / 2] = frag(va, HT_A + wn * 8192 + (r / 2) * 1024);
        }
        if constexpr (!(ABL & 1) && (s_ & 3) == WR) issue_one(std::integral_constant<int, (s_ / 4) / 2>{}, hpair + ((s_ / 4) & 1), cur == 1);
      };
      auto run = [&](auto self, auto K) {
        constexpr int k = decltype(K)::value;
        if constexpr (k < 64) {
          constexpr int gi = k / 8, ni = k % 8;
          if constexpr (first) mfma_a0(acc[ni][gi], fb[cur][ni], fa[cur][gi]);
          else mfma_a(acc[ni][gi], fb[cur][ni], fa[cur][gi]);
          if constexpr (k % 2 == 1) {
            __builtin_amdgcn_sched_barrier(0);
            filler(std::integral_constant<int, k / 2>{});
            __builtin_amdgcn_sched_barrier(0);
          }
          self(self, std::integral_constant<int, k + 1>{});
        }
      };
      run(run, std::integral_constant<int, 0>{});
      if constexpr (ABL & 2) {
        if (p == 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { fa[nxt][i] = fa[cur][i]; fb[nxt][i] = fb[cur][i]; }
        }
      }
      if (cur == 1 && !(ABL & 1) && hpair < np) mneed = issued;
    };
    step(std::integral_constant<int, 0>{}, std::true_type{}, 0);
    step(std::integral_constant<int, 1>{}, std::false_type{}, 1);
    for (int p = 2; p < np; p += 2) {
      step(std::integral_constant<int, 0>{}, std::false_type{}, p);
      step(std::integral_constant<int, 1>{}, std::false_type{}, p + 1);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (the asm MFMAs' results: wait states the compiler does not count)

    w += gridDim.x;
    const bool more = w < total;
    if (more) {
      tid = decode_item(w, total, g.tiles_m, g.tiles_n, g.band_m);
      prologue(tid);
    }
    if constexpr (EPI == 2) {
      int lane_e = threadIdx.x & 63;
      asm volatile("" : "+v"(lane_e));
      const int q = lane_e >> 4;
      bf16_t *crow = g.C + (m0 + wm * 128 + (lane_e & 15)) * g.ldc + n0 + wn * 128 + (q & 1) * 16 + (q >> 1) * 8;
      const int64_t rstep = 16 * g.ldc;
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f4_t a = acc[2 * t][mi], b = acc[2 * t + 1][mi];
          const unsigned a0 = pack_bf16x2(a[0], a[1]), a1 = pack_bf16x2(a[2], a[3]);
          const unsigned b0 = pack_bf16x2(b[0], b[1]), b1 = pack_bf16x2(b[2], b[3]);
          const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
          const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
          u4v_t pk;
          pk[0] = r0[0]; pk[1] = r1[0]; pk[2] = r0[1]; pk[3] = r1[1];
          __builtin_nontemporal_store(pk, reinterpret_cast<u4v_t *>(crow + t * 32));
          issued += 1;
          __builtin_amdgcn_sched_barrier(0);
        }
        crow += rstep;
      }
    } else {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3]; if ((j & 1) && (i & 1)) __builtin_amdgcn_sched_barrier(0); }
      if (s == 12345.678f) g.C[0] = 1;
    }
    if (!more) break;
  }
}
template <int EPI, int ABL>
__global__ __launch_bounds__(256, 1) void g4w3_k(PArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave == 0) g4w3_body<EPI, ABL, 0>(g, smem);
  else if (wave == 1) g4w3_body<EPI, ABL, 1>(g, smem);
  else if (wave == 2) g4w3_body<EPI, ABL, 2>(g, smem);
  else g4w3_body<EPI, ABL, 3>(g, smem);
}


// Fourth structure: as the third, with at most two instructions per LDS-DMA in the MFMA stream (s_mov m0 +
// the request: the piece bases are scalar registers set once per item, the half-tile's k offset is added to
// the lane offset once per step, the second half-tile of a pair is addressed by the instruction's immediate
// against an M0 value 64 bytes lower), steps that send nothing compiled separately (no branch per gap), and
// the accumulators read out of the accumulation registers where they are packed.
// K must be a multiple of 64 and at least 192.
template <int EPI, int ABL, int WR>
__device__ __forceinline__ void g4w4_body(const PArgs &g, char *smem) {
  constexpr int wave = WR, wm = WR >> 1, wn = WR & 1;
  const int total = g.tiles_m * g.tiles_n;
  const int np = (int)(g.K / KH);
  int w = blockIdx.x;
  if (w >= total) return;
  const unsigned lds_base = (unsigned)(uintptr_t)LDS_PTR(smem);
  int lane = threadIdx.x & 63;
  const unsigned lane_rd = (unsigned)((lane & 15) * 64 + (((lane >> 4) ^ swz_kc(lane & 15)) << 4));
  const unsigned laneA = (unsigned)((lane >> 2) * (int)g.lda * 2 + (((lane & 3) ^ swz_kc(lane >> 2)) << 4));
  const unsigned laneB = (unsigned)((lane >> 2) * (int)g.ldb * 2 + (((lane & 3) ^ swz_kc(lane >> 2)) << 4));
  const int64_t pieceA = (int64_t)32 * g.lda - 1024, pieceB = (int64_t)32 * g.ldb - 1024;
  const char *bA[4], *bB[4];   // scalar: piece i of this wave in the item's A / B panel, less the immediate i * 1024
  int issued = 0;
  // one request: half-tile `half` of a pair, piece i; M0 and the lane offset (lane part + k offset) are set by the caller
  auto dma = [&](auto I, auto HALF, bool opB, unsigned voff) {
    constexpr int i = decltype(I)::value, half = decltype(HALF)::value;
    if constexpr (ABL & 64) {
      const unsigned v2 = voff + half * 64;
      asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(v2), "s"(opB ? bB[i] : bA[i]), "n"(i * 1024) : "memory");
    } else
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(opB ? bB[i] : bA[i]), "n"(i * 1024 + half * 64) : "memory");
    issued += 1;
  };
  // (an LDS-DMA right behind the write of M0 needs one wait state)
  auto set_m0 = [&](unsigned v) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(v) : "memory"); };
  auto set_m0_early = [&](unsigned v) { asm volatile("s_mov_b32 m0, %0" ::"s"(v) : "memory"); };
  // M0 of half-tile h (first of a pair) / h + 1 (second: 64 lower, the immediate adds it back)
  auto m0_of = [&](int h, bool opB, int half) { return lds_base + ((h + half) % NSLOT) * HT + (opB ? HT_A : 0) + wave * 4096 - ((ABL & 64) ? 0 : half * 64); };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, 2> I2;
  typedef std::integral_constant<int, 3> I3;
  int mark01 = 0, mark23 = 0;
  auto prologue = [&](TileId t) {
    const char *tA = (const char *)(g.A + ((int64_t)t.tm * BM + wave * 64) * g.lda);
    const char *tB = (const char *)(g.B + ((int64_t)t.tn * BN + wave * 64) * g.ldb);
#pragma unroll
    for (int i = 0; i < 4; ++i) { bA[i] = tA + i * pieceA; bB[i] = tB + i * pieceB; }
    auto pair = [&](int h) {
      const unsigned va = laneA + h * 64, vb = laneB + h * 64;
      const unsigned a0 = m0_of(h, false, 0), a1 = m0_of(h, false, 1), b0 = m0_of(h, true, 0), b1 = m0_of(h, true, 1);
      set_m0(a0); dma(I0{}, I0{}, false, va); set_m0(a1); dma(I0{}, I1{}, false, va);
      set_m0(a0); dma(I1{}, I0{}, false, va); set_m0(a1); dma(I1{}, I1{}, false, va);
      set_m0(a0); dma(I2{}, I0{}, false, va); set_m0(a1); dma(I2{}, I1{}, false, va);
      set_m0(a0); dma(I3{}, I0{}, false, va); set_m0(a1); dma(I3{}, I1{}, false, va);
      set_m0(b0); dma(I0{}, I0{}, true, vb); set_m0(b1); dma(I0{}, I1{}, true, vb);
      set_m0(b0); dma(I1{}, I0{}, true, vb); set_m0(b1); dma(I1{}, I1{}, true, vb);
      set_m0(b0); dma(I2{}, I0{}, true, vb); set_m0(b1); dma(I2{}, I1{}, true, vb);
      set_m0(b0); dma(I3{}, I0{}, true, vb); set_m0(b1); dma(I3{}, I1{}, true, vb);
    };
    pair(0);
    mark01 = issued;
    pair(2);
    mark23 = issued;
  };
  auto wait_mark = [&](int mark) {
    int young = issued - mark;
    if (young > 63) young = 63;
    wait_vm_sw<0, 63>(young);
  };
  auto frag = [&](unsigned va, int off) {
    return *reinterpret_cast<const __attribute__((address_space(3))) bf8_t *>((uintptr_t)(va + off));
  };

  TileId tid = decode_item(w, total, g.tiles_m, g.tiles_n, g.band_m);
  if constexpr (ABL & 16) {
    // start stagger: workgroup b waits (b / 8 % 32) x ~0.3 us, so that the workgroups' store phases do not coincide
    const int n = ((blockIdx.x >> 3) & 31) * 8;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
  }
  prologue(tid);

  while (true) {
    const int64_t m0 = (int64_t)tid.tm * BM, n0 = (int64_t)tid.tn * BN;
    f4_t acc[8][8];
    bf8_t fa[2][8], fb[2][8];

    wait_mark(mark01);
    __builtin_amdgcn_s_waitcnt(0xC07F);
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
    int mneed = mark23;

    auto step = [&](auto PAR, auto FIRST, auto ISSUE, int p) {
      constexpr int cur = decltype(PAR)::value, nxt = cur ^ 1;
      constexpr bool first = decltype(FIRST)::value;
      constexpr bool do_issue = decltype(ISSUE)::value && !(ABL & 1);
      if (cur == 1) wait_mark(mneed);
      // even step p: A pieces of pair (p+4, p+5); odd step p: B pieces of pair (p+3, p+4)
      const int hpair = cur == 0 ? p + 4 : p + 3;
      unsigned voff = (cur == 1 ? laneB : laneA) + hpair * 64;
      unsigned mA = m0_of(hpair, cur == 1, 0), mB = m0_of(hpair, cur == 1, 1);
      unsigned va = lane_rd + lds_base + ((p + 1) % NSLOT) * HT;
      asm volatile("" : "+v"(va), "+v"(voff), "+s"(mA), "+s"(mB));
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's reads of half-tile p are complete
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      auto filler = [&](auto S) {
        constexpr int s_ = decltype(S)::value;
        // 32 gaps (one behind every second MFMA): a fragment read in every even gap; the wave's j-th request of
        // the step (piece j / 2, half j % 2) in gap 4 j + WR, its M0 one gap earlier
        if constexpr (s_ % 2 == 0 && !(ABL & 2)) {
          constexpr int r = s_ / 2;
          if constexpr (r % 2 == 0) fa[nxt][r / 2] = frag(va, wm * 8192 + (r / 2) * 1024);
          else fb[nxt][r / 2] = frag(va, HT_A + wn * 8192 + (r / 2) * 1024);
        }
        if constexpr (do_issue) {
          constexpr int sn = s_ + 1;   // the gap M0 is set for
          if constexpr (!(ABL & 128)) {
            if constexpr (sn < 32 && (sn & 3) == WR) set_m0_early(((sn / 4) & 1) ? mB : mA);
            if constexpr (s_ == 0 && WR == 0) set_m0(mA);
          } else {
            if constexpr ((s_ & 3) == WR) set_m0(((s_ / 4) & 1) ? mB : mA);
          }
          if constexpr ((s_ & 3) == WR)
            dma(std::integral_constant<int, (s_ / 4) / 2>{}, std::integral_constant<int, (s_ / 4) & 1>{}, cur == 1, voff);
        }
      };
      auto run = [&](auto self, auto K) {
        constexpr int k = decltype(K)::value;
        if constexpr (k < 64) {
          constexpr int gi = k / 8, ni = k % 8;
          if constexpr (first) mfma_a0(acc[ni][gi], fb[cur][ni], fa[cur][gi]);
          else mfma_a(acc[ni][gi], fb[cur][ni], fa[cur][gi]);
          if constexpr (k % 2 == 1) {
            __builtin_amdgcn_sched_barrier(0);
            filler(std::integral_constant<int, k / 2>{});
            __builtin_amdgcn_sched_barrier(0);
          }
          self(self, std::integral_constant<int, k + 1>{});
        }
      };
      run(run, std::integral_constant<int, 0>{});
      if constexpr (ABL & 2) {
        if (p == 0) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { fa[nxt][i] = fa[cur][i]; fb[nxt][i] = fb[cur][i]; }
        }
      }
      if (cur == 1 && do_issue) mneed = issued;
    };
    typedef std::true_type T_;
    typedef std::false_type F_;
    // steps 0 .. np-5 send (pairs up to (np-2, np-1)); the last four send nothing
    step(I0{}, T_{}, T_{}, 0);
    step(I1{}, F_{}, T_{}, 1);
    int p = 2;
    for (; p < np - 4; p += 2) {
      step(I0{}, F_{}, T_{}, p);
      step(I1{}, F_{}, T_{}, p + 1);
    }
    for (; p < np; p += 2) {
      step(I0{}, F_{}, F_{}, p);
      step(I1{}, F_{}, F_{}, p + 1);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (the asm MFMAs' results: wait states the compiler does not count)

    w += gridDim.x;
    const bool more = w < total;
    if (more) {
      tid = decode_item(w, total, g.tiles_m, g.tiles_n, g.band_m);
      prologue(tid);
    }
    if constexpr (EPI == 2 || EPI == 3) {
      int lane_e = threadIdx.x & 63;
      asm volatile("" : "+v"(lane_e));
      const int q = lane_e >> 4;
      bf16_t *crow = g.C + (m0 + wm * 128 + (lane_e & 15)) * g.ldc + n0 + wn * 128 + (q & 1) * 16 + (q >> 1) * 8;
      if constexpr (ABL & 8)   // (ablation: every item of a workgroup stores to the same 128 KiB)
        crow = g.C + ((int64_t)blockIdx.x * 256 + wm * 128 + (lane_e & 15)) * g.ldc + wn * 128 + (q & 1) * 16 + (q >> 1) * 8;
      const int64_t rstep = 16 * g.ldc;
      auto rd = [&](const f4_t &x) {
        f4_t r;
        asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                     : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]) : "a"(x[0]), "a"(x[1]), "a"(x[2]), "a"(x[3]));
        return r;
      };
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f4_t a = EPI == 3 ? acc[2 * t][mi] : rd(acc[2 * t][mi]), b = EPI == 3 ? acc[2 * t + 1][mi] : rd(acc[2 * t + 1][mi]);
          const unsigned a0 = pack_bf16x2(a[0], a[1]), a1 = pack_bf16x2(a[2], a[3]);
          const unsigned b0 = pack_bf16x2(b[0], b[1]), b1 = pack_bf16x2(b[2], b[3]);
          const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
          const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
          u4v_t pk;
          pk[0] = r0[0]; pk[1] = r1[0]; pk[2] = r0[1]; pk[3] = r1[1];
          if constexpr (ABL & 256) {   // (ablation: everything but the store)
            asm volatile("" ::"v"(pk), "v"(crow));
          } else {
            if constexpr (ABL & 512) *reinterpret_cast<u4v_t *>(crow + t * 32) = pk;
            else __builtin_nontemporal_store(pk, reinterpret_cast<u4v_t *>(crow + t * 32));
            issued += 1;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        crow += rstep;
      }
    } else {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3]; if ((j & 1) && (i & 1)) __builtin_amdgcn_sched_barrier(0); }
      if (s == 12345.678f) g.C[0] = 1;
    }
    if (!more) break;
  }
}
template <int EPI, int ABL>
__global__ __launch_bounds__(256, 1) void g4w4_k(PArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave == 0) g4w4_body<EPI, ABL, 0>(g, smem);
  else if (wave == 1) g4w4_body<EPI, ABL, 1>(g, smem);
  else if (wave == 2) g4w4_body<EPI, ABL, 2>(g, smem);
  else g4w4_body<EPI, ABL, 3>(g, smem);
}


// Fifth structure: the fourth, with the item's stores spread over the NEXT item's first steps.  A CU's store
// path takes about 64 cycles per 1-KiB store instruction whatever the destination, i.e. ~8k cycles for a
// 256 x 256 bf16 tile; issued back to back after the main loop that time is exposed.  Here the accumulators
// are packed to bf16 and exchanged into 16-byte runs right behind the main loop (32 quads = 128 registers
// per lane; the accumulators live in the accumulation registers, so the vector registers have room), the
// first NOW of them are stored while the rest is being packed, and the others go out as fillers of the next
// item's first steps (SPS per step).  Stores are buffer stores: the first item's descriptor has zero
// records, so its (empty) deferred stores are dropped by the range check and no step needs a branch.
typedef int i4s_t __attribute__((ext_vector_type(4)));
template <int SPS, int ABL, int WR>
__device__ __forceinline__ void g4w5_body(const PArgs &g, char *smem) {
  constexpr int wave = WR, wm = WR >> 1, wn = WR & 1;
  constexpr int NOW = 8;                      // stores issued during packing
  constexpr int NSS = (32 - NOW) / SPS;       // steps that carry deferred stores
  static_assert((32 - NOW) % SPS == 0 && NSS % 2 == 0, "whole pairs of store steps");
  const int total = g.tiles_m * g.tiles_n;
  const int np = (int)(g.K / KH);
  int w = blockIdx.x;
  if (w >= total) return;
  const unsigned lds_base = (unsigned)(uintptr_t)LDS_PTR(smem);
  int lane = threadIdx.x & 63;
  const unsigned lane_rd = (unsigned)((lane & 15) * 64 + (((lane >> 4) ^ swz_kc(lane & 15)) << 4));
  const unsigned laneA = (unsigned)((lane >> 2) * (int)g.lda * 2 + (((lane & 3) ^ swz_kc(lane >> 2)) << 4));
  const unsigned laneB = (unsigned)((lane >> 2) * (int)g.ldb * 2 + (((lane & 3) ^ swz_kc(lane >> 2)) << 4));
  // store: lane part of the byte offset inside the wave's 128 x 128 block (row lane & 15 of a 16-row block,
  // 16-byte run (q & 1) * 32 + (q >> 1) * 16 of a 64-byte group, q = lane >> 4)
  const unsigned lane_st = (unsigned)((lane & 15) * (int)g.ldc * 2 + ((lane >> 4) & 1) * 32 + (lane >> 5) * 16);
  const unsigned ldc16 = (unsigned)(g.ldc * 32);   // bytes per 16 rows of C
  const int64_t pieceA = (int64_t)32 * g.lda - 1024, pieceB = (int64_t)32 * g.ldb - 1024;
  const char *bA[4], *bB[4];
  int issued = 0;
  auto dma = [&](auto I, bool opB, unsigned voff) {
    constexpr int i = decltype(I)::value;
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(opB ? bB[i] : bA[i]), "n"(i * 1024) : "memory");
    issued += 1;
  };
  auto set_m0 = [&](unsigned v) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(v) : "memory"); };
  auto set_m0_early = [&](unsigned v) { asm volatile("s_mov_b32 m0, %0" ::"s"(v) : "memory"); };
  auto m0_of = [&](int h, bool opB) { return lds_base + (h % NSLOT) * HT + (opB ? HT_A : 0) + wave * 4096; };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, 2> I2;
  typedef std::integral_constant<int, 3> I3;
  int mark01 = 0, mark23 = 0;
  auto prologue = [&](TileId t) {
    const char *tA = (const char *)(g.A + ((int64_t)t.tm * BM + wave * 64) * g.lda);
    const char *tB = (const char *)(g.B + ((int64_t)t.tn * BN + wave * 64) * g.ldb);
#pragma unroll
    for (int i = 0; i < 4; ++i) { bA[i] = tA + i * pieceA; bB[i] = tB + i * pieceB; }
    auto pair = [&](int h) {
      const unsigned va = laneA + h * 64, vb = laneB + h * 64;
      const unsigned a0 = m0_of(h, false), a1 = m0_of(h + 1, false), b0 = m0_of(h, true), b1 = m0_of(h + 1, true);
      set_m0(a0); dma(I0{}, false, va); set_m0(a1); dma(I0{}, false, va + 64);
      set_m0(a0); dma(I1{}, false, va); set_m0(a1); dma(I1{}, false, va + 64);
      set_m0(a0); dma(I2{}, false, va); set_m0(a1); dma(I2{}, false, va + 64);
      set_m0(a0); dma(I3{}, false, va); set_m0(a1); dma(I3{}, false, va + 64);
      set_m0(b0); dma(I0{}, true, vb); set_m0(b1); dma(I0{}, true, vb + 64);
      set_m0(b0); dma(I1{}, true, vb); set_m0(b1); dma(I1{}, true, vb + 64);
      set_m0(b0); dma(I2{}, true, vb); set_m0(b1); dma(I2{}, true, vb + 64);
      set_m0(b0); dma(I3{}, true, vb); set_m0(b1); dma(I3{}, true, vb + 64);
    };
    pair(0);
    mark01 = issued;
    pair(2);
    mark23 = issued;
  };
  auto wait_mark = [&](int mark) {
    int young = issued - mark;
    if (young > 63) young = 63;
    wait_vm_sw<0, 63>(young);
  };
  auto frag = [&](unsigned va, int off) {
    return *reinterpret_cast<const __attribute__((address_space(3))) bf8_t *>((uintptr_t)(va + off));
  };
  // descriptor of the wave's block of C in the PREVIOUS item (zero records before the first item)
  i4s_t srd_prev = {0, 0, 0, 0x00020000};
  u4v_t pk[32];
  auto store_q = [&](const u4v_t &q, const i4s_t &srd, unsigned soff, auto T) {
    constexpr int t = decltype(T)::value;
    if constexpr (ABL & 1024)
      asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:%4 nt\n\ts_nop 0" ::"v"(q), "v"(lane_st), "s"(srd), "s"(soff), "n"(t * 64) : "memory");
    else
      asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:%4\n\ts_nop 0" ::"v"(q), "v"(lane_st), "s"(srd), "s"(soff), "n"(t * 64) : "memory");
    issued += 1;
  };

  TileId tid = decode_item(w, total, g.tiles_m, g.tiles_n, g.band_m);
  prologue(tid);
  bool first_item = true;

  while (true) {
    const int64_t m0 = (int64_t)tid.tm * BM, n0 = (int64_t)tid.tn * BN;
    f4_t acc[8][8];
    bf8_t fa[2][8], fb[2][8];

    // half-tiles 0, 1 landed: younger than their requests are the pair (2, 3) and, except before the first
    // item, the NOW stores of the packing phase
    if (first_item) wait_vm<16>(); else wait_vm<16 + NOW>();
    first_item = false;
    __builtin_amdgcn_s_waitcnt(0xC07F);
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
    int mneed = mark23;

    // SQ: index of the first quad this step stores (-1: none)
    auto step = [&](auto PAR, auto FIRST, auto ISSUE, auto SQ, int p) {
      constexpr int cur = decltype(PAR)::value, nxt = cur ^ 1;
      constexpr bool first = decltype(FIRST)::value;
      constexpr bool do_issue = decltype(ISSUE)::value && !(ABL & 1);
      constexpr int sq = decltype(SQ)::value;
      // An odd step needs the pair (p+1, p+2) complete.  Its last request went out in step p-2 (or before the
      // item); younger are at least the eight requests of step p-1 and that step's deferred stores (counting only
      // those is the stricter wait); the last four steps have sent nothing.
      if constexpr (cur == 1) {
        if constexpr (!decltype(ISSUE)::value) wait_vm<0>();
        else wait_vm<8 + (sq >= 0 ? SPS : 0)>();
      }
      const int hpair = cur == 0 ? p + 4 : p + 3;
      constexpr bool opb = (ABL & 2048) ? cur == 0 : cur == 1;
      unsigned voff = (opb ? laneB : laneA) + hpair * 64;
      unsigned voff2 = voff + 64;
      unsigned mA = m0_of(hpair, opb), mB = m0_of(hpair + 1, opb);
      unsigned va = lane_rd + lds_base + ((p + 1) % NSLOT) * HT;
      unsigned soff = (unsigned)(sq >= 0 ? sq / 4 : 0) * ldc16;
      asm volatile("" : "+v"(va), "+v"(voff), "+v"(voff2), "+s"(mA), "+s"(mB), "+s"(soff));
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // Gap s (behind MFMA 2s+1), period of eight gaps, g = s % 8, j = s / 8, at most one instruction per gap:
      //   LDS-DMA of piece j: first half-tile in gap g = 2 WR, second in g = 2 WR + 1 (M0 between the gap's two MFMAs);
      //   deferred store u = j in gap g = (2 WR + 5) % 8;  fragment reads in the first four of the other five gaps.
      // Over the four waves every gap carries one LDS-DMA and every second gap one store.
      auto mid = [&](auto S) {
        constexpr int s_ = decltype(S)::value, gq = s_ % 8;
        if constexpr (do_issue && (gq == 2 * WR || gq == 2 * WR + 1)) set_m0_early(gq == 2 * WR ? mA : mB);
      };
      auto filler = [&](auto S) {
        constexpr int s_ = decltype(S)::value, gq = s_ % 8, j = s_ / 8;
        constexpr int gst = (2 * WR + 5) % 8;
        if constexpr (gq == 2 * WR || gq == 2 * WR + 1) {
          if constexpr (do_issue) dma(std::integral_constant<int, j>{}, opb, gq == 2 * WR ? voff : voff2);
        } else if constexpr (gq == gst) {
          if constexpr (sq >= 0 && j < SPS) store_q(pk[sq + j], srd_prev, SPS == 4 ? soff : (unsigned)((sq + j) / 4) * ldc16, std::integral_constant<int, (sq + j) % 4>{});
        } else if constexpr (!(ABL & 2)) {
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
          if constexpr (first) mfma_a0(acc[ni][gi], fb[cur][ni], fa[cur][gi]);
          else mfma_a(acc[ni][gi], fb[cur][ni], fa[cur][gi]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (k % 2 == 0) mid(std::integral_constant<int, k / 2>{});
          else filler(std::integral_constant<int, k / 2>{});
          __builtin_amdgcn_sched_barrier(0);
          self(self, std::integral_constant<int, k + 1>{});
        }
      };
      run(run, std::integral_constant<int, 0>{});
      if (cur == 1 && do_issue) mneed = issued;
    };
    typedef std::true_type T_;
    typedef std::false_type F_;
    typedef std::integral_constant<int, -1> NS;
    // steps 0 .. NSS-1 carry the previous item's deferred stores
    auto store_steps = [&](auto self, auto P) {
      constexpr int p = decltype(P)::value;
      if constexpr (p < NSS) {
        if constexpr (p == 0) step(I0{}, T_{}, T_{}, std::integral_constant<int, NOW>{}, 0);
        else step(std::integral_constant<int, p & 1>{}, F_{}, T_{}, std::integral_constant<int, NOW + SPS * p>{}, p);
        self(self, std::integral_constant<int, p + 1>{});
      }
    };
    store_steps(store_steps, I0{});
    int p = NSS;
    for (; p < np - 4; p += 2) {
      step(I0{}, F_{}, T_{}, NS{}, p);
      step(I1{}, F_{}, T_{}, NS{}, p + 1);
    }
    for (; p < np; p += 2) {
      step(I0{}, F_{}, F_{}, NS{}, p);
      step(I1{}, F_{}, F_{}, NS{}, p + 1);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    w += gridDim.x;
    const bool more = w < total;
    if (more) {
      tid = decode_item(w, total, g.tiles_m, g.tiles_n, g.band_m);
      prologue(tid);
    }
    // pack: quad j = (16-row block j / 4, 64-byte group j % 4)
    {
      const char *cb = (const char *)(g.C + (m0 + wm * 128) * g.ldc + n0 + wn * 128);
      i4s_t srd;
      srd[0] = (int)(uintptr_t)cb;
      srd[1] = (int)(((uintptr_t)cb >> 32) & 0xffff);
      srd[2] = -1;
      srd[3] = 0x00020000;
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
          const f4_t a = rd(acc[2 * t][mi]), b = rd(acc[2 * t + 1][mi]);
          const unsigned a0 = pack_bf16x2(a[0], a[1]), a1 = pack_bf16x2(a[2], a[3]);
          const unsigned b0 = pack_bf16x2(b[0], b[1]), b1 = pack_bf16x2(b[2], b[3]);
          const auto r0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
          const auto r1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
          pk[j][0] = r0[0]; pk[j][1] = r1[0]; pk[j][2] = r0[1]; pk[j][3] = r1[1];
          // the first NOW quads leave while the others are packed: one store behind every fourth packed quad
          if constexpr (j % 4 == 3 && j / 4 < NOW) store_q(pk[j / 4], srd, (unsigned)((j / 4) / 4) * ldc16, std::integral_constant<int, (j / 4) % 4>{});
          __builtin_amdgcn_sched_barrier(0);
          self(self, std::integral_constant<int, j + 1>{});
        }
      };
      packq(packq, I0{});
      srd_prev = srd;
      if (!more) {
        // last item: the rest at once
        auto flush = [&](auto self, auto J) {
          constexpr int j = decltype(J)::value;
          if constexpr (j < 32) {
            store_q(pk[j], srd, (unsigned)(j / 4) * ldc16, std::integral_constant<int, j % 4>{});
            self(self, std::integral_constant<int, j + 1>{});
          }
        };
        flush(flush, std::integral_constant<int, NOW>{});
      }
    }
    if (!more) break;
  }
}
template <int SPS, int ABL>
__global__ __launch_bounds__(256, 1) void g4w5_k(PArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave == 0) g4w5_body<SPS, ABL, 0>(g, smem);
  else if (wave == 1) g4w5_body<SPS, ABL, 1>(g, smem);
  else if (wave == 2) g4w5_body<SPS, ABL, 2>(g, smem);
  else g4w5_body<SPS, ABL, 3>(g, smem);
}

__global__ void ref_k(const bf16_t *A, const bf16_t *B, float *C, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                      const int *rows, int nrows) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)nrows * N) return;
  const int64_t m = rows[idx / N], n = idx % N;
  float s = 0.f;
  for (int64_t k = 0; k < K; ++k)
    s += __uint_as_float((unsigned)A[m * lda + k] << 16) * __uint_as_float((unsigned)B[n * ldb + k] << 16);
  C[idx] = s;
}

static bf16_t f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u = (u + 0x7fff + ((u >> 16) & 1)) >> 16;
  return (bf16_t)u;
}
static float bf2f(bf16_t b) {
  unsigned u = (unsigned)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

typedef int (*ldm_gemm_t)(int, int, int, int, int64_t, int64_t, int64_t, float, const void *, int64_t, const void *,
                          int64_t, void *, int64_t, const float *, const void *, int64_t, void *);

template <typename F>
static double time_ms(F f, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

template <int PAIR, int RPG, int NOEPI, int ABL = 0>
static void launch(const PArgs &g) {
  static bool attr = false;
  if (!attr) {
    CK(hipFuncSetAttribute((const void *)g4w_k<PAIR, RPG, NOEPI, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr = true;
  }
  const int64_t items = (int64_t)g.tiles_m * g.tiles_n;
  const int nwg = items < 256 ? (int)items : 256;
  hipLaunchKernelGGL((g4w_k<PAIR, RPG, NOEPI, ABL>), dim3(nwg), dim3(256), LDS_BYTES, 0, g);
}

template <int FPM, int NOEPI, int ABL = 0>
static void launch2(const PArgs &g) {
  static bool attr = false;
  if (!attr) {
    CK(hipFuncSetAttribute((const void *)g4w2_k<FPM, NOEPI, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr = true;
  }
  const int64_t items = (int64_t)g.tiles_m * g.tiles_n;
  const int nwg = items < 256 ? (int)items : 256;
  hipLaunchKernelGGL((g4w2_k<FPM, NOEPI, ABL>), dim3(nwg), dim3(256), LDS_BYTES, 0, g);
}

template <int EPI, int ABL = 0>
static void launch3(const PArgs &g) {
  static bool attr = false;
  if (!attr) {
    CK(hipFuncSetAttribute((const void *)g4w3_k<EPI, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr = true;
  }
  const int64_t items = (int64_t)g.tiles_m * g.tiles_n;
  const int nwg = items < 256 ? (int)items : 256;
  hipLaunchKernelGGL((g4w3_k<EPI, ABL>), dim3(nwg), dim3(256), LDS_BYTES, 0, g);
}

template <int EPI, int ABL = 0>
static void launch4(const PArgs &g) {
  static bool attr = false;
  if (!attr) {
    CK(hipFuncSetAttribute((const void *)g4w4_k<EPI, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr = true;
  }
  const int64_t items = (int64_t)g.tiles_m * g.tiles_n;
  const int nwg = items < 256 ? (int)items : 256;
  hipLaunchKernelGGL((g4w4_k<EPI, ABL>), dim3(nwg), dim3(256), LDS_BYTES, 0, g);
}

template <int SPS, int ABL = 0>
static void launch5(const PArgs &g) {
  static bool attr = false;
  if (!attr) {
    CK(hipFuncSetAttribute((const void *)g4w5_k<SPS, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr = true;
  }
  const int64_t items = (int64_t)g.tiles_m * g.tiles_n;
  const int nwg = items < 256 ? (int)items : 256;
  hipLaunchKernelGGL((g4w5_k<SPS, ABL>), dim3(nwg), dim3(256), LDS_BYTES, 0, g);
}

int main(int argc, char **argv) {
  std::vector<int64_t> shapes;
  for (int i = 1; i + 2 < argc; i += 3) { shapes.push_back(atoll(argv[i])); shapes.push_back(atoll(argv[i + 1])); shapes.push_back(atoll(argv[i + 2])); }
  if (shapes.empty()) shapes = {65536, 2304, 768, 65536, 768, 768, 65536, 2048, 768, 65536, 768, 1024, 65536, 2304, 1536, 8192, 8192, 8192};
  void *lib = dlopen("language-diffusion-model_amd/lib/libldm_hip.so", RTLD_NOW);
  ldm_gemm_t ldm_gemm = lib ? (ldm_gemm_t)dlsym(lib, "ldm_gemm") : nullptr;
  if (!ldm_gemm) printf("(library kernel not available: %s)\n", dlerror());
  const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 20;
  for (size_t si = 0; si + 2 < shapes.size(); si += 3) {
    const int64_t M = shapes[si], N = shapes[si + 1], K = shapes[si + 2];
    std::vector<bf16_t> hA((size_t)M * K), hB((size_t)N * K);
    unsigned s = 12345u + (unsigned)si;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto &v : hA) v = f2bf(rnd());
    for (auto &v : hB) v = f2bf(rnd());
    bf16_t *dA, *dB, *dC, *dC2;
    CK(hipMalloc(&dA, hA.size() * 2));
    CK(hipMalloc(&dB, hB.size() * 2));
    CK(hipMalloc(&dC, (size_t)M * N * 2));
    CK(hipMalloc(&dC2, (size_t)M * N * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xff, (size_t)M * N * 2));
    PArgs g;
    g.M = M; g.N = N; g.K = K; g.A = dA; g.lda = K; g.B = dB; g.ldb = K; g.C = dC; g.ldc = N;
    g.tiles_m = (int)(M / 256); g.tiles_n = (int)(N / 256);
    int b = 16;
    while (b > 2 && (int64_t)b * 512 * K > (int64_t)8 << 20) b >>= 1;
    g.band_m = getenv("BAND") ? atoi(getenv("BAND")) : b;
    const double flops = 2.0 * M * N * K;
    printf("M=%lld N=%lld K=%lld band=%d\n", (long long)M, (long long)N, (long long)K, g.band_m);
    // reference rows
    const int nrows = 96;
    std::vector<int> hrows(nrows);
    for (int i = 0; i < nrows; ++i) hrows[i] = (int)(((int64_t)i * 7919 * 131 + i) % M);
    int *drows; float *dref;
    CK(hipMalloc(&drows, nrows * 4));
    CK(hipMalloc(&dref, (size_t)nrows * N * 4));
    CK(hipMemcpy(drows, hrows.data(), nrows * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_k, dim3((unsigned)(((int64_t)nrows * N + 255) / 256)), dim3(256), 0, 0, dA, dB, dref, N, K, K, K, drows, nrows);
    CK(hipDeviceSynchronize());
    std::vector<float> href((size_t)nrows * N);
    CK(hipMemcpy(href.data(), dref, href.size() * 4, hipMemcpyDeviceToHost));
    auto check = [&](bf16_t *dev, const char *name) {
      std::vector<bf16_t> row(N);
      double worst = 0;
      for (int i = 0; i < nrows; ++i) {
        CK(hipMemcpy(row.data(), dev + (size_t)hrows[i] * N, N * 2, hipMemcpyDeviceToHost));
        for (int64_t n = 0; n < N; ++n) {
          const double r = href[(size_t)i * N + n], v = bf2f(row[n]);
          const double e = fabs(v - r) / (fabs(r) + 1.0);
          if (!(e <= worst)) worst = e;   // NaN propagates
        }
      }
      printf("  %-28s max rel err %.3e %s\n", name, worst, worst < 1.5e-2 ? "ok" : "WRONG");
    };
    auto run = [&](const char *name, auto fn, bool chk) {
      CK(hipMemset(dC, 0xff, (size_t)M * N * 2));
      fn();
      CK(hipDeviceSynchronize());
      if (chk) check(dC, name);
      const double ms = time_ms(fn, reps);
      printf("  %-28s %8.1f us  %7.1f TFLOP/s\n", name, ms * 1e3, flops / ms * 1e-9);
    };
    for (int round = 0; round < 2; ++round) {
      if (ldm_gemm)
        run("library 8-wave", [&]() { int rc = ldm_gemm(1, 1, 0, 1, M, N, K, 1.0f, dA, K, dB, K, dC, N, nullptr, nullptr, 0, nullptr); if (rc) { printf("ldm_gemm rc %d\n", rc); exit(1); } }, round == 0);
      run("v5 deferred plain stores, 4/step", [&]() { launch5<4>(g); }, round == 0);
      run("v5, A pieces 2 steps ahead, B 3", [&]() { launch5<4, 2048>(g); }, round == 0);
      run("v5, no DMA in the loop", [&]() { launch5<4, 1>(g); }, false);
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dC2)); CK(hipFree(drows)); CK(hipFree(dref));
  }
  return 0;
}
