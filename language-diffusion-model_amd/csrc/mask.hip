// K1: forward masking in one coalesced pass (HBM-bound: 17 B/token, 21 with an
// injected draw).  Follows RandomMaskingStrategy.apply_random_mask
// (src/masking_strategy.py:104-114) and the t draw of src/model.py:59-68.
#include "common.h"
#include "philox.h"

namespace {

struct MaskArgs {
  const int64_t *ids;
  const float *draw;
  int t_mode;
  float t_scalar;
  float *t_seq;
  float eps;
  int64_t n, seq;
  int64_t bos, eos, pad, mask_id;
  unsigned k0, k1, o0, o1;
  int64_t *masked;
  uint8_t *mask;
};

// stream 1 of the Philox counter space holds the per-sequence t draws
__device__ __forceinline__ float seq_t(const MaskArgs &a, int64_t b) {
  if (a.t_mode == 0) return a.t_scalar;
  if (a.t_mode == 1) return a.t_seq[b];
  U4 r = philox4x32_10(U4{(unsigned)b, (unsigned)((unsigned long long)b >> 32), a.o0 ^ 0x80000000u, a.o1}, a.k0, a.k1);
  return fmaf(u01(r.x), 1.0f - a.eps, a.eps);
}

__device__ __forceinline__ void mask_one(const MaskArgs &a, int64_t id, float u, float t,
                                         int64_t &out, uint8_t &m) {
  bool elig = (id != a.bos) & (id != a.eos) & (id != a.pad);
  bool mk = (u < t) & elig;
  out = mk ? a.mask_id : id;
  m = mk ? 1 : 0;
}

// 4 tokens per thread: 2 x 16-B id loads, one 16-B draw load (or one Philox call),
// 2 x 16-B id stores, one 4-B mask store.  Group g covers tokens 4g..4g+3 and uses
// Philox counter (g, offset) so the draw of token i is word i&3 of block i>>2.
__global__ __launch_bounds__(256) void mask_fwd_vec4(MaskArgs a) {
  const int64_t ngroups = a.n >> 2;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups;
       g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = g << 2;
    const longlong2 v0 = *reinterpret_cast<const longlong2 *>(a.ids + i0);
    const longlong2 v1 = *reinterpret_cast<const longlong2 *>(a.ids + i0 + 2);
    float u[4];
    if (a.draw) {
      float4 d = *reinterpret_cast<const float4 *>(a.draw + i0);
      u[0] = d.x; u[1] = d.y; u[2] = d.z; u[3] = d.w;
    } else {
      U4 r = philox4x32_10(U4{(unsigned)g, (unsigned)((unsigned long long)g >> 32), a.o0, a.o1}, a.k0, a.k1);
      u[0] = u01(r.x); u[1] = u01(r.y); u[2] = u01(r.z); u[3] = u01(r.w);
    }
    int64_t id[4] = {v0.x, v0.y, v1.x, v1.y};
    int64_t o[4];
    uint8_t m[4];
    const int64_t b0 = i0 / a.seq;
    const int64_t rem = i0 - b0 * a.seq;
    float tb = seq_t(a, b0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = tb;
      if (rem + j >= a.seq) t = seq_t(a, (i0 + j) / a.seq);   // group straddles sequences
      mask_one(a, id[j], u[j], t, o[j], m[j]);
    }
    longlong2 w0, w1;
    w0.x = o[0]; w0.y = o[1]; w1.x = o[2]; w1.y = o[3];
    *reinterpret_cast<longlong2 *>(a.masked + i0) = w0;
    *reinterpret_cast<longlong2 *>(a.masked + i0 + 2) = w1;
    uchar4 mm;
    mm.x = m[0]; mm.y = m[1]; mm.z = m[2]; mm.w = m[3];
    *reinterpret_cast<uchar4 *>(a.mask + i0) = mm;
  }
}

// scalar form: tails and misaligned views
__global__ __launch_bounds__(256) void mask_fwd_scalar(MaskArgs a, int64_t first) {
  for (int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float u;
    if (a.draw) {
      u = a.draw[i];
    } else {
      const int64_t g = i >> 2;
      U4 r = philox4x32_10(U4{(unsigned)g, (unsigned)((unsigned long long)g >> 32), a.o0, a.o1}, a.k0, a.k1);
      const unsigned w[4] = {r.x, r.y, r.z, r.w};
      u = u01(w[i & 3]);
    }
    int64_t o;
    uint8_t m;
    mask_one(a, a.ids[i], u, seq_t(a, i / a.seq), o, m);
    a.masked[i] = o;
    a.mask[i] = m;
  }
}

// t_mode 2: publish the per-sequence t that the masking kernel used
__global__ void mask_write_t(MaskArgs a, int64_t batch) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < batch) {
    U4 r = philox4x32_10(U4{(unsigned)b, (unsigned)((unsigned long long)b >> 32), a.o0 ^ 0x80000000u, a.o1}, a.k0, a.k1);
    a.t_seq[b] = fmaf(u01(r.x), 1.0f - a.eps, a.eps);
  }
}

}  // namespace

extern "C" int ldm_mask_fwd(const int64_t *ids, const float *draw, int t_mode, float t_scalar,
                            float *t_seq, float eps, int64_t batch, int64_t seq, int64_t bos_id,
                            int64_t eos_id, int64_t pad_id, int64_t mask_id, uint64_t seed,
                            uint64_t offset, int64_t *masked_out, uint8_t *mask_out, void *stream) {
  if (batch < 0 || seq < 0) return LDM_E_BADARG;
  const int64_t n = batch * seq;
  if (n == 0) return LDM_OK;
  if (!ids || !masked_out || !mask_out) return LDM_E_BADARG;
  if (t_mode < 0 || t_mode > 2) return LDM_E_BADARG;
  if (t_mode != 0 && !t_seq) return LDM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  MaskArgs a;
  a.ids = ids; a.draw = draw; a.t_mode = t_mode; a.t_scalar = t_scalar; a.t_seq = t_seq;
  a.eps = eps; a.n = n; a.seq = seq; a.bos = bos_id; a.eos = eos_id; a.pad = pad_id;
  a.mask_id = mask_id;
  a.k0 = (unsigned)seed; a.k1 = (unsigned)(seed >> 32);
  a.o0 = (unsigned)offset & 0x7fffffffu; a.o1 = (unsigned)(offset >> 31);
  a.masked = masked_out; a.mask = mask_out;

  if (t_mode == 2) {
    // written by a separate tiny launch BEFORE the mask kernel reads nothing from it:
    // the mask kernel recomputes t from Philox, so there is no ordering hazard.
    hipLaunchKernelGGL(mask_write_t, dim3((unsigned)ceil_div(batch, 256)), dim3(256), 0, s, a, batch);
    LDM_CHECK_LAUNCH();
  }
  const bool vec_ok = aligned_to(ids, 16) && aligned_to(masked_out, 16) && aligned_to(mask_out, 4) &&
                      (!draw || aligned_to(draw, 16));
  int64_t done = 0;
  if (vec_ok && n >= 4) {
    const int64_t groups = n >> 2;
    int64_t blocks = ceil_div(groups, 256);
    if (blocks > 2048) blocks = 2048;   // 256 CUs x 8 resident blocks, grid-stride beyond
    hipLaunchKernelGGL(mask_fwd_vec4, dim3((unsigned)blocks), dim3(256), 0, s, a);
    LDM_CHECK_LAUNCH();
    done = groups << 2;
  }
  if (done < n) {
    int64_t blocks = ceil_div(n - done, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(mask_fwd_scalar, dim3((unsigned)blocks), dim3(256), 0, s, a, done);
    LDM_CHECK_LAUNCH();
  }
  return LDM_OK;
}
