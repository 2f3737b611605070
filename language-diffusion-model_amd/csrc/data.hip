// Device-side batch producer for the {"input_ids", "attention_mask"} contract of the training step.
//   collate_pad_k : what collate_batch (src/dataset.py:26-48: pad_sequence of ids and masks with 0)
//                   produces, from the UNPADDED tokens of the batch: only the real tokens cross PCIe
//   synth_tokens_k: synthetic batches generated on the device (Philox4x32-10), optional ragged lengths
// Integer work, HBM-bound: 16 B written per token, 8 B (collate) or 0 B (synthetic) read.
#include "common.h"
#include "philox.h"

namespace {

// One wave per 64 consecutive positions of one row; grid.y = row.
__global__ __launch_bounds__(256) void collate_pad_k(const int64_t *flat, const int64_t *offsets, int64_t batch,
                                                     int64_t smax, int64_t pad, int64_t *ids, int64_t *mask) {
  const int64_t b = blockIdx.y;
  const int64_t start = offsets[b], len = offsets[b + 1] - start;
  for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < smax; s += (int64_t)gridDim.x * 256) {
    const bool real = s < len;
    ids[b * smax + s] = real ? flat[start + s] : pad;
    mask[b * smax + s] = real ? 1 : 0;
  }
}

struct SynthArgs {
  int64_t batch, seq, lo, span, bos, eos, pad, min_len;
  unsigned k0, k1, o0, o1;
};

__device__ __forceinline__ int64_t synth_len(const SynthArgs &a, int64_t b) {
  if (a.min_len >= a.seq) return a.seq;
  const U4 r = philox4x32_10(U4{(unsigned)b, (unsigned)((unsigned long long)b >> 32), a.o0 ^ 0x80000000u, a.o1}, a.k0, a.k1);
  return a.min_len + (int64_t)(((unsigned long long)r.x * (unsigned long long)(a.seq - a.min_len + 1)) >> 32);
}

// 4 consecutive flat positions per thread: one Philox block (counter = position >> 2) gives their
// four 32-bit words; id = lo + (word * span) >> 32.
__global__ __launch_bounds__(256) void synth_tokens_k(SynthArgs a, int64_t *ids, int64_t *mask) {
  const int64_t total = a.batch * a.seq;
  const int64_t ngroups = (total + 3) >> 2;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * 256) {
    const U4 r = philox4x32_10(U4{(unsigned)g, (unsigned)((unsigned long long)g >> 32), a.o0, a.o1}, a.k0, a.k1);
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
    int64_t row = -1, len = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = g * 4 + j;
      if (i >= total) break;
      const int64_t b = i / a.seq, s = i - b * a.seq;
      if (b != row) { row = b; len = synth_len(a, b); }
      int64_t id = a.lo + (int64_t)(((unsigned long long)w[j] * (unsigned long long)a.span) >> 32);
      if (s == 0) id = a.bos;
      else if (s == len - 1) id = a.eos;
      if (s >= len) id = a.pad;
      ids[i] = id;
      mask[i] = s < len ? 1 : 0;
    }
  }
}

}  // namespace

extern "C" int ldm_collate_pad(const int64_t *flat, const int64_t *offsets, int64_t batch, int64_t smax,
                               int64_t pad_id, int64_t *ids_out, int64_t *mask_out, void *stream) {
  if (batch < 0 || smax < 0) return LDM_E_BADARG;
  if (batch == 0 || smax == 0) return LDM_OK;
  if (!flat || !offsets || !ids_out || !mask_out) return LDM_E_BADARG;
  if (batch > 65535) return LDM_E_SHAPE;
  unsigned gx = (unsigned)ceil_div(smax, 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(collate_pad_k, dim3(gx, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, flat, offsets, batch,
                     smax, pad_id, ids_out, mask_out);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

extern "C" int ldm_synth_tokens(int64_t batch, int64_t seq, int64_t id_lo, int64_t id_hi, int64_t bos_id,
                                int64_t eos_id, int64_t pad_id, int64_t min_len, uint64_t seed, uint64_t offset,
                                int64_t *ids_out, int64_t *mask_out, void *stream) {
  if (batch < 0 || seq < 0 || id_hi <= id_lo || id_hi - id_lo > 0xffffffffll) return LDM_E_BADARG;
  if (batch == 0 || seq == 0) return LDM_OK;
  if (!ids_out || !mask_out) return LDM_E_BADARG;
  if (min_len < 2 && seq >= 2) min_len = 2;      // BOS and EOS always fit
  if (min_len > seq) min_len = seq;
  SynthArgs a;
  a.batch = batch; a.seq = seq; a.lo = id_lo; a.span = id_hi - id_lo; a.bos = bos_id; a.eos = eos_id; a.pad = pad_id;
  a.min_len = min_len;
  a.k0 = (unsigned)seed; a.k1 = (unsigned)(seed >> 32);
  a.o0 = (unsigned)(offset & 0x7fffffffu); a.o1 = (unsigned)(offset >> 31);
  int64_t blocks = ceil_div(ceil_div(batch * seq, 4), 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(synth_tokens_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, ids_out, mask_out);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}
