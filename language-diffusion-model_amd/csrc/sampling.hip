// Reverse-process token selection (HBM-bound): what DiffusionLanguageModel.generate / sample
// do after the transformer forward (src/model.py:93-105, 167-177) and what
// RandomRemaskingStrategy.apply_remask does (src/masking_strategy.py:160-198), without the
// [T, V] probability tensor, the boolean index copies and the nonzero() of the reference.
#include <limits.h>

#include "common.h"
#include "philox.h"

namespace {

struct Best { float v; int i; };
__device__ __forceinline__ Best better(Best a, Best b) {
  // the larger score; of equal scores the smaller column (torch.argmax returns the first maximum)
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

struct SampleArgs {
  int64_t ld, T, V;
  const int64_t *ids;
  int64_t mask_id;
  int all_rows;
  float inv_temp;            // 0: greedy
  unsigned k0, k1, o0, o1;   // Philox key = seed, counter words 2,3 = call offset
  int64_t *filled;           // ids with the mask positions replaced by the prediction (may be null)
  int64_t *pred;             // prediction of every visited row (may be null)
};

// Gumbel noise of column c in row r: word c&3 of the Philox block with counter (c>>2, r, offset).
__device__ __forceinline__ void gumbel4(const SampleArgs &a, int64_t r, int64_t g, float (&n)[4]) {
  const U4 w = philox4x32_10(U4{(unsigned)g, (unsigned)r, a.o0, a.o1}, a.k0, a.k1);
  const unsigned x[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float u = ((float)(x[j] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // in (0, 1)
    n[j] = -logf(-logf(u));
  }
}

// One 256-thread block per row; rows that are not mask positions leave after one id read.
// temperature 0: argmax.  temperature > 0: argmax(logit / temperature + Gumbel noise), which
// is a draw from softmax(logits / temperature) (the reference draws it with torch.multinomial).
template <typename T>
__global__ __launch_bounds__(256) void sample_rows_k(const T *logits, SampleArgs a) {
  const int64_t r = blockIdx.x;
  const int64_t id = a.ids[r];
  const bool is_mask = id == a.mask_id;
  if (!a.all_rows && !is_mask) {
    if (threadIdx.x == 0 && a.filled) a.filled[r] = id;
    return;
  }
  const T *row = logits + r * a.ld;
  Best b{-INFINITY, INT_MAX};
  const bool vec = (a.ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(logits) % 16) == 0);
  const int64_t ng = (a.V + 3) >> 2;
  for (int64_t g = threadIdx.x; g < ng; g += 256) {
    const int64_t c0 = g << 2;
    float x[4];
    if (vec && c0 + 4 <= a.V) {
      const f4 v = ld4(row + c0);
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = v.v[j];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = c0 + j < a.V ? Elt<T>::ld(row + c0 + j) : -INFINITY;
    }
    if (a.inv_temp > 0.f) {
      float n[4];
      gumbel4(a, r, g, n);
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = fmaf(x[j], a.inv_temp, n[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c0 + j < a.V && x[j] > b.v) b = Best{x[j], (int)(c0 + j)};
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) b = better(b, Best{__shfl_xor(b.v, o, 64), __shfl_xor(b.i, o, 64)});
  __shared__ float sv[4];
  __shared__ int si[4];
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = b.v;
    si[threadIdx.x >> 6] = b.i;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) b = better(b, Best{sv[w], si[w]});
    const int64_t p = b.i == INT_MAX ? 0 : b.i;   // a row of -inf / NaN only: column 0
    if (a.pred) a.pred[r] = p;
    if (a.filled) a.filled[r] = is_mask ? p : id;
  }
}

struct RemaskArgs {
  int64_t n;
  const int64_t *ids;
  const uint8_t *was_masked;   // mask_indices of the step
  const uint8_t *prompt;       // may be null
  int64_t bos, eos, pad;
};
__device__ __forceinline__ bool remask_eligible(const RemaskArgs &a, int64_t i) {
  const int64_t id = a.ids[i];
  bool ok = id != a.bos && id != a.eos && id != a.pad && a.was_masked[i] != 0;
  if (a.prompt) ok = ok && a.prompt[i] == 0;
  return ok;
}

// pos[0..count) = flat indices of the eligible positions in ascending order (the order of
// Tensor.nonzero()).  One 1024-thread block: ballot + popcount inside a wave, wave counts via LDS.
__global__ __launch_bounds__(1024) void remask_compact_k(RemaskArgs a, int64_t *pos, int *count) {
  __shared__ int wsum[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int base = 0;
  for (int64_t g0 = 0; g0 < a.n; g0 += 1024) {
    const int64_t i = g0 + threadIdx.x;
    const bool act = i < a.n && remask_eligible(a, i);
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
    if (act) pos[base + before + __popcll(m & ((1ull << lane) - 1ull))] = i;
    base += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) count[0] = base;
}

// out[pos[perm[j]]] = mask_id for j < k  (out already holds a copy of ids)
__global__ __launch_bounds__(256) void remask_scatter_k(const int64_t *pos, const int64_t *perm, int64_t k,
                                                        int64_t num_eligible, int64_t mask_id, int64_t *out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= k) return;
  const int64_t e = perm[j];
  if (e >= 0 && e < num_eligible) out[pos[e]] = mask_id;
}

}  // namespace

extern "C" int ldm_sample_tokens(const void *logits, int dtype, int64_t T, int64_t V, int64_t ld,
                                 const int64_t *ids, int64_t mask_id, int all_rows, float temperature,
                                 uint64_t seed, uint64_t offset, int64_t *filled_out, int64_t *pred_out,
                                 void *stream) {
  if (T < 0 || V <= 0 || ld < V || V > INT_MAX || temperature < 0.f) return LDM_E_BADARG;
  if (T == 0) return LDM_OK;
  if (!logits || !ids || (!filled_out && !pred_out)) return LDM_E_BADARG;
  SampleArgs a;
  a.ld = ld; a.T = T; a.V = V; a.ids = ids; a.mask_id = mask_id; a.all_rows = all_rows;
  a.inv_temp = temperature > 0.f ? 1.0f / temperature : 0.f;
  a.k0 = (unsigned)seed; a.k1 = (unsigned)(seed >> 32);
  a.o0 = (unsigned)offset; a.o1 = (unsigned)(offset >> 32);
  a.filled = filled_out; a.pred = pred_out;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == LDM_F32)
    hipLaunchKernelGGL(sample_rows_k<float>, dim3((unsigned)T), dim3(256), 0, s, (const float *)logits, a);
  else if (dtype == LDM_BF16)
    hipLaunchKernelGGL(sample_rows_k<bf16_t>, dim3((unsigned)T), dim3(256), 0, s, (const bf16_t *)logits, a);
  else
    return LDM_E_BADARG;
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

extern "C" int ldm_remask_eligible(const int64_t *ids, const uint8_t *was_masked, const uint8_t *prompt,
                                   int64_t n, int64_t bos_id, int64_t eos_id, int64_t pad_id,
                                   int64_t *pos_out, int32_t *count_out, void *stream) {
  if (n < 0 || !count_out) return LDM_E_BADARG;
  if (n > 0 && (!ids || !was_masked || !pos_out)) return LDM_E_BADARG;
  RemaskArgs a;
  a.n = n; a.ids = ids; a.was_masked = was_masked; a.prompt = prompt;
  a.bos = bos_id; a.eos = eos_id; a.pad = pad_id;
  hipLaunchKernelGGL(remask_compact_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, pos_out, count_out);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}

extern "C" int ldm_remask_apply(const int64_t *ids, int64_t n, const int64_t *pos, int64_t num_eligible,
                                const int64_t *perm, int64_t k, int64_t mask_id, int64_t *out, void *stream) {
  if (n < 0 || k < 0 || k > num_eligible) return LDM_E_BADARG;
  if (n == 0) return LDM_OK;
  if (!ids || !out || (k > 0 && (!pos || !perm))) return LDM_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  if (out != ids) {
    hipError_t e = hipMemcpyAsync(out, ids, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return (int)e;
  }
  if (k > 0) {
    hipLaunchKernelGGL(remask_scatter_k, dim3((unsigned)ceil_div(k, 256)), dim3(256), 0, s, pos, perm, k,
                       num_eligible, mask_id, out);
    LDM_CHECK_LAUNCH();
  }
  return LDM_OK;
}
