// Multi-tensor AdamW step (decoupled weight decay, bias correction) over fp32 master parameters:
// the update torch.optim.AdamW applies for the optimizer built at main.py:87-91 of the reference,
// for up to LDM_ADAMW_MAX tensors per launch.  HBM-bound: p, g, m, v read once, p, m, v written
// once; the gradient can be pre-scaled (1 / world size after a summing all-reduce) and the
// compute-dtype copy of the parameter (the bf16 weight shadow the next forward reads) is written
// by the same pass.
#include "common.h"

namespace {

constexpr int MAXT = 24;          // tensors per launch (kernel arguments by value)
constexpr int CHUNK = 256 * 16;   // elements per workgroup: 256 threads x 4 x float4

struct AdamTensors {
  float *p[MAXT];
  const float *g[MAXT];
  float *m[MAXT];
  float *v[MAXT];
  void *shadow[MAXT];
  int64_t n[MAXT];
  int64_t vstart[MAXT + 1];       // first virtual chunk of tensor i (tensors never share a chunk)
  int shadow_kind[MAXT];          // 0 none, 1 bf16, 2 fp32
  int count;
};

struct AdamScalars {
  float lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale;
  const float *gate;        // nullable: the whole update is skipped when *gate == 0
  const int32_t *skipped;   // nullable: updates skipped so far (they did not advance the step count)
  int64_t step;
};

__global__ void adamw_gate_commit_k(const float *gate, int32_t *skipped) {
  if (gate[0] == 0.0f) skipped[0] += 1;
}

__global__ __launch_bounds__(256) void adamw_multi_k(AdamTensors t, AdamScalars s) {
  if (s.gate && s.gate[0] == 0.0f) return;
  if (s.skipped) {
    const int32_t sk = s.skipped[0];
    if (sk > 0) {   // bias corrections of the step count without the skipped updates
      const double k = (double)(s.step - sk > 1 ? s.step - sk : 1);
      s.bc1 = (float)(1.0 - pow((double)s.beta1, k));
      s.bc2_sqrt = (float)sqrt(1.0 - pow((double)s.beta2, k));
    }
  }
  // locate the tensor of this workgroup's chunk (count <= 24: a scan over scalar registers)
  const int64_t chunk = blockIdx.x;
  int ti = 0;
#pragma unroll 1
  for (int i = 1; i < t.count; ++i)
    if (chunk >= t.vstart[i]) ti = i;
  const int64_t base = (chunk - t.vstart[ti]) * CHUNK;
  const int64_t n = t.n[ti];
  float *p = t.p[ti];
  const float *g = t.g[ti];
  float *m = t.m[ti], *v = t.v[ti];
  const int kind = t.shadow_kind[ti];
  const float decay = 1.0f - s.lr * s.weight_decay;
  const float step_size = s.lr / s.bc1;
  const float omb1 = 1.0f - s.beta1, omb2 = 1.0f - s.beta2;
  const bool vec_ok = (n % 4 == 0);   // tensors start 16-byte aligned (checked on the host)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = base + ((int64_t)u * 256 + threadIdx.x) * 4;
    if (i >= n) break;
    if (vec_ok || i + 4 <= n) {
      float4 pv = *reinterpret_cast<const float4 *>(p + i);
      const float4 gv = *reinterpret_cast<const float4 *>(g + i);
      float4 mv = *reinterpret_cast<const float4 *>(m + i);
      float4 vv = *reinterpret_cast<const float4 *>(v + i);
      float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
      float mm[4] = {mv.x, mv.y, mv.z, mv.w}, ww[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gj = gg[j] * s.grad_scale;
        pp[j] *= decay;
        mm[j] = mm[j] + (gj - mm[j]) * omb1;
        ww[j] = ww[j] * s.beta2 + omb2 * gj * gj;
        const float denom = sqrtf(ww[j]) / s.bc2_sqrt + s.eps;
        pp[j] -= step_size * (mm[j] / denom);
      }
      *reinterpret_cast<float4 *>(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
      *reinterpret_cast<float4 *>(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
      *reinterpret_cast<float4 *>(v + i) = make_float4(ww[0], ww[1], ww[2], ww[3]);
      if (kind == 1) {
        uint2 o;
        o.x = pack_bf16x2(pp[0], pp[1]);
        o.y = pack_bf16x2(pp[2], pp[3]);
        *reinterpret_cast<uint2 *>((bf16_t *)t.shadow[ti] + i) = o;
      } else if (kind == 2) {
        *reinterpret_cast<float4 *>((float *)t.shadow[ti] + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
      }
    } else {
      for (int64_t e = i; e < n; ++e) {
        const float gj = g[e] * s.grad_scale;
        float pe = p[e] * decay;
        const float me = m[e] + (gj - m[e]) * omb1;
        const float ve = v[e] * s.beta2 + omb2 * gj * gj;
        pe -= step_size * (me / (sqrtf(ve) / s.bc2_sqrt + s.eps));
        p[e] = pe; m[e] = me; v[e] = ve;
        if (kind == 1) ((bf16_t *)t.shadow[ti])[e] = f32_to_bf16(pe);
        else if (kind == 2) ((float *)t.shadow[ti])[e] = pe;
      }
    }
  }
}

}  // namespace

extern "C" int ldm_adamw_step_gated(const ldm_adamw_tensor *tensors, int64_t count, int64_t step, float lr,
                                    float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                                    const float *gate, const int32_t *skipped, void *stream) {
  if (count < 0 || step < 1) return LDM_E_BADARG;
  if (count == 0) return LDM_OK;
  if (!tensors) return LDM_E_BADARG;
  for (int64_t i = 0; i < count; ++i) {
    const ldm_adamw_tensor &x = tensors[i];
    if (x.n < 0 || x.shadow_kind < 0 || x.shadow_kind > 2) return LDM_E_BADARG;
    if (x.n == 0) continue;
    if (!x.p || !x.g || !x.m || !x.v || (x.shadow_kind && !x.shadow)) return LDM_E_BADARG;
    if (!aligned_to(x.p, 16) || !aligned_to(x.g, 16) || !aligned_to(x.m, 16) || !aligned_to(x.v, 16) ||
        (x.shadow_kind == 1 && !aligned_to(x.shadow, 8)) || (x.shadow_kind == 2 && !aligned_to(x.shadow, 16)))
      return LDM_E_ALIGN;
  }
  AdamScalars s;
  s.lr = lr; s.beta1 = beta1; s.beta2 = beta2; s.eps = eps; s.weight_decay = weight_decay; s.grad_scale = grad_scale;
  s.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  s.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  s.gate = gate; s.skipped = skipped; s.step = step;
  hipStream_t st = (hipStream_t)stream;
  int64_t i = 0;
  while (i < count) {
    AdamTensors t;
    t.count = 0;
    int64_t chunks = 0;
    while (i < count && t.count < MAXT) {
      const ldm_adamw_tensor &x = tensors[i++];
      if (x.n == 0) continue;
      const int k = t.count++;
      t.p[k] = x.p; t.g[k] = x.g; t.m[k] = x.m; t.v[k] = x.v; t.shadow[k] = x.shadow; t.n[k] = x.n;
      t.shadow_kind[k] = x.shadow_kind;
      t.vstart[k] = chunks;
      chunks += ceil_div(x.n, CHUNK);
    }
    if (t.count == 0) break;
    t.vstart[t.count] = chunks;
    if (chunks > 0x7fffffff) return LDM_E_SHAPE;
    hipLaunchKernelGGL(adamw_multi_k, dim3((unsigned)chunks), dim3(256), 0, st, t, s);
    LDM_CHECK_LAUNCH();
  }
  return LDM_OK;
}

extern "C" int ldm_adamw_step(const ldm_adamw_tensor *tensors, int64_t count, int64_t step, float lr, float beta1,
                              float beta2, float eps, float weight_decay, float grad_scale, void *stream) {
  return ldm_adamw_step_gated(tensors, count, step, lr, beta1, beta2, eps, weight_decay, grad_scale, nullptr, nullptr,
                              stream);
}

extern "C" int ldm_adamw_gate_commit(const float *gate, int32_t *skipped, void *stream) {
  if (!gate || !skipped) return LDM_E_BADARG;
  hipLaunchKernelGGL(adamw_gate_commit_k, dim3(1), dim3(1), 0, (hipStream_t)stream, gate, skipped);
  LDM_CHECK_LAUNCH();
  return LDM_OK;
}
