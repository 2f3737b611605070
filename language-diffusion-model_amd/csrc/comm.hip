// Gradient all-reduce over RCCL behind the C-ABI: what DistributedDataParallel does for the reference
// (main.py:105-108 wraps the model, src/trainer.py:136 runs the bucketed all-reduce inside backward).
// One process per GPU; a bucket (a flat fp32 or bf16 buffer of gradients) is summed in place across the
// ranks on a communication stream that is joined to the producing stream by events, so the collective
// overlaps the rest of backward.
//
// RCCL is looked up at run time (dlopen): the library has no link-time dependency on it, and a process that
// already holds an RCCL (PyTorch's) gets that same instance, so a communicator created by the caller's
// RCCL can be handed in.
#include "common.h"
#include <dlfcn.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <mutex>

namespace {

typedef struct ncclComm *comm_t;
struct unique_id_t { char internal[128]; };

// ncclResult_t / ncclDataType_t / ncclRedOp_t values of rccl.h (ABI-stable since NCCL 2.x)
constexpr int kSuccess = 0, kFloat32 = 7, kBfloat16 = 9, kSum = 0;

struct Rccl {
  void *handle = nullptr;
  int (*GetUniqueId)(unique_id_t *) = nullptr;
  int (*CommInitRank)(comm_t *, int, unique_id_t, int) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  int (*CommCount)(comm_t, int *) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
};

char g_err[256] = "no error";
void set_err(const char *what, const char *detail) { snprintf(g_err, sizeof g_err, "%s: %s", what, detail ? detail : "?"); }

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *env = getenv("LDM_RCCL_LIB");
    // an instance that is already in the process first (RTLD_NOLOAD), then the system's
    const char *names[] = {env, "librccl.so", "librccl.so.1"};
    for (int pass = 0; pass < 2 && !r.handle; ++pass)
      for (const char *n : names) {
        if (!n || !*n) continue;
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
        if (r.handle) break;
      }
    if (!r.handle) { set_err("dlopen(librccl)", dlerror()); return; }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))dlsym(r.handle, "ncclCommCount");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommCount && r.AllReduce;
    if (!r.ok) set_err("dlsym(librccl)", "a collective entry point is missing");
  });
  return r;
}

// one event per thread and device, re-recorded by every call (a wait that was queued on an earlier record
// keeps that record's position)
hipEvent_t order_event() {
  thread_local hipEvent_t ev[64] = {};
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return nullptr;
  if (!ev[d] && hipEventCreateWithFlags(&ev[d], hipEventDisableTiming) != hipSuccess) ev[d] = nullptr;
  return ev[d];
}

// work queued on `after` from now on runs behind what `before` holds now
int order_streams(hipStream_t before, hipStream_t after) {
  if (before == after) return LDM_OK;
  hipEvent_t ev = order_event();
  if (!ev) return LDM_E_COMM;
  hipError_t e = hipEventRecord(ev, before);
  if (e == hipSuccess) e = hipStreamWaitEvent(after, ev, 0);
  return e == hipSuccess ? LDM_OK : (int)e;
}

int fail(const Rccl &r, const char *what, int rc) {
  set_err(what, r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
  return LDM_E_COMM;
}

}  // namespace

extern "C" const char *ldm_comm_last_error(void) { return g_err; }

extern "C" int ldm_comm_unique_id(void *id128) {
  if (!id128) return LDM_E_BADARG;
  Rccl &r = rccl();
  if (!r.ok) return LDM_E_COMM;
  unique_id_t id;
  const int rc = r.GetUniqueId(&id);
  if (rc != kSuccess) return fail(r, "ncclGetUniqueId", rc);
  memcpy(id128, id.internal, sizeof id.internal);
  return LDM_OK;
}

extern "C" int ldm_comm_init_rank(void **comm, int nranks, const void *id128, int rank) {
  if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return LDM_E_BADARG;
  Rccl &r = rccl();
  if (!r.ok) return LDM_E_COMM;
  unique_id_t id;
  memcpy(id.internal, id128, sizeof id.internal);
  comm_t c = nullptr;
  const int rc = r.CommInitRank(&c, nranks, id, rank);
  if (rc != kSuccess) return fail(r, "ncclCommInitRank", rc);
  *comm = (void *)c;
  return LDM_OK;
}

extern "C" int ldm_comm_destroy(void *comm) {
  if (!comm) return LDM_OK;
  Rccl &r = rccl();
  if (!r.ok) return LDM_E_COMM;
  const int rc = r.CommDestroy((comm_t)comm);
  return rc == kSuccess ? LDM_OK : fail(r, "ncclCommDestroy", rc);
}

extern "C" int ldm_comm_size(void *comm, int *nranks) {
  if (!comm || !nranks) return LDM_E_BADARG;
  Rccl &r = rccl();
  if (!r.ok) return LDM_E_COMM;
  const int rc = r.CommCount((comm_t)comm, nranks);
  return rc == kSuccess ? LDM_OK : fail(r, "ncclCommCount", rc);
}

// sum buf[0..count) over the communicator's ranks, in place, on comm_stream, behind everything queued on
// producer_stream so far (the kernels that wrote the gradients).  Returns at once.
extern "C" int ldm_allreduce_bucket(void *comm, void *buf, int64_t count, int dtype, void *producer_stream,
                                    void *comm_stream) {
  if (!comm || count < 0) return LDM_E_BADARG;
  if (dtype != LDM_F32 && dtype != LDM_BF16) return LDM_E_DTYPE;
  if (count == 0) return LDM_OK;
  if (!buf) return LDM_E_BADARG;
  if (!aligned_to(buf, 16)) return LDM_E_ALIGN;
  Rccl &r = rccl();
  if (!r.ok) return LDM_E_COMM;
  hipStream_t ps = (hipStream_t)producer_stream, cs = (hipStream_t)comm_stream;
  const int oc = order_streams(ps, cs);
  if (oc != LDM_OK) return oc;
  const int rc = r.AllReduce(buf, buf, (size_t)count, dtype == LDM_F32 ? kFloat32 : kBfloat16, kSum, (comm_t)comm, cs);
  return rc == kSuccess ? LDM_OK : fail(r, "ncclAllReduce", rc);
}

// everything queued on consumer_stream after this call runs behind what comm_stream holds now
extern "C" int ldm_allreduce_join(void *comm_stream, void *consumer_stream) {
  return order_streams((hipStream_t)comm_stream, (hipStream_t)consumer_stream);
}
