"""Data-parallel gradient synchronisation: bucketed all-reduce overlapped with backward.

Stands in for what the reference gets from ``Accelerator().prepare(model, ...)``
(main.py:105-108 -> torch DistributedDataParallel, 25 MiB buckets) and
``accelerator.backward(loss)`` (src/trainer.py:74,136).  One process per GPU;
``torch.distributed`` backend ``nccl`` (RCCL over xGMI on ROCm) or ``gloo`` (CPU).

Semantics kept from DDP: every rank ends backward with grad = mean over ranks of
its local gradients (each rank's loss is normalised by ITS OWN masked count).

Mechanics:
  * parameters are assigned to flat fp32 buckets in REVERSE registration order
    (LM head and final norm first, embedding last), the order backward produces them;
  * a post-accumulate hook copies each finished gradient into its bucket slice and
    repoints ``param.grad`` at the slice; when a bucket's last gradient arrives its
    all-reduce is launched asynchronously -- under the RCCL backend through the library's
    ``ldm_allreduce_bucket`` (ncclAllReduce on a communication stream of this wrapper, joined
    to the compute stream by events), otherwise through ``torch.distributed`` --, so
    communication of late layers overlaps the backward of early ones;
  * an engine callback queued on the first hook waits for all buckets at the end of
    backward and applies the 1/world_size scaling; parameters that produced no
    gradient contribute zeros so that ranks stay in lock-step.
Buckets default to 64 MiB: xGMI is point-to-point, a ring step is bound by one
~153 GB/s link and has a fixed latency per collective, so fewer/larger collectives
than DDP's 25 MiB are the better trade on one MI355X node.
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.autograd import Variable


class _Bucket:
    def __init__(self, params, device, wire_dtype=torch.float32):
        self.params = params
        self.wire_dtype = wire_dtype
        # every slice starts on a 256-byte boundary (kernels that take the gradient views, e.g. the
        # AdamW pass, want 16-byte aligned pointers; a vocabulary bias of 30522 floats is not a
        # multiple of that); the padding elements stay zero
        self.offsets = {}
        off = 0
        for p in params:
            self.offsets[p] = off
            off += (p.numel() + 63) // 64 * 64
        self.numel = off
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=device)
        # the buffer that goes over the wire when the bucket is exchanged in a narrower type (bf16 halves
        # the bytes per step: 16 GB instead of 32 GB at the 8B shape); gradients are kept and applied in fp32
        self.wire = torch.zeros(self.numel, dtype=wire_dtype, device=device) if wire_dtype != torch.float32 else None
        self.pending = len(params)
        self.ready = set()
        self.work = None
        self.host = None

    def view_for(self, p):
        off = self.offsets[p]
        return self.flat[off:off + p.numel()].view_as(p)


class GradBucketDataParallel(nn.Module):
    def __init__(self, module: nn.Module, process_group=None, bucket_bytes: int = 64 << 20,
                 wire_dtype: torch.dtype = torch.float32):
        super().__init__()
        self.module = module
        self.process_group = process_group
        if wire_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("wire_dtype must be torch.float32 or torch.bfloat16")
        self.wire_dtype = wire_dtype
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # LDM_DP_FORCE=1 keeps the bucket/all-reduce machinery active on a single rank (rehearsal)
        self._active = self.world_size > 1 or (dist.is_initialized() and os.environ.get("LDM_DP_FORCE") == "1")
        params = [p for p in module.parameters() if p.requires_grad]
        self._buckets: list[_Bucket] = []
        self._bucket_of = {}
        cur, cur_bytes = [], 0
        for p in reversed(params):
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                self._add_bucket(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._add_bucket(cur)
        self._callback_queued = False
        self._next = 0
        # True: leave the SUM over ranks in the buckets; the optimizer applies 1 / world_size inside its
        # update (FusedAdamW.grad_scale), which saves one pass over every gradient
        self.defer_scale = False
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]
        # identical starting point on every rank (DDP broadcasts rank 0's state too)
        # gloo + device tensors: the payload is staged through host memory HERE, on the current stream,
        # and gloo is handed a host tensor.  gloo's own device path (pinned staging on its side streams,
        # completion events recorded from its worker thread) completes isolated collectives between two
        # processes that share a GPU (scripts/gloo_dev_bcast_probe.py), but the pattern of this wrapper --
        # dozens of back-to-back broadcasts with no device synchronisation in between, then asynchronous
        # all-reduces issued from autograd hooks -- does not return on this ROCm stack (observed twice:
        # both ranks parked in broadcast()'s wait).  gloo never exchanges device handles, so this is not
        # an IPC matter, and the RCCL path (the one a multi-GPU node runs) is not involved.
        # LDM_DP_HOST_STAGE=0 hands gloo the device tensors (for re-testing on another stack).
        self._host_staged = (dist.is_initialized() and dist.get_backend(process_group) == "gloo"
                             and os.environ.get("LDM_DP_HOST_STAGE", "1") != "0")
        if self.world_size > 1:
            src = dist.get_global_rank(process_group, 0) if process_group else 0
            for t in list(module.parameters()) + list(module.buffers()):
                if self._host_staged and t.is_cuda:
                    h = t.data.cpu()
                    dist.broadcast(h, src=src, group=process_group)
                    t.data.copy_(h)
                else:
                    dist.broadcast(t.data, src=src, group=process_group)
        # RCCL backend: the buckets go through the library's own entry point (ldm_allreduce_bucket: ncclAllReduce
        # on a communication stream, ordered against the compute stream by events) on a communicator of its
        # own; torch.distributed stays the channel for set-up and the fallback.  LDM_DP_NATIVE=0 switches it off.
        self._comm = None
        self._comm_stream = None
        if (self._active and dist.get_backend(process_group) == "nccl" and params and params[0].is_cuda
                and os.environ.get("LDM_DP_NATIVE", "1") != "0"):
            self._setup_native_comm(params[0].device)

    def _setup_native_comm(self, device):
        """Create this wrapper's RCCL communicator and prove it with one small all-reduce.  Every rank takes
        the same decision: if any rank failed, all of them stay on torch.distributed."""
        import ctypes as C
        import warnings
        from . import _native as N
        lib = N.lib()
        group = self.process_group
        rank = dist.get_rank(group)
        ok, comm, why = 1, C.c_void_p(), ""
        ident = [None]
        if rank == 0:
            buf = C.create_string_buffer(128)
            if lib.ldm_comm_unique_id(buf) == 0:
                ident[0] = buf.raw
        dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group else 0, group=group)
        stream = torch.cuda.Stream(device=device)
        # two communicators (torch.distributed's and this one) never have collectives in flight at the same
        # time: the device is idle before this one's first collective, and again before torch's next one
        torch.cuda.synchronize(device)
        try:
            if ident[0] is None:
                raise RuntimeError("rank 0 could not create a communicator id: " + lib.ldm_comm_last_error().decode())
            with torch.cuda.device(device):
                N.check(lib.ldm_comm_init_rank(C.byref(comm), self.world_size, ident[0], rank), "ldm_comm_init_rank")
                probe = torch.ones(64, dtype=torch.float32, device=device)
                cur = torch.cuda.current_stream(device).cuda_stream
                N.check(lib.ldm_allreduce_bucket(comm, probe.data_ptr(), probe.numel(), N.F32, cur, stream.cuda_stream),
                        "ldm_allreduce_bucket")
                N.check(lib.ldm_allreduce_join(stream.cuda_stream, cur), "ldm_allreduce_join")
                if not bool((probe == float(self.world_size)).all().item()):
                    raise RuntimeError("probe all-reduce returned a wrong sum")
        except Exception as e:       # noqa: BLE001 -- any failure means: use torch.distributed
            ok, why = 0, str(e)
        torch.cuda.synchronize(device)
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 1:
            self._comm, self._comm_stream = comm, stream
        else:
            if comm.value:
                lib.ldm_comm_destroy(comm)
            warnings.warn("native RCCL bucket all-reduce unavailable, using torch.distributed"
                          + (f" ({why})" if why else " (another rank failed)"))

    def _add_bucket(self, params):
        b = _Bucket(list(params), params[0].device, self.wire_dtype)
        self._buckets.append(b)
        for p in params:
            self._bucket_of[p] = b

    # attribute passthrough: the trainer reads model.tokenizer, calls model.train()/eval()
    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(super().__getattr__("module"), name)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    @property
    def num_buckets(self) -> int:
        return len(self._buckets)

    def _on_grad(self, p):
        if not self._active:
            return
        if not self._callback_queued:
            Variable._execution_engine.queue_callback(self._finalize)
            self._callback_queued = True
        b = self._bucket_of[p]
        if p in b.ready:          # a second accumulation into the same grad in one backward
            return
        view = b.view_for(p)
        if p.grad.data_ptr() != view.data_ptr():
            view.copy_(p.grad)
            p.grad = view
        b.ready.add(p)
        self._launch_ready()

    def _launch_ready(self):
        # Collectives must be issued in the same order on every rank: bucket i+1 is launched
        # only after bucket i, whatever order the gradients arrive in on this rank.
        while self._next < len(self._buckets):
            b = self._buckets[self._next]
            if len(b.ready) != len(b.params):
                break
            payload = b.flat
            if b.wire is not None:
                b.wire.copy_(b.flat)           # one rounding per rank before the sum
                payload = b.wire
            if self._comm is not None:
                from . import _native as N
                b.host = None
                b.work = None
                N.check(N.lib().ldm_allreduce_bucket(self._comm, payload.data_ptr(), payload.numel(),
                                                     N.dtype_code(payload.dtype), N.stream(),
                                                     self._comm_stream.cuda_stream), "ldm_allreduce_bucket")
            elif self._host_staged and payload.is_cuda:
                b.host = payload.cpu()
                b.work = dist.all_reduce(b.host, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)
            else:
                b.host = None
                b.work = dist.all_reduce(payload, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)
            self._next += 1

    def _finalize(self):
        self._callback_queued = False
        for b in self._buckets[self._next:]:
            # parameters that produced no gradient this step contribute zeros
            for p in b.params:
                if p not in b.ready:
                    view = b.view_for(p)
                    view.zero_()
                    p.grad = view
                    b.ready.add(p)
        self._launch_ready()
        self._next = 0
        inv = 1.0 / self.world_size
        if self._comm is not None:
            from . import _native as N
            # the copies, the scaling and the optimizer step queued from here on run behind every bucket's sum
            N.check(N.lib().ldm_allreduce_join(self._comm_stream.cuda_stream, N.stream()), "ldm_allreduce_join")
        for b in self._buckets:
            if b.work is not None:
                b.work.wait()
            if b.host is not None:
                (b.wire if b.wire is not None else b.flat).copy_(b.host)
                b.host = None
            if b.wire is not None:
                b.flat.copy_(b.wire)
            if not self.defer_scale:
                b.flat.mul_(inv)
            b.work = None
            b.ready.clear()

    def close(self):
        """Release this wrapper's RCCL communicator (every rank calls it, before the process group goes)."""
        comm, self._comm = self._comm, None
        if comm is not None:
            from . import _native as N
            torch.cuda.synchronize()
            N.lib().ldm_comm_destroy(comm)

    def finish_grad_sync(self):
        """Explicit form of the end-of-backward callback (idempotent)."""
        if self._callback_queued:
            self._finalize()


class HipAccelerator:
    """The small part of ``accelerate.Accelerator`` that the training step touches.

    ``prepare`` moves modules to this rank's GPU, selects the compute dtype
    (``mixed_precision="bf16"`` is what the reference's Accelerate config would
    request) and wraps the model for gradient all-reduce when more than one rank is
    running; ``backward`` is ``loss.backward()`` (accelerate/accelerator.py backward
    with gradient_accumulation_steps == 1)."""

    def __init__(self, mixed_precision: str = "no", device=None, bucket_bytes: int = 64 << 20,
                 grad_wire_dtype: torch.dtype = torch.float32):
        self.mixed_precision = mixed_precision
        self.bucket_bytes = bucket_bytes
        self.grad_wire_dtype = grad_wire_dtype   # torch.bfloat16: gradient buckets cross the links in bf16
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        self.num_processes = dist.get_world_size() if dist.is_initialized() else 1
        self.process_index = dist.get_rank() if dist.is_initialized() else 0
        self.is_main_process = self.process_index == 0

    def prepare(self, *objs):
        from .optim import FusedAdamW
        out = []
        for o in objs:
            if isinstance(o, FusedAdamW):
                # the optimizer keeps the packed weight copies of the last prepared model current and,
                # under data parallelism, takes over the 1 / world_size scaling of the summed gradients
                model = getattr(self, "_last_model", None)
                if model is not None:
                    inner = model.module if isinstance(model, GradBucketDataParallel) else model
                    o.attach(inner)
                    if isinstance(model, GradBucketDataParallel) and model._active:
                        model.defer_scale = True
                        o.grad_scale = 1.0 / model.world_size
            elif (isinstance(o, torch.optim.Optimizer) and not getattr(o, "_ldm_epoch_hook", False)
                  and hasattr(o, "_optimizer_step_post_hooks")):
                # fused torch optimizers update parameters without moving their version counters.  (Wrappers
                # that never ran Optimizer.__init__ have no hook registry; the trainer marks the weight
                # copies stale after their step() itself.)
                from . import ops
                o.register_step_post_hook(lambda *_a, **_k: ops.parameters_changed())
                o._ldm_epoch_hook = True
            if isinstance(o, nn.Module):
                o = o.to(self.device)
                dtype = torch.bfloat16 if self.mixed_precision == "bf16" else torch.float32
                for m in o.modules():
                    if hasattr(m, "set_compute_dtype"):
                        m.set_compute_dtype(dtype)
                    # accelerate wraps a prepared model's forward in convert_outputs_to_fp32 under mixed
                    # precision: the callers of the reference (src/trainer.py:68-69, the sampling loop)
                    # see fp32 logits.  The fused training loss never forms them, so it is unaffected.
                    if self.mixed_precision == "bf16" and hasattr(m, "fp32_logits"):
                        m.fp32_logits = True
                if self.num_processes > 1 or (dist.is_initialized() and os.environ.get("LDM_DP_FORCE") == "1"):
                    o = GradBucketDataParallel(o, bucket_bytes=self.bucket_bytes, wire_dtype=self.grad_wire_dtype)
                self._last_model = o
            out.append(o)
        return out[0] if len(out) == 1 else tuple(out)

    def backward(self, loss, **kwargs):
        loss.backward(**kwargs)
