"""Autograd operators over the HIP kernels.

Each ``torch.autograd.Function`` here owns its save-for-backward state and calls
the C ABI with raw device pointers on torch's current stream.  PyTorch only
provides device memory, streams and the autograd tape.
"""

from __future__ import annotations

import math
import os

import torch
from torch import Tensor

from . import _native as N

PAD_N = 64  # output columns of a projection are padded to this multiple in memory

# LM-head backward over the participating rows only (no host sync: the row count stays on the
# device).  A _HeadLink is created per forward call of the vocabulary projection and travels WITH the
# logits tensor (attribute ``_ldm_head_link`` of the tensor object the projection returns) to the
# masked-CE that consumes it; the CE backward leaves its compact gradient (or the note that its dense
# gradient buffer has zeroed padding columns) on the link, and the projection's backward -- which holds
# the same link in its autograd context -- picks it up.  Nothing is keyed by addresses and nothing
# outlives the tensors of one forward/backward pair, so several models, or a forward that is never
# followed by its backward, cannot see each other's records.
class _HeadLink:
    __slots__ = ("y_ptr", "ld", "compact_ok", "grad", "zero_padded")

    def __init__(self):
        self.y_ptr = 0            # data pointer and row stride of the logits buffer this link describes
        self.ld = 0
        self.compact_ok = False   # the projection's backward can take compact gradient rows
        self.grad = None          # (dlogits_compact, rows, count, dummy) left by the masked-CE backward
        self.zero_padded = None   # (data_ptr, row stride) of a dense gradient whose padding columns are zero


COMPACT_HEAD = True   # set False to force the dense LM-head backward
FUSED_UNROPE = True     # rotary backward inside the attention-backward stores (False: separate pass)
FUSED_BIAS_GRAD = True  # bias gradient from the weight-gradient GEMM (False: separate column-sum kernel)
COMPACT_STATS = {"used": 0}


def _expand_compact(dlc, rows, count, T, n_pad):
    """Dense [T, n_pad] form of a compact gradient (fallback path; synchronises with the host)."""
    cnt = int(count.item())
    g = torch.zeros(T, n_pad, dtype=dlc.dtype, device=dlc.device)
    if cnt:
        g[rows[:cnt].long()] = dlc[:cnt]
    return g


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _rows2d(x: Tensor) -> Tensor:
    """[..., C] -> [rows, C] view with unit column stride (copies only if needed)."""
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < x2.shape[1]):
        x2 = x2.contiguous()
    return x2


# ----------------------------------------------------------------------------
# weight shadows: compute-dtype copies of the fp32 master parameters, packed so
# that projections sharing an input run as ONE GEMM.  Rebuilt when a parameter's
# version counter moves (i.e. after every optimizer step).
# ----------------------------------------------------------------------------
# Bumped whenever parameters may have changed WITHOUT their version counters moving: fused torch
# optimizers (torch.optim.AdamW(fused=True)) and raw kernels update parameters in place and leave
# ``_version`` alone.  The accelerator / trainer call ``parameters_changed()`` after such a step.
WEIGHTS_EPOCH = 0


def parameters_changed() -> None:
    """Mark every packed compute-dtype weight copy stale (rebuilt by the next forward)."""
    global WEIGHTS_EPOCH
    WEIGHTS_EPOCH += 1


# Moves whenever the CONTENT of a packed copy may have changed without ``PackedWeights.get`` rebuilding it:
# FusedAdamW refreshes the copies inside its update kernel.  The transposed copies (``get_t``) follow it.
SHADOW_GEN = 0


def shadows_refreshed() -> None:
    global SHADOW_GEN
    SHADOW_GEN += 1


# dx = dy @ W with a K-contiguous W^T copy (the same operand layout as the forward, which the four-wave
# GEMM takes) instead of the strided read of W; LDM_DX_WT=0 keeps the strided form (A/B measurements)
DX_TRANSPOSED_WEIGHTS = os.environ.get("LDM_DX_WT", "1") != "0"


class PackedWeights:
    def __init__(self, weights, biases, dtype: torch.dtype, name: str = "linear", pad: int = PAD_N):
        """``pad``: the packed copies hold the output columns padded (zero rows / zero bias) to this multiple
        of PAD_N; the vocabulary projection pads to 256 so that its products are whole 256-wide tiles."""
        if pad % PAD_N:
            raise ValueError(f"pad must be a multiple of {PAD_N}")
        self.name = name
        self.compact_rows = name == "head"   # its gradient comes from the masked CE: mostly zero rows
        self.weights = list(weights)
        self.biases = list(biases)
        self.dtype = dtype
        self.n_each = [int(w.shape[0]) for w in self.weights]
        self.n = sum(self.n_each)
        self.k = int(self.weights[0].shape[1])
        self.n_pad = _round_up(self.n, pad)
        self._key = None
        self.w = None
        self.b = None
        self._gen = 0        # rebuilds of w by get()
        self.wt = None       # transposed copy (get_t)
        self._t_key = None

    def get(self):
        key = (WEIGHTS_EPOCH,) + tuple((p._version, p.data_ptr()) for p in self.weights + self.biases)
        if key != self._key or self.w is None or self.w.device != self.weights[0].device:
            dev = self.weights[0].device
            if self.w is None or self.w.device != dev:
                self.w = torch.zeros(self.n_pad, self.k, dtype=self.dtype, device=dev)
                self.b = torch.zeros(self.n_pad, dtype=torch.float32, device=dev)
            off = 0
            for wt, bs, n in zip(self.weights, self.biases, self.n_each):
                wd = wt.detach()
                if self.dtype == torch.bfloat16:
                    if not wd.is_contiguous():
                        wd = wd.contiguous()
                    N.check(N.lib().ldm_cast_f32_to_bf16(wd.data_ptr(), self.w[off:off + n].data_ptr(),
                                                         wd.numel(), N.stream()), "ldm_cast_f32_to_bf16")
                else:
                    self.w[off:off + n].copy_(wd)
                self.b[off:off + n].copy_(bs.detach())
                off += n
            self._key = key
            self._gen += 1
        return self.w, self.b

    def get_t(self):
        """bf16 W^T [k, n_pad] (row stride n_pad) of the packed weights, for the input-gradient product
        dx[T, k] = dy[T, n_pad] @ W[n_pad, k]: with W^T both operands are read along the reduction, the layout
        of the forward GEMM.  Rebuilt (one transpose pass over the weights) when the packed copy changed.
        None for fp32 packs."""
        w, _ = self.get()
        if self.dtype != torch.bfloat16:
            return None
        key = (SHADOW_GEN, self._gen, w.data_ptr())
        if key != self._t_key or self.wt is None or self.wt.device != w.device:
            if self.wt is None or self.wt.device != w.device:
                self.wt = torch.empty(self.k, self.n_pad, dtype=self.dtype, device=w.device)
            N.check(N.lib().ldm_transpose_bf16(w.data_ptr(), self.k, self.wt.data_ptr(), self.n_pad, self.n_pad, self.k,
                                               N.stream()), "ldm_transpose_bf16")
            self._t_key = key
        return self.wt


class _ZeroPool:
    """Zero-filled fp32 buffers for the weight gradients of ONE backward pass, carved from a single
    allocation: one memset instead of one fill per projection.  ``begin()`` marks a new training
    step; the size asked for in the previous step is what the next one allocates up front.  A
    request that does not fit (first step, changed shapes, a second backward) gets its own
    ``torch.zeros``."""

    def __init__(self):
        self.want = 0
        self.used = 0
        self.buf = None
        self.off = 0

    def begin(self):
        self.want, self.used, self.buf, self.off = self.used, 0, None, 0

    def take(self, numel: int, device) -> Tensor:
        step = _round_up(numel, 64)
        self.used += step
        if self.want:
            if self.buf is None:
                self.buf = torch.zeros(self.want, dtype=torch.float32, device=device)
                self.off = 0
            if self.buf.device == device and self.off + numel <= self.buf.numel():
                v = self.buf[self.off:self.off + numel]
                self.off += step
                return v
        return torch.zeros(numel, dtype=torch.float32, device=device)


ZERO_POOL = _ZeroPool()


def begin_step():
    """Called once per training step, before the forward (model.forward / training_loss)."""
    ZERO_POOL.begin()


# Optional per-launch timing for the benchmark: when set to a dict, every instrumented launch gets a
# pair of events recorded around it on the launching stream (torch's current stream, the one the
# kernels are enqueued on); KERNEL_EVENTS[tag] collects (start, end, work, count) tuples, where
# ``work`` is the launch's algorithmic flops or bytes and ``count`` is None, or a device int32
# tensor that the work is to be multiplied by (row counts that exist only on the device).
KERNEL_EVENTS: dict | None = None


class _Timed:
    __slots__ = ("lst", "e0", "work", "count", "declined")

    def __init__(self, tag, work, count=None):
        self.lst = None
        self.declined = False        # set inside the block when the kernel declined the problem: no record
        if KERNEL_EVENTS is not None:
            self.lst = KERNEL_EVENTS.setdefault(tag, [])
            self.work, self.count = work, count

    def __enter__(self):
        if self.lst is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.lst is not None and not self.declined:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.lst.append((self.e0, e1, self.work, self.count))
        return False


def _gemm(in_dt, out_dt, ta, tb, m, n, k, a, lda, b, ldb, c, ldc, bias=None, res=None, ldr=0, alpha=1.0,
          tag="gemm.fwd"):
    with _Timed(tag, 2.0 * m * n * k):
        N.check(N.lib().ldm_gemm(N.dtype_code(in_dt), N.dtype_code(out_dt), ta, tb, m, n, k, alpha,
                                 a.data_ptr(), lda, b.data_ptr(), ldb, c.data_ptr(), ldc,
                                 N.ptr(bias), N.ptr(res), ldr, N.stream()), "ldm_gemm")


def _input_grad(pack, dt, T, k, n_pad, g, w):
    """dx[T, k] = g[T, n_pad] @ W[n_pad, k] (nn.Linear's input gradient)."""
    dx = torch.empty(T, k, dtype=dt, device=g.device)
    wt = pack.get_t() if (DX_TRANSPOSED_WEIGHTS and dt == torch.bfloat16 and T >= 256) else None
    if wt is not None:
        _gemm(dt, dt, 0, 1, T, k, n_pad, g, n_pad, wt, n_pad, dx, k, tag="gemm.dx")
    else:
        _gemm(dt, dt, 0, 0, T, k, n_pad, g, n_pad, w, k, dx, k, tag="gemm.dx")
    return dx


FUSED_RESIDUAL_DROPOUT = os.environ.get("LDM_FUSED_RES_DROPOUT", "1") != "0"   # residual dropout inside the projection's epilogue


def _draw_seed() -> int:
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def _project_residual(dt, T, n_pad, k, a, lda, w, y, b, res, drop):
    """y[T, n_pad] = res + dropout(a @ w^T + b): the `src + self.dropout(sublayer(...))` lines of the
    reference layer (src/transformer.py:352,361) with the sublayer's last projection.  ``drop`` = (p, seed)
    or None (plain residual add in the GEMM epilogue).  One GEMM where the kernel takes the shape
    (ldm_gemm_dropout_residual); otherwise the projection and ldm_dropout_add, same mask."""
    ldr = res.stride(0) if res is not None else 0
    if drop is None:
        _gemm(dt, dt, 0, 1, T, n_pad, k, a, lda, w, k, y, n_pad, bias=b, res=res, ldr=ldr, tag="gemm.fwd")
        return
    p, seed = drop
    rc = -3
    if FUSED_RESIDUAL_DROPOUT and dt == torch.bfloat16 and res is not None:
        with _Timed("gemm.fwd", 2.0 * T * n_pad * k) as tm:
            rc = N.lib().ldm_gemm_dropout_residual(T, n_pad, k, a.data_ptr(), lda, w.data_ptr(), k, N.ptr(b),
                                                   res.data_ptr(), ldr, y.data_ptr(), n_pad, float(p), int(seed), N.stream())
            tm.declined = rc in (-2, -3)
        if rc not in (-2, -3):
            N.check(rc, "ldm_gemm_dropout_residual")
    if rc in (-2, -3):
        _gemm(dt, dt, 0, 1, T, n_pad, k, a, lda, w, k, y, n_pad, bias=b, tag="gemm.fwd")
        rc_ = res if (res is None or (res.is_contiguous() and res.shape[1] == n_pad)) else None
        if res is not None and rc_ is None:     # a residual narrower than the padded row, or strided
            rc_ = torch.zeros(T, n_pad, dtype=dt, device=y.device)
            rc_[:, :res.shape[1]].copy_(res)
        N.check(N.lib().ldm_dropout_add(y.data_ptr(), N.ptr(rc_), y.data_ptr(), T, n_pad, N.dtype_code(dt), float(p),
                                        int(seed), N.stream()), "ldm_dropout_add")


FUSED_DROP_GRAD = os.environ.get("LDM_FUSED_DROP_GRAD", "1") != "0"   # masked gradient written by the norm's backward


class _Drop:
    """One residual dropout: (p, seed), plus the hand-off of its backward.  The tensor res + dropout(y) is
    normally read by exactly one RMSNorm; that norm's backward kernel produces the total gradient dx of the
    tensor and can write keep * dx / (1 - p) beside it (ldm_rmsnorm_bwd_dropgrad), which is the gradient
    the projection's backward needs.  The norm leaves both here; the projection takes the masked copy only
    if the gradient autograd hands it IS that dx (same storage, unmodified), and masks it itself otherwise.
    Holding ``dx`` keeps its address from being reused and keeps autograd from accumulating into it."""
    __slots__ = ("p", "seed", "dx", "dx_drop", "ver")

    def __init__(self, p: float, seed: int):
        self.p, self.seed = p, seed
        self.dx = self.dx_drop = self.ver = None

    def __iter__(self):
        return iter((self.p, self.seed))

    def take(self, gy, T: int, n: int):
        dx, dxd, ver = self.dx, self.dx_drop, self.ver
        self.dx = self.dx_drop = self.ver = None
        if (dx is not None and gy.data_ptr() == dx.data_ptr() and gy.dtype == dx.dtype and gy.is_contiguous()
                and gy.numel() == dx.numel() == T * n and dx._version == ver):
            return dxd.view(T, n)
        return None


def _new_drop(residual, residual_dropout: float):
    if residual is None or not residual_dropout > 0.0:
        return None
    return _Drop(float(residual_dropout), _draw_seed())


def _tag_drop(out, drop):
    if drop is not None and FUSED_DROP_GRAD:
        out._ldm_drop = drop     # read by the RMSNorm that is handed THIS tensor object
    return out


def _drop_grad(g, drop, gy=None):
    """Gradient w.r.t. the projection output under the residual dropout: keep * g / (1 - p), same mask.
    ``gy``: the gradient as autograd delivered it (see _Drop)."""
    if drop is None:
        return g
    if gy is not None:
        handed = drop.take(gy, g.shape[0], g.shape[1])
        if handed is not None:
            return handed
    p, seed = drop
    gc_ = g if g.is_contiguous() else g.contiguous()
    out = torch.empty_like(gc_)
    N.check(N.lib().ldm_dropout_add(gc_.data_ptr(), None, out.data_ptr(), gc_.shape[0], gc_.shape[1],
                                    N.dtype_code(gc_.dtype), float(p), int(seed), N.stream()), "ldm_dropout_add(backward)")
    return out


def _weight_bias_grads(pack, g, x_in, ld_x, T, k, dev, dt):
    """dW [n_pad, k] and db [n_pad] of a projection from its output gradient g [T, n_pad] and its input
    (one zeroed buffer; for bf16 one kernel produces both), split per packed parameter."""
    n_pad = pack.n_pad
    dwb = ZERO_POOL.take(n_pad * k + n_pad, dev)
    dw, db = dwb[:n_pad * k].view(n_pad, k), dwb[n_pad * k:]
    rc = -2
    if dt == torch.bfloat16 and FUSED_BIAS_GRAD:
        with _Timed("gemm.dw", 2.0 * n_pad * k * T) as tm:
            rc = N.lib().ldm_gemm_acc_rowsum(1, 0, n_pad, k, T, 1.0, g.data_ptr(), n_pad, x_in.data_ptr(), ld_x,
                                             dw.data_ptr(), k, db.data_ptr(), N.stream())
            tm.declined = rc in (-2, -3)
        if rc not in (-2, -3):
            N.check(rc, "ldm_gemm_acc_rowsum")
    if rc in (-2, -3):
        with _Timed("gemm.dw", 2.0 * n_pad * k * T):
            N.check(N.lib().ldm_gemm_acc(N.dtype_code(dt), 1, 0, n_pad, k, T, 1.0, g.data_ptr(), n_pad,
                                         x_in.data_ptr(), ld_x, dw.data_ptr(), k, N.stream()), "ldm_gemm_acc")
        ws = torch.empty(N.lib().ldm_colsum_ws_floats(T, n_pad), dtype=torch.float32, device=dev)
        N.check(N.lib().ldm_colsum(g.data_ptr(), N.dtype_code(dt), T, n_pad, n_pad, db.data_ptr(),
                                   ws.data_ptr(), N.stream()), "ldm_colsum")
    grads_w, grads_b = [], []
    off = 0
    for ne in pack.n_each:
        grads_w.append(dw[off:off + ne])
        grads_b.append(db[off:off + ne])
        off += ne
    return grads_w, grads_b


FUSED_QKV_ROPE = os.environ.get("LDM_FUSED_QKV_ROPE", "1") != "0"   # rotary embedding inside the q/k/v GEMM epilogue


class RopeSpec:
    """What the q/k/v projection needs to apply the rotary embedding in its epilogue: fp32 tables
    cos/sin [S, hd/2], the 16-position angle-addition step [2, hd/2], sequence length, head dim and
    the query scale.  ``applied`` is set by the projection: True when its output holds the ROTATED,
    scaled q and k (the attention call must then not rotate again)."""
    __slots__ = ("cos", "sin", "step", "seq", "hd", "q_scale", "applied")

    def __init__(self, cos, sin, step, seq, hd, q_scale):
        self.cos, self.sin, self.step, self.seq, self.hd, self.q_scale = cos, sin, step, seq, hd, q_scale
        self.applied = False


class _PackedLinear(torch.autograd.Function):
    """y[T, n] = x[T, k] @ Wpacked^T + b (+ residual);  grads for every packed weight/bias.

    ``link`` (a _HeadLink or None) is filled in for the vocabulary projection: see _HeadLink.
    ``rope`` (a RopeSpec or None): the pack is [Wq; Wk; Wv] and the rotary embedding + query scale may
    be applied in the GEMM epilogue (ldm_gemm_qkv_rope).  The backward is unchanged by that: the
    attention backward hands back the gradient w.r.t. the projection output BEFORE the rotation."""

    @staticmethod
    def forward(ctx, x2, residual, pack: PackedWeights, link, rope, drop, *params):
        N.require_device(x2)
        w, b = pack.get()
        T, k = x2.shape
        dt = pack.dtype
        if x2.dtype != dt:
            x2 = x2.to(dt)
        y = torch.empty(T, pack.n_pad, dtype=dt, device=x2.device)
        res = None
        if residual is not None:
            res = _rows2d(residual)
            if res.dtype != dt:
                res = res.to(dt)
        rc = -3
        if (rope is not None and FUSED_QKV_ROPE and dt == torch.bfloat16 and res is None and pack.n_pad == pack.n
                and len(pack.n_each) == 3 and pack.n_each[0] == pack.n_each[1] == pack.n_each[2]):
            with _Timed("gemm.fwd", 2.0 * T * pack.n_pad * k) as tm:
                rc = N.lib().ldm_gemm_qkv_rope(T, pack.n_each[0], k, rope.hd, rope.seq, x2.data_ptr(), x2.stride(0),
                                               w.data_ptr(), k, N.ptr(b), rope.cos.data_ptr(), rope.sin.data_ptr(),
                                               rope.step.data_ptr(), float(rope.q_scale), y.data_ptr(), pack.n_pad,
                                               N.stream())
                tm.declined = rc in (-2, -3)
            if rc not in (-2, -3):
                N.check(rc, "ldm_gemm_qkv_rope")
                rope.applied = True
        if rc in (-2, -3):
            _project_residual(dt, T, pack.n_pad, k, x2, x2.stride(0), w, y, b, res, drop)
        ctx.drop = drop
        ctx.pack = pack
        ctx.res_shape = tuple(residual.shape) if residual is not None else None
        ctx.save_for_backward(x2, w)
        ctx.link = link
        if link is not None:
            link.y_ptr, link.ld = y.data_ptr(), pack.n_pad
            link.compact_ok = bool(
                COMPACT_HEAD and pack.compact_rows and dt == torch.bfloat16 and T >= 1024 and k >= 128
                and k % 8 == 0 and x2.stride(0) % 4 == 0 and (pack.n_pad // 256) * ((k + 255) // 256) >= 200
                )
        return y[:, :pack.n] if pack.n != pack.n_pad else y

    @staticmethod
    def backward(ctx, gy):
        x2, w = ctx.saved_tensors
        pack = ctx.pack
        dt = pack.dtype
        T, k = x2.shape
        n, n_pad = pack.n, pack.n_pad
        link = ctx.link
        entry = None
        if link is not None:
            entry, link.grad = link.grad, None
        if entry is not None:
            dlc, rows, count, dummy = entry
            if gy.data_ptr() == dummy.data_ptr() and all(st == 0 for st in gy.stride()):
                out = _PackedLinear._backward_compact(ctx, x2, w, dlc, rows, count)
                if out is not None:
                    return out
                gy = _expand_compact(dlc, rows, count, T, n_pad)[:, :n]
            else:   # other gradients were added to the logits' gradient: go dense
                gy = gy + _expand_compact(dlc, rows, count, T, n_pad)[:, :n].view_as(gy)
        # gradient with the padded row stride and zero padding columns
        row_ok = gy.dtype == dt and gy.stride(-1) == 1 and (T == 1 or gy.stride(0) == n_pad)
        zp = None
        if link is not None:
            zp, link.zero_padded = link.zero_padded, None
        if row_ok and (n_pad == n or zp == (gy.data_ptr(), n_pad)):
            g = gy
        else:
            g = torch.zeros(T, n_pad, dtype=dt, device=gy.device)
            g[:, :n].copy_(gy)
        g = _drop_grad(g, ctx.drop, gy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _input_grad(pack, dt, T, k, n_pad, g, w)
        grads_w, grads_b = _weight_bias_grads(pack, g, x2, x2.stride(0), T, k, g.device, dt)
        dres = gy.reshape(ctx.res_shape) if ctx.res_shape is not None else None
        return (dx, dres, None, None, None, None, *grads_w, *grads_b)


def _compact_input_grad(pack, gc, rows, count, T, k, w):
    """dx [T, k] of a projection from COMPACT output-gradient rows: row rows[i] of dx = gc[i] @ W for
    i < *count, every other row zero.  With the transposed weight copy the product is computed compact
    (both operands read along the reduction, the four-wave kernel's form) and its rows are then moved to
    their places by one scatter pass; otherwise the GEMM's mapped-row form writes them directly."""
    lib = N.lib()
    st = N.stream()
    n_pad = pack.n_pad
    dev = gc.device
    dx = torch.zeros(T, k, dtype=torch.bfloat16, device=dev)
    rows_alloc = gc.shape[0]
    wt = pack.get_t() if (DX_TRANSPOSED_WEIGHTS and rows_alloc % 256 == 0 and k % 256 == 0 and n_pad % 64 == 0) else None
    if wt is not None:
        dxc = torch.empty(rows_alloc, k, dtype=torch.bfloat16, device=dev)
        with _Timed("gemm.dx", 2.0 * k * n_pad, count) as tm:
            rc = lib.ldm_gemm_rows_dyn(N.BF16, 1, rows_alloc, k, n_pad, gc.data_ptr(), n_pad, wt.data_ptr(), n_pad,
                                       dxc.data_ptr(), k, None, count.data_ptr(), None, st)
            tm.declined = rc in (-2, -3)
        if rc not in (-2, -3):
            N.check(rc, "ldm_gemm_rows_dyn")
            N.check(lib.ldm_scatter_rows(dxc.data_ptr(), N.BF16, k, rows.data_ptr(), count.data_ptr(), rows_alloc, T, k,
                                         dx.data_ptr(), k, st), "ldm_scatter_rows")
            return dx
    with _Timed("gemm.dx", 2.0 * k * n_pad, count) as tm:
        rc = lib.ldm_gemm_rows_dyn(N.BF16, 0, T, k, n_pad, gc.data_ptr(), n_pad, w.data_ptr(), k, dx.data_ptr(), k,
                                   None, count.data_ptr(), rows.data_ptr(), st)
        tm.declined = rc in (-2, -3)
    if rc in (-2, -3):
        return None
    N.check(rc, "ldm_gemm_rows_dyn")
    return dx


def _backward_compact(ctx, x2, w, dlc, rows, count):
    """dx, dW, db of a projection from the compact gradient rows (None if the kernels decline)."""
    pack = ctx.pack
    lib = N.lib()
    T, k = x2.shape
    n_pad = pack.n_pad
    Tp = _round_up(T, 64)
    dev = x2.device
    st = N.stream()
    dx = None
    if ctx.needs_input_grad[0]:
        dx = _compact_input_grad(pack, dlc, rows, count, T, k, w)
        if dx is None:
            return None
    xc = torch.empty(Tp, k, dtype=torch.bfloat16, device=dev)
    N.check(lib.ldm_gather_rows(x2.data_ptr(), N.BF16, x2.stride(0), rows.data_ptr(), count.data_ptr(), T, k,
                                xc.data_ptr(), st), "ldm_gather_rows")
    dwb = ZERO_POOL.take(n_pad * k + n_pad, dev)
    dw, db = dwb[:n_pad * k].view(n_pad, k), dwb[n_pad * k:]
    with _Timed("gemm.dw", 2.0 * n_pad * k, count) as tm:
        rc = lib.ldm_gemm_acc_kdyn(1, 0, n_pad, k, Tp, dlc.data_ptr(), n_pad, xc.data_ptr(), k, dw.data_ptr(), k,
                                   count.data_ptr(), db.data_ptr() if FUSED_BIAS_GRAD else None, st)
        tm.declined = rc in (-2, -3)
    if rc in (-2, -3):
        return None
    N.check(rc, "ldm_gemm_acc_kdyn")
    if not FUSED_BIAS_GRAD:
        ws = torch.empty(lib.ldm_colsum_ws_floats(T, n_pad), dtype=torch.float32, device=dev)
        N.check(lib.ldm_colsum_rows_dyn(dlc.data_ptr(), N.BF16, T, n_pad, n_pad, db.data_ptr(), ws.data_ptr(),
                                        count.data_ptr(), st), "ldm_colsum_rows_dyn")
    grads_w, grads_b = [], []
    off = 0
    for ne in pack.n_each:
        grads_w.append(dw[off:off + ne])
        grads_b.append(db[off:off + ne])
        off += ne
    COMPACT_STATS["used"] += 1
    return (dx, None, None, None, None, None, *grads_w, *grads_b)


_PackedLinear._backward_compact = staticmethod(_backward_compact)


def packed_linear(x: Tensor, pack: PackedWeights, residual: Tensor | None = None, rope: RopeSpec | None = None,
                  residual_dropout: float = 0.0) -> Tensor:
    """``residual_dropout`` p > 0 (with a residual): residual + dropout(projection, p), training semantics."""
    shape = x.shape
    x2 = _rows2d(x)
    link = _HeadLink() if pack.compact_rows else None
    drop = _new_drop(residual, residual_dropout)
    y = _PackedLinear.apply(x2, residual, pack, link, rope, drop, *pack.weights, *pack.biases)
    out = y.reshape(*shape[:-1], pack.n) if y.is_contiguous() else y.view(*shape[:-1], pack.n)
    _tag_drop(out, drop)
    if link is not None:
        out._ldm_head_link = link    # read by masked_ce when THIS tensor object is handed to the loss
    return out


# ----------------------------------------------------------------------------
class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, table, out_dtype, hot_id):
        N.require_device(ids, table)
        ids_c = ids.contiguous()
        T = ids_c.numel()
        V, d = table.shape
        out = torch.empty(*ids.shape, d, dtype=out_dtype, device=table.device)
        tb = table.detach()
        if not tb.is_contiguous():
            tb = tb.contiguous()
        N.check(N.lib().ldm_embed_fwd(ids_c.data_ptr(), tb.data_ptr(), T, d, V, out.data_ptr(),
                                      N.dtype_code(out_dtype), N.stream()), "ldm_embed_fwd")
        ctx.save_for_backward(ids_c)
        ctx.shape = (V, d)
        ctx.hot_id = hot_id
        return out

    @staticmethod
    def backward(ctx, g):
        (ids_c,) = ctx.saved_tensors
        V, d = ctx.shape
        g = g.contiguous()
        dtab = ZERO_POOL.take(V * d, g.device).view(V, d)
        N.check(N.lib().ldm_embed_bwd(ids_c.data_ptr(), g.data_ptr(), N.dtype_code(g.dtype), ids_c.numel(), d, V,
                                      ctx.hot_id, dtab.data_ptr(), N.stream()), "ldm_embed_bwd")
        return None, dtab, None, None


def embedding(ids: Tensor, table: Tensor, out_dtype: torch.dtype, hot_id: int = 103) -> Tensor:
    return _Embedding.apply(ids, table, out_dtype, hot_id)


# ----------------------------------------------------------------------------
def _rmsnorm_forward(xc, w, eps, out_dtype):
    """(y, rstd, mixed): y in ``out_dtype`` (None: x's).  mixed: fp32 x -> bf16 y out of one kernel (the norm
    between an fp32 residual stream and a bf16 projection); other combinations are the norm in x's dtype
    followed by a cast."""
    d = xc.shape[-1]
    T = xc.numel() // d
    lib = N.lib()
    rstd = torch.empty(T, dtype=torch.float32, device=xc.device)
    if out_dtype is not None and out_dtype != xc.dtype and xc.dtype == torch.float32 and out_dtype == torch.bfloat16:
        y = torch.empty(xc.shape, dtype=torch.bfloat16, device=xc.device)
        with _Timed("rmsnorm.fwd", 6.0 * T * d) as tm:
            rc = lib.ldm_rmsnorm_fwd_f32_bf16(xc.data_ptr(), w.data_ptr(), T, d, eps, y.data_ptr(), rstd.data_ptr(), N.stream())
            tm.declined = rc in (-2, -3)
        if rc not in (-2, -3):
            N.check(rc, "ldm_rmsnorm_fwd_f32_bf16")
            return y, rstd, True
    y = torch.empty_like(xc)
    with _Timed("rmsnorm.fwd", 2.0 * T * d * xc.element_size()):
        N.check(lib.ldm_rmsnorm_fwd(xc.data_ptr(), w.data_ptr(), T, d, eps, N.dtype_code(xc.dtype), y.data_ptr(),
                                    rstd.data_ptr(), N.stream()), "ldm_rmsnorm_fwd")
    if out_dtype is not None and out_dtype != y.dtype:
        y = y.to(out_dtype)
    return y, rstd, False


def _rmsnorm_backward(xc, w, rstd, g, r, drop, mixed=False):
    """dx (+ r), dw of the norm; with ``drop`` (a _Drop) also the masked copy of dx, left in ``drop``.
    ``mixed``: the forward was the fp32 -> bf16 kernel; g is then taken in bf16, x / r / dx are fp32."""
    d = xc.shape[-1]
    T = xc.numel() // d
    es = xc.element_size()
    lib = N.lib()
    dx = torch.empty_like(xc)
    dw = torch.empty(d, dtype=torch.float32, device=xc.device)
    ws = torch.empty(lib.ldm_rmsnorm_bwd_ws_floats(T, d), dtype=torch.float32, device=xc.device)
    nbuf = 4.0 if r is not None else 3.0
    if mixed:
        if g.dtype != torch.bfloat16:
            g = g.to(torch.bfloat16)
        with _Timed("rmsnorm.bwd", ((nbuf - 1.0) * 4.0 + 2.0) * T * d):
            N.check(lib.ldm_rmsnorm_bwd_f32_bf16(xc.data_ptr(), w.data_ptr(), rstd.data_ptr(), g.data_ptr(), N.ptr(r), T, d,
                                                 dx.data_ptr(), dw.data_ptr(), ws.data_ptr(), N.stream()),
                    "ldm_rmsnorm_bwd_f32_bf16")
        return dx, dw
    if g.dtype != xc.dtype:
        g = g.to(xc.dtype)
    if drop is not None and FUSED_DROP_GRAD and T > 0:
        dxd = torch.empty_like(xc)
        with _Timed("rmsnorm.bwd", (nbuf + 1.0) * T * d * es) as tm:
            rc = lib.ldm_rmsnorm_bwd_dropgrad(xc.data_ptr(), w.data_ptr(), rstd.data_ptr(), g.data_ptr(), N.ptr(r), T, d,
                                              N.dtype_code(xc.dtype), dx.data_ptr(), dxd.data_ptr(), float(drop.p),
                                              int(drop.seed), dw.data_ptr(), ws.data_ptr(), N.stream())
            tm.declined = rc in (-2, -3)
        if rc not in (-2, -3):
            N.check(rc, "ldm_rmsnorm_bwd_dropgrad")
            drop.dx, drop.dx_drop, drop.ver = dx, dxd, dx._version
            return dx, dw
    with _Timed("rmsnorm.bwd", nbuf * T * d * es):
        N.check(lib.ldm_rmsnorm_bwd(xc.data_ptr(), w.data_ptr(), rstd.data_ptr(), g.data_ptr(), N.ptr(r), T, d,
                                    N.dtype_code(xc.dtype), dx.data_ptr(), dw.data_ptr(), ws.data_ptr(), N.stream()),
                "ldm_rmsnorm_bwd")
    return dx, dw


def _input_drop(x):
    return getattr(x, "_ldm_drop", None) if torch.is_grad_enabled() else None


class _RMSNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, eps, drop=None, out_dtype=None):
        N.require_device(x, weight)
        xc = x.contiguous()
        w = weight.detach().float().contiguous()
        y, rstd, ctx.mixed = _rmsnorm_forward(xc, w, eps, out_dtype)
        ctx.save_for_backward(xc, w, rstd)
        ctx.drop = drop if xc.data_ptr() == x.data_ptr() else None
        return y

    @staticmethod
    def backward(ctx, gy):
        xc, w, rstd = ctx.saved_tensors
        dx, dw = _rmsnorm_backward(xc, w, rstd, gy.contiguous(), None, ctx.drop, ctx.mixed)
        return dx, dw, None, None, None


def rms_norm(x: Tensor, weight: Tensor, eps: float, out_dtype: torch.dtype | None = None) -> Tensor:
    """``out_dtype``: dtype of the result (default x's); fp32 x with a bf16 result is one kernel."""
    return _RMSNorm.apply(x, weight, eps, _input_drop(x), out_dtype)


class _RMSNormRes(torch.autograd.Function):
    """(y, x) = (rms_norm(x), x): the second output carries the residual stream, so that the
    gradient arriving through the residual connection is added inside the norm's backward
    kernel instead of by a separate elementwise pass."""

    @staticmethod
    def forward(ctx, x, weight, eps, drop=None, out_dtype=None):
        N.require_device(x, weight)
        xc = x.contiguous()
        w = weight.detach().float().contiguous()
        y, rstd, ctx.mixed = _rmsnorm_forward(xc, w, eps, out_dtype)
        ctx.save_for_backward(xc, w, rstd)
        ctx.drop = drop if xc.data_ptr() == x.data_ptr() else None
        return y, xc.view_as(xc)

    @staticmethod
    def backward(ctx, gy, gres):
        xc, w, rstd = ctx.saved_tensors
        if gy is None:
            return gres, None, None, None, None
        g = gy.contiguous()
        r = None
        if gres is not None:
            r = gres.contiguous()
            if r.dtype != xc.dtype:
                r = r.to(xc.dtype)
        dx, dw = _rmsnorm_backward(xc, w, rstd, g, r, ctx.drop, ctx.mixed)
        return dx, dw, None, None, None


def rms_norm_with_residual(x: Tensor, weight: Tensor, eps: float,
                           out_dtype: torch.dtype | None = None) -> tuple[Tensor, Tensor]:
    """(rms_norm(x), x); use the second value as the residual-connection operand."""
    return _RMSNormRes.apply(x, weight, eps, _input_drop(x), out_dtype)


# ----------------------------------------------------------------------------
class _SwiGLUAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ag):
        N.require_device(ag)
        ag = ag.contiguous()
        F2 = ag.shape[-1]
        F = F2 // 2
        T = ag.numel() // F2
        h = torch.empty(*ag.shape[:-1], F, dtype=ag.dtype, device=ag.device)
        N.check(N.lib().ldm_swiglu_fwd(ag.data_ptr(), T, F, N.dtype_code(ag.dtype), h.data_ptr(), N.stream()),
                "ldm_swiglu_fwd")
        ctx.save_for_backward(ag)
        return h

    @staticmethod
    def backward(ctx, gh):
        (ag,) = ctx.saved_tensors
        F2 = ag.shape[-1]
        F = F2 // 2
        T = ag.numel() // F2
        g = gh.contiguous()
        if g.dtype != ag.dtype:
            g = g.to(ag.dtype)
        dag = torch.empty_like(ag)
        N.check(N.lib().ldm_swiglu_bwd(ag.data_ptr(), g.data_ptr(), T, F, N.dtype_code(ag.dtype), dag.data_ptr(),
                                       N.stream()), "ldm_swiglu_bwd")
        return dag


def swiglu_act(ag: Tensor) -> Tensor:
    """h = a * silu(g) for the fused activation ag = [a | g] (w1 branch | w2 branch)."""
    return _SwiGLUAct.apply(ag)


FUSED_SWIGLU_DOWN = os.environ.get("LDM_FUSED_SWIGLU", "1") != "0"   # SwiGLU backward inside the w3 input-gradient GEMM
# SwiGLU forward inside the gate/up projection GEMM (LDM_FUSED_SWIGLU_UP=0: separate elementwise pass)
FUSED_SWIGLU_UP = os.environ.get("LDM_FUSED_SWIGLU_UP", "1") != "0"


class _SwiGLUUp(torch.autograd.Function):
    """(ag, h) = ([w1(x) | w2(x)], w1(x) * silu(w2(x))) from ONE GEMM over the packed gate/up weights
    (src/transformer.py:91-93).  ``h`` is an auxiliary, non-differentiable output: the down projection
    takes it as a ready-made operand and differentiates through ``ag`` (its backward re-derives
    everything from ag), so the gradient of this node arrives through ``ag`` alone."""

    @staticmethod
    def forward(ctx, x2, pack: PackedWeights, *params):
        N.require_device(x2)
        w, b = pack.get()
        T, k = x2.shape
        dt = pack.dtype
        if x2.dtype != dt:
            x2 = x2.to(dt)
        F = pack.n // 2
        y = torch.empty(T, pack.n_pad, dtype=dt, device=x2.device)
        h = torch.empty(T, F, dtype=dt, device=x2.device)
        rc = -3
        if FUSED_SWIGLU_UP and dt == torch.bfloat16 and pack.n_pad == pack.n:
            with _Timed("gemm.fwd", 2.0 * T * pack.n_pad * k) as tm:
                rc = N.lib().ldm_gemm_swiglu_fwd(T, F, k, x2.data_ptr(), x2.stride(0), w.data_ptr(), k, N.ptr(b),
                                                 y.data_ptr(), pack.n_pad, h.data_ptr(), F, N.stream())
                tm.declined = rc in (-2, -3)
            if rc not in (-2, -3):
                N.check(rc, "ldm_gemm_swiglu_fwd")
        if rc in (-2, -3):
            _gemm(dt, dt, 0, 1, T, pack.n_pad, k, x2, x2.stride(0), w, k, y, pack.n_pad, bias=b, tag="gemm.fwd")
            ag = y if pack.n_pad == pack.n else y[:, :pack.n].contiguous()
            N.check(N.lib().ldm_swiglu_fwd(ag.data_ptr(), T, F, N.dtype_code(dt), h.data_ptr(), N.stream()),
                    "ldm_swiglu_fwd")
        ctx.pack = pack
        ctx.save_for_backward(x2, w)
        ctx.mark_non_differentiable(h)
        # without this autograd hands backward a zero tensor of h's shape for the non-differentiable output:
        # a [T, F] fill per layer and step
        ctx.set_materialize_grads(False)
        return (y[:, :pack.n] if pack.n != pack.n_pad else y), h

    @staticmethod
    def backward(ctx, gy, _gh):
        x2, w = ctx.saved_tensors
        pack = ctx.pack
        if gy is None:
            return (None, None) + (None,) * (2 * len(pack.weights))
        dt = pack.dtype
        T, k = x2.shape
        n, n_pad = pack.n, pack.n_pad
        if gy.dtype == dt and gy.stride(-1) == 1 and n_pad == n and (T == 1 or gy.stride(0) == n_pad):
            g = gy
        else:
            g = torch.zeros(T, n_pad, dtype=dt, device=gy.device)
            g[:, :n].copy_(gy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _input_grad(pack, dt, T, k, n_pad, g, w)
        grads_w, grads_b = _weight_bias_grads(pack, g, x2, x2.stride(0), T, k, g.device, dt)
        return (dx, None, *grads_w, *grads_b)


def swiglu_up(x: Tensor, pack: PackedWeights) -> tuple[Tensor, Tensor]:
    """([a | g], a * silu(g)) for the packed [w1; w2] projection of x; both [..., 2F] / [..., F]."""
    assert len(pack.weights) == 2 and pack.n_each[0] == pack.n_each[1]
    shape = x.shape
    ag, h = _SwiGLUUp.apply(_rows2d(x), pack, *pack.weights, *pack.biases)
    return ag.reshape(*shape[:-1], pack.n), h


class _SwiGLUDown(torch.autograd.Function):
    """y = (a * silu(g)) @ W3^T + b (+ residual) for ag = [a | g]: SwiGLU.forward after the two up
    projections (src/transformer.py:93-95).  ``h`` = a * silu(g) when the caller already has it
    (swiglu_up), else None.  Backward never writes the gradient of the hidden activation: dag comes
    out of the epilogue of the GEMM dy @ W3 (ldm_gemm_swiglu_bwd)."""

    @staticmethod
    def forward(ctx, ag, h, residual, pack: PackedWeights, drop, *params):
        N.require_device(ag)
        lib = N.lib()
        w, b = pack.get()
        dt = pack.dtype
        F2 = ag.shape[-1]
        F = F2 // 2
        ag2 = _rows2d(ag)
        if ag2.dtype != dt:
            ag2 = ag2.to(dt)
        if ag2.stride(0) != F2:
            ag2 = ag2.contiguous()
        T = ag2.shape[0]
        if h is None or tuple(h.shape) != (T, F) or h.dtype != dt or not h.is_contiguous():
            h = torch.empty(T, F, dtype=dt, device=ag2.device)
            N.check(lib.ldm_swiglu_fwd(ag2.data_ptr(), T, F, N.dtype_code(dt), h.data_ptr(), N.stream()), "ldm_swiglu_fwd")
        y = torch.empty(T, pack.n_pad, dtype=dt, device=ag2.device)
        res = None
        if residual is not None:
            res = _rows2d(residual)
            if res.dtype != dt:
                res = res.to(dt)
        _project_residual(dt, T, pack.n_pad, F, h, F, w, y, b, res, drop)
        ctx.drop = drop
        ctx.pack = pack
        ctx.res_shape = tuple(residual.shape) if residual is not None else None
        ctx.out_shape = tuple(ag.shape[:-1]) + (pack.n,)
        ctx.ag_shape = tuple(ag.shape)
        ctx.save_for_backward(ag2, h, w)
        y = y[:, :pack.n] if pack.n != pack.n_pad else y
        return y.reshape(ctx.out_shape) if y.is_contiguous() else y.view(ctx.out_shape)

    @staticmethod
    def backward(ctx, gy):
        ag2, h, w = ctx.saved_tensors
        pack = ctx.pack
        lib = N.lib()
        dt = pack.dtype
        T, F = h.shape
        n, n_pad = pack.n, pack.n_pad
        gy2 = gy.reshape(T, n)
        if gy2.dtype == dt and gy2.stride(-1) == 1 and n_pad == n and (T == 1 or gy2.stride(0) == n_pad):
            g = gy2
        else:
            g = torch.zeros(T, n_pad, dtype=dt, device=gy.device)
            g[:, :n].copy_(gy2)
        g = _drop_grad(g, ctx.drop, gy)
        dag = None
        if ctx.needs_input_grad[0]:
            dag = torch.empty_like(ag2)
            rc = -3
            if dt == torch.bfloat16 and FUSED_SWIGLU_DOWN:
                with _Timed("gemm.dx", 2.0 * T * F * n_pad) as tm:
                    rc = lib.ldm_gemm_swiglu_bwd(T, F, n_pad, g.data_ptr(), n_pad, w.data_ptr(), F, ag2.data_ptr(), 2 * F,
                                                 dag.data_ptr(), 2 * F, N.stream())
                    tm.declined = rc in (-2, -3)
                if rc not in (-2, -3):
                    N.check(rc, "ldm_gemm_swiglu_bwd")
            if rc in (-2, -3):
                dh = torch.empty(T, F, dtype=dt, device=g.device)
                _gemm(dt, dt, 0, 0, T, F, n_pad, g, n_pad, w, F, dh, F, tag="gemm.dx")
                N.check(lib.ldm_swiglu_bwd(ag2.data_ptr(), dh.data_ptr(), T, F, N.dtype_code(dt), dag.data_ptr(), N.stream()),
                        "ldm_swiglu_bwd")
            dag = dag.view(ctx.ag_shape)
        grads_w, grads_b = _weight_bias_grads(pack, g, h, F, T, F, g.device, dt)
        dres = gy.reshape(ctx.res_shape) if ctx.res_shape is not None else None
        return (dag, None, dres, None, None, grads_w[0], grads_b[0])


def swiglu_down(ag: Tensor, pack: PackedWeights, residual: Tensor | None = None, h: Tensor | None = None,
                residual_dropout: float = 0.0) -> Tensor:
    """w3(a * silu(g)) (+ residual) for the fused activation ag = [a | g]; one weight in ``pack``.
    ``h``: a * silu(g) if the caller has it (the second output of swiglu_up).  ``residual_dropout`` p > 0
    (with a residual): residual + dropout(w3(...), p)."""
    assert len(pack.weights) == 1
    drop = _new_drop(residual, residual_dropout)
    return _tag_drop(_SwiGLUDown.apply(ag, h, residual, pack, drop, pack.weights[0], pack.biases[0]), drop)


# ----------------------------------------------------------------------------
class _DropoutAdd(torch.autograd.Function):
    """res + dropout(y) in one pass; the mask is a hash of (seed, row, column), regenerated in
    backward (src/transformer.py:352,361 of the reference: ``src + self.dropout(...)``)."""

    @staticmethod
    def forward(ctx, y, res, p, seed):
        N.require_device(y, res)
        yc = y.contiguous()
        rc = res.contiguous()
        if rc.dtype != yc.dtype:
            rc = rc.to(yc.dtype)
        d = yc.shape[-1]
        out = torch.empty_like(yc)
        N.check(N.lib().ldm_dropout_add(yc.data_ptr(), rc.data_ptr(), out.data_ptr(), yc.numel() // d, d,
                                        N.dtype_code(yc.dtype), float(p), int(seed), N.stream()), "ldm_dropout_add")
        ctx.args = (float(p), int(seed), res.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        p, seed, res_dtype = ctx.args
        gc_ = g.contiguous()
        d = gc_.shape[-1]
        gy = torch.empty_like(gc_)
        N.check(N.lib().ldm_dropout_add(gc_.data_ptr(), None, gy.data_ptr(), gc_.numel() // d, d,
                                        N.dtype_code(gc_.dtype), p, seed, N.stream()), "ldm_dropout_add(backward)")
        return gy, (g if g.dtype == res_dtype else g.to(res_dtype)), None, None


def dropout_add(y: Tensor, residual: Tensor, p: float, seed: int | None = None) -> Tensor:
    """``residual + dropout(y, p)`` (training semantics); ``seed`` defaults to a draw from torch's
    CPU generator."""
    if p <= 0.0:
        return residual + y
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    return _DropoutAdd.apply(y, residual, float(p), int(seed))


# ----------------------------------------------------------------------------
class _RopeAttention(torch.autograd.Function):
    """RoPE(q,k), q/sqrt(hd), dropout(softmax(q k^T + key mask)) v on a fused qkv activation.

    ``dropout_p`` > 0 drops attention probabilities (src/transformer.py:278 of the reference) with a
    mask the kernels regenerate from ``seed`` and the element coordinates; nothing is stored."""

    @staticmethod
    def forward(ctx, qkv, key_valid, B, S, H, hd, cos, sin, dropout_p=0.0, seed=0, pre_rotated=False, add_mask=None):
        N.require_device(qkv)
        lib = N.lib()
        dt = qkv.dtype
        code = N.dtype_code(dt)
        D = H * hd
        qkv = qkv.contiguous() if qkv.stride(-1) != 1 else qkv
        q2 = qkv.reshape(B * S, 3 * D)
        if q2.data_ptr() != qkv.data_ptr() or q2.stride(0) != 3 * D:
            q2 = q2.contiguous()
        # the rotation is applied in place; the pre-rotation values are not needed again
        # (the projection's backward uses only its own input and weights)
        # (pre_rotated: the projection's epilogue has already rotated and scaled q and k)
        if not pre_rotated:
            with _Timed("rope", 2.0 * B * S * 2 * D * q2.element_size()):
                N.check(lib.ldm_rope_qk(q2.data_ptr(), code, B, S, H, hd, 3 * D, cos.data_ptr(), sin.data_ptr(),
                                        1.0 / math.sqrt(hd), 0, N.stream()), "ldm_rope_qk")
        o = torch.empty(B * S, D, dtype=dt, device=qkv.device)
        saved = torch.empty(lib.ldm_attn_saved_bytes(code, B, H, S, hd), dtype=torch.uint8, device=qkv.device)
        if add_mask is not None:
            # scores + attn_mask (src/transformer.py:267-268): on the path that materialises the scores
            ws = (torch.empty(lib.ldm_attn_bwd_ws_bytes(code, B, H, S, hd), dtype=torch.uint8, device=qkv.device)
                  if dropout_p > 0.0 else None)
            sb, sh, sr, _ = add_mask.stride()
            with _Timed("attn.fwd", 4.0 * B * H * S * S * hd):
                N.check(lib.ldm_attn_fwd_addmask(code, q2.data_ptr(), 3 * D, N.ptr(key_valid), add_mask.data_ptr(), sb, sh,
                                                 sr, B, H, S, hd, o.data_ptr(), D, saved.data_ptr(), N.ptr(ws),
                                                 float(dropout_p), int(seed), N.stream()), "ldm_attn_fwd_addmask")
        elif dropout_p > 0.0:
            # the fp32 path materialises the probabilities: it needs room for their dropped copy
            ws = (torch.empty(lib.ldm_attn_bwd_ws_bytes(code, B, H, S, hd), dtype=torch.uint8, device=qkv.device)
                  if dt == torch.float32 else None)
            with _Timed("attn.fwd", 4.0 * B * H * S * S * hd):
                N.check(lib.ldm_attn_fwd_dropout(code, q2.data_ptr(), 3 * D, N.ptr(key_valid), B, H, S, hd, o.data_ptr(), D,
                                                 saved.data_ptr(), N.ptr(ws), float(dropout_p), int(seed), N.stream()),
                        "ldm_attn_fwd_dropout")
        else:
            with _Timed("attn.fwd", 4.0 * B * H * S * S * hd):
                N.check(lib.ldm_attn_fwd(code, q2.data_ptr(), 3 * D, N.ptr(key_valid), B, H, S, hd, o.data_ptr(), D,
                                         saved.data_ptr(), N.stream()), "ldm_attn_fwd")
        ctx.save_for_backward(q2, o, saved, cos, sin)
        ctx.key_valid = key_valid
        ctx.dropout = (float(dropout_p), int(seed))
        ctx.dims = (B, S, H, hd)
        return o.view(B, S, D)

    @staticmethod
    def backward(ctx, go):
        q2, o, saved, cos, sin = ctx.saved_tensors
        B, S, H, hd = ctx.dims
        D = H * hd
        lib = N.lib()
        dt = q2.dtype
        code = N.dtype_code(dt)
        g = go.reshape(B * S, D)
        if g.dtype != dt:
            g = g.to(dt)
        g = g.contiguous()
        dqkv = torch.empty_like(q2)
        ws = torch.empty(lib.ldm_attn_bwd_ws_bytes(code, B, H, S, hd), dtype=torch.uint8, device=q2.device)
        p_drop, seed = ctx.dropout
        if p_drop > 0.0:
            fused = FUSED_UNROPE and dt == torch.bfloat16 and hd in (32, 64, 96, 128)
            with _Timed("attn.bwd", 10.0 * B * H * S * S * hd):
                N.check(lib.ldm_attn_bwd_dropout(code, q2.data_ptr(), 3 * D, N.ptr(ctx.key_valid), B, H, S, hd, o.data_ptr(),
                                                 g.data_ptr(), D, saved.data_ptr(), dqkv.data_ptr(), ws.data_ptr(),
                                                 cos.data_ptr() if fused else None, sin.data_ptr() if fused else None,
                                                 1.0 / math.sqrt(hd), p_drop, seed, N.stream()), "ldm_attn_bwd_dropout")
            if not fused:
                N.check(lib.ldm_rope_qk(dqkv.data_ptr(), code, B, S, H, hd, 3 * D, cos.data_ptr(), sin.data_ptr(),
                                        1.0 / math.sqrt(hd), 1, N.stream()), "ldm_rope_qk(inverse)")
            return dqkv.view(B, S, 3 * D), None, None, None, None, None, None, None, None, None, None, None
        # attention backward with the rotary backward fused into its stores where the kernel has it
        rc = -3
        if FUSED_UNROPE and dt == torch.bfloat16:
            with _Timed("attn.bwd", 10.0 * B * H * S * S * hd) as tm:
                rc = lib.ldm_attn_bwd_unrope(code, q2.data_ptr(), 3 * D, N.ptr(ctx.key_valid), B, H, S, hd, o.data_ptr(),
                                             g.data_ptr(), D, saved.data_ptr(), dqkv.data_ptr(), ws.data_ptr(),
                                             cos.data_ptr(), sin.data_ptr(), 1.0 / math.sqrt(hd), N.stream())
                tm.declined = rc in (-2, -3)
            if rc not in (-2, -3):
                N.check(rc, "ldm_attn_bwd_unrope")
        if rc in (-2, -3):
            with _Timed("attn.bwd", 10.0 * B * H * S * S * hd):
                N.check(lib.ldm_attn_bwd(code, q2.data_ptr(), 3 * D, N.ptr(ctx.key_valid), B, H, S, hd, o.data_ptr(),
                                         g.data_ptr(), D, saved.data_ptr(), dqkv.data_ptr(), ws.data_ptr(), N.stream()),
                        "ldm_attn_bwd")
            N.check(lib.ldm_rope_qk(dqkv.data_ptr(), code, B, S, H, hd, 3 * D, cos.data_ptr(), sin.data_ptr(),
                                    1.0 / math.sqrt(hd), 1, N.stream()), "ldm_rope_qk(inverse)")
        return dqkv.view(B, S, 3 * D), None, None, None, None, None, None, None, None, None, None, None


def rope_attention(qkv: Tensor, key_valid: Tensor | None, B: int, S: int, H: int, hd: int, cos: Tensor,
                   sin: Tensor, dropout_p: float = 0.0, seed: int | None = None, pre_rotated: bool = False,
                   attn_mask: Tensor | None = None) -> Tensor:
    """``dropout_p`` > 0: attention-probability dropout; ``seed`` (63 bits) defaults to a draw from
    torch's CPU generator, so ``torch.manual_seed`` reproduces the masks.  ``pre_rotated``: q and k in
    ``qkv`` already carry the rotary embedding and the query scale (the projection applied them); the
    gradient handed back is still the one w.r.t. the un-rotated projection output.

    ``attn_mask``: anything that broadcasts against the scores [B, H, S, S], added to them before the
    key mask and the softmax (src/transformer.py:267-268).  Served by the fp32 kernels that
    materialise the scores -- a bf16 ``qkv`` is up-cast for the attention core and the result cast
    back -- so it costs O(S^2) memory; the model never passes one.  No gradient reaches the mask."""
    if dropout_p > 0.0 and seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    if attn_mask is None:
        return _RopeAttention.apply(qkv, key_valid, B, S, H, hd, cos, sin, float(dropout_p), int(seed or 0),
                                    bool(pre_rotated))
    if attn_mask.requires_grad:
        raise NotImplementedError("a gradient with respect to attn_mask is not implemented")
    m = attn_mask.detach().to(device=qkv.device, dtype=torch.float32)
    m = torch.broadcast_to(m, (B, H, S, S))
    if m.stride(3) != 1:
        m = m.contiguous()
    out = _RopeAttention.apply(qkv.float(), key_valid, B, S, H, hd, cos, sin, float(dropout_p), int(seed or 0),
                               bool(pre_rotated), m)
    return out.to(qkv.dtype)


# ----------------------------------------------------------------------------
# fp32[1] on the device: the number of positions the most recent masked-CE forward averaged over.
# The trainer hands it to FusedAdamW.gate() so that a step whose loss had no position changes nothing.
LAST_LOSS_COUNT = None


class _MaskedCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, masked_ids, mask_token_id, mask_bool, att_valid, seq_weight, norm_mode, link):
        N.require_device(logits)
        lib = N.lib()
        V = logits.shape[-1]
        l2 = logits.reshape(-1, V)
        if l2.stride(-1) != 1 or l2.dtype not in N.DTYPE_CODE:
            l2 = l2.float().contiguous() if l2.dtype not in N.DTYPE_CODE else l2.contiguous()
        T = l2.shape[0]
        seq = logits.shape[-2] if logits.dim() >= 2 else 1
        dev = logits.device
        tg = targets.reshape(-1).contiguous()
        mi = masked_ids.reshape(-1).contiguous() if masked_ids is not None else None
        mb = mask_bool.reshape(-1).contiguous().view(torch.uint8) if mask_bool is not None else None
        av = att_valid.reshape(-1).contiguous().view(torch.uint8) if att_valid is not None else None
        sw = seq_weight.reshape(-1).float().contiguous() if seq_weight is not None else None
        lse = torch.empty(T, dtype=torch.float32, device=dev)
        tmp = torch.empty(T, dtype=torch.float32, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        ld = l2.stride(0) if T > 1 else max(V, l2.stride(0))
        N.check(lib.ldm_ce_fwd(l2.data_ptr(), N.dtype_code(l2.dtype), ld, T, V, seq, tg.data_ptr(), N.ptr(mi),
                               int(mask_token_id), N.ptr(mb), N.ptr(av), N.ptr(sw), norm_mode, lse.data_ptr(),
                               tmp.data_ptr(), out[0:1].data_ptr(), out[1:2].data_ptr(), N.stream()), "ldm_ce_fwd")
        ctx.compact = None
        # the logits came straight from the vocabulary projection (same buffer, same row stride) and its
        # backward can take compact rows
        if link is not None and not (link.y_ptr == l2.data_ptr() and link.ld == ld):
            link = None
        ctx.link = link
        if (link is not None and link.compact_ok and l2.dtype == torch.bfloat16 and T > 1 and ld % PAD_N == 0
                and ld >= V):
            rows = torch.empty(T, dtype=torch.int32, device=dev)
            cnt = torch.empty(1, dtype=torch.int32, device=dev)
            N.check(lib.ldm_ce_compact_rows(T, seq, N.ptr(mi), int(mask_token_id), N.ptr(mb), N.ptr(av),
                                            rows.data_ptr(), cnt.data_ptr(), N.stream()), "ldm_ce_compact_rows")
            ctx.compact = (rows, cnt)
        ctx.save_for_backward(l2, tg, lse, out)
        ctx.aux = (mi, mb, av, sw, int(mask_token_id), norm_mode, seq, ld, logits.shape, logits.dtype)
        global LAST_LOSS_COUNT
        LAST_LOSS_COUNT = out[1:2]
        return out[0].clone()

    @staticmethod
    def backward(ctx, gout):
        l2, tg, lse, out = ctx.saved_tensors
        mi, mb, av, sw, mask_token_id, norm_mode, seq, ld, shape, in_dtype = ctx.aux
        lib = N.lib()
        T, V = l2.shape
        g = gout.reshape(1).float().contiguous()
        link = ctx.link
        # row stride of the gradient: that of the projection's output when the logits came straight from it
        # (its backward then takes the buffer as it is), else V rounded up to the padding unit
        Vp = ld if (link is not None and ld % PAD_N == 0 and ld >= V) else _round_up(V, PAD_N)
        if ctx.compact is not None:
            rows, cnt = ctx.compact
            dlc = torch.empty(_round_up(T, 256), Vp, dtype=l2.dtype, device=l2.device)
            N.check(lib.ldm_ce_bwd_compact(l2.data_ptr(), N.dtype_code(l2.dtype), ld, T, V, seq, tg.data_ptr(),
                                           N.ptr(sw), norm_mode, lse.data_ptr(), out[1:2].data_ptr(), g.data_ptr(),
                                           rows.data_ptr(), cnt.data_ptr(), dlc.data_ptr(), Vp, N.stream()),
                    "ldm_ce_bwd_compact")
            dummy = torch.zeros(1, dtype=l2.dtype, device=l2.device)
            link.grad = (dlc, rows, cnt, dummy)
            return dummy.expand(tuple(shape)), None, None, None, None, None, None, None, None
        dl = torch.empty(T, Vp, dtype=l2.dtype, device=l2.device)
        N.check(lib.ldm_ce_bwd(l2.data_ptr(), N.dtype_code(l2.dtype), ld, T, V, seq, tg.data_ptr(), N.ptr(mi),
                               mask_token_id, N.ptr(mb), N.ptr(av), N.ptr(sw), norm_mode, lse.data_ptr(),
                               out[1:2].data_ptr(), g.data_ptr(), dl.data_ptr(), Vp, N.stream()), "ldm_ce_bwd")
        d = dl[:, :V] if Vp != V else dl
        d = d.view(*shape[:-1], V) if d.is_contiguous() else torch.as_strided(
            dl, tuple(shape), _padded_strides(shape, Vp))
        if d.dtype != in_dtype:
            d = d.to(in_dtype)
        elif link is not None:
            link.zero_padded = (dl.data_ptr(), Vp)
        return d, None, None, None, None, None, None, None, None


# ----------------------------------------------------------------------------
FUSED_HEAD_LOSS = True   # set False to always materialise the full [T, V] logits in the training step
# loss and d(loss)/d(logits) in ONE pass over the compact logits when a backward will follow
# (False: log-sum-exp pass in forward, softmax-minus-onehot pass in backward)
FUSED_CE_GRAD = os.environ.get("LDM_FUSED_CE", "1") != "0"
FUSED_HEAD_STATS = {"used": 0}


def head_loss_eligible(x2: Tensor, pack: PackedWeights) -> bool:
    """Whether head_masked_ce can take this problem (bf16 MFMA forms with device-side sizes)."""
    T, k = x2.shape
    return (FUSED_HEAD_LOSS and pack.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and T >= 1024
            and k >= 128 and k % 32 == 0 and x2.stride(1) == 1 and x2.stride(0) % 4 == 0
            and (pack.n_pad // 256) * ((k + 255) // 256) >= 200)


class _HeadMaskedCE(torch.autograd.Function):
    """mean CE over the participating rows of (x @ W^T + b) without forming the other rows.

    The vocabulary projection (src/transformer.py:441) followed by
    variational_lower_bound_loss (src/loss.py:59-65) reads only the logits of positions whose
    masked id is the mask token (and attention_mask == 1).  Forward: compact those rows of x,
    one GEMM over them (row count on the device, no host sync), streamed log-sum-exp.
    Backward: softmax-minus-one-hot written over the compact logits in place, then dx (scattered
    back to the source rows), dW and db from the compact rows.  Loss and every gradient equal
    the dense computation's: the skipped rows contribute exact zeros there."""

    @staticmethod
    def forward(ctx, x2, pack: PackedWeights, targets, masked_ids, att_valid, mask_token_id, want_grad, seq_weight,
                seq, norm_mode, *params):
        N.require_device(x2)
        lib = N.lib()
        st = N.stream()
        w, b = pack.get()
        T, k = x2.shape
        dev = x2.device
        n_pad = pack.n_pad
        Tp = _round_up(T, 256)
        tg = targets.reshape(-1).contiguous()
        mi = masked_ids.reshape(-1).contiguous()
        av = att_valid.reshape(-1).contiguous().view(torch.uint8) if att_valid is not None else None
        rows = torch.zeros(T, dtype=torch.int32, device=dev)
        cnt = torch.empty(1, dtype=torch.int32, device=dev)
        N.check(lib.ldm_ce_compact_rows(T, 1, mi.data_ptr(), int(mask_token_id), None, N.ptr(av), rows.data_ptr(),
                                        cnt.data_ptr(), st), "ldm_ce_compact_rows")
        xc = torch.empty(Tp, k, dtype=torch.bfloat16, device=dev)
        N.check(lib.ldm_gather_rows(x2.data_ptr(), N.BF16, x2.stride(0), rows.data_ptr(), cnt.data_ptr(), T, k,
                                    xc.data_ptr(), st), "ldm_gather_rows")
        rows_l = rows.long()
        tgc = tg[rows_l]                                            # entries past the count repeat row 0: unused
        # weight of compact row i = weight of the sequence its source row belongs to (1/t in the paper form)
        wc = None
        if seq_weight is not None:
            wc = seq_weight.reshape(-1).float()[rows_l // int(seq)].contiguous()
        norm_T = float(T)
        lc = torch.empty(Tp, n_pad, dtype=torch.bfloat16, device=dev)
        with _Timed("gemm.fwd", 2.0 * n_pad * k, cnt):              # flops per row; rows = cnt (device)
            # (M = the rows the compact buffers hold, a multiple of 256: whole tiles for the four-wave kernel)
            N.check(lib.ldm_gemm_rows_dyn(N.BF16, 1, Tp, n_pad, k, xc.data_ptr(), k, w.data_ptr(), k, lc.data_ptr(), n_pad,
                                          b.data_ptr(), cnt.data_ptr(), None, st), "ldm_gemm_rows_dyn")
        lse = torch.empty(T, dtype=torch.float32, device=dev)
        tmp = torch.empty(T, dtype=torch.float32, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        ctx.grad_ready = bool(want_grad and FUSED_CE_GRAD and pack.n <= 131072)
        if ctx.grad_ready:
            # one read + one write of the compact logits: they leave this call as d(loss)/d(logits)
            with _Timed("ce", 4.0 * n_pad, cnt):                    # bytes per row: read + write, bf16
                N.check(lib.ldm_ce_fwd_grad(lc.data_ptr(), N.BF16, n_pad, Tp, pack.n, tgc.data_ptr(), cnt.data_ptr(),
                                            int(norm_mode), norm_T, N.ptr(wc), lse.data_ptr(), tmp.data_ptr(),
                                            out[0:1].data_ptr(), out[1:2].data_ptr(), st), "ldm_ce_fwd_grad")
        else:
            valid = torch.arange(T, dtype=torch.int32, device=dev) < cnt  # compact row i exists
            N.check(lib.ldm_ce_fwd(lc.data_ptr(), N.BF16, n_pad, T, pack.n, 1, tgc.data_ptr(), None, 0,
                                   valid.view(torch.uint8).data_ptr(), None, N.ptr(wc), int(norm_mode), lse.data_ptr(),
                                   tmp.data_ptr(), out[0:1].data_ptr(), out[1:2].data_ptr(), st), "ldm_ce_fwd")
        ctx.pack = pack
        ctx.save_for_backward(xc, w, lc, tgc, lse, out, rows, cnt)
        ctx.wc = wc
        ctx.norm_mode = int(norm_mode)
        ctx.dims = (T, k)
        ctx.done = False
        FUSED_HEAD_STATS["used"] += 1
        global LAST_LOSS_COUNT
        LAST_LOSS_COUNT = out[1:2]
        return out[0].clone()

    @staticmethod
    def backward(ctx, gout):
        if ctx.done:
            raise RuntimeError("the fused LM-head loss overwrites its logits in backward: backward twice "
                               "through the same forward is not supported")
        ctx.done = True
        xc, w, lc, tgc, lse, out, rows, cnt = ctx.saved_tensors
        pack = ctx.pack
        lib = N.lib()
        st = N.stream()
        T, k = ctx.dims
        n_pad = pack.n_pad
        dev = xc.device
        g = gout.reshape(1).float().contiguous()
        if ctx.grad_ready:
            # the forward left the gradient for an upstream gradient of 1: a no-op launch in that case
            N.check(lib.ldm_scale_rows_dev(lc.data_ptr(), N.BF16, n_pad, lc.shape[0], n_pad, cnt.data_ptr(),
                                           g.data_ptr(), st), "ldm_scale_rows_dev")
        else:
            ident = torch.arange(T, dtype=torch.int32, device=dev)
            # dlogits over the logits, in place (each element is read and written by the same thread)
            N.check(lib.ldm_ce_bwd_compact(lc.data_ptr(), N.BF16, n_pad, T, pack.n, 1, tgc.data_ptr(), N.ptr(ctx.wc),
                                           ctx.norm_mode, lse.data_ptr(), out[1:2].data_ptr(), g.data_ptr(),
                                           ident.data_ptr(), cnt.data_ptr(), lc.data_ptr(), n_pad, st),
                    "ldm_ce_bwd_compact")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _compact_input_grad(pack, lc, rows, cnt, T, k, w)
            if dx is None:
                raise RuntimeError("ldm_gemm_rows_dyn declined the LM-head input gradient")
        dwb = ZERO_POOL.take(n_pad * k + n_pad, dev)
        dw, db = dwb[:n_pad * k].view(n_pad, k), dwb[n_pad * k:]
        with _Timed("gemm.dw", 2.0 * n_pad * k, cnt):
            N.check(lib.ldm_gemm_acc_kdyn(1, 0, n_pad, k, _round_up(T, 64), lc.data_ptr(), n_pad, xc.data_ptr(), k,
                                          dw.data_ptr(), k, cnt.data_ptr(), db.data_ptr() if FUSED_BIAS_GRAD else None,
                                          st), "ldm_gemm_acc_kdyn")
        if not FUSED_BIAS_GRAD:
            ws = torch.empty(lib.ldm_colsum_ws_floats(T, n_pad), dtype=torch.float32, device=dev)
            N.check(lib.ldm_colsum_rows_dyn(lc.data_ptr(), N.BF16, T, n_pad, n_pad, db.data_ptr(), ws.data_ptr(),
                                            cnt.data_ptr(), st), "ldm_colsum_rows_dyn")
        grads_w, grads_b = [], []
        off = 0
        for ne in pack.n_each:
            grads_w.append(dw[off:off + ne])
            grads_b.append(db[off:off + ne])
            off += ne
        return (dx, None, None, None, None, None, None, None, None, None, *grads_w, *grads_b)


def head_masked_ce(x: Tensor, pack: PackedWeights, targets: Tensor, masked_ids: Tensor, mask_token_id: int,
                   att_valid: Tensor | None = None, seq_weight: Tensor | None = None, seq: int = 1,
                   norm_mode: int = 0) -> Tensor:
    """variational_lower_bound_loss(linear(x), ...) computed on the participating rows only.
    ``seq_weight`` [B] (rows of sequence b weigh seq_weight[b]; ``seq`` = rows per sequence) and
    ``norm_mode`` (0: number of participating rows, 1: all rows) are the paper-form extensions."""
    x2 = _rows2d(x)
    want_grad = torch.is_grad_enabled() and (x2.requires_grad or any(p.requires_grad for p in pack.weights + pack.biases))
    return _HeadMaskedCE.apply(x2, pack, targets, masked_ids, att_valid, mask_token_id, want_grad, seq_weight,
                               int(seq), int(norm_mode), *pack.weights, *pack.biases)


def _padded_strides(shape, ld):
    strides = [1] * len(shape)
    acc = ld
    for i in range(len(shape) - 2, -1, -1):
        strides[i] = acc
        acc *= shape[i]
    return tuple(strides)


def masked_ce(logits, targets, *, masked_ids=None, mask_token_id=0, mask_bool=None, att_valid=None,
              seq_weight=None, norm_mode=0) -> Tensor:
    link = getattr(logits, "_ldm_head_link", None) if torch.is_grad_enabled() else None
    return _MaskedCE.apply(logits, targets, masked_ids, mask_token_id, mask_bool, att_valid, seq_weight, norm_mode,
                           link)
