"""AdamW on one HIP kernel per (up to 24) tensors.

``FusedAdamW(params, lr, betas, eps, weight_decay)`` applies the update of
``torch.optim.AdamW`` -- the optimizer the reference builds at main.py:87-91 and steps at
src/trainer.py:75,137 -- through ``ldm_adamw_step``: parameters, gradients and both moments are
read once and written once, ``grad_scale`` folds a gradient scaling (1 / world size after a summing
all-reduce) into the same pass, and the compute-dtype copies of the projection weights
(``ops.PackedWeights``) are refreshed by it, so the next forward does not re-cast them.

State layout and names follow torch (``step``, ``exp_avg``, ``exp_avg_sq`` per parameter), so
``state_dict()`` round-trips with ``torch.optim.AdamW``.  Parameters must be contiguous fp32 tensors
on a ROCm device; there is no CPU path.
"""

from __future__ import annotations

import ctypes as C

import torch

from . import _native as N


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.grad_scale = 1.0        # multiplies every gradient inside the update
        self._shadows: dict = {}     # id(param) -> (tensor slice, kind)
        self._shadow_sig = None
        self._gate = None            # device fp32[1]: the next step() changes nothing when it holds 0
        self._skipped = None         # device int32[1]: steps gated off since the last reconcile

    # ------------------------------------------------------------------ empty-loss gate
    def gate(self, count: torch.Tensor | None) -> None:
        """Make the next ``step()`` a no-op when ``count`` (device fp32[1], the number of positions
        the loss averaged over) is 0 -- what the reference's step does when the loss had no masked
        position: src/loss.py:18-19 returns a constant, every ``.grad`` stays None, and
        torch.optim.AdamW skips such parameters (no weight decay, no moment decay, no step count).
        Decided on the device; the host never waits.  ``None`` removes the gate."""
        self._gate = count
        if count is not None and self._skipped is None:
            self._skipped = torch.zeros(1, dtype=torch.int32, device=count.device)

    def _reconcile(self) -> None:
        """Take the gated-off steps out of the per-parameter step counts (reads one int32 back)."""
        if self._skipped is None:
            return
        sk = int(self._skipped.item())
        if sk:
            for st in self.state.values():
                if "step" in st:
                    st["step"] -= min(float(sk), float(st["step"]))
            self._skipped.zero_()

    def state_dict(self):
        self._reconcile()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        self._reconcile()
        return super().load_state_dict(state_dict)

    # ------------------------------------------------------------------ weight shadows
    def attach(self, module: torch.nn.Module) -> "FusedAdamW":
        """Let the step keep the packed compute-dtype weight copies of ``module`` current."""
        self._module = module
        self._shadow_sig = None
        return self

    def _refresh_shadows(self) -> None:
        module = getattr(self, "_module", None)
        if module is None:
            return
        # only the packs of each module's CURRENT compute dtype are refreshed by the update kernel; a
        # module that was switched (bf16 training, then an fp32 evaluation) also holds packs of the other
        # dtype: those are marked stale after every step and re-cast when they are next used
        packs, stale = [], []
        for m in module.modules():
            cur = getattr(m, "compute_dtype", None)
            for pk in getattr(m, "_packs", {}).values():
                (packs if (cur is None or pk.dtype == cur) else stale).append(pk)
        self._stale_packs = stale
        sig = tuple((id(pk), id(pk.w)) for pk in packs)
        if sig == self._shadow_sig:
            return
        self._shadows = {}
        for pk in packs:
            w, b = pk.get()                       # builds the copies if they do not exist yet
            kind = 1 if w.dtype == torch.bfloat16 else 2
            off = 0
            for wt, bs, n in zip(pk.weights, pk.biases, pk.n_each):
                self._shadows.setdefault(id(wt), []).append((w[off:off + n], kind))
                self._shadows.setdefault(id(bs), []).append((b[off:off + n], 2))
                off += n
        self._shadow_sig = tuple((id(pk), id(pk.w)) for pk in packs)

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._refresh_shadows()
        if getattr(self, "_module", None) is None:
            from . import ops
            ops.parameters_changed()     # nobody keeps the packed weight copies current: re-cast lazily
        lib = N.lib()
        gate, self._gate = self._gate, None          # a gate holds for one step
        gate_ptr = gate.data_ptr() if gate is not None else None
        skipped_ptr = self._skipped.data_ptr() if self._skipped is not None else None
        for group in self.param_groups:
            entries = []
            step_no = None
            for p in group["params"]:
                if p.grad is None:
                    continue
                N.require_device(p, p.grad)
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise TypeError("FusedAdamW needs contiguous fp32 parameters")
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                if g.data_ptr() % 16:          # a view at an odd offset of a larger buffer
                    g = g.clone()
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                k = int(st["step"].item())
                sh = self._shadows.get(id(p), [])
                if len(sh) > 1:
                    raise RuntimeError("a parameter belongs to more than one weight pack of its module's compute dtype")
                entries.append((k, p, g, st["exp_avg"], st["exp_avg_sq"], sh[0] if sh else None))
            # tensors of one launch share the step count (bias correction); group by it
            by_step: dict = {}
            for e in entries:
                by_step.setdefault(e[0], []).append(e)
            for step_no, es in by_step.items():
                arr = (N.AdamWTensor * len(es))()
                keep = []
                for i, (_, p, g, m, v, sh) in enumerate(es):
                    arr[i].p, arr[i].g, arr[i].m, arr[i].v = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
                    arr[i].n = p.numel()
                    if sh is not None:
                        arr[i].shadow, arr[i].shadow_kind = sh[0].data_ptr(), sh[1]
                    else:
                        arr[i].shadow, arr[i].shadow_kind = None, 0
                    keep.append(g)
                b1, b2 = group["betas"]
                N.check(lib.ldm_adamw_step_gated(arr, len(es), step_no, float(group["lr"]), float(b1), float(b2),
                                                 float(group["eps"]), float(group["weight_decay"]),
                                                 float(self.grad_scale), gate_ptr, skipped_ptr, N.stream()),
                        "ldm_adamw_step_gated")
        if gate is not None:
            N.check(lib.ldm_adamw_gate_commit(gate_ptr, skipped_ptr, N.stream()), "ldm_adamw_gate_commit")
        for pk in getattr(self, "_stale_packs", ()):
            pk._key = None
        if self._shadows:
            from . import ops
            ops.shadows_refreshed()      # the packed copies changed under their keys: transposed copies follow
        return loss
