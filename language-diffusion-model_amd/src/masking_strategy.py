"""Forward-process masking (K1) and reverse-process remasking on HIP kernels.

Mirrors the reference's ``src/masking_strategy.py``: ``MaskingStrategy`` /
``RemaskingStrategy`` protocols (:9-62), ``RandomMaskingStrategy`` (:65-114),
``RandomRemaskingStrategy`` (:117-198), ``create_masking_strategy`` (:201-231) and
``create_remasking_strategy`` (:234-264).
"""

from __future__ import annotations

from typing import Literal, Protocol

import torch
from torch import Tensor

from . import _native as N


class MaskingStrategy(Protocol):
    """Interface of a forward-process masking strategy (src/masking_strategy.py:9-31)."""

    def apply_random_mask(self, input_ids: Tensor, mask_prob: float) -> tuple[Tensor, Tensor]:
        ...


class RemaskingStrategy(Protocol):
    """Interface of a reverse-process remasking strategy (src/masking_strategy.py:34-62)."""

    def apply_remask(self, input_ids: Tensor, logits: Tensor, mask_indices: Tensor, remask_ratio: float,
                     prompt_positions: Tensor | None = None) -> Tensor:
        ...


class RandomMaskingStrategy:
    """Bernoulli(mask_prob) masking of every token that is not BOS, EOS or PAD.

    ``apply_random_mask`` keeps the reference signature and semantics
    (src/masking_strategy.py:88-114): ``eligible & (rand < mask_prob)`` with the
    comparison in fp32, input left untouched, returns ``(masked_input, mask_indices)``.

    draw source:
      * default -- ``torch.rand_like(input_ids, dtype=torch.float)``, i.e. the SAME
        torch generator stream the reference consumes (src/masking_strategy.py:110),
        handed to the kernel as an injected draw;
      * ``fused_rng=True`` -- Philox4x32-10 inside the kernel (no draw tensor is
        materialised: one 17 B/token pass), keyed by ``seed`` and a per-call counter.
    ``mask_prob`` may also be a per-sequence tensor ([B] or [B,1]).
    """

    def __init__(self, bos_token_id: int = 101, eos_token_id: int = 102, pad_token_id: int = 0,
                 mask_token_id: int = 103, fused_rng: bool = False, seed: int = 0):
        self.bos_token_id = bos_token_id
        self.eos_token_id = eos_token_id
        self.pad_token_id = pad_token_id
        self.mask_token_id = mask_token_id
        self.fused_rng = fused_rng
        self.seed = seed
        self.calls = 0

    def _launch(self, input_ids: Tensor, t_mode: int, t_scalar: float, t_seq: Tensor | None, eps: float,
                draw: Tensor | None, seed: int, offset: int) -> tuple[Tensor, Tensor]:
        N.require_device(input_ids)
        if input_ids.dtype != torch.int64:
            raise TypeError(f"input_ids must be int64, got {input_ids.dtype}")
        ids = input_ids.contiguous()
        if ids.dim() == 1:
            batch, seq = 1, ids.shape[0]
        else:
            seq = ids.shape[-1]
            batch = ids.numel() // max(seq, 1) if seq else 0
        masked = torch.empty_like(ids)
        mask = torch.empty(ids.shape, dtype=torch.bool, device=ids.device)
        if draw is not None:
            draw = draw.to(torch.float32).contiguous()
        N.check(N.lib().ldm_mask_fwd(ids.data_ptr(), N.ptr(draw), t_mode, float(t_scalar), N.ptr(t_seq), eps,
                                     batch, seq, self.bos_token_id, self.eos_token_id, self.pad_token_id,
                                     self.mask_token_id, seed, offset, masked.data_ptr(), mask.data_ptr(),
                                     N.stream()), "ldm_mask_fwd")
        return masked, mask

    def apply_random_mask(self, input_ids: Tensor, mask_prob, draw: Tensor | None = None) -> tuple[Tensor, Tensor]:
        t_mode, t_scalar, t_seq = 0, 0.0, None
        if isinstance(mask_prob, Tensor):
            if mask_prob.numel() == 1:
                t_scalar = float(mask_prob.item())
            elif input_ids.dim() == 2 and mask_prob.numel() == input_ids.shape[0]:
                t_mode = 1
                t_seq = mask_prob.reshape(-1).to(device=input_ids.device, dtype=torch.float32).contiguous()
            else:
                raise ValueError("mask_prob tensor must have 1 element or one element per sequence")
        else:
            t_scalar = float(mask_prob)
        offset = self.calls
        self.calls += 1
        if draw is None and not self.fused_rng:
            draw = torch.rand_like(input_ids, dtype=torch.float)
        return self._launch(input_ids, t_mode, t_scalar, t_seq, 0.0, draw, self.seed, offset)

    def apply_per_sequence_mask(self, input_ids: Tensor, eps: float = 1e-3, seed: int | None = None,
                                offset: int | None = None) -> tuple[Tensor, Tensor, Tensor]:
        """Paper-form masking: t_b ~ U(eps, 1) per sequence, drawn inside the kernel.

        Not the reference's behaviour (it draws ONE scalar per call, src/model.py:59-60);
        returns ``(masked_input, mask_indices, t[B])``."""
        t_seq = torch.empty(input_ids.shape[0], dtype=torch.float32, device=input_ids.device)
        if offset is None:
            offset = self.calls
            self.calls += 1
        masked, mask = self._launch(input_ids, 2, 0.0, t_seq, eps, None, self.seed if seed is None else seed,
                                    offset)
        return masked, mask, t_seq


class RandomRemaskingStrategy:
    """Puts the mask token back on a random ``int(num_eligible * remask_ratio)`` of the positions
    that were filled in this step (src/masking_strategy.py:140-198).

    Eligible = was masked this step, not BOS/EOS/PAD, not a prompt position; the choice is the
    first ``k`` entries of ``torch.randperm(num_eligible, device=...)`` applied to the eligible
    positions in row-major order -- the same generator call the reference makes (:189), so a
    seeded run consumes the same stream.  ``logits`` is accepted and unused, as in the reference.
    One host read of the eligible count per call (the reference has the same ``.item()``, :178)."""

    def __init__(self, bos_token_id: int = 101, eos_token_id: int = 102, pad_token_id: int = 0,
                 mask_token_id: int = 103):
        self.bos_token_id = bos_token_id
        self.eos_token_id = eos_token_id
        self.pad_token_id = pad_token_id
        self.mask_token_id = mask_token_id

    def apply_remask(self, input_ids: Tensor, logits: Tensor | None, mask_indices: Tensor, remask_ratio: float,
                     prompt_positions: Tensor | None = None) -> Tensor:
        N.require_device(input_ids)
        if input_ids.dtype != torch.int64:
            raise TypeError(f"input_ids must be int64, got {input_ids.dtype}")
        ids = input_ids.contiguous()
        n = ids.numel()
        out = torch.empty_like(ids)
        if n == 0:
            return out
        dev = ids.device
        was = mask_indices.to(device=dev, dtype=torch.bool).contiguous()
        prompt = None
        if prompt_positions is not None:
            prompt = prompt_positions.to(device=dev, dtype=torch.bool).contiguous()
        pos = torch.empty(n, dtype=torch.int64, device=dev)
        count = torch.empty(1, dtype=torch.int32, device=dev)
        lib, st = N.lib(), N.stream()
        N.check(lib.ldm_remask_eligible(ids.data_ptr(), was.data_ptr(), N.ptr(prompt), n, self.bos_token_id,
                                        self.eos_token_id, self.pad_token_id, pos.data_ptr(), count.data_ptr(), st),
                "ldm_remask_eligible")
        num_eligible = int(count.item())
        k, perm = 0, None
        if num_eligible > 0 and remask_ratio > 0:
            k = int(num_eligible * remask_ratio)
            if k > 0:
                perm = torch.randperm(num_eligible, device=dev)
        N.check(lib.ldm_remask_apply(ids.data_ptr(), n, pos.data_ptr(), num_eligible, N.ptr(perm), k,
                                     self.mask_token_id, out.data_ptr(), st), "ldm_remask_apply")
        return out


def create_masking_strategy(strategy_name: Literal["random"], bos_token_id: int, eos_token_id: int,
                            pad_token_id: int, mask_token_id: int) -> MaskingStrategy:
    """Factory with the reference's signature and error (src/masking_strategy.py:201-231)."""
    if strategy_name == "random":
        return RandomMaskingStrategy(bos_token_id=bos_token_id, eos_token_id=eos_token_id,
                                     pad_token_id=pad_token_id, mask_token_id=mask_token_id)
    raise ValueError(f"Unsupported masking strategy: {strategy_name}")


def create_remasking_strategy(strategy_name: Literal["random"], bos_token_id: int, eos_token_id: int,
                              pad_token_id: int, mask_token_id: int) -> RemaskingStrategy:
    """Factory with the reference's signature and error (src/masking_strategy.py:234-264)."""
    if strategy_name == "random":
        return RandomRemaskingStrategy(bos_token_id=bos_token_id, eos_token_id=eos_token_id,
                                       pad_token_id=pad_token_id, mask_token_id=mask_token_id)
    raise ValueError(f"Unsupported remasking strategy: {strategy_name}")
