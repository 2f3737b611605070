"""Forward-process masking on the fused HIP kernel (K1).

Mirrors the forward half of the reference's ``src/masking_strategy.py``
(``MaskingStrategy`` protocol :9-31, ``RandomMaskingStrategy`` :65-114,
``create_masking_strategy`` :201-231).  The reverse-process remasking strategy
is sampling/evaluation code and is not part of the training step.
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


def create_masking_strategy(strategy_name: Literal["random"], bos_token_id: int, eos_token_id: int,
                            pad_token_id: int, mask_token_id: int) -> MaskingStrategy:
    """Factory with the reference's signature and error (src/masking_strategy.py:201-231)."""
    if strategy_name == "random":
        return RandomMaskingStrategy(bos_token_id=bos_token_id, eos_token_id=eos_token_id,
                                     pad_token_id=pad_token_id, mask_token_id=mask_token_id)
    raise ValueError(f"Unsupported masking strategy: {strategy_name}")
