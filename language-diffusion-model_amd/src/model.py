"""Model wrapper of the masked-diffusion training step.

Mirrors ``DiffusionLanguageModel.__init__`` / ``forward`` of the reference
(src/model.py:22-72): draw t, mask, run the transformer, return
``(logits, masked_input_ids, mask_indices)``.  ``generate`` / ``sample`` are
reverse-process inference and are not part of this package.
"""

from __future__ import annotations

import random

import torch
import torch.nn as nn
from torch import Tensor

from . import ops


class DiffusionLanguageModel(nn.Module):
    def __init__(self, transformer, tokenizer, masking_strategy, remasking_strategy=None) -> None:
        super().__init__()
        self.transformer = transformer
        self.tokenizer = tokenizer
        self.masking_strategy = masking_strategy
        self.remasking_strategy = remasking_strategy

    def forward(self, input_ids: Tensor, attention_mask: Tensor | None = None,
                t: Tensor | None = None) -> tuple[Tensor, Tensor, Tensor]:
        # src/model.py:59-64: one python-RNG scalar per call unless t is given; a
        # one-element tensor is unwrapped, anything else is passed through.
        if torch.is_grad_enabled():
            ops.begin_step()
        if t is None:
            mask_prob = random.random()
        else:
            mask_prob = t.item() if (isinstance(t, torch.Tensor) and t.numel() == 1) else t
        masked_input_ids, mask_indices = self.masking_strategy.apply_random_mask(
            input_ids=input_ids, mask_prob=mask_prob)
        logits = self.transformer(input_ids=masked_input_ids, attention_mask=attention_mask)
        return logits, masked_input_ids, mask_indices

    def training_loss(self, input_ids: Tensor, attention_mask: Tensor | None = None,
                      t: Tensor | None = None) -> Tensor:
        """The loss of one training step: forward() followed by variational_lower_bound_loss
        as the reference trainer composes them (src/trainer.py:126-134), with the LM head
        evaluated only at the positions the loss reads.  Consumes the python RNG exactly as
        forward() does (one random.random() per call)."""
        if torch.is_grad_enabled():
            ops.begin_step()
        if t is None:
            mask_prob = random.random()
        else:
            mask_prob = t.item() if (isinstance(t, torch.Tensor) and t.numel() == 1) else t
        masked_input_ids, _ = self.masking_strategy.apply_random_mask(input_ids=input_ids, mask_prob=mask_prob)
        return self.transformer.masked_token_loss(masked_input_ids, input_ids, self.tokenizer.mask_token_id,
                                                  attention_mask)

    def generate(self, *args, **kwargs):
        raise NotImplementedError("reverse-process generation (src/model.py:74-106) is outside the "
                                  "training-step scope of this package")

    def sample(self, *args, **kwargs):
        raise NotImplementedError("reverse-process sampling (src/model.py:108-189) is outside the "
                                  "training-step scope of this package")
