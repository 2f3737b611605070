"""Model wrapper of the masked-diffusion training step.

Mirrors ``DiffusionLanguageModel.__init__`` / ``forward`` of the reference
(src/model.py:22-72): draw t, mask, run the transformer, return
``(logits, masked_input_ids, mask_indices)``, and the reverse process
``generate`` / ``sample`` (src/model.py:74-189) with the token choice and the
remasking on HIP kernels.
"""

from __future__ import annotations

import random

import torch
import torch.nn as nn
from torch import Tensor

from . import _native as N
from . import ops


class DiffusionLanguageModel(nn.Module):
    def __init__(self, transformer, tokenizer, masking_strategy, remasking_strategy=None) -> None:
        super().__init__()
        self.transformer = transformer
        self.tokenizer = tokenizer
        self.masking_strategy = masking_strategy
        self.remasking_strategy = remasking_strategy
        self.sampling_rng = "philox"   # "torch": softmax + torch.multinomial, the reference's generator stream
        self.sampling_seed = 0
        self._sampling_calls = 0

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
                      t: Tensor | None = None, *, objective: str = "reference", eps: float = 1e-3) -> Tensor:
        """The loss of one training step: forward() followed by variational_lower_bound_loss
        as the reference trainer composes them (src/trainer.py:126-134), with the LM head
        evaluated only at the positions the loss reads.  Consumes the python RNG exactly as
        forward() does (one random.random() per call).

        ``objective="reference"`` (default) is the reference's step: one scalar t per call, unweighted
        mean over the masked positions.  ``objective="paper"`` is the LLaDA form, which the reference
        does not implement (so it has no reference parity; it is checked against the CPU restatement
        of the same formula): t_b ~ U(eps, 1) per sequence drawn inside the masking kernel (or the
        per-sequence ``t`` handed in), every masked position of sequence b weighted by 1 / t_b, the
        sum divided by B * L."""
        if objective not in ("reference", "paper"):
            raise ValueError(f"objective must be 'reference' or 'paper', got {objective!r}")
        if torch.is_grad_enabled():
            ops.begin_step()
        if objective == "paper":
            if t is None:
                masked_input_ids, _, t_seq = self.masking_strategy.apply_per_sequence_mask(input_ids, eps=eps)
            else:
                t_seq = t.reshape(-1).to(device=input_ids.device, dtype=torch.float32)
                if t_seq.numel() == 1:
                    t_seq = t_seq.expand(input_ids.shape[0]).contiguous()
                masked_input_ids, _ = self.masking_strategy.apply_random_mask(input_ids=input_ids, mask_prob=t_seq)
            return self.transformer.masked_token_loss(masked_input_ids, input_ids, self.tokenizer.mask_token_id,
                                                      attention_mask, seq_weight=1.0 / t_seq, normalize_by="tokens")
        if t is None:
            mask_prob = random.random()
        else:
            mask_prob = t.item() if (isinstance(t, torch.Tensor) and t.numel() == 1) else t
        masked_input_ids, _ = self.masking_strategy.apply_random_mask(input_ids=input_ids, mask_prob=mask_prob)
        return self.transformer.masked_token_loss(masked_input_ids, input_ids, self.tokenizer.mask_token_id,
                                                  attention_mask)

    # ------------------------------------------------------------------ reverse process
    def _pick_tokens(self, seq: Tensor, logits: Tensor, temperature: float) -> Tensor:
        """seq with every mask position replaced by a token chosen from its logits row: argmax
        for temperature 0, otherwise a draw from softmax(logits / temperature)
        (src/model.py:93-105).  One kernel over the rows that hold a mask; no [T, V] softmax.

        ``sampling_rng == "torch"`` keeps the reference's operators for the draw (softmax +
        torch.multinomial, i.e. torch's generator stream); the default draws in the kernel from
        Philox keyed by ``sampling_seed`` and a per-call counter."""
        N.require_device(seq)
        ids = seq.contiguous()
        V = logits.shape[-1]
        l2 = logits.reshape(-1, V)
        if l2.stride(-1) != 1:
            l2 = l2.contiguous()
        if temperature > 0 and self.sampling_rng == "torch":
            probs = torch.softmax(logits / temperature, dim=-1)
            predictions = torch.multinomial(probs.reshape(-1, V), num_samples=1).reshape(ids.shape)
            pos = ids == self.tokenizer.mask_token_id
            out = ids.clone()
            out[pos] = predictions[pos]
            return out
        out = torch.empty_like(ids)
        offset = self._sampling_calls
        self._sampling_calls += 1
        N.check(N.lib().ldm_sample_tokens(l2.data_ptr(), N.dtype_code(l2.dtype), ids.numel(), V, l2.stride(0),
                                          ids.data_ptr(), self.tokenizer.mask_token_id, 0, float(temperature),
                                          self.sampling_seed, offset, out.data_ptr(), None, N.stream()),
                "ldm_sample_tokens")
        return out

    def generate(self, input_ids: Tensor, temperature: float = 1.0) -> Tensor:
        """One denoising pass: every mask token of ``input_ids`` is replaced by a prediction
        (src/model.py:74-106; all positions attended, as there)."""
        attention_mask = torch.ones_like(input_ids)
        with torch.no_grad():
            logits = self.transformer(input_ids=input_ids, attention_mask=attention_mask)
            return self._pick_tokens(input_ids, logits, temperature)

    def sample(self, prompt: Tensor | None = None, num_steps: int = 20, seq_length: int = 32,
               temperature: float = 1.0) -> Tensor:
        """Reverse diffusion (src/model.py:108-189): start from mask tokens (after the prompt, or
        after a BOS), and ``num_steps`` times predict every masked position, then put the mask
        back on a ``next_t / t`` share of the positions just filled, t = 1 - step/(num_steps-1).
        Without a prompt the sequence is created on the transformer's device."""
        mask_id = self.tokenizer.mask_token_id
        if prompt is not None:
            seq = prompt
            batch_size = seq.size(0)
            if seq.size(1) < seq_length:
                padding = torch.full((batch_size, seq_length - seq.size(1)), mask_id, dtype=torch.long,
                                     device=seq.device)
                seq = torch.cat([seq, padding], dim=1)
            else:
                seq = seq[:, :seq_length]
        else:
            device = next(self.transformer.parameters()).device
            seq = torch.full((1, seq_length), mask_id, dtype=torch.long, device=device)
            seq[:, 0] = self.tokenizer.bos_token_id
        seq = seq.contiguous()
        attention_mask = torch.ones_like(seq)
        prompt_positions = torch.zeros_like(seq, dtype=torch.bool)
        if prompt is not None:
            prompt_positions[:, :min(prompt.size(1), seq_length)] = True

        for step in range(num_steps):
            t = 1.0 - (step / (num_steps - 1))
            next_t = 1.0 - ((step + 1) / (num_steps - 1)) if step < num_steps - 1 else 0.0
            with torch.no_grad():
                logits = self.transformer(input_ids=seq, attention_mask=attention_mask)
                mask_positions = seq == mask_id
                seq = self._pick_tokens(seq, logits, temperature)
                if step < num_steps - 1:
                    seq = self.remasking_strategy.apply_remask(input_ids=seq, logits=logits,
                                                               mask_indices=mask_positions,
                                                               remask_ratio=next_t / t,
                                                               prompt_positions=prompt_positions)
        return seq
