"""CPU restatement of the reference's reverse process (TEST INFRASTRUCTURE ONLY).

Restates, over plain torch-CPU / numpy values:
  * the token choice of ``DiffusionLanguageModel.generate`` / ``sample`` (src/model.py:93-105,
    167-177): greedy = ``argmax``; with a temperature the reference draws with
    ``torch.multinomial(softmax(logits / temperature))`` -- here the draw is
    ``argmax(logits / temperature + Gumbel)`` with the SAME Philox noise the kernel uses
    (a draw from the same distribution; with respect to the reference's generator stream the
    sampled tokens are "parity unpinned");
  * ``RandomRemaskingStrategy.apply_remask`` (src/masking_strategy.py:160-198) with the
    permutation handed in (the reference takes it from ``torch.randperm``, :189);
  * the loop of ``DiffusionLanguageModel.sample`` (src/model.py:127-189).
Pinned against the real reference by ``oracle/make_golden.py`` (greedy choice, remasking and the
whole greedy sample loop, with the reference's own permutations recorded).
"""

from __future__ import annotations

import numpy as np
import torch

from . import philox
from . import step_oracle as so


def gumbel_noise(rows, vocab, seed, offset):
    """Noise of column c in row r: word c&3 of the Philox block with counter
    (c>>2, r, offset low, offset high), key = seed; u = ((x>>8) + 0.5) * 2^-24; -log(-log(u))."""
    ng = (vocab + 3) // 4
    g = np.tile(np.arange(ng, dtype=np.uint32), rows)
    r = np.repeat(np.arange(rows, dtype=np.uint32), ng)
    w = philox.philox4x32_10(g, r, np.full(g.shape, offset & 0xFFFFFFFF, dtype=np.uint32),
                             np.full(g.shape, (offset >> 32) & 0xFFFFFFFF, dtype=np.uint32),
                             seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    x = np.stack(w, axis=1).reshape(rows, ng * 4)[:, :vocab]
    u = ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    return (-np.log(-np.log(u.astype(np.float64)))).astype(np.float32)


def token_scores(logits, temperature, seed=0, offset=0):
    """[T, V] scores whose row-wise argmax is the chosen token."""
    l2 = logits.reshape(-1, logits.shape[-1]).to(torch.float32)
    if temperature > 0:
        noise = torch.from_numpy(gumbel_noise(l2.shape[0], l2.shape[1], seed, offset))
        return l2 * np.float32(1.0 / temperature) + noise
    return l2


def pick_tokens(ids, logits, mask_id=so.MASK, temperature=0.0, seed=0, offset=0):
    """src/model.py:93-105: predictions at the mask positions, everything else unchanged."""
    pred = torch.argmax(token_scores(logits, temperature, seed, offset), dim=-1).reshape(ids.shape)
    out = ids.clone()
    pos = ids == mask_id
    out[pos] = pred[pos]
    return out


def eligible_positions(ids, was_masked, prompt_positions=None, bos=so.BOS, eos=so.EOS, pad=so.PAD):
    """src/masking_strategy.py:165-175, as flat row-major indices (the order of nonzero())."""
    ok = (ids != bos) & (ids != eos) & (ids != pad) & was_masked.bool()
    if prompt_positions is not None:
        ok &= ~prompt_positions.bool()
    return torch.nonzero(ok.reshape(-1), as_tuple=False).reshape(-1)


def apply_remask(ids, was_masked, remask_ratio, prompt_positions=None, perm=None, mask_id=so.MASK,
                 bos=so.BOS, eos=so.EOS, pad=so.PAD):
    """src/masking_strategy.py:160-198 with the permutation of :189 given as ``perm``."""
    out = ids.clone()
    pos = eligible_positions(ids, was_masked, prompt_positions, bos, eos, pad)
    n = int(pos.numel())
    if n > 0 and remask_ratio > 0:
        k = int(n * remask_ratio)
        if k > 0:
            flat = out.reshape(-1)
            flat[pos[perm[:k]]] = mask_id
    return out


def sample_loop(params, cfg, prompt, num_steps, seq_length, perms, temperature=0.0, seed=0,
                mask_id=so.MASK, bos=so.BOS, eos=so.EOS, pad=so.PAD):
    """src/model.py:127-189.  ``perms``: the permutations of the successive remasking calls that
    draw one (in call order)."""
    if prompt is not None:
        seq = prompt
        if seq.size(1) < seq_length:
            seq = torch.cat([seq, torch.full((seq.size(0), seq_length - seq.size(1)), mask_id, dtype=torch.long)], 1)
        else:
            seq = seq[:, :seq_length]
    else:
        seq = torch.full((1, seq_length), mask_id, dtype=torch.long)
        seq[:, 0] = bos
    att = torch.ones_like(seq)
    prompt_positions = torch.zeros_like(seq, dtype=torch.bool)
    if prompt is not None:
        prompt_positions[:, :min(prompt.size(1), seq_length)] = True
    perms = list(perms)
    calls = 0
    for step in range(num_steps):
        t = 1.0 - (step / (num_steps - 1))
        next_t = 1.0 - ((step + 1) / (num_steps - 1)) if step < num_steps - 1 else 0.0
        logits = so.transformer_forward(params, cfg, seq, att)
        was = seq == mask_id
        seq = pick_tokens(seq, logits, mask_id, temperature, seed, calls)
        calls += 1
        if step < num_steps - 1:
            ratio = next_t / t
            n = int(eligible_positions(seq, was, prompt_positions, bos, eos, pad).numel())
            perm = perms.pop(0) if (n > 0 and ratio > 0 and int(n * ratio) > 0) else None
            seq = apply_remask(seq, was, ratio, prompt_positions, perm, mask_id, bos, eos, pad)
    return seq
