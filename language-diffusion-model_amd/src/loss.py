"""Masked cross-entropy of the diffusion objective on the streamed-logsumexp kernel (K4).

Same functions and signatures as the reference's ``src/loss.py``
(``masked_cross_entropy_loss`` :8-28, ``variational_lower_bound_loss`` :31-65).
Only the participating rows of ``logits`` are read; the "no row participates"
case is decided on the device (no host synchronisation) and yields 0 with zero
gradients.  The reference returns a constant there (:18-19), so its optimizer
step changes nothing; the trainer gets the same effect by handing the loss's row
count (``ops.LAST_LOSS_COUNT``) to ``FusedAdamW.gate`` when it runs as one process.
"""

from __future__ import annotations

from torch import Tensor

from . import ops


def _ensure_grad(loss: Tensor) -> Tensor:
    # src/loss.py:25-26: the returned loss always has requires_grad=True
    if not loss.requires_grad:
        loss = loss.clone().detach().requires_grad_(True)
    return loss


def masked_cross_entropy_loss(logits: Tensor, targets: Tensor, mask_indices: Tensor) -> Tensor:
    """Average cross-entropy over the positions where ``mask_indices`` is True."""
    loss = ops.masked_ce(logits, targets, mask_bool=mask_indices.bool())
    return _ensure_grad(loss)


def variational_lower_bound_loss(logits: Tensor, targets: Tensor, masked_input_ids: Tensor, mask_token_id: int,
                                 attention_mask: Tensor | None = None, *, seq_weight: Tensor | None = None,
                                 normalize_by: str = "masked") -> Tensor:
    """Mean CE over positions whose (masked) input id equals ``mask_token_id`` and, when an
    attention mask is given, whose ``attention_mask == 1`` (src/loss.py:59-65).

    Keyword-only extensions (not in the reference; defaults reproduce it):
    ``seq_weight`` -- per-sequence weight [B], e.g. 1/t of the paper's objective;
    ``normalize_by`` -- ``"masked"`` (reference: number of participating rows) or
    ``"tokens"`` (B*L as in the paper).
    """
    att_valid = None
    if attention_mask is not None:
        att_valid = attention_mask == 1
    if normalize_by not in ("masked", "tokens"):
        raise ValueError(f"normalize_by must be 'masked' or 'tokens', got {normalize_by!r}")
    loss = ops.masked_ce(logits, targets, masked_ids=masked_input_ids, mask_token_id=mask_token_id,
                         att_valid=att_valid, seq_weight=seq_weight,
                         norm_mode=0 if normalize_by == "masked" else 1)
    return _ensure_grad(loss)
