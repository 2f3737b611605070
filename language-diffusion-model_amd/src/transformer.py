"""Bidirectional (non-causal) transformer of the masked-diffusion model on HIP kernels.

Same classes, constructor signatures, sub-module names and ``state_dict`` keys as
the reference's ``src/transformer.py`` (RMSNorm :27-60, SwiGLU :63-95,
RotaryPositionEmbedding :104-185, RotaryMultiheadAttention :188-291,
TransformerEncoderLayer :294-361, DiffusionTransformer :364-444); the arithmetic
runs in the gfx950 library instead of ATen:

* RMSNorm            -> ``ldm_rmsnorm_fwd/bwd``
* q/k/v projections  -> ONE packed GEMM (weights stay three parameters) whose epilogue also applies
                        RoPE and q/sqrt(hd) (``ldm_gemm_qkv_rope``; ``ldm_rope_qk`` in place otherwise)
* scores/softmax/PV  -> ``ldm_attn_fwd/bwd`` (flash attention in bf16)
* out-proj, w3       -> GEMM with bias + residual epilogue; in training with dropout > 0 the epilogue also
                        applies the layer's residual dropout (``ldm_gemm_dropout_residual``)
* w1/w2              -> ONE packed GEMM that also emits ``a * silu(g)`` (``ldm_gemm_swiglu_fwd``)

``compute_dtype`` selects the arithmetic: ``torch.float32`` (default; every
accumulation fp32, matches the reference to ~1e-6) or ``torch.bfloat16``: bf16 MFMA
with fp32 accumulation, fp32 softmax / log-sum-exp / norm statistics, and bf16
STORAGE of every activation, the residual stream, the norm outputs and the logits
included.  That is narrower than what the reference gets from Accelerate's bf16
autocast, which keeps the residual stream, RMSNorm and the logits in fp32; measured
against the fp32 oracle on the full-depth default model the two land at the same
error (whole gradient 1.2e-2 here, 1.1e-2 for the autocast run; worst tensor 3.1e-2
vs 2.8e-2; tests/test_gpu_fullsize.py).  Parameters are always fp32 masters.
"""

from __future__ import annotations

import math
import os
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: N812
from torch import Tensor

from . import _native as N
from . import ops


def _default_dtype() -> torch.dtype:
    v = os.environ.get("LDM_COMPUTE_DTYPE", "fp32").lower()
    return torch.bfloat16 if v in ("bf16", "bfloat16") else torch.float32


@dataclass
class DiffusionTransformerConfig:
    """Same fields and defaults as the reference (src/transformer.py:13-24)."""

    vocab_size: int = 30522
    d_model: int = 768
    nhead: int = 12
    num_layers: int = 12
    dim_feedforward: int = 3072
    dropout: float = 0.1
    layer_norm_eps: float = 1e-12
    max_position_embeddings: int = 512


class RMSNorm(nn.Module):
    """y = weight * x / sqrt(mean(x^2) + eps)  (src/transformer.py:54-58)."""

    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x: Tensor, out_dtype: torch.dtype | None = None) -> Tensor:
        """``out_dtype`` (an extension): dtype of the result, default x's."""
        return ops.rms_norm(x, self.weight, self.eps, out_dtype)

    def with_residual(self, x: Tensor, out_dtype: torch.dtype | None = None) -> tuple[Tensor, Tensor]:
        """(norm(x), x): feeding the second value to the residual connection lets the norm's
        backward kernel add the residual-stream gradient itself."""
        return ops.rms_norm_with_residual(x, self.weight, self.eps, out_dtype)


class SwiGLU(nn.Module):
    """w3( w1(x) * silu(w2(x)) ) -- the gate is w2 (src/transformer.py:91-95)."""

    def __init__(self, in_features: int, hidden_features: int, out_features: int):
        super().__init__()
        self.w1 = nn.Linear(in_features, hidden_features)
        self.w2 = nn.Linear(in_features, hidden_features)
        self.w3 = nn.Linear(hidden_features, out_features)
        self.compute_dtype = _default_dtype()
        self._packs: dict = {}

    def _pack(self, which: str) -> ops.PackedWeights:
        key = (which, self.compute_dtype)
        if key not in self._packs:
            if which == "in":
                self._packs[key] = ops.PackedWeights([self.w1.weight, self.w2.weight],
                                                     [self.w1.bias, self.w2.bias], self.compute_dtype, "w12")
            else:
                self._packs[key] = ops.PackedWeights([self.w3.weight], [self.w3.bias], self.compute_dtype, "w3")
        return self._packs[key]

    def forward(self, x: Tensor, residual: Tensor | None = None, residual_dropout: float = 0.0) -> Tensor:
        ag, h = ops.swiglu_up(x, self._pack("in"))      # h = a * silu(g) out of the same GEMM's epilogue
        return ops.swiglu_down(ag, self._pack("out"), residual=residual, h=h, residual_dropout=residual_dropout)


def rotate_half(x: Tensor) -> Tensor:
    """cat(-x2, x1) over the two contiguous halves (src/transformer.py:98-101)."""
    x1, x2 = x.chunk(2, dim=-1)
    return torch.cat((-x2, x1), dim=-1)


class RotaryPositionEmbedding(nn.Module):
    """cos/sin tables of src/transformer.py:110-160 (base 10000, cat(freqs, freqs))."""

    def __init__(self, dim: int, max_position_embeddings: int = 2048, base: int = 10000):
        super().__init__()
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq)
        self.max_seq_len = max_position_embeddings
        self.cached_seq_len: int | None = None
        self.cached_cos: Tensor | None = None
        self.cached_sin: Tensor | None = None
        self._half: tuple[Tensor, Tensor] | None = None
        self._step: Tensor | None = None

    def forward(self, x: Tensor, seq_len: int) -> tuple[Tensor, Tensor]:
        if seq_len > self.max_seq_len:
            raise ValueError(f"Sequence length {seq_len} exceeds maximum length {self.max_seq_len}")
        if (self.cached_seq_len != seq_len or self.cached_cos is None
                or self.cached_cos.device != x.device):
            self.cached_seq_len = seq_len
            t = torch.arange(seq_len, device=x.device, dtype=self.inv_freq.dtype)
            freqs = torch.einsum("i,j->ij", t, self.inv_freq.to(x.device))
            emb = torch.cat((freqs, freqs), dim=-1)
            self.cached_cos = emb.cos()
            self.cached_sin = emb.sin()
            half = freqs.shape[-1]
            self._half = (self.cached_cos[:, :half].contiguous(), self.cached_sin[:, :half].contiguous())
            # cos / sin of a 16-position advance, per rotation pair (double precision, then fp32): the
            # projection epilogue steps down its rows with it (ops.RopeSpec)
            ang = 16.0 * self.inv_freq.to(device=x.device, dtype=torch.float64)
            self._step = torch.stack((ang.cos(), ang.sin())).to(torch.float32).contiguous()
        assert self.cached_cos is not None and self.cached_sin is not None
        return self.cached_cos, self.cached_sin

    def half_tables(self, x: Tensor, seq_len: int) -> tuple[Tensor, Tensor]:
        """fp32 [seq, dim/2] tables for the kernel (the two halves of the reference's are equal)."""
        self.forward(x, seq_len)
        assert self._half is not None
        return self._half

    def step_table(self, x: Tensor, seq_len: int) -> Tensor:
        """fp32 [2, dim/2]: cos and sin of the angle of 16 positions, per rotation pair."""
        self.forward(x, seq_len)
        assert self._step is not None
        return self._step

    @staticmethod
    def apply_rotary_pos_emb(q: Tensor, k: Tensor, cos: Tensor, sin: Tensor) -> tuple[Tensor, Tensor]:
        """q*cos + rotate_half(q)*sin for q, k of shape [batch, seq, heads, dim]
        (src/transformer.py:178-185).  The fused attention path does not come through
        here (it rotates in place inside the qkv activation); this standalone form is
        kept for API compatibility and runs as plain tensor ops."""
        cos = cos.unsqueeze(0).unsqueeze(2)
        sin = sin.unsqueeze(0).unsqueeze(2)
        return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


class RotaryMultiheadAttention(nn.Module):
    """Self-attention with rotary embeddings (src/transformer.py:188-291)."""

    def __init__(self, d_model: int, nhead: int, dropout: float = 0.0):
        super().__init__()
        assert d_model % nhead == 0, "d_model must be divisible by nhead"
        self.d_model = d_model
        self.nhead = nhead
        self.head_dim = d_model // nhead
        self.q_proj = nn.Linear(d_model, d_model)
        self.k_proj = nn.Linear(d_model, d_model)
        self.v_proj = nn.Linear(d_model, d_model)
        self.out_proj = nn.Linear(d_model, d_model)
        self.dropout = nn.Dropout(dropout)
        self.rotary_emb = RotaryPositionEmbedding(self.head_dim)
        self.compute_dtype = _default_dtype()
        self._packs: dict = {}

    def _pack(self, which: str) -> ops.PackedWeights:
        key = (which, self.compute_dtype)
        if key not in self._packs:
            if which == "qkv":
                self._packs[key] = ops.PackedWeights(
                    [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight],
                    [self.q_proj.bias, self.k_proj.bias, self.v_proj.bias], self.compute_dtype, "qkv")
            else:
                self._packs[key] = ops.PackedWeights([self.out_proj.weight], [self.out_proj.bias],
                                                     self.compute_dtype, "out")
        return self._packs[key]

    def forward(self, query: Tensor, key: Tensor, value: Tensor, attn_mask: Tensor | None = None,
                key_padding_mask: Tensor | None = None, residual: Tensor | None = None,
                residual_dropout: float = 0.0) -> Tensor:
        if key is not query or value is not query:
            raise NotImplementedError("the HIP path implements self-attention (query is key is value), "
                                      "which is the only form the model uses (src/transformer.py:347-353)")
        bsz, seq, _ = query.shape
        cos, sin = self.rotary_emb.half_tables(query, seq)
        key_valid = None
        if key_padding_mask is not None:
            # the model converts the mask once per forward and hangs the result on the tensor
            key_valid = getattr(key_padding_mask, "_ldm_key_valid", None)
            if key_valid is None:
                key_valid = _key_valid_bytes(key_padding_mask)
        # the rotary embedding and q / sqrt(hd) (src/transformer.py:255-261) ride in the projection's
        # epilogue where the kernel takes the shape; otherwise the attention call rotates in place
        rope = ops.RopeSpec(cos, sin, self.rotary_emb.step_table(query, seq), seq, self.head_dim,
                            1.0 / math.sqrt(self.head_dim))
        qkv = ops.packed_linear(query, self._pack("qkv"), rope=rope)
        # dropout on the attention probabilities (src/transformer.py:278) happens inside the kernels
        p_drop = self.dropout.p if self.training else 0.0
        # attn_mask (src/transformer.py:267-268; the model passes none, :435) goes to the fp32 kernels
        attn = ops.rope_attention(qkv, key_valid, bsz, seq, self.nhead, self.head_dim, cos, sin, dropout_p=p_drop,
                                  pre_rotated=rope.applied, attn_mask=attn_mask)
        return ops.packed_linear(attn, self._pack("out"), residual=residual, residual_dropout=residual_dropout)


def _key_valid_bytes(key_padding_mask: Tensor) -> Tensor:
    """True-means-padding mask [B,S] -> the uint8 'key is valid' bytes the attention kernels read."""
    return (~key_padding_mask.bool()).to(torch.uint8).contiguous()


class TransformerEncoderLayer(nn.Module):
    """Pre-norm block: x + attn(norm1(x)); x + swiglu(norm2(x))  (src/transformer.py:346-361)."""

    def __init__(self, d_model: int, nhead: int, dim_feedforward: int, dropout: float = 0.1,
                 layer_norm_eps: float = 1e-6):
        super().__init__()
        self.self_attn = RotaryMultiheadAttention(d_model, nhead, dropout)
        self.norm1 = RMSNorm(d_model, eps=layer_norm_eps)
        self.norm2 = RMSNorm(d_model, eps=layer_norm_eps)
        self.swiglu = SwiGLU(d_model, dim_feedforward, d_model)
        self.dropout = nn.Dropout(dropout)
        self.fp32_residual = False      # see DiffusionTransformer.fp32_residual

    def forward(self, src: Tensor, src_mask: Tensor | None = None,
                src_key_padding_mask: Tensor | None = None) -> Tensor:
        if self.fp32_residual and src.dtype == torch.float32 and self.self_attn.compute_dtype != torch.float32:
            # fp32 residual stream around bf16 sublayers, as the reference runs under bf16 autocast (its
            # RMSNorm and the additions of src/transformer.py:346-361 stay in fp32, only the Linears and the
            # attention products are cast): norm of the fp32 stream, sublayer in bf16, fp32 addition
            # (the additions promote the bf16 sublayer output to the stream's fp32 in the same pass)
            dt = self.self_attn.compute_dtype
            q, src = self.norm1.with_residual(src, dt)
            h = self.self_attn(q, q, q, attn_mask=src_mask, key_padding_mask=src_key_padding_mask)
            src = src + self.dropout(h)
            q, src = self.norm2.with_residual(src, dt)
            return src + self.dropout(self.swiglu(q))
        if self.training and self.dropout.p > 0:
            # the two residual dropouts (src/transformer.py:352,361), each fused with its addition
            # and with the sublayer's last projection: the mask is applied in that GEMM's epilogue
            p = self.dropout.p
            src_norm, src = self.norm1.with_residual(src)
            src = self.self_attn(src_norm, src_norm, src_norm, attn_mask=src_mask,
                                 key_padding_mask=src_key_padding_mask, residual=src, residual_dropout=p)
            src_norm, src = self.norm2.with_residual(src)
            return self.swiglu(src_norm, residual=src, residual_dropout=p)
        # residual add fused into the projection epilogue (forward) and into the norm's
        # backward kernel (gradient of the residual stream)
        src_norm, src = self.norm1.with_residual(src)
        src = self.self_attn(src_norm, src_norm, src_norm, attn_mask=src_mask,
                             key_padding_mask=src_key_padding_mask, residual=src)
        src_norm, src = self.norm2.with_residual(src)
        return self.swiglu(src_norm, residual=src)


class DiffusionTransformer(nn.Module):
    """Embedding -> L x encoder layer -> RMSNorm -> biased LM head (src/transformer.py:364-444)."""

    def __init__(self, config: DiffusionTransformerConfig):
        super().__init__()
        self.config = config
        self.embedding = nn.Embedding(config.vocab_size, config.d_model)
        self.layers = nn.ModuleList([
            TransformerEncoderLayer(d_model=config.d_model, nhead=config.nhead,
                                    dim_feedforward=config.dim_feedforward, dropout=config.dropout,
                                    layer_norm_eps=config.layer_norm_eps)
            for _ in range(config.num_layers)
        ])
        self.norm = RMSNorm(config.d_model, eps=config.layer_norm_eps)
        self.output_layer = nn.Linear(config.d_model, config.vocab_size)
        self._init_parameters()
        # The reference builds its rotary tables with the constructor default of 2048
        # positions and never reads config.max_position_embeddings
        # (src/transformer.py:217); longer configs are accepted here.
        for layer in self.layers:
            layer.self_attn.rotary_emb.max_seq_len = max(2048, int(config.max_position_embeddings))
        self.compute_dtype = _default_dtype()
        self.hot_token_id = 103   # embedding-gradient rows of this id are pre-reduced per workgroup
        self._packs: dict = {}
        # True: keep only each layer's INPUT for backward and re-run the layer's forward there
        # (torch.utils.checkpoint): activation memory per layer drops from ~19 [T, d]-sized tensors to one,
        # for one extra forward pass.  Meant for the 8B shape at long batch; off by default.
        self.recompute_layers = False
        # True (with a bf16 compute dtype): the residual stream -- embedding output, the two additions per layer,
        # the norms' inputs and outputs -- is kept in fp32 and only the projections and the attention run in
        # bf16, which is what the reference does under bf16 autocast.  The default keeps the stream in bf16 (the
        # additions ride in the projection epilogues); at 16 layers that costs up to 2.5 x the reference's own
        # bf16 deviation on single deep-layer gradient tensors, while loss and whole gradient stay within 2e-2.
        # The fp32 stream runs its additions and casts as separate passes.
        self.fp32_residual = False

    def _init_parameters(self) -> None:
        """Xavier-uniform on every parameter with dim() > 1 (src/transformer.py:407-411)."""
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def set_compute_dtype(self, dtype: torch.dtype) -> "DiffusionTransformer":
        N.dtype_code(dtype)
        self.compute_dtype = dtype
        for m in self.modules():
            if isinstance(m, (SwiGLU, RotaryMultiheadAttention)):
                m.compute_dtype = dtype
        return self

    def _head_pack(self) -> ops.PackedWeights:
        key = self.compute_dtype
        if key not in self._packs:
            # padded to whole 256-column tiles (30522 -> 30720): the head's products then run on the four-wave
            # GEMM; the padding columns are zero weights and zero bias, and the loss kernels skip them
            self._packs[key] = ops.PackedWeights([self.output_layer.weight], [self.output_layer.bias], key, "head",
                                                 pad=256)
        return self._packs[key]

    def hidden_states(self, input_ids: Tensor, attention_mask: Tensor | None = None) -> Tensor:
        """Everything of forward() up to and including the final norm: [B,S,d]."""
        stream32 = self.fp32_residual and self.compute_dtype != torch.float32
        x = ops.embedding(input_ids, self.embedding.weight, torch.float32 if stream32 else self.compute_dtype,
                          self.hot_token_id)
        for layer in self.layers:
            layer.fp32_residual = stream32
        key_padding_mask = None
        if attention_mask is not None:
            key_padding_mask = attention_mask == 0
            key_padding_mask._ldm_key_valid = _key_valid_bytes(key_padding_mask)   # shared by all layers
        if self.recompute_layers and torch.is_grad_enabled() and self.training:
            from torch.utils.checkpoint import checkpoint
            key_valid = None if key_padding_mask is None else key_padding_mask._ldm_key_valid

            def run(layer, x_in):
                kpm = None
                if key_padding_mask is not None:      # tensor attributes do not survive the checkpoint's re-run
                    kpm = attention_mask == 0
                    kpm._ldm_key_valid = key_valid
                return layer(x_in, src_key_padding_mask=kpm)
            # preserve_rng_state: every dropout seed of a layer (attention probabilities, the two residual
            # dropouts) is drawn from torch's CPU generator inside ``run``; the re-run in backward regenerates
            # the SAME masks only because checkpoint restores the generator state it recorded in forward
            for layer in self.layers:
                x = checkpoint(run, layer, x, use_reentrant=False, preserve_rng_state=True)
        else:
            for layer in self.layers:
                x = layer(x, src_key_padding_mask=key_padding_mask)
        # (the head and its fused loss take the compute dtype)
        return self.norm(x, self.compute_dtype) if stream32 else self.norm(x)

    # Accelerate wraps the reference's forward so that the bf16 logits of its autocast Linear come back as
    # fp32 (``convert_outputs_to_fp32``): the same values, up-cast after the fact.  ``fp32_logits = True``
    # does exactly that here (one more pass over [B,S,V]; the trainer's fused head loss never forms them).
    fp32_logits: bool = False

    def forward(self, input_ids: Tensor, attention_mask: Tensor | None = None) -> Tensor:
        """input_ids [B,S] (+ attention_mask [B,S]) -> logits [B,S,V] in compute_dtype (fp32 with
        ``fp32_logits``).

        The logits rows are stored with a stride padded to 64 columns; the returned
        tensor is the [B,S,V] view of that buffer."""
        logits = ops.packed_linear(self.hidden_states(input_ids, attention_mask), self._head_pack())
        if self.fp32_logits and logits.dtype != torch.float32:
            return logits.float()
        return logits

    def masked_token_loss(self, masked_input_ids: Tensor, targets: Tensor, mask_token_id: int,
                          attention_mask: Tensor | None = None, *, seq_weight: Tensor | None = None,
                          normalize_by: str = "masked") -> Tensor:
        """variational_lower_bound_loss(self(masked_input_ids, attention_mask), targets, ...)
        (src/loss.py:31-65 on src/transformer.py:413-444) without the logits of positions the
        loss does not read; falls back to exactly that composition when the fused kernels do
        not take the problem.  ``seq_weight`` / ``normalize_by``: the keyword-only extensions of
        variational_lower_bound_loss (per-sequence weight, "masked" or "tokens" normaliser)."""
        from .loss import variational_lower_bound_loss
        if normalize_by not in ("masked", "tokens"):
            raise ValueError(f"normalize_by must be 'masked' or 'tokens', got {normalize_by!r}")
        x = self.hidden_states(masked_input_ids, attention_mask)
        pack = self._head_pack()
        x2 = x.reshape(-1, x.shape[-1])
        if ops.head_loss_eligible(x2, pack):
            att_valid = (attention_mask == 1) if attention_mask is not None else None
            return ops.head_masked_ce(x2, pack, targets, masked_input_ids, mask_token_id, att_valid,
                                      seq_weight=seq_weight, seq=masked_input_ids.shape[-1],
                                      norm_mode=0 if normalize_by == "masked" else 1)
        logits = ops.packed_linear(x, pack)
        return variational_lower_bound_loss(logits=logits, targets=targets, masked_input_ids=masked_input_ids,
                                            mask_token_id=mask_token_id, attention_mask=attention_mask,
                                            seq_weight=seq_weight, normalize_by=normalize_by)


__all__ = ["DiffusionTransformerConfig", "RMSNorm", "SwiGLU", "rotate_half", "RotaryPositionEmbedding",
           "RotaryMultiheadAttention", "TransformerEncoderLayer", "DiffusionTransformer", "F"]
