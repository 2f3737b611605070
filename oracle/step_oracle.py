"""CPU restatement of the reference's masked-diffusion training step.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  This file restates, as
plain functional torch-CPU code over a flat parameter dict, the arithmetic of
the reference hot path.  It is pinned against the real reference by
``oracle/make_golden.py`` (run in the build container, where the reference is
importable) and by the committed fixtures under ``tests/golden/``.

Parameter names are the reference's ``state_dict`` keys
(``embedding.weight``, ``layers.{i}.self_attn.q_proj.weight`` ...), so a
reference ``state_dict`` can be fed in unchanged.

Every function cites the reference lines it follows (paths are relative to the
reference checkout).
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

BOS, EOS, PAD, MASK = 101, 102, 0, 103


@dataclass
class OracleConfig:
    """Mirror of the hyper-parameters of src/transformer.py:13-24."""

    vocab_size: int = 30522
    d_model: int = 768
    nhead: int = 8
    num_layers: int = 8
    dim_feedforward: int = 1024
    layer_norm_eps: float = 1e-12
    rope_base: float = 10000.0


# --------------------------------------------------------------------------
# forward masking
# --------------------------------------------------------------------------
def forward_mask(input_ids, t, draw, bos=BOS, eos=EOS, pad=PAD, mask_id=MASK):
    """Forward masking with an injected uniform draw.

    src/masking_strategy.py:104-114: eligible = id not in {bos,eos,pad};
    mask = (rand < mask_prob) & eligible; masked = where(mask, mask_id, id).
    ``t`` is a python float (the reference form, one scalar per call,
    src/model.py:59-68) or a per-sequence fp32 tensor [B] (extension).
    The comparison happens in fp32: torch rounds a python scalar to the
    tensor's dtype before ``<``.
    """
    ids = input_ids
    draw = draw.to(torch.float32)
    if isinstance(t, torch.Tensor):
        thr = t.to(torch.float32).reshape(-1, 1)
    else:
        thr = torch.tensor(float(t), dtype=torch.float32)
    eligible = (ids != bos) & (ids != eos) & (ids != pad)
    mask = (draw < thr) & eligible
    masked = torch.where(mask, torch.full_like(ids, mask_id), ids)
    return masked, mask


# --------------------------------------------------------------------------
# transformer forward
# --------------------------------------------------------------------------
def rms_norm(x, weight, eps):
    """src/transformer.py:54-58: w * x / sqrt(mean(x^2) + eps)."""
    denom = torch.sqrt(torch.mean(x ** 2, dim=-1, keepdim=True) + eps)
    return weight * (x / denom)


def rope_tables(seq_len, head_dim, base=10000.0):
    """src/transformer.py:123,150-156: cos/sin of cat(freqs, freqs), fp32."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    pos = torch.arange(seq_len).type_as(inv_freq)
    freqs = torch.einsum("i,j->ij", pos, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def _rot_half(x):
    """src/transformer.py:98-101: cat(-x2, x1) over the two contiguous halves."""
    a, b = x.chunk(2, dim=-1)
    return torch.cat((-b, a), dim=-1)


def attention(p, prefix, x, key_padding_mask, cfg):
    """src/transformer.py:239-291 (dropout 0)."""
    bsz, seq, _ = x.shape
    nh = cfg.nhead
    hd = cfg.d_model // nh
    q = F.linear(x, p[prefix + "q_proj.weight"], p[prefix + "q_proj.bias"])
    k = F.linear(x, p[prefix + "k_proj.weight"], p[prefix + "k_proj.bias"])
    v = F.linear(x, p[prefix + "v_proj.weight"], p[prefix + "v_proj.bias"])
    q = q.view(bsz, seq, nh, hd)
    k = k.view(bsz, seq, nh, hd)
    v = v.view(bsz, seq, nh, hd)
    cos, sin = rope_tables(seq, hd, cfg.rope_base)
    cos = cos[None, :, None, :]
    sin = sin[None, :, None, :]
    q = (q * cos) + (_rot_half(q) * sin)
    k = (k * cos) + (_rot_half(k) * sin)
    q = q.transpose(1, 2)
    k = k.transpose(1, 2)
    v = v.transpose(1, 2)
    q = q / math.sqrt(hd)
    scores = torch.matmul(q, k.transpose(-2, -1))
    if key_padding_mask is not None:
        scores = scores.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    probs = F.softmax(scores, dim=-1)
    out = torch.matmul(probs, v)
    out = out.transpose(1, 2).contiguous().view(bsz, seq, cfg.d_model)
    return F.linear(out, p[prefix + "out_proj.weight"], p[prefix + "out_proj.bias"])


def swiglu(p, prefix, x):
    """src/transformer.py:91-95: w3( w1(x) * silu(w2(x)) ) -- the gate is w2."""
    a = F.linear(x, p[prefix + "w1.weight"], p[prefix + "w1.bias"])
    g = F.linear(x, p[prefix + "w2.weight"], p[prefix + "w2.bias"])
    return F.linear(a * F.silu(g), p[prefix + "w3.weight"], p[prefix + "w3.bias"])


def transformer_forward(p, cfg, input_ids, attention_mask=None):
    """src/transformer.py:413-444 with the layer body of :346-361 (dropout 0)."""
    x = F.embedding(input_ids, p["embedding.weight"])
    kpm = None
    if attention_mask is not None:
        kpm = attention_mask == 0
    for i in range(cfg.num_layers):
        pre = f"layers.{i}."
        h = rms_norm(x, p[pre + "norm1.weight"], cfg.layer_norm_eps)
        x = x + attention(p, pre + "self_attn.", h, kpm, cfg)
        h = rms_norm(x, p[pre + "norm2.weight"], cfg.layer_norm_eps)
        x = x + swiglu(p, pre + "swiglu.", h)
    x = rms_norm(x, p["norm.weight"], cfg.layer_norm_eps)
    return F.linear(x, p["output_layer.weight"], p["output_layer.bias"])


# --------------------------------------------------------------------------
# loss
# --------------------------------------------------------------------------
def masked_ce(logits, targets, mask):
    """src/loss.py:12-28: mean CE over masked rows; empty mask -> constant 0."""
    v = logits.shape[-1]
    lf = logits.reshape(-1, v)
    tf = targets.reshape(-1)
    mf = mask.reshape(-1)
    if int(mf.sum()) == 0:
        return torch.tensor(0.0, requires_grad=True)
    return F.cross_entropy(lf[mf], tf[mf])


def vlb_loss(logits, targets, masked_input_ids, mask_token_id=MASK, attention_mask=None):
    """src/loss.py:59-65: mask re-derived from ``== mask_token_id`` & attention."""
    m = masked_input_ids == mask_token_id
    if attention_mask is not None:
        m = m & (attention_mask == 1)
    return masked_ce(logits, targets, m)


def vlb_loss_paper(logits, targets, masked_input_ids, t_seq, mask_token_id=MASK, attention_mask=None):
    """The LLaDA-paper form of the objective -- NOT in the reference (its loss is src/loss.py:20-22:
    unweighted mean over the masked positions): with t_b the masking ratio of sequence b,

        loss = (1 / (B * L)) * sum_b (1 / t_b) * sum_{i in masked(b)} CE(logits[b, i], targets[b, i])

    over the same re-derived mask as ``vlb_loss``.  An empty mask gives 0.  This restates the
    formula the HIP kernels implement for ``objective="paper"``; there is no reference output to
    pin it against (parity unpinned w.r.t. the reference)."""
    m = masked_input_ids == mask_token_id
    if attention_mask is not None:
        m = m & (attention_mask == 1)
    bsz, seq = masked_input_ids.shape
    v = logits.shape[-1]
    ce = F.cross_entropy(logits.reshape(-1, v).float(), targets.reshape(-1), reduction="none").reshape(bsz, seq)
    w = (1.0 / t_seq.to(torch.float32)).reshape(bsz, 1)
    return (ce * w * m.to(ce.dtype)).sum() / float(bsz * seq)


# --------------------------------------------------------------------------
# one step (forward + loss + backward), src/trainer.py:119-137
# --------------------------------------------------------------------------
def step_loss_and_grads(params, cfg, input_ids, attention_mask, t, draw, autocast_bf16=False, objective="reference"):
    """Returns (loss, grads dict, masked ids, mask, logits) for one step.

    ``objective="paper"``: ``t`` is a per-sequence fp32 tensor [B] and the loss is ``vlb_loss_paper``.

    ``params`` is a dict name -> fp32 tensor (no grad required on entry).
    With ``autocast_bf16`` the forward+loss run under
    ``torch.autocast('cpu', torch.bfloat16)``, which is how Accelerate's mixed
    precision wraps the reference forward (accelerate/accelerator.py autocast
    wrapper entered at main.py:105-108).
    """
    leaf = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "inv_freq" not in k)
            for k, v in params.items()}
    masked, mask = forward_mask(input_ids, t, draw)
    if autocast_bf16:
        with torch.autocast("cpu", dtype=torch.bfloat16):
            logits = transformer_forward(leaf, cfg, masked, attention_mask)
        logits = logits.float()
    else:
        logits = transformer_forward(leaf, cfg, masked, attention_mask)
    if objective == "paper":
        loss = vlb_loss_paper(logits, input_ids, masked, t, MASK, attention_mask)
    else:
        loss = vlb_loss(logits, input_ids, masked, MASK, attention_mask)
    grads = {}
    if loss.grad_fn is not None:
        loss.backward()
        grads = {k: v.grad for k, v in leaf.items() if v.requires_grad and v.grad is not None}
    return loss.detach(), grads, masked, mask, logits.detach()


def init_params(cfg, seed=42):
    """Random parameters with the reference's init rule.

    src/transformer.py:407-411: Xavier-uniform on every dim()>1 parameter
    (embedding included); nn.Linear biases keep their default
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)); RMSNorm weights are ones.
    The RNG stream is NOT the reference's (parameter creation order differs);
    parity tests load reference-generated weights from fixtures instead.
    """
    g = torch.Generator().manual_seed(seed)
    d, f, v = cfg.d_model, cfg.dim_feedforward, cfg.vocab_size

    def xavier(o, i):
        a = math.sqrt(6.0 / (o + i))
        return (torch.rand(o, i, generator=g) * 2 - 1) * a

    def bias(o, i):
        a = 1.0 / math.sqrt(i)
        return (torch.rand(o, generator=g) * 2 - 1) * a

    p = {"embedding.weight": xavier(v, d)}
    for l in range(cfg.num_layers):
        pre = f"layers.{l}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            p[pre + f"self_attn.{nm}.weight"] = xavier(d, d)
            p[pre + f"self_attn.{nm}.bias"] = bias(d, d)
        p[pre + "norm1.weight"] = torch.ones(d)
        p[pre + "norm2.weight"] = torch.ones(d)
        p[pre + "swiglu.w1.weight"] = xavier(f, d)
        p[pre + "swiglu.w1.bias"] = bias(f, d)
        p[pre + "swiglu.w2.weight"] = xavier(f, d)
        p[pre + "swiglu.w2.bias"] = bias(f, d)
        p[pre + "swiglu.w3.weight"] = xavier(d, f)
        p[pre + "swiglu.w3.bias"] = bias(d, f)
    p["norm.weight"] = torch.ones(d)
    p["output_layer.weight"] = xavier(v, d)
    p["output_layer.bias"] = bias(v, d)
    return p


def synthetic_batch(batch, seq, vocab, seed=1234, ragged=False, dirty=False):
    """Synthetic tokens: BOS in column 0, EOS last, body uniform in [1000, V).

    ``ragged`` adds pad tails with attention_mask 0; ``dirty`` sprinkles
    special ids (incl. a stray mask id) into the body -- parity-only inputs.
    """
    g = torch.Generator().manual_seed(seed)
    lo = 1000 if vocab > 2000 else 110
    ids = torch.randint(lo, vocab, (batch, seq), generator=g, dtype=torch.int64)
    ids[:, 0] = BOS
    att = torch.ones(batch, seq, dtype=torch.int64)
    if ragged:
        lens = torch.randint(max(3, seq // 2), seq + 1, (batch,), generator=g)
        lens[0] = seq
        for b in range(batch):
            n = int(lens[b])
            ids[b, n - 1] = EOS
            ids[b, n:] = PAD
            att[b, n:] = 0
    else:
        ids[:, -1] = EOS
    if dirty and seq >= 8:
        ids[0, 3] = MASK      # a stray mask id in the ORIGINAL input
        ids[-1, 4] = EOS      # a special id inside the body
        ids[-1, 5] = PAD
    return ids, att
