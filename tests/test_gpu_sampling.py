"""Reverse process on a real MI355X: token choice and remasking kernels against the CPU oracle
and the reference-generated fixtures (tests/golden/sampling.npz), through the drop-in modules.

Greedy choice, remasking (with the reference's recorded permutations) and the greedy sample loop
must be bit-exact.  The temperature > 0 choice draws from the kernel's own Philox stream: it is
checked against the oracle's restatement of that stream and against the softmax distribution; with
respect to the reference's torch.multinomial stream it is parity-unpinned.
"""
import types

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _sample_tokens(logits, ids, mask_id=103, all_rows=False, temperature=0.0, seed=0, offset=0, want_pred=False):
    from src import _native as N
    V = logits.shape[-1]
    l2 = logits.reshape(-1, V)
    filled = torch.empty_like(ids)
    pred = torch.full_like(ids, -1) if want_pred else None
    N.check(N.lib().ldm_sample_tokens(l2.data_ptr(), N.dtype_code(l2.dtype), ids.numel(), V, l2.stride(0),
                                      ids.data_ptr(), mask_id, int(all_rows), float(temperature), seed, offset,
                                      filled.data_ptr(), N.ptr(pred), N.stream()), "ldm_sample_tokens")
    return (filled, pred) if want_pred else filled


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T,V,pad", [(1, 1, 0), (7, 5, 0), (33, 200, 0), (64, 30522, 6), (5, 1001, 3)])
def test_greedy_choice_is_first_argmax(dev, dtype, T, V, pad):
    torch.manual_seed(T * 1000 + V)
    full = torch.randn(T, V + pad, device=dev).to(dtype)
    logits = full[:, :V]                    # padded row stride, as the LM head writes it
    if V > 3:                               # ties: the first maximum wins
        logits[0, 2] = 100.0
        logits[0, V - 1] = 100.0
        logits[T - 1, :] = 0.25
    ids = torch.randint(1000, 2000, (T,), device=dev)
    ids[::2] = 103
    filled, pred = _sample_tokens(logits, ids, all_rows=True, want_pred=True)
    ref = torch.argmax(logits.float(), -1)
    assert torch.equal(pred, ref)
    assert torch.equal(filled, torch.where(ids == 103, ref, ids))
    filled2, pred2 = _sample_tokens(logits, ids, all_rows=False, want_pred=True)
    assert torch.equal(filled2, filled)
    assert torch.equal(pred2[ids == 103], ref[ids == 103]) and bool((pred2[ids != 103] == -1).all())


def test_temperature_choice_matches_oracle_stream(dev):
    from oracle import sampling_oracle as smp
    torch.manual_seed(3)
    T, V = 40, 777
    logits = torch.randn(T, V, device=dev) * 2.0
    ids = torch.full((T,), 103, dtype=torch.int64, device=dev)
    for temp, seed, offset in [(1.0, 0, 0), (0.8, 12345678901, 7), (0.3, 5, (1 << 33) + 2)]:
        got = _sample_tokens(logits, ids, temperature=temp, seed=seed, offset=offset).cpu()
        scores = smp.token_scores(logits.cpu(), temp, seed, offset)
        # device logf and numpy log differ in the last bits: the pick must be (within 1e-4 of) the best score
        best = scores.max(-1).values
        assert float((best - scores[torch.arange(T), got]).max()) < 1e-4
        assert float((torch.argmax(scores, -1) == got).float().mean()) > 0.9


def test_temperature_choice_follows_softmax(dev):
    logits1 = torch.tensor([2.0, 1.0, 0.0, -1.0, 0.5, 0.5, -3.0], device=dev)
    rows = 20000
    logits = logits1.repeat(rows, 1).contiguous()
    ids = torch.full((rows,), 103, dtype=torch.int64, device=dev)
    got = _sample_tokens(logits, ids, temperature=0.7, seed=99, offset=1)
    freq = torch.bincount(got, minlength=7).double().cpu()
    p = torch.softmax(logits1.double().cpu() / 0.7, 0)
    chi2 = float(((freq - rows * p) ** 2 / (rows * p)).sum())
    assert chi2 < 30.0, chi2
    # another call offset gives another stream
    again = _sample_tokens(logits, ids, temperature=0.7, seed=99, offset=2)
    assert float((again == got).float().mean()) < 0.9


class _Perms:
    """Stands in for torch.randperm: returns the recorded permutations in call order."""

    def __init__(self, perms):
        self.perms = list(perms)

    def __call__(self, n, device=None, **kw):
        p = self.perms.pop(0)
        assert p.numel() == n
        return p.to(device) if device is not None else p


def test_remask_matches_reference(dev, monkeypatch):
    from src.masking_strategy import RandomRemaskingStrategy, create_remasking_strategy
    g = load_golden("sampling.npz")
    rs = create_remasking_strategy("random", 101, 102, 0, 103)
    assert isinstance(rs, RandomRemaskingStrategy)
    with pytest.raises(ValueError):
        create_remasking_strategy("confidence", 101, 102, 0, 103)
    for j in g["rm_cases"]:
        ids = torch.from_numpy(g[f"rm{j}_ids"]).to(dev)
        before = ids.clone()
        was = torch.from_numpy(g[f"rm{j}_was"]).to(dev)
        prompt = torch.from_numpy(g[f"rm{j}_prompt"]).to(dev) if g[f"rm{j}_prompt"].size else None
        monkeypatch.setattr(torch, "randperm", _Perms([torch.from_numpy(g[f"rm{j}_perm"])]))
        out = rs.apply_remask(input_ids=ids, logits=None, mask_indices=was, remask_ratio=float(g[f"rm{j}_ratio"][0]),
                              prompt_positions=prompt)
        monkeypatch.undo()
        assert np.array_equal(out.cpu().numpy(), g[f"rm{j}_out"]), j
        assert torch.equal(ids, before)            # input untouched (the reference clones, :160)


def test_remask_count_and_eligibility_at_size(dev):
    """Full-size property: exactly int(n * ratio) new mask tokens, all on eligible positions."""
    from oracle import sampling_oracle as smp
    from src.masking_strategy import RandomRemaskingStrategy
    torch.manual_seed(0)
    B, S = 64, 512
    ids = torch.randint(0, 30522, (B, S), device=dev)
    ids[:, 0] = 101
    ids[:, -1] = 102
    ids[ids == 103] = 104
    was = torch.rand(B, S, device=dev) < 0.5
    prompt = torch.zeros(B, S, dtype=torch.bool, device=dev)
    prompt[:, :17] = True
    out = RandomRemaskingStrategy().apply_remask(ids, None, was, 0.37, prompt)
    pos = smp.eligible_positions(ids.cpu(), was.cpu(), prompt.cpu())
    changed = torch.nonzero((out != ids).reshape(-1).cpu()).reshape(-1)
    assert changed.numel() == int(pos.numel() * 0.37)
    assert bool(torch.isin(changed, pos).all()) and bool((out.reshape(-1)[changed.to(dev)] == 103).all())


def _model(g, dev, dtype=torch.float32):
    from src.masking_strategy import create_masking_strategy, create_remasking_strategy
    from src.model import DiffusionLanguageModel
    from src.transformer import DiffusionTransformer, DiffusionTransformerConfig
    V, d, H, L, Fd = [int(x) for x in g["cfg"]]
    tr = DiffusionTransformer(DiffusionTransformerConfig(vocab_size=V, d_model=d, nhead=H, num_layers=L,
                                                         dim_feedforward=Fd, dropout=0.0, layer_norm_eps=1e-12,
                                                         max_position_embeddings=512))
    tr.load_state_dict({k[len("param/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")})
    tr.to(dev).set_compute_dtype(dtype)
    tok = types.SimpleNamespace(mask_token_id=103, bos_token_id=101, eos_token_id=102, pad_token_id=0,
                                decode=lambda seq, skip_special_tokens=False: " ".join(
                                    str(t) for t in seq if not (skip_special_tokens and t in (0, 101, 102, 103))))
    model = DiffusionLanguageModel(tr, tok, create_masking_strategy("random", 101, 102, 0, 103),
                                   create_remasking_strategy("random", 101, 102, 0, 103))
    model.eval()
    return model


def test_generate_matches_reference(dev):
    g = load_golden("sampling.npz")
    model = _model(g, dev)
    ids = torch.from_numpy(g["gen_ids"]).to(dev)
    out = model.generate(ids, temperature=0)
    assert np.array_equal(out.cpu().numpy(), g["gen_out"])
    # with a temperature only the mask positions change, and they leave the mask token
    out_t = model.generate(ids, temperature=1.0)
    keep = ids != 103
    assert torch.equal(out_t[keep], ids[keep])
    model.sampling_rng = "torch"
    out_m = model.generate(ids, temperature=1.0)
    assert torch.equal(out_m[keep], ids[keep])


def test_sample_loop_matches_reference(dev, monkeypatch):
    g = load_golden("sampling.npz")
    model = _model(g, dev)
    for j in range(3):
        prompt = torch.from_numpy(g[f"sm{j}_prompt"]).to(dev) if g[f"sm{j}_prompt"].size else None
        steps, length = [int(x) for x in g[f"sm{j}_steps"]]
        perms = [torch.from_numpy(g[f"sm{j}_perm{i}"]) for i in range(int(g[f"sm{j}_nperm"][0]))]
        monkeypatch.setattr(torch, "randperm", _Perms(perms))
        out = model.sample(prompt=prompt, num_steps=steps, seq_length=length, temperature=0)
        monkeypatch.undo()
        assert np.array_equal(out.cpu().numpy(), g[f"sm{j}_out"]), j
    # sampled run: shape, prompt kept, no mask token left after the last step
    prompt = torch.from_numpy(g["sm0_prompt"]).to(dev)
    out = model.sample(prompt=prompt, num_steps=5, seq_length=20, temperature=0.8)
    keep = prompt != 103        # a mask token inside the prompt is filled in like any other (src/model.py:175-177)
    assert out.shape == (prompt.shape[0], 20) and torch.equal(out[:, :prompt.shape[1]][keep], prompt[keep])
    assert not bool((out == 103).any())


def test_evaluate_reports_examples(dev):
    from src.trainer import DiffusionLanguageModelTrainer
    g = load_golden("sampling.npz")
    model = _model(g, dev)
    torch.manual_seed(1)
    batches = []
    for _ in range(2):
        ids = torch.randint(104, 200, (3, 16), device=dev)
        ids[:, 0] = 101
        ids[:, -1] = 102
        batches.append({"input_ids": ids, "attention_mask": torch.ones_like(ids)})
    tr = DiffusionLanguageModelTrainer(model, None, None, None)
    res = tr.evaluate(batches, num_examples=1, mask_ratio=0.5, sample_steps=4)
    assert set(res) == {"loss", "accuracy", "examples"} and len(res["examples"]) == 1
    assert set(res["examples"][0]) == {"original", "masked", "predicted", "generated"}
    assert res["loss"] > 0 and 0.0 <= res["accuracy"] <= 1.0
