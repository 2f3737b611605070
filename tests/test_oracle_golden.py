"""The CPU oracle reproduces the reference-generated golden vectors (CPU only)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import step_oracle as so


def test_masking_known_answers_and_seeded_cases():
    g = load_golden("masking.npz")
    ids = torch.from_numpy(g["ka_ids"])
    # reference tests/test_masking_strategy.py:22-49: p=1 masks every eligible token, p=0 none
    m, i = so.forward_mask(ids, 1.0, torch.rand(ids.shape))
    assert np.array_equal(m.numpy(), g["ka_masked_p1"]) and np.array_equal(i.numpy(), g["ka_mask_p1"])
    m, i = so.forward_mask(ids, 0.0, torch.rand(ids.shape))
    assert np.array_equal(m.numpy(), g["ka_masked_p0"]) and not i.any()
    for j in range(5):
        ids = torch.from_numpy(g[f"r{j}_ids"])
        m, i = so.forward_mask(ids, float(g[f"r{j}_p"][0]), torch.from_numpy(g[f"r{j}_draw"]))
        assert np.array_equal(m.numpy(), g[f"r{j}_masked"]), j
        assert np.array_equal(i.numpy(), g[f"r{j}_mask"]), j


def test_loss_golden():
    g = load_golden("loss.npz")
    logits = torch.from_numpy(g["t_logits"])
    targets = torch.from_numpy(g["t_targets"])
    assert abs(float(so.masked_ce(logits, targets, torch.from_numpy(g["t_mask"]))) - float(g["t_loss_mce"])) < 1e-6
    mids = torch.from_numpy(g["t_masked_ids"])
    assert abs(float(so.vlb_loss(logits, targets, mids, 103)) - float(g["t_loss_vlb"])) < 1e-6
    assert abs(float(so.vlb_loss(logits, targets, mids, 103, torch.from_numpy(g["t_att2"]))) - float(g["t_loss_vlb_att2"])) < 1e-6
    assert float(so.masked_ce(logits, targets, torch.zeros(2, 5, dtype=torch.bool))) == 0.0
    lv = so.vlb_loss(torch.from_numpy(g["v_logits"]), torch.from_numpy(g["v_targets"]),
                     torch.from_numpy(g["v_masked_ids"]), 103, torch.from_numpy(g["v_att"]))
    assert abs(float(lv) - float(g["v_loss"])) < 1e-5


@pytest.mark.parametrize("name", ["c1_tiny", "hd96", "nomask_att"])
def test_step_golden(name):
    g = load_golden(f"step_{name}.npz")
    V, d, H, L, Fd = [int(x) for x in g["cfg"]]
    cfg = so.OracleConfig(vocab_size=V, d_model=d, nhead=H, num_layers=L, dim_feedforward=Fd)
    params = {k[len("param/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")}
    ids = torch.from_numpy(g["input_ids"])
    att = torch.from_numpy(g["attention_mask"])
    # fp32 must reproduce the recorded reference run to rounding.  The bf16-autocast run goes through
    # oneDNN's bf16 matmul, whose accumulation order depends on the host CPU (AVX512-BF16 / AMX or not):
    # a rounding of one bf16 product that flips moves the small q/k gradients by a few 1e-3, so that mode
    # is held to bf16's own resolution, not to fp32's.
    for mode in ("fp32", "bf16"):
        ltol, gtol = (1e-6, 2e-6) if mode == "fp32" else (1e-4, 1e-2)
        loss, grads, masked, mask, logits = so.step_loss_and_grads(
            params, cfg, ids, att, float(g["t"][0]), torch.from_numpy(g["draw"]), autocast_bf16=(mode == "bf16"))
        assert np.array_equal(masked.numpy(), g["masked_input_ids"])
        assert np.array_equal(mask.numpy(), g["mask_indices"])
        assert rel_err(loss, g[f"loss_{mode}"]) < ltol
        if mode == "fp32":
            assert rel_err(logits, g["logits_fp32"]) < 1e-6
        # k_proj.bias has a gradient of (mathematically) zero: its absolute error is held to gtol times the
        # size of the query bias gradients of the same run, not to a bound that any value would pass
        qb = max(float(np.abs(g[f"grad_{mode}/{k}"]).max()) for k in grads if "q_proj.bias" in k)
        num = den = 0.0
        for k, v in grads.items():
            ref = g[f"grad_{mode}/{k}"]
            if "k_proj.bias" in k:
                assert float(np.abs(v.numpy() - ref).max()) < gtol * qb, (mode, k)
            else:
                assert rel_err(v, ref) < gtol, (mode, k)
                num += float(((v.numpy().astype(np.float64) - ref) ** 2).sum())
                den += float((ref.astype(np.float64) ** 2).sum())
        # the whole gradient is held tighter than any single tensor
        assert (num / den) ** 0.5 < (2e-6 if mode == "fp32" else 3e-3), mode


def test_sampling_golden():
    """Reverse process: greedy token choice, remasking with the recorded permutations and the
    greedy sample loop reproduce what the reference returned (fixtures of oracle/make_golden.py)."""
    from oracle import sampling_oracle as smp
    g = load_golden("sampling.npz")
    ids = torch.from_numpy(g["gen_ids"])
    assert np.array_equal(smp.pick_tokens(ids, torch.from_numpy(g["gen_logits"]), 103, 0.0).numpy(), g["gen_out"])
    for j in g["rm_cases"]:
        prompt = torch.from_numpy(g[f"rm{j}_prompt"]) if g[f"rm{j}_prompt"].size else None
        out = smp.apply_remask(torch.from_numpy(g[f"rm{j}_ids"]), torch.from_numpy(g[f"rm{j}_was"]),
                               float(g[f"rm{j}_ratio"][0]), prompt, torch.from_numpy(g[f"rm{j}_perm"]))
        assert np.array_equal(out.numpy(), g[f"rm{j}_out"]), j
    V, d, H, L, Fd = [int(x) for x in g["cfg"]]
    cfg = so.OracleConfig(vocab_size=V, d_model=d, nhead=H, num_layers=L, dim_feedforward=Fd)
    params = {k[len("param/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param/")}
    for j in range(3):
        prompt = torch.from_numpy(g[f"sm{j}_prompt"]) if g[f"sm{j}_prompt"].size else None
        steps, length = [int(x) for x in g[f"sm{j}_steps"]]
        perms = [torch.from_numpy(g[f"sm{j}_perm{i}"]) for i in range(int(g[f"sm{j}_nperm"][0]))]
        out = smp.sample_loop(params, cfg, prompt, steps, length, perms)
        assert np.array_equal(out.numpy(), g[f"sm{j}_out"]), j


def test_gumbel_choice_follows_softmax():
    """The temperature > 0 choice of the sampling oracle (argmax of logits/T + Philox Gumbel
    noise) is distributed as softmax(logits/T): chi-square against the exact probabilities."""
    from oracle import sampling_oracle as smp
    logits = torch.tensor([2.0, 1.0, 0.0, -1.0, 0.5, 0.5, -3.0])
    rows = 20000
    picks = torch.argmax(smp.token_scores(logits.repeat(rows, 1), 0.7, seed=11, offset=3), -1)
    freq = torch.bincount(picks, minlength=7).double()
    p = torch.softmax(logits.double() / 0.7, 0)
    chi2 = float(((freq - rows * p) ** 2 / (rows * p)).sum())
    assert chi2 < 30.0, chi2   # 6 degrees of freedom: P(chi2 > 30) ~ 4e-5


def test_adamw_oracle_matches_torch_adamw():
    """oracle/adamw_oracle.py against torch.optim.AdamW (the optimizer of main.py:87-91) on CPU."""
    import numpy as np
    import torch
    from oracle.adamw_oracle import adamw_step
    torch.manual_seed(3)
    shapes = [(7, 5), (33,), (1,), (4, 3, 2)]
    ps = [torch.randn(*s).requires_grad_(True) for s in shapes]
    opt = torch.optim.AdamW(ps, lr=5e-4, weight_decay=0.1)
    mine = [(p.detach().numpy().copy(), np.zeros(s, np.float32), np.zeros(s, np.float32)) for p, s in zip(ps, shapes)]
    for step in range(1, 8):
        gs = [torch.randn(*s) * (0.1 + step) for s in shapes]
        for p, g in zip(ps, gs):
            p.grad = g.clone()
        opt.step()
        mine = [adamw_step(p, g.numpy(), m, v, step, lr=5e-4, weight_decay=0.1) for (p, m, v), g in zip(mine, gs)]
        for p, (q, _, _) in zip(ps, mine):
            assert np.allclose(p.detach().numpy(), q, rtol=2e-6, atol=1e-8), step


def test_synth_token_oracle_contract():
    """oracle/philox.py synth_tokens: the batch contract (src/dataset.py:48) and determinism."""
    import numpy as np
    from oracle.philox import synth_tokens
    ids, mask = synth_tokens(6, 50, 1000, 30522, 101, 102, 0, 7, seed=11, offset=3)
    ids2, mask2 = synth_tokens(6, 50, 1000, 30522, 101, 102, 0, 7, seed=11, offset=3)
    assert np.array_equal(ids, ids2) and np.array_equal(mask, mask2)
    assert ids.dtype == np.int64 and mask.dtype == np.int64 and ids.shape == (6, 50)
    lens = mask.sum(1)
    assert (lens >= 7).all() and (lens <= 50).all() and len(set(lens.tolist())) > 1
    assert (ids[:, 0] == 101).all() and (ids[np.arange(6), lens - 1] == 102).all()
    assert (ids[mask == 0] == 0).all()
    full, fmask = synth_tokens(2, 9, 1000, 30522, 101, 102, 0, 9, seed=11, offset=0)
    assert fmask.all() and (full[:, -1] == 102).all()


def test_dropout_mask_oracle_properties():
    """The dropout-mask restatement: deterministic in (seed, coordinates), keep rate 1 - p, rows and
    heads decorrelated, threshold / scale arithmetic of the C side."""
    from oracle import dropout_oracle as do
    assert do.threshold(0.0) == 0 and do.threshold(0.5) == 2 ** 31 and do.threshold(0.99999999) <= 2 ** 32 - 1
    assert abs(do.scale(0.1) - 1.0 / (1.0 - do.threshold(0.1) / 2.0 ** 32)) < 1e-6
    m = do.keep_mask(0.1, 12345, 6, 64, 128)
    assert m.shape == (6, 64, 128) and m.dtype == bool
    assert np.array_equal(m, do.keep_mask(0.1, 12345, 6, 64, 128))
    assert not np.array_equal(m, do.keep_mask(0.1, 12346, 6, 64, 128))
    assert abs(m.mean() - 0.9) < 0.01
    # no two (batch*head, query) rows share a mask, and the per-key keep rate is flat
    rows = {r.tobytes() for r in m.reshape(-1, 128)}
    assert len(rows) == 6 * 64
    assert np.all(np.abs(m.reshape(-1, 128).mean(0) - 0.9) < 0.08)
    assert do.keep_mask(0.0, 1, 2, 8, 8).all()
