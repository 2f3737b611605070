"""Generate golden vectors from the real reference and pin the oracle to it.

TEST INFRASTRUCTURE ONLY.  Runs in the build container, where the reference
checkout is importable (``/root/reference``); the reference itself never
travels, only the ``.npz`` data written under ``tests/golden/`` does.

    python oracle/make_golden.py [--reference /root/reference]

For every case it (1) runs the reference's own modules
(DiffusionLanguageModel.forward -> variational_lower_bound_loss -> backward),
(2) runs ``oracle/step_oracle.py`` on the same weights / ids / t / draw and
asserts agreement (masking bit-exact, loss+grads <= 1e-6 rel in fp32),
(3) stores inputs and expected outputs.
"""

from __future__ import annotations

import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _load_oracle():
    spec = importlib.util.spec_from_file_location("step_oracle", os.path.join(HERE, "step_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["step_oracle"] = mod
    spec.loader.exec_module(mod)
    return mod


def _load_sampling_oracle():
    """``oracle`` as a package (its modules import each other relatively) without putting the
    repository root, which also holds a package called ``src``, on sys.path."""
    spec = importlib.util.spec_from_file_location("oracle", os.path.join(HERE, "__init__.py"),
                                                  submodule_search_locations=[HERE])
    pkg = importlib.util.module_from_spec(spec)
    sys.modules["oracle"] = pkg
    spec.loader.exec_module(pkg)
    return importlib.import_module("oracle.sampling_oracle")


def _rel(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def build_case(ref, so, name, *, d, H, L, Fd, V, B, S, t, seed, ragged, dirty, out_dir):
    from src.loss import variational_lower_bound_loss
    from src.masking_strategy import create_masking_strategy, create_remasking_strategy
    from src.model import DiffusionLanguageModel
    from src.transformer import DiffusionTransformer, DiffusionTransformerConfig

    torch.manual_seed(seed)
    cfg = DiffusionTransformerConfig(vocab_size=V, d_model=d, nhead=H, num_layers=L,
                                     dim_feedforward=Fd, dropout=0.0, layer_norm_eps=1e-12,
                                     max_position_embeddings=512)
    tr = DiffusionTransformer(cfg)
    # Perturb the RMSNorm gains and give biases some size so that every
    # parameter's gradient path is exercised (ones would hide a missing gain).
    with torch.no_grad():
        for n, p_ in tr.named_parameters():
            if n.endswith("norm1.weight") or n.endswith("norm2.weight") or n == "norm.weight":
                p_.add_(0.1 * torch.randn_like(p_))
    tok = types.SimpleNamespace(mask_token_id=103, bos_token_id=101, eos_token_id=102, pad_token_id=0)
    ms = create_masking_strategy("random", 101, 102, 0, 103)
    rs = create_remasking_strategy("random", 101, 102, 0, 103)
    model = DiffusionLanguageModel(tr, tok, ms, rs)
    model.train()

    ids, att = so.synthetic_batch(B, S, V, seed=seed + 1, ragged=ragged, dirty=dirty)
    draw_seed = seed + 2
    torch.manual_seed(draw_seed)
    draw = torch.rand(ids.shape, dtype=torch.float32)

    results = {}
    for mode in ("fp32", "bf16"):
        model.zero_grad(set_to_none=True)
        torch.manual_seed(draw_seed)
        if mode == "bf16":
            with torch.autocast("cpu", dtype=torch.bfloat16):
                logits, masked, mask = model(input_ids=ids, attention_mask=att, t=torch.tensor([t]))
            logits = logits.float()
        else:
            logits, masked, mask = model(input_ids=ids, attention_mask=att, t=torch.tensor([t]))
        loss = variational_lower_bound_loss(logits=logits, targets=ids, masked_input_ids=masked,
                                            attention_mask=att, mask_token_id=103)
        loss.backward()
        grads = {n: p_.grad.detach().clone() for n, p_ in tr.named_parameters()}
        results[mode] = (loss.detach(), grads, masked, mask, logits.detach())

    # the injected draw reproduces the reference's mask exactly
    t32 = float(torch.tensor([t]).item())
    om, omask = so.forward_mask(ids, t32, draw)
    assert torch.equal(om, results["fp32"][2]) and torch.equal(omask, results["fp32"][3]), "mask draw not reproduced"

    # pin the oracle to the reference
    params = {k: v.detach().clone() for k, v in tr.state_dict().items()}
    ocfg = so.OracleConfig(vocab_size=V, d_model=d, nhead=H, num_layers=L, dim_feedforward=Fd,
                           layer_norm_eps=1e-12)
    for mode in ("fp32", "bf16"):
        ol, og, omk, omsk, ologits = so.step_loss_and_grads(params, ocfg, ids, att, t32, draw,
                                                            autocast_bf16=(mode == "bf16"))
        rl, rg, rmk, rmsk, rlogits = results[mode]
        assert torch.equal(omk, rmk) and torch.equal(omsk, rmsk)
        tol = 1e-6 if mode == "fp32" else 1e-6
        assert _rel(ol, rl) <= tol, (name, mode, "loss", _rel(ol, rl))
        assert _rel(ologits, rlogits) <= tol, (name, mode, "logits", _rel(ologits, rlogits))
        for k in rg:
            e = _rel(og[k], rg[k])
            # k_proj.bias has a mathematically zero gradient (softmax shift invariance)
            if "k_proj.bias" in k:
                assert float((og[k] - rg[k]).abs().max()) < 1e-6
            else:
                assert e <= tol, (name, mode, k, e)
        print(f"[{name}/{mode}] oracle == reference   loss={float(rl):.6f}  M={int(rmsk.sum())}")

    out = {
        "cfg": np.array([V, d, H, L, Fd], dtype=np.int64),
        "input_ids": ids.numpy(), "attention_mask": att.numpy(),
        "t": np.array([t32], dtype=np.float32), "draw": draw.numpy(),
        "masked_input_ids": results["fp32"][2].numpy(),
        "mask_indices": results["fp32"][3].numpy(),
        "logits_fp32": results["fp32"][4].numpy(),
        "loss_fp32": results["fp32"][0].numpy(), "loss_bf16": results["bf16"][0].numpy(),
    }
    for k, v in params.items():
        out["param/" + k] = v.numpy()
    for k, v in results["fp32"][1].items():
        out["grad_fp32/" + k] = v.numpy()
    for k, v in results["bf16"][1].items():
        out["grad_bf16/" + k] = v.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(out_dir, f"step_{name}.npz"), **out)


def build_masking(so, out_dir):
    """Known answers of tests/test_masking_strategy.py:7-49 plus seeded draws
    pushed through the reference's RandomMaskingStrategy."""
    from src.masking_strategy import RandomMaskingStrategy

    st = RandomMaskingStrategy(bos_token_id=101, eos_token_id=102, pad_token_id=0, mask_token_id=103)
    ka_ids = torch.tensor([[101, 450, 231, 102, 0], [101, 450, 231, 450, 102]])
    m1, i1 = st.apply_random_mask(ka_ids, mask_prob=1.0)
    m0, i0 = st.apply_random_mask(ka_ids, mask_prob=0.0)
    exp = torch.tensor([[False, True, True, False, False], [False, True, True, True, False]])
    assert torch.equal(i1, exp) and torch.equal(m0, ka_ids) and not i0.any()
    out = {"ka_ids": ka_ids.numpy(), "ka_masked_p1": m1.numpy(), "ka_mask_p1": i1.numpy(),
           "ka_masked_p0": m0.numpy(), "ka_mask_p0": i0.numpy()}
    for j, (B, S, p, seed) in enumerate([(4, 33, 0.3, 7), (7, 129, 0.77, 8), (2, 1024, 0.5, 9),
                                         (1, 1, 0.5, 10), (5, 64, 0.999999, 11)]):
        ids, _ = so.synthetic_batch(B, S, 30522, seed=seed, ragged=S > 8, dirty=S > 8)
        torch.manual_seed(seed)
        masked, mask = st.apply_random_mask(ids, mask_prob=p)
        torch.manual_seed(seed)
        draw = torch.rand(ids.shape, dtype=torch.float32)
        if j == 2:
            # draws sitting exactly on the fp32-rounded threshold must not mask
            draw[0, 10] = float(np.float32(p))
            draw[0, 11] = float(np.nextafter(np.float32(p), np.float32(0)))
            masked, mask = so.forward_mask(ids, p, draw)
            chk_m = ids.clone()
            elig = (ids != 101) & (ids != 102) & (ids != 0)
            ref_mask = (draw < p) & elig          # torch semantics, reference line 111
            chk_m[ref_mask] = 103
            assert torch.equal(chk_m, masked) and torch.equal(ref_mask, mask)
        om, omask = so.forward_mask(ids, p, draw)
        assert torch.equal(om, masked) and torch.equal(omask, mask)
        out[f"r{j}_ids"] = ids.numpy()
        out[f"r{j}_p"] = np.array([p], dtype=np.float64)
        out[f"r{j}_draw"] = draw.numpy()
        out[f"r{j}_masked"] = masked.numpy()
        out[f"r{j}_mask"] = mask.numpy()
    np.savez_compressed(os.path.join(out_dir, "masking.npz"), **out)
    print("[masking] oracle == reference on known answers and 5 seeded cases")


def build_loss(so, out_dir):
    """tests/test_loss.py:29-67 fixture values through the reference loss, a
    large-vocab case, an attention-masked case and the empty-mask case."""
    from src.loss import masked_cross_entropy_loss, variational_lower_bound_loss

    out = {}
    torch.manual_seed(42)
    B, S, V = 2, 5, 150
    logits = torch.randn(B, S, V)
    targets = torch.randint(0, V - 100, (B, S))
    mask = torch.zeros(B, S, dtype=torch.bool)
    mask[:, 1] = True
    mask[:, 3] = True
    mids = targets.clone()
    mids[mask] = 103
    att = torch.ones(B, S)
    att2 = att.clone()
    att2[:, 3] = 0
    lg = logits.clone().requires_grad_(True)
    l_mce = masked_cross_entropy_loss(lg, targets, mask)
    l_mce.backward()
    out.update(t_logits=logits.numpy(), t_targets=targets.numpy(), t_mask=mask.numpy(),
               t_masked_ids=mids.numpy(), t_att=att.numpy(), t_att2=att2.numpy(),
               t_loss_mce=l_mce.detach().numpy(), t_grad_mce=lg.grad.numpy(),
               t_loss_vlb=variational_lower_bound_loss(logits, targets, mids, 103).detach().numpy(),
               t_loss_vlb_att=variational_lower_bound_loss(logits, targets, mids, 103, att).detach().numpy(),
               t_loss_vlb_att2=variational_lower_bound_loss(logits, targets, mids, 103, att2).detach().numpy(),
               t_loss_empty=masked_cross_entropy_loss(logits, targets, torch.zeros_like(mask)).detach().numpy())
    assert float(out["t_loss_empty"]) == 0.0
    assert abs(float(so.masked_ce(logits, targets, mask)) - float(l_mce)) < 1e-7

    g = torch.Generator().manual_seed(5)
    B, S, V = 2, 6, 30522
    logits = torch.randn(B, S, V, generator=g) * 3.0
    targets = torch.randint(0, V, (B, S), generator=g)
    mids = targets.clone()
    mids[0, 1] = 103
    mids[0, 4] = 103
    mids[1, 2] = 103
    mids[1, 5] = 103
    att = torch.ones(B, S, dtype=torch.int64)
    att[1, 5] = 0
    lg = logits.clone().requires_grad_(True)
    lv = variational_lower_bound_loss(lg, targets, mids, 103, att)
    lv.backward()
    rows = (mids == 103) & (att == 1)
    out.update(v_logits=logits.numpy(), v_targets=targets.numpy(), v_masked_ids=mids.numpy(),
               v_att=att.numpy(), v_loss=lv.detach().numpy(),
               v_grad_rows=lg.grad[rows].numpy(), v_rows=rows.numpy())
    assert abs(float(so.vlb_loss(logits, targets, mids, 103, att)) - float(lv)) < 1e-6
    np.savez_compressed(os.path.join(out_dir, "loss.npz"), **out)
    print("[loss] oracle == reference on the reference-test fixture and a V=30522 case")


def build_sampling(smp, so, out_dir):
    """Reverse process: greedy generate(), apply_remask() and the greedy sample() loop of the
    reference, with the permutations its torch.randperm calls returned recorded next to the
    outputs (src/model.py:74-189, src/masking_strategy.py:140-198)."""
    from src.masking_strategy import create_masking_strategy, create_remasking_strategy
    from src.model import DiffusionLanguageModel
    from src.transformer import DiffusionTransformer, DiffusionTransformerConfig

    V, d, H, L, Fd = 200, 32, 4, 2, 64
    torch.manual_seed(21)
    cfg = DiffusionTransformerConfig(vocab_size=V, d_model=d, nhead=H, num_layers=L, dim_feedforward=Fd,
                                     dropout=0.0, layer_norm_eps=1e-12, max_position_embeddings=512)
    tr = DiffusionTransformer(cfg)
    tok = types.SimpleNamespace(mask_token_id=103, bos_token_id=101, eos_token_id=102, pad_token_id=0)
    model = DiffusionLanguageModel(tr, tok, create_masking_strategy("random", 101, 102, 0, 103),
                                   create_remasking_strategy("random", 101, 102, 0, 103))
    model.eval()
    params = {k: v.detach().clone() for k, v in tr.state_dict().items()}
    ocfg = so.OracleConfig(vocab_size=V, d_model=d, nhead=H, num_layers=L, dim_feedforward=Fd, layer_norm_eps=1e-12)
    out = {"cfg": np.array([V, d, H, L, Fd], dtype=np.int64)}
    for k, v in params.items():
        out["param/" + k] = v.numpy()

    def min_gap(logits, rows):
        top = torch.topk(logits.reshape(-1, V)[rows.reshape(-1)], 2, dim=-1).values
        return float((top[:, 0] - top[:, 1]).min()) if top.numel() else 1.0

    # (1) greedy generate
    ids, _ = so.synthetic_batch(3, 12, V, seed=5, ragged=False, dirty=True)
    g = torch.Generator().manual_seed(6)
    holes = (torch.rand(ids.shape, generator=g) < 0.4) & (ids != 101) & (ids != 102) & (ids != 0)
    ids = torch.where(holes, torch.full_like(ids, 103), ids)
    with torch.no_grad():
        gen = model.generate(ids, temperature=0)
        logits = tr(input_ids=ids, attention_mask=torch.ones_like(ids))
    assert min_gap(logits, ids == 103) > 1e-3
    assert torch.equal(smp.pick_tokens(ids, logits, 103, 0.0), gen)
    assert torch.equal(smp.pick_tokens(ids, so.transformer_forward(params, ocfg, ids, torch.ones_like(ids)), 103, 0.0), gen)
    out.update(gen_ids=ids.numpy(), gen_logits=logits.numpy(), gen_out=gen.numpy())

    # (2) remasking, permutations recorded from the generator stream the reference consumed
    rs = model.remasking_strategy
    cases = []
    for j, (B, S, ratio, with_prompt, seed) in enumerate([(2, 16, 0.5, True, 3), (3, 33, 0.25, False, 4),
                                                          (1, 8, 1.0, True, 5), (2, 16, 0.0, False, 6),
                                                          (2, 9, 0.01, False, 7), (4, 64, 0.77, True, 8)]):
        cur, _ = so.synthetic_batch(B, S, V, seed=seed, ragged=S > 9, dirty=True)
        gg = torch.Generator().manual_seed(seed + 100)
        was = torch.rand(cur.shape, generator=gg) < 0.6
        prompt = None
        if with_prompt:
            prompt = torch.zeros_like(cur, dtype=torch.bool)
            prompt[:, :3] = True
        torch.manual_seed(seed)
        res = rs.apply_remask(input_ids=cur, logits=None, mask_indices=was, remask_ratio=ratio, prompt_positions=prompt)
        n = int(smp.eligible_positions(cur, was, prompt).numel())
        torch.manual_seed(seed)
        perm = torch.randperm(n) if (n > 0 and ratio > 0 and int(n * ratio) > 0) else torch.zeros(0, dtype=torch.long)
        assert torch.equal(smp.apply_remask(cur, was, ratio, prompt, perm), res), j
        out.update({f"rm{j}_ids": cur.numpy(), f"rm{j}_was": was.numpy(), f"rm{j}_ratio": np.array([ratio]),
                    f"rm{j}_prompt": (prompt.numpy() if prompt is not None else np.zeros(0, dtype=bool)),
                    f"rm{j}_perm": perm.numpy(), f"rm{j}_out": res.numpy()})
        cases.append(j)
    out["rm_cases"] = np.array(cases)

    # (3) greedy sample loop; torch.randperm is wrapped for the duration to record what it returned
    orig_randperm = torch.randperm
    for j, (prompt, steps, length, seed) in enumerate([(ids[:2, :4].clone(), 6, 16, 33), (None, 5, 12, 34),
                                                       (ids[:1, :12].clone(), 3, 8, 35)]):
        rec = []

        def recording(*a, **k):
            p_ = orig_randperm(*a, **k)
            rec.append(p_.clone())
            return p_

        torch.randperm = recording
        try:
            torch.manual_seed(seed)
            with torch.no_grad():
                res = model.sample(prompt=prompt, num_steps=steps, seq_length=length, temperature=0)
        finally:
            torch.randperm = orig_randperm
        ores = smp.sample_loop(params, ocfg, prompt, steps, length, rec, temperature=0.0)
        assert torch.equal(ores, res), (j, ores, res)
        out.update({f"sm{j}_prompt": (prompt.numpy() if prompt is not None else np.zeros((0, 0), dtype=np.int64)),
                    f"sm{j}_steps": np.array([steps, length]), f"sm{j}_out": res.numpy(),
                    f"sm{j}_nperm": np.array([len(rec)])})
        for i, p_ in enumerate(rec):
            out[f"sm{j}_perm{i}"] = p_.numpy()
    np.savez_compressed(os.path.join(out_dir, "sampling.npz"), **out)
    print("[sampling] oracle == reference on greedy generate, 6 remasking cases and 3 greedy sample loops")


SCHEDULE_CASES = [
    # create_scheduler(total_steps, warmup_steps, initial_lr, peak_lr, stable_lr, final_lr), optimizer lr
    dict(total_steps=100, warmup_steps=10, initial_lr=0.0, peak_lr=None, stable_lr=None, final_lr=None, lr=5e-4),
    dict(total_steps=37, warmup_steps=2000, initial_lr=0.0, peak_lr=None, stable_lr=None, final_lr=None, lr=5e-4),
    dict(total_steps=64, warmup_steps=0, initial_lr=0.0, peak_lr=4e-4, stable_lr=1e-4, final_lr=1e-5, lr=1e-3),
    dict(total_steps=3, warmup_steps=1, initial_lr=1e-6, peak_lr=None, stable_lr=None, final_lr=None, lr=3e-4),
    dict(total_steps=1000, warmup_steps=333, initial_lr=1e-5, peak_lr=4e-4, stable_lr=None, final_lr=1e-7, lr=2e-4),
]


def build_schedule(out_dir):
    """Learning rates the reference's create_scheduler / WarmupStableDecayLR (src/scheduler.py:11-159)
    hand the optimizer, step by step, for a few splits (incl. a warmup longer than the run and
    phases of zero length), 20 % past the end of the run."""
    from src.scheduler import create_scheduler

    out = {"n_cases": np.array([len(SCHEDULE_CASES)])}
    for j, c in enumerate(SCHEDULE_CASES):
        w = torch.nn.Parameter(torch.zeros(3))
        opt = torch.optim.SGD([w], lr=c["lr"])
        sch = create_scheduler(opt, **{k: v for k, v in c.items() if k != "lr"})
        lrs = [opt.param_groups[0]["lr"]]
        for _ in range(c["total_steps"] + c["total_steps"] // 5 + 2):
            opt.step()
            sch.step()
            lrs.append(opt.param_groups[0]["lr"])
        out[f"s{j}_lrs"] = np.array(lrs, dtype=np.float64)
        out[f"s{j}_phases"] = np.array([sch.warmup_steps, sch.stable_steps, sch.middle_decay_steps,
                                        sch.final_decay_steps], dtype=np.int64)
        out[f"s{j}_levels"] = np.array([sch.initial_lr, sch.peak_lr, sch.stable_lr, sch.final_lr], dtype=np.float64)
    np.savez_compressed(os.path.join(out_dir, "schedule.npz"), **out)
    print(f"[schedule] {len(SCHEDULE_CASES)} learning-rate curves recorded from the reference")


class WordHashTokenizer:
    """Stand-in for the reference's tokenizer wrapper, whose vocabulary needs a download: BOS, one id
    per whitespace-separated word (a hash of its bytes into [1000, 30522)), EOS, cut to
    ``max_length``.  Same class in tests/test_host_logic.py."""

    def encode(self, text, max_length=512, return_tensors=False):
        ids = [101]
        for w in text.split():
            h = 2166136261
            for b in w.encode("utf-8"):
                h = ((h ^ b) * 16777619) & 0xFFFFFFFF
            ids.append(1000 + h % (30522 - 1000))
        ids = ids[:max_length - 1] + [102]
        return {"input_ids": ids, "attention_mask": [1] * len(ids)}


DATASET_LINES = [
    "q1@@what is the capital of france@@paris",
    "",
    "no delimiter on this line",
    "q2@@how many legs does a spider have",
    "   q3@@why@@because@@really   ",
    "q4@@" + " ".join(f"w{i}" for i in range(40)),
    "q5@@",
    "q6@@name three primary colours please@@red green blue",
    "q7@@who wrote hamlet",
    "q8@@is water wet",
    "q9@@a b c d e f g",
    "q10@@last one here",
]


def build_dataset(out_dir):
    """The reference's QuestionDataset / collate_batch / create_question_dataloaders
    (src/dataset.py:26-202) over a 12-line file and the stand-in tokenizer: the parsed questions,
    every item, and the train / validation batches drawn after torch.manual_seed(3)."""
    import tempfile

    from src.dataset import QuestionDataset, create_question_dataloaders

    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "questions.txt")
        with open(path, "w", encoding="utf-8") as f:
            f.write("\n".join(DATASET_LINES) + "\n")
        tok = WordHashTokenizer()
        ds = QuestionDataset(path, tok, max_length=16)
        out = {"lines": np.array(DATASET_LINES), "questions": np.array(ds.questions), "n_items": np.array([len(ds)])}
        for i in range(len(ds)):
            out[f"item{i}_ids"] = ds[i]["input_ids"].numpy()
        torch.manual_seed(3)
        tr, va = create_question_dataloaders(path, tok, batch_size=4, max_length=16, train_ratio=0.75,
                                             shuffle=True, num_workers=0)
        for name, dl in (("train", tr), ("val", va)):
            n = 0
            for b in dl:
                out[f"{name}{n}_ids"] = b["input_ids"].numpy()
                out[f"{name}{n}_mask"] = b["attention_mask"].numpy()
                n += 1
            out[f"n_{name}"] = np.array([n])
    np.savez_compressed(os.path.join(out_dir, "dataset.npz"), **out)
    print(f"[dataset] {len(ds)} questions, {int(out['n_train'][0])} train + {int(out['n_val'][0])} validation batches recorded")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only-sampling", action="store_true", help="rewrite sampling.npz only")
    ap.add_argument("--only-schedule", action="store_true", help="rewrite schedule.npz only")
    ap.add_argument("--only-dataset", action="store_true", help="rewrite dataset.npz only")
    args = ap.parse_args()
    so = _load_oracle()
    # the reference's package is called ``src``; import it from ITS checkout
    sys.path = [args.reference] + [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
    os.chdir(args.reference)
    os.makedirs(args.out, exist_ok=True)
    smp = _load_sampling_oracle()
    if "--only-sampling" in sys.argv:
        build_sampling(smp, so, args.out)
        return
    if args.only_schedule:
        build_schedule(args.out)
        return
    if args.only_dataset:
        build_dataset(args.out)
        return
    build_masking(so, args.out)
    build_loss(so, args.out)
    build_schedule(args.out)
    build_dataset(args.out)
    build_sampling(smp, so, args.out)
    build_case(None, so, "c1_tiny", d=32, H=4, L=2, Fd=64, V=200, B=3, S=10, t=0.6, seed=42,
               ragged=True, dirty=True, out_dir=args.out)
    build_case(None, so, "hd96", d=192, H=2, L=1, Fd=128, V=300, B=2, S=48, t=0.5, seed=7,
               ragged=True, dirty=True, out_dir=args.out)
    build_case(None, so, "nomask_att", d=64, H=2, L=2, Fd=96, V=256, B=2, S=70, t=0.35, seed=11,
               ragged=False, dirty=False, out_dir=args.out)


if __name__ == "__main__":
    main()
