"""Host-side mirror of the reference interface: names, errors, init rule (CPU only)."""
import inspect

import numpy as np
import pytest
import torch

from conftest import load_golden


def test_factory_and_signatures():
    from src.masking_strategy import RandomMaskingStrategy, create_masking_strategy
    s = create_masking_strategy("random", 101, 102, 0, 103)
    assert isinstance(s, RandomMaskingStrategy)
    assert (s.bos_token_id, s.eos_token_id, s.pad_token_id, s.mask_token_id) == (101, 102, 0, 103)
    with pytest.raises(ValueError, match="Unsupported masking strategy"):
        create_masking_strategy("nope", 101, 102, 0, 103)
    params = list(inspect.signature(s.apply_random_mask).parameters)
    assert params[:2] == ["input_ids", "mask_prob"]
    from src.loss import masked_cross_entropy_loss, variational_lower_bound_loss
    assert list(inspect.signature(masked_cross_entropy_loss).parameters) == ["logits", "targets", "mask_indices"]
    assert list(inspect.signature(variational_lower_bound_loss).parameters)[:5] == [
        "logits", "targets", "masked_input_ids", "mask_token_id", "attention_mask"]


def test_state_dict_keys_and_init_match_reference_fixture():
    from src.transformer import DiffusionTransformer, DiffusionTransformerConfig
    g = load_golden("step_c1_tiny.npz")
    V, d, H, L, Fd = [int(x) for x in g["cfg"]]
    torch.manual_seed(42)   # the fixture's reference model was built right after this seed
    m = DiffusionTransformer(DiffusionTransformerConfig(vocab_size=V, d_model=d, nhead=H, num_layers=L,
                                                        dim_feedforward=Fd, dropout=0.0, layer_norm_eps=1e-12,
                                                        max_position_embeddings=512))
    ref_keys = sorted(k[len("param/"):] for k in g.files if k.startswith("param/"))
    assert sorted(m.state_dict().keys()) == ref_keys
    # same construction order + same init rule => bit-identical weights (the fixture's
    # norm gains were perturbed afterwards, so they are skipped)
    for k, v in m.state_dict().items():
        if "norm" in k:
            continue
        assert np.array_equal(v.numpy(), g["param/" + k]), k


def test_errors_like_reference():
    from src.transformer import RotaryMultiheadAttention, RotaryPositionEmbedding
    with pytest.raises(AssertionError):
        RotaryMultiheadAttention(10, 3)
    r = RotaryPositionEmbedding(8)
    assert r.max_seq_len == 2048
    with pytest.raises(ValueError, match="exceeds maximum length"):
        r(torch.zeros(1), 4096)
    cos, sin = r(torch.zeros(1), 16)
    assert cos.shape == (16, 8) and torch.equal(cos[:, :4], cos[:, 4:])


def test_cpu_tensors_fail_loudly():
    """There is no CPU fallback: ops refuse host tensors instead of silently computing."""
    from src.loss import masked_cross_entropy_loss
    from src.masking_strategy import RandomMaskingStrategy
    from src.transformer import RMSNorm
    with pytest.raises(RuntimeError, match="ROCm device"):
        RandomMaskingStrategy().apply_random_mask(torch.tensor([[101, 5, 102]]), 0.5)
    with pytest.raises(RuntimeError, match="ROCm device"):
        RMSNorm(8)(torch.zeros(2, 8))
    with pytest.raises(RuntimeError, match="ROCm device"):
        masked_cross_entropy_loss(torch.zeros(1, 2, 4), torch.zeros(1, 2, dtype=torch.long),
                                  torch.ones(1, 2, dtype=torch.bool))


def test_weight_copy_epoch_and_loss_value_host_logic():
    """Host logic that needs no GPU: packed weight copies are rebuilt after parameters_changed()
    (optimizers that do not move version counters), and trainer.loss_value falls back to item()."""
    import torch
    from src import ops
    from src.trainer import DiffusionLanguageModelTrainer
    lin = torch.nn.Linear(8, 4)
    pack = ops.PackedWeights([lin.weight], [lin.bias], torch.float32)
    w0, _ = pack.get()
    assert torch.equal(w0[:4], lin.weight.detach())
    with torch.no_grad():
        lin.weight.data_ptr()                      # raw in-place change that leaves _version alone
        torch._foreach_mul_([lin.weight.data], 2.0)
    stale, _ = pack.get()
    v_before = lin.weight._version
    ops.parameters_changed()
    fresh, _ = pack.get()
    assert lin.weight._version == v_before
    assert torch.equal(fresh[:4], lin.weight.detach())
    assert stale is fresh                          # same buffer, refreshed in place

    class _Opt(torch.optim.SGD):
        pass
    opt = _Opt(lin.parameters(), lr=0.1)
    tr = DiffusionLanguageModelTrainer(lin, opt, None, None)
    assert tr._marks_stale                         # the trainer marks copies stale after every step() it makes
    assert tr.loss_value(torch.tensor(1.5)) == 1.5


def test_trainer_accepts_an_accelerate_prepared_optimizer():
    """The reference's main.py hands the trainer what Accelerator.prepare returns.  accelerate's
    AcceleratedOptimizer subclasses torch.optim.Optimizer without running its __init__ (no hook
    registry, and its step() fires no hooks): the trainer must not register hooks on it and must
    mark the packed weight copies stale itself after each step."""
    import pytest
    import torch
    accelerate = pytest.importorskip("accelerate")
    from src import ops
    from src.trainer import DiffusionLanguageModelTrainer

    class _Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(8, 4)
            self.tokenizer = None

        def training_loss(self, input_ids, attention_mask=None):
            return self.lin(input_ids).pow(2).mean()

    model = _Toy()
    acc = accelerate.Accelerator(cpu=True)
    opt = acc.prepare(torch.optim.AdamW(model.parameters(), lr=1e-2))
    assert not hasattr(opt, "_optimizer_step_post_hooks")      # the situation this test is about
    tr = DiffusionLanguageModelTrainer(model, opt, None, acc)
    assert tr._marks_stale
    w0 = model.lin.weight.detach().clone()
    e0 = ops.WEIGHTS_EPOCH
    loss = tr._step({"input_ids": torch.randn(3, 8)})
    assert ops.WEIGHTS_EPOCH == e0 + 1
    assert not torch.equal(w0, model.lin.weight.detach())
    assert tr.loss_value(loss) == loss.item()
    # the accelerator's own prepare() tolerates the same object
    from src.dp import HipAccelerator
    HipAccelerator(device="cpu").prepare(opt)


def test_head_link_travels_with_the_tensor_not_with_addresses():
    """The compact-gradient hand-over between the masked CE and the vocabulary projection is a
    per-call object referenced by the logits tensor and by the projection's autograd node; nothing
    process-global is keyed by device addresses."""
    from src import ops
    for name in ("_COMPACT_OK", "_COMPACT_GRAD", "_ZERO_PADDED", "_SWIGLU_H"):
        assert not hasattr(ops, name)
    a, b = ops._HeadLink(), ops._HeadLink()
    a.grad = ("x",)
    assert b.grad is None and b.zero_padded is None and not b.compact_ok


def test_dropout_gradient_handoff_is_taken_only_for_the_very_gradient_it_was_made_from():
    """The masked gradient a norm's backward leaves for the projection behind a residual dropout (ops._Drop) is
    used only when the gradient autograd delivers is the norm's dx itself: same storage, same size and dtype,
    not modified since; it is handed out once and every other case falls back to the masking pass."""
    import torch
    from src import ops
    T, n = 6, 8

    def filled():
        d = ops._Drop(0.1, 1234)
        dx, dxd = torch.randn(T, n), torch.randn(T, n)
        d.dx, d.dx_drop, d.ver = dx, dxd, dx._version
        return d, dx, dxd

    assert tuple(ops._Drop(0.25, 7)) == (0.25, 7)
    assert ops._Drop(0.25, 7).take(torch.zeros(T, n), T, n) is None          # nothing was left
    d, dx, dxd = filled()
    got = d.take(dx.view(2, 3, n), T, n)                                     # (a view of dx: what a reshape's backward delivers)
    assert got is not None and got.data_ptr() == dxd.data_ptr() and got.shape == (T, n)
    assert d.dx is None and d.take(dx, T, n) is None                         # handed out once
    d, dx, _ = filled()
    assert d.take(dx + 1.0, T, n) is None                                    # a summed gradient is another tensor
    d, dx, _ = filled()
    dx.add_(1.0)
    assert d.take(dx, T, n) is None                                          # modified in place since
    d, dx, _ = filled()
    assert d.take(dx[:3], 3, n) is None and d.dx is None                     # another size; the link is cleared anyway
    d, dx, _ = filled()
    assert d.take(dx.t(), n, T) is None                                      # not the row layout
    # a tensor without the tag, or with gradients off, hands the norm nothing
    x = torch.zeros(2, 3)
    assert ops._input_drop(x) is None
    ops._tag_drop(x, d)
    assert ops._input_drop(x) is d
    with torch.no_grad():
        assert ops._input_drop(x) is None
    assert ops._new_drop(None, 0.1) is None and ops._new_drop(x, 0.0) is None and ops._new_drop(x, 0.1).p == 0.1


def test_bench_launches_its_own_ranks(tmp_path, monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment starts N ranks under
    torch.distributed.run from a parent that has not imported torch, and returns their exit code."""
    import importlib
    import subprocess
    import sys
    bench = importlib.import_module("bench")
    seen = {}

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    import pytest
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # the parent's module-level imports do not pull in torch
    src_text = open(bench.__file__).read()
    head = src_text[:src_text.index("def main()")]
    assert "\nimport torch" not in head


def test_fused_adamw_refuses_host_tensors():
    import pytest
    import torch
    from src.optim import FusedAdamW
    p = torch.nn.Parameter(torch.randn(4))
    p.grad = torch.randn(4)
    opt = FusedAdamW([p], lr=1e-3)
    with pytest.raises(RuntimeError):
        opt.step()
    with pytest.raises(ValueError):
        FusedAdamW([p], lr=-1.0)


def test_learning_rate_schedule_matches_reference_curves():
    """Bit-exact against curves recorded from the reference's create_scheduler (src/scheduler.py:97-159);
    the cases are those of oracle/make_golden.py::SCHEDULE_CASES."""
    from src.scheduler import WarmupStableDecayLR, create_scheduler
    cases = [
        dict(total_steps=100, warmup_steps=10, initial_lr=0.0, peak_lr=None, stable_lr=None, final_lr=None, lr=5e-4),
        dict(total_steps=37, warmup_steps=2000, initial_lr=0.0, peak_lr=None, stable_lr=None, final_lr=None, lr=5e-4),
        dict(total_steps=64, warmup_steps=0, initial_lr=0.0, peak_lr=4e-4, stable_lr=1e-4, final_lr=1e-5, lr=1e-3),
        dict(total_steps=3, warmup_steps=1, initial_lr=1e-6, peak_lr=None, stable_lr=None, final_lr=None, lr=3e-4),
        dict(total_steps=1000, warmup_steps=333, initial_lr=1e-5, peak_lr=4e-4, stable_lr=None, final_lr=1e-7, lr=2e-4),
    ]
    g = load_golden("schedule.npz")
    assert int(g["n_cases"][0]) == len(cases)
    for j, c in enumerate(cases):
        w = torch.nn.Parameter(torch.zeros(3))
        opt = torch.optim.SGD([w], lr=c["lr"])
        sch = create_scheduler(opt, **{k: v for k, v in c.items() if k != "lr"})
        assert isinstance(sch, WarmupStableDecayLR)
        assert [sch.warmup_steps, sch.stable_steps, sch.middle_decay_steps, sch.final_decay_steps] == \
            [int(x) for x in g[f"s{j}_phases"]]
        assert [sch.initial_lr, sch.peak_lr, sch.stable_lr, sch.final_lr] == [float(x) for x in g[f"s{j}_levels"]]
        want = g[f"s{j}_lrs"]
        got = [opt.param_groups[0]["lr"]]
        for _ in range(len(want) - 1):
            opt.step()
            sch.step()
            got.append(opt.param_groups[0]["lr"])
        assert np.array_equal(np.array(got, dtype=np.float64), want), (j, got[:5], want[:5])
    # the state dict carries the reference's attribute names, so checkpoints move between the two
    keys = set(sch.state_dict())
    assert {"warmup_steps", "stable_steps", "middle_decay_steps", "final_decay_steps", "phase2_start",
            "phase3_start", "phase4_start", "phase5_start", "initial_lr", "peak_lr", "stable_lr", "final_lr",
            "last_epoch", "base_lrs"} <= keys
    sch2 = create_scheduler(torch.optim.SGD([w], lr=1.0), total_steps=10)
    sch2.load_state_dict(sch.state_dict())
    assert sch2.lr_at(500) == sch.lr_at(500)


class WordHashTokenizer:
    """Stand-in for the reference's tokenizer wrapper (same class as in oracle/make_golden.py)."""

    def encode(self, text, max_length=512, return_tensors=False):
        ids = [101]
        for w in text.split():
            h = 2166136261
            for b in w.encode("utf-8"):
                h = ((h ^ b) * 16777619) & 0xFFFFFFFF
            ids.append(1000 + h % (30522 - 1000))
        ids = ids[:max_length - 1] + [102]
        return {"input_ids": ids, "attention_mask": [1] * len(ids)}


def write_question_file(tmp_path, g):
    path = tmp_path / "questions.txt"
    path.write_text("\n".join(str(x) for x in g["lines"]) + "\n", encoding="utf-8")
    return str(path)


def test_question_dataset_and_loaders_match_reference_batches(tmp_path):
    """Parsing, items, split and padded batches equal those recorded from the reference's
    src/dataset.py over the same file, tokenizer stand-in and seed."""
    from src.dataset import QuestionDataset, collate_batch, create_question_dataloaders
    g = load_golden("dataset.npz")
    path = write_question_file(tmp_path, g)
    ds = QuestionDataset(path, WordHashTokenizer(), max_length=16)
    assert ds.questions == [str(x) for x in g["questions"]] and len(ds) == int(g["n_items"][0])
    for i in range(len(ds)):
        item = ds[i]
        assert item["input_ids"].dtype == torch.long and item["input_ids"].dim() == 1
        assert np.array_equal(item["input_ids"].numpy(), g[f"item{i}_ids"])
        assert item["attention_mask"].tolist() == [1] * item["input_ids"].numel()
    assert QuestionDataset(path, WordHashTokenizer(), transform=lambda it: {**it, "extra": 1})[0]["extra"] == 1
    torch.manual_seed(3)
    tr, va = create_question_dataloaders(path, WordHashTokenizer(), batch_size=4, max_length=16, train_ratio=0.75,
                                         shuffle=True, num_workers=0)
    for name, dl in (("train", tr), ("val", va)):
        batches = list(dl)
        assert len(batches) == int(g[f"n_{name}"][0])
        for n, b in enumerate(batches):
            assert np.array_equal(b["input_ids"].numpy(), g[f"{name}{n}_ids"])
            assert np.array_equal(b["attention_mask"].numpy(), g[f"{name}{n}_mask"])
    _, none = create_question_dataloaders(path, WordHashTokenizer(), train_ratio=1.0, num_workers=0)
    assert none is None
    # collate_batch against torch's pad_sequence on ragged items
    items = [{"input_ids": torch.arange(1, n + 1), "attention_mask": torch.ones(n, dtype=torch.long)} for n in (3, 7, 1)]
    out = collate_batch(items)
    for k in ("input_ids", "attention_mask"):
        assert torch.equal(out[k], torch.nn.utils.rnn.pad_sequence([it[k] for it in items], batch_first=True))
