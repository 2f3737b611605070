"""Device-side batch producers against the reference's collate semantics and the Philox oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _items(lengths, seed=0, vocab=30522):
    g = torch.Generator().manual_seed(seed)
    out = []
    for n in lengths:
        ids = torch.randint(1000, vocab, (n,), generator=g, dtype=torch.int64)
        out.append({"input_ids": ids, "attention_mask": torch.ones(n, dtype=torch.int64)})
    return out


@pytest.mark.parametrize("lengths", [[5, 1, 9, 3], [512] * 3, [1], [64, 300, 17, 511, 2, 128], []])
def test_collate_batch_device_equals_reference_collate(dev, lengths):
    """src/dataset.py:26-48: pad_sequence(batch_first=True, padding_value=0) of ids and of masks."""
    from torch.nn.utils.rnn import pad_sequence
    from src.data import collate_batch_device
    items = _items(lengths)
    got = collate_batch_device(items, dev)
    if not lengths:
        assert got["input_ids"].numel() == 0
        return
    want_ids = pad_sequence([it["input_ids"] for it in items], batch_first=True, padding_value=0)
    want_mask = pad_sequence([it["attention_mask"] for it in items], batch_first=True, padding_value=0)
    assert got["input_ids"].dtype == torch.int64 and got["attention_mask"].dtype == torch.int64
    assert torch.equal(got["input_ids"].cpu(), want_ids) and torch.equal(got["attention_mask"].cpu(), want_mask)


@pytest.mark.parametrize("B,S,min_len", [(4, 512, None), (3, 10, 4), (5, 127, 2), (64, 512, 100), (1, 1, None), (2, 4096, 1)])
def test_synth_tokens_bit_exact(dev, B, S, min_len):
    from oracle.philox import synth_tokens
    from src.data import SyntheticTokenBatches
    src = SyntheticTokenBatches(3, B, S, 30522, seed=99, device=dev, min_len=min_len)
    for i, batch in enumerate(src):
        ids, mask = synth_tokens(B, S, 1000, 30522, 101, 102, 0, S if min_len is None else min_len, 99, i)
        assert np.array_equal(batch["input_ids"].cpu().numpy(), ids), i
        assert np.array_equal(batch["attention_mask"].cpu().numpy(), mask), i
    b0 = src.batch(0)
    assert (b0["input_ids"][:, 0] == 101).all()
    lens = b0["attention_mask"].sum(1)
    if S >= 2:
        assert (b0["input_ids"][torch.arange(B), lens - 1] == 102).all() and (lens >= 2).all()
    body = b0["input_ids"][b0["attention_mask"] == 1]
    assert ((body >= 1000) | (body == 101) | (body == 102)).all() and (body < 30522).all()
    assert (b0["input_ids"][b0["attention_mask"] == 0] == 0).all()


def test_device_prefetcher_order_and_content(dev):
    from torch.nn.utils.rnn import pad_sequence
    from src.data import DevicePrefetcher
    raw = [_items([7, 3, 12], seed=s) for s in range(5)]
    # host-collated dict batches (the reference's DataLoader output) ...
    host = [{"input_ids": pad_sequence([it["input_ids"] for it in b], batch_first=True),
             "attention_mask": pad_sequence([it["attention_mask"] for it in b], batch_first=True)} for b in raw]
    got = list(DevicePrefetcher(host, dev, depth=2))
    assert len(got) == 5
    for g, h in zip(got, host):
        assert g["input_ids"].is_cuda and torch.equal(g["input_ids"].cpu(), h["input_ids"])
        assert torch.equal(g["attention_mask"].cpu(), h["attention_mask"])
    # ... and un-collated item lists, padded on the device
    got2 = list(DevicePrefetcher(raw, dev, depth=3))
    for g, h in zip(got2, host):
        assert torch.equal(g["input_ids"].cpu(), h["input_ids"]) and torch.equal(g["attention_mask"].cpu(), h["attention_mask"])


def test_training_step_on_ragged_device_batches(dev):
    """A trainer epoch over ragged synthetic device batches: finite losses, pad rows never counted."""
    import random
    import types
    from src.data import SyntheticTokenBatches
    from src.dp import HipAccelerator
    from src.masking_strategy import create_masking_strategy
    from src.model import DiffusionLanguageModel
    from src.optim import FusedAdamW
    from src.trainer import DiffusionLanguageModelTrainer
    from src.transformer import DiffusionTransformer, DiffusionTransformerConfig
    torch.manual_seed(0)
    random.seed(0)
    tr = DiffusionTransformer(DiffusionTransformerConfig(vocab_size=2000, d_model=64, nhead=2, num_layers=2,
                                                         dim_feedforward=128, dropout=0.0, layer_norm_eps=1e-12,
                                                         max_position_embeddings=128))
    tok = types.SimpleNamespace(mask_token_id=103, bos_token_id=101, eos_token_id=102, pad_token_id=0)
    model = DiffusionLanguageModel(tr, tok, create_masking_strategy("random", 101, 102, 0, 103), None)
    acc = HipAccelerator(mixed_precision="bf16", device=dev)
    model = acc.prepare(model)
    opt = acc.prepare(FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.1))
    trainer = DiffusionLanguageModelTrainer(model, opt, None, acc)
    batches = SyntheticTokenBatches(4, 8, 96, 2000, seed=5, device=dev, id_lo=200, min_len=10)
    loss = trainer.train_epoch(batches)
    assert np.isfinite(loss) and trainer.global_step == 4


def test_question_loaders_with_device_padding_match_reference_batches(dev, tmp_path):
    """create_question_dataloaders(device=...) -- items un-collated from the workers, padded by
    ldm_collate_pad in HBM -- yields the batches recorded from the reference's host loaders
    (src/dataset.py:137-202) for the same file, tokenizer stand-in and seed."""
    from conftest import load_golden
    from test_host_logic import WordHashTokenizer, write_question_file
    from src.dataset import create_question_dataloaders
    g = load_golden("dataset.npz")
    path = write_question_file(tmp_path, g)
    torch.manual_seed(3)
    tr, va = create_question_dataloaders(path, WordHashTokenizer(), batch_size=4, max_length=16, train_ratio=0.75,
                                         shuffle=True, num_workers=0, device=dev)
    for name, dl in (("train", tr), ("val", va)):
        assert len(dl) == int(g[f"n_{name}"][0])
        batches = list(dl)
        assert len(batches) == int(g[f"n_{name}"][0])
        for n, b in enumerate(batches):
            assert b["input_ids"].is_cuda and b["input_ids"].dtype == torch.int64
            assert np.array_equal(b["input_ids"].cpu().numpy(), g[f"{name}{n}_ids"])
            assert np.array_equal(b["attention_mask"].cpu().numpy(), g[f"{name}{n}_mask"])
