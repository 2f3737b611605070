"""How many weight-gradient buffers of a training step come out of the zero pool, and how many fall back to torch.zeros."""
import os, sys, types, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from src import ops
from src.dp import HipAccelerator
from src.masking_strategy import create_masking_strategy
from src.model import DiffusionLanguageModel
from src.trainer import DiffusionLanguageModelTrainer
from src.transformer import DiffusionTransformer, DiffusionTransformerConfig
from src.optim import FusedAdamW
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["default"]
torch.manual_seed(42); random.seed(42)
tr = DiffusionTransformer(DiffusionTransformerConfig(dropout=0.0, layer_norm_eps=1e-12, max_position_embeddings=512, **cfg))
tok = types.SimpleNamespace(mask_token_id=103, bos_token_id=101, eos_token_id=102, pad_token_id=0)
model = DiffusionLanguageModel(tr, tok, create_masking_strategy("random", 101, 102, 0, 103), None)
acc = HipAccelerator(mixed_precision="bf16", device=dev)
model = acc.prepare(model)
opt = acc.prepare(FusedAdamW(model.parameters(), lr=5e-4, weight_decay=0.1))
trainer = DiffusionLanguageModelTrainer(model, opt, None, acc)
model.train()
ids, att = bench.synthetic_tokens(16, 512, cfg["vocab_size"], 1234)
batch = {"input_ids": ids.to(dev), "attention_mask": att.to(dev)}
stats = {"pool": 0, "fallback": 0}
orig = ops._ZeroPool.take
def take(self, numel, device):
    before = self.off
    v = orig(self, numel, device)
    stats["pool" if (self.buf is not None and self.off != before) else "fallback"] += 1
    return v
ops._ZeroPool.take = take
zeros = torch.zeros
calls = {"n": 0}
def counting_zeros(*a, **k):
    calls["n"] += 1
    return zeros(*a, **k)
torch.zeros = counting_zeros
for i in range(4):
    stats["pool"] = stats["fallback"] = 0; calls["n"] = 0
    trainer._step(batch).item()
    print("step", i, stats, "torch.zeros calls", calls["n"], "want", ops.ZERO_POOL.want, "used", ops.ZERO_POOL.used)
