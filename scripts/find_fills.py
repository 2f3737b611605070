"""Which Python lines launch fill kernels in one training step (torch profiler with stacks)."""
import os, sys, types, random, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from src.dp import HipAccelerator
from src.masking_strategy import create_masking_strategy
from src.model import DiffusionLanguageModel
from src.trainer import DiffusionLanguageModelTrainer
from src.transformer import DiffusionTransformer, DiffusionTransformerConfig
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["default"]
torch.manual_seed(42); random.seed(42)
tr = DiffusionTransformer(DiffusionTransformerConfig(dropout=0.0, layer_norm_eps=1e-12, max_position_embeddings=512, **cfg))
tok = types.SimpleNamespace(mask_token_id=103, bos_token_id=101, eos_token_id=102, pad_token_id=0)
model = DiffusionLanguageModel(tr, tok, create_masking_strategy("random", 101, 102, 0, 103), None)
acc = HipAccelerator(mixed_precision="bf16", device=dev)
model = acc.prepare(model)
from src.optim import FusedAdamW
opt = acc.prepare(FusedAdamW(model.parameters(), lr=5e-4, weight_decay=0.1))
trainer = DiffusionLanguageModelTrainer(model, opt, None, acc)
model.train()
ids, att = bench.synthetic_tokens(16, 512, cfg["vocab_size"], 1234)
batch = {"input_ids": ids.to(dev), "attention_mask": att.to(dev)}
for _ in range(2):
    trainer._step(batch).item()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    trainer._step(batch).item()
rows = []
for ev in prof.key_averages(group_by_stack_n=12):
    if ev.key in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::ones", "aten::full", "aten::zeros_like"):
        st = [f for f in (ev.stack or []) if "dist-packages/torch" not in f and "built-in" not in f]
        rows.append((ev.count, ev.key, (st[0] if st else (ev.stack[0] if ev.stack else "?"))[-120:]))
for r in sorted(rows, reverse=True)[:25]:
    print(*r)
