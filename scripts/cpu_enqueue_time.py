"""How long the host needs to enqueue one training step (vs. the GPU time of the step)."""
import os, sys, time, types, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
if os.environ.get("OPT", "hip") == "hip":
    from src.optim import FusedAdamW
    opt = acc.prepare(FusedAdamW(model.parameters(), lr=5e-4, weight_decay=0.1))
else:
    opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=0.1, fused=True)
trainer = DiffusionLanguageModelTrainer(model, opt, None, acc)
model.train()
B = int(os.environ.get("B", 128))
ids, att = bench.synthetic_tokens(B, 512, cfg["vocab_size"], 1234)
batch = {"input_ids": ids.to(dev), "attention_mask": att.to(dev)}
for _ in range(3):
    trainer._step(batch).item()
enq, tot = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = trainer._step(batch)
    t1 = time.perf_counter()
    loss.item()
    t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
print(f"B={B}: host enqueue {1e3 * sum(enq) / len(enq):.2f} ms/step, step {1e3 * sum(tot) / len(tot):.2f} ms")

# optimizer step alone: host time of the call, and GPU time between events
torch.cuda.synchronize()
hs = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); e0.record(); opt.step(); e1.record(); t1 = time.perf_counter()
    torch.cuda.synchronize()
    hs.append((1e3 * (t1 - t0), e0.elapsed_time(e1)))
print("optimizer.step(): host ms, gpu ms:", [(round(a, 2), round(b, 2)) for a, b in hs])
if os.environ.get("CPROFILE"):
    import cProfile, pstats
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        trainer.loss_value(trainer._step(batch))
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
