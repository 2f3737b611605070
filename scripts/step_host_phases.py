"""Host-side time of each phase of consecutive training steps (forward enqueue, backward enqueue,
optimizer enqueue, wait for the loss value): shows where the host stalls when a step is slow."""
import gc, os, sys, time, types, random
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
if os.environ.get("NOGC"):
    gc.disable()
ids, att = bench.synthetic_tokens(128, 512, cfg["vocab_size"], 1234)
batch = {"input_ids": ids.to(dev), "attention_mask": att.to(dev)}
pc = time.perf_counter
for i in range(16):
    t0 = pc()
    opt.zero_grad()
    loss = model.training_loss(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"])
    trainer._stage_loss(loss)
    t1 = pc()
    acc.backward(loss)
    t2 = pc()
    opt.step()
    t3 = pc()
    v = trainer.loss_value(loss)
    t4 = pc()
    print(f"step {i:2d} fwd {1e3*(t1-t0):6.2f} bwd {1e3*(t2-t1):6.2f} opt {1e3*(t3-t2):6.2f} wait {1e3*(t4-t3):6.2f} total {1e3*(t4-t0):6.2f} loss {v:.4f} mem {torch.cuda.memory_reserved()/2**30:.1f} GiB", flush=True)
torch.cuda.synchronize()
