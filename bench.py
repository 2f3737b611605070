"""Masked-diffusion training-step benchmark (one process per GPU).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is the reference trainer's step (src/trainer.py:119-140 of the reference):
zero_grad -> forward masking -> transformer forward -> masked CE -> backward (with bucketed
RCCL gradient all-reduce when N > 1) -> AdamW.step -> loss.item(), on synthetic tokens that are
resident in HBM before the timed region.  Workload: the reference's configs/default.yaml model
(d=768, 8 heads of 96, 8 layers, ff=1024, vocab 30522) at sequence length 512, bf16 compute,
dropout 0.0 (--dropout), per-GPU batch --batch (default 512 sequences, the YAML's own batch_size, = 262,144 tokens per
GPU per step; weak scaling).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import random
import sys
import time
import types

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak (spec), per GPU
PEAK_HBM_GBS = 8000.0       # MI355X HBM3E peak (spec), per GPU

CONFIGS = {
    # reference configs/default.yaml:8-15 with len(BertTokenizer) = 30522
    "default": dict(vocab_size=30522, d_model=768, nhead=8, num_layers=8, dim_feedforward=1024),
    "1b": dict(vocab_size=32000, d_model=2048, nhead=16, num_layers=16, dim_feedforward=5632),
    # LLaDA-8B layer shape at reduced depth (a single-GPU sanity / kernel-shape run, not a benchmark)
    "8b-2layer": dict(vocab_size=126464, d_model=4096, nhead=32, num_layers=2, dim_feedforward=12288),
    "tiny": dict(vocab_size=200, d_model=64, nhead=2, num_layers=2, dim_feedforward=128),
}


def train_flops_per_token(c, seq):
    """3 * (2*N_mm + 4*S*d*L), N_mm = L*(4d^2 + 3dF) + d*V: the algorithmic matmul+attention work."""
    d, L, Fd, V = c["d_model"], c["num_layers"], c["dim_feedforward"], c["vocab_size"]
    n_mm = L * (4 * d * d + 3 * d * Fd) + d * V
    return 3.0 * (2.0 * n_mm + 4.0 * seq * d * L)


def synthetic_tokens(batch, seq, vocab, seed):
    g = torch.Generator().manual_seed(seed)
    lo = 1000 if vocab > 2000 else 110
    ids = torch.randint(lo, vocab, (batch, seq), generator=g, dtype=torch.int64)
    ids[:, 0] = 101
    ids[:, -1] = 102
    return ids, torch.ones(batch, seq, dtype=torch.int64)


def cpu_baseline(cfg, seq, steps=3):
    """The CPU restatement of the reference step (oracle/step_oracle.py: fp32, torch CPU ops,
    autograd backward, AdamW) on a bounded sample: B=4 sequences per step."""
    from oracle import step_oracle as so
    ncores = os.cpu_count() or 1
    threads = min(ncores, int(os.environ.get("OMP_NUM_THREADS", ncores)))
    torch.set_num_threads(threads)
    ocfg = so.OracleConfig(vocab_size=cfg["vocab_size"], d_model=cfg["d_model"], nhead=cfg["nhead"],
                           num_layers=cfg["num_layers"], dim_feedforward=cfg["dim_feedforward"])
    params = {k: v.requires_grad_(True) for k, v in so.init_params(ocfg, seed=42).items()}
    opt = torch.optim.AdamW(list(params.values()), lr=5e-4, weight_decay=0.1)
    B = 4
    ids, att = synthetic_tokens(B, seq, cfg["vocab_size"], 1234)
    rng = random.Random(42)
    g = torch.Generator().manual_seed(7)
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        masked, _ = so.forward_mask(ids, rng.random(), torch.rand(ids.shape, generator=g))
        logits = so.transformer_forward(params, ocfg, masked, att)
        loss = so.vlb_loss(logits, ids, masked, 103, att)
        if loss.grad_fn is not None:
            loss.backward()
        opt.step()
        float(loss)
        if i > 0:
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": B * seq / med, "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": f"{steps} timed steps (median) of B={B}, S={seq}, fp32, oracle/step_oracle.py + AdamW"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="default", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=512,
                    help="sequences per GPU per step (configs/default.yaml:23 batch_size = 512)")
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dropout", type=float, default=0.0,
                    help="residual dropout probability (the two per-layer residual dropouts; the fused "
                         "attention kernel applies none to the attention probabilities)")
    ap.add_argument("--bucket-mb", type=int, default=64)
    ap.add_argument("--optimizer", default="hip", choices=["hip", "torch"],
                    help="hip: src.optim.FusedAdamW (one HIP pass, weight copies refreshed in it); torch: torch.optim.AdamW(fused=True)")
    args = ap.parse_args()

    if os.environ.get("LDM_FAULT_TIMEOUT"):     # debugging aid: dump python stacks if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["LDM_FAULT_TIMEOUT"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if os.environ.get("LDM_DP_SAME_GPU"):      # rehearsal of the multi-rank path on a one-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        backend = os.environ.get("LDM_DP_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from src import ops
    from src.dp import HipAccelerator
    from src.masking_strategy import create_masking_strategy
    from src.model import DiffusionLanguageModel
    from src.trainer import DiffusionLanguageModelTrainer
    from src.transformer import DiffusionTransformer, DiffusionTransformerConfig

    cfg = CONFIGS[args.config]
    torch.manual_seed(42)
    random.seed(42)     # the reference seeds both identically on every rank (main.py:39-40)
    tr = DiffusionTransformer(DiffusionTransformerConfig(dropout=args.dropout, layer_norm_eps=1e-12,
                                                         max_position_embeddings=max(512, args.seq), **cfg))
    tok = types.SimpleNamespace(mask_token_id=103, bos_token_id=101, eos_token_id=102, pad_token_id=0)
    model = DiffusionLanguageModel(tr, tok, create_masking_strategy("random", 101, 102, 0, 103), None)
    acc = HipAccelerator(mixed_precision="bf16" if args.dtype == "bf16" else "no", device=dev,
                         bucket_bytes=args.bucket_mb << 20)
    model = acc.prepare(model)
    if args.optimizer == "hip":
        from src.optim import FusedAdamW
        opt = acc.prepare(FusedAdamW(model.parameters(), lr=5e-4, weight_decay=0.1))
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=0.1, fused=True)
    trainer = DiffusionLanguageModelTrainer(model, opt, None, acc)
    model.train()

    ids, att = synthetic_tokens(args.batch, args.seq, cfg["vocab_size"], 1234 + rank)
    batch = {"input_ids": ids.to(dev), "attention_mask": att.to(dev)}

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    last = 0.0
    # loss_value(): the host reads every step's loss before it starts the next one, as the
    # reference's `loss.item()` does; the copy is queued right behind the loss kernels
    read = (lambda l: l.item()) if os.environ.get("LDM_ITEM_SYNC") else trainer.loss_value   # A/B switch
    for _ in range(args.warmup):
        last = read(trainer._step(batch))
    sync()
    # everything alive now (modules, torch internals) moves to the collector's permanent generation:
    # a full collection over it pauses the host for ~80 ms, long enough to drain the GPU queue
    gc.collect()
    gc.freeze()
    ops.GEMM_EVENTS = {"head.fwd": []}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = read(trainer._step(batch))
    sync()
    elapsed = time.perf_counter() - t0
    events = ops.GEMM_EVENTS["head.fwd"]
    ops.GEMM_EVENTS = None
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        tokens = args.batch * args.seq * world * args.steps
        tps = tokens / elapsed
        fpt = train_flops_per_token(cfg, args.seq)
        # dominant kernel: the LM-head forward GEMM ([B*S, d] x [d, V]), one launch per step
        ms = [e[0].elapsed_time(e[1]) for e in events]
        # fused head loss: the launch covers only the rows the loss reads (count on the device)
        fl = [e[2] * (float(e[3].item()) if len(e) > 3 else 1.0) for e in events]
        flops = sum(fl) / max(len(fl), 1)
        avg_ms = sum(ms) / max(len(ms), 1)
        achieved = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        head_rows = (sum(float(e[3].item()) for e in events) / (len(events) * args.batch * args.seq)
                     if events and len(events[0]) > 3 else 1.0)
        d_, V_ = cfg["d_model"], cfg["vocab_size"]
        # work actually executed: the three LM-head GEMMs run on the participating rows only
        fpt_exec = fpt - 6.0 * d_ * V_ * (1.0 - head_rows)
        # memory-side bytes of that kernel per launch, from the committed PMC passes (per-row rates
        # measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, scaled to this run's row count)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r1_head_fwd_traffic.json")
        if os.path.exists(tpath) and args.config == "default" and args.dtype == "bf16":
            tj = json.load(open(tpath))
            rows_m = sum(l["rows"] for l in tj["launches"])
            per_row = sum(l["fetch_kib_raw"] * tj["fetch_correction"] + l["write_kib"] for l in tj["launches"]) * 1024.0 / rows_m
            traffic = per_row * head_rows * args.batch * args.seq
        out = {
            "metric": "masked-diffusion train tokens/sec (node)", "value": tps, "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"configs/default.yaml model, seq {args.seq}" if args.config == "default"
                       else f"{args.config} model, seq {args.seq}",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "seq_len": args.seq,
                       "parallelism": f"dp{world}", "dropout": args.dropout, "optimizer": "AdamW(lr=5e-4, wd=0.1)",
                       **cfg},
            "loss_last_step": last,
            "train_flops_per_token": fpt,
            # fraction of the bf16 MFMA peak: with the flops of the reference's step (LM head on every
            # position) and with the flops executed here (LM head on the rows the loss reads)
            "mfma_fraction_algorithmic": tps * fpt / (world * PEAK_BF16_TFLOPS * 1e12),
            "mfma_fraction_of_step": tps * fpt_exec / (world * PEAK_BF16_TFLOPS * 1e12),
            "executed_flops_per_token": fpt_exec, "lm_head_rows_fraction": head_rows,
            "roofline": {"kernel": "gemm256_k (LM-head forward GEMM over the rows the loss reads)", "bound": "mfma",
                         "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic,
                         "flops_per_launch": flops, "avg_launch_ms": avg_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args.seq)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
