"""Masked-diffusion training-step benchmark (one process per GPU).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus 8 --steps 10 --warmup 3          # starts its 8 ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is the reference trainer's step (src/trainer.py:119-140 of the reference):
zero_grad -> forward masking -> transformer forward -> masked CE -> backward (with bucketed
RCCL gradient all-reduce when N > 1) -> AdamW.step -> loss.item(), on synthetic tokens that are
resident in HBM before the timed region.  Workload: the reference's configs/default.yaml model
(d=768, 8 heads of 96, 8 layers, ff=1024, vocab 30522) at sequence length 512, bf16 compute,
dropout 0.0 (--dropout; the YAML trains with 0.1, see the "dropout" key of the line), per-GPU batch
--batch (default 512 sequences, the YAML's own batch_size, = 262,144 tokens per GPU per step; weak
scaling).  Rank 0 prints ONE JSON line.

The line's "roofline" block is about the kernel class the step spends most of its time in (the
256-row MFMA GEMMs, eight-wave gemm256_k and four-wave gemm4w_k, in all their roles); "roofline_by_kernel" lists every instrumented class.  Both
come from HIP events recorded around each launch on the launching stream inside the timed region
(src/ops.py KERNEL_EVENTS); LDM_BENCH_EVENTS=0 switches the events off (A/B of their cost).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import random
import socket
import subprocess
import sys
import time
import types

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

# kernel class -> (what bounds it, printable name)
CLASSES = {
    "gemm.fwd": ("mfma", "gemm256_k forward (qkv, out+residual, gate/up+SwiGLU, down+residual, LM head on masked rows)"),
    "gemm.dx": ("mfma", "gemm4w_k (plain input gradients over the W^T copies) and gemm256_k (the SwiGLU-backward epilogue variant, LM head)"),
    "gemm.dw": ("mfma", "gemm256_k weight+bias gradients (split-K, fp32 atomics)"),
    "attn.fwd": ("mfma", "attention forward"),
    "attn.bwd": ("mfma", "attention backward (dQ and dK/dV kernels, one entry point)"),
    "rmsnorm.fwd": ("hbm", "rmsnorm_fwd8_k"),
    "rmsnorm.bwd": ("hbm", "rmsnorm_bwd8_k + dw second stage"),
    "rope": ("hbm", "rope_qk8_k"),
    "ce": ("hbm", "ce_fwd_grad_k (one-pass masked CE over the compact logits)"),
}


def train_flops_per_token(c, seq):
    """3 * (2*N_mm + 4*S*d*L), N_mm = L*(4d^2 + 3dF) + d*V: the algorithmic matmul+attention work."""
    d, L, Fd, V = c["d_model"], c["num_layers"], c["dim_feedforward"], c["vocab_size"]
    n_mm = L * (4 * d * d + 3 * d * Fd) + d * V
    return 3.0 * (2.0 * n_mm + 4.0 * seq * d * L)


def synthetic_tokens(batch, seq, vocab, seed):
    import torch
    g = torch.Generator().manual_seed(seed)
    lo = 1000 if vocab > 2000 else 110
    ids = torch.randint(lo, vocab, (batch, seq), generator=g, dtype=torch.int64)
    ids[:, 0] = 101
    ids[:, -1] = 102
    return ids, torch.ones(batch, seq, dtype=torch.int64)


def cpu_baseline(cfg, seq, steps=3):
    """The CPU restatement of the reference step (oracle/step_oracle.py: fp32, torch CPU ops,
    autograd backward, AdamW) on a bounded sample: B=4 sequences per step."""
    import torch
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


def launch_ranks(n: int) -> int:
    """``python bench.py --gpus N`` typed as a plain command: start the N ranks as children (one process
    per GPU under torch.distributed.run, the launcher the reference's README reaches through
    ``accelerate launch``) and hand back their exit status.  This parent never touches the GPU -- it
    does not even import torch -- so the ranks are ordinary children of a GPU-free process; rank 0
    prints the single JSON line straight to the inherited stdout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def summarize_events(events, steps):
    """KERNEL_EVENTS -> {class: {launches_per_step, ms_per_step, work_per_step}} (device-side counts read back)."""
    out = {}
    for tag, lst in events.items():
        if not lst:
            continue
        ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in lst)
        work = sum(w * (float(c.item()) if c is not None else 1.0) for _, _, w, c in lst)
        out[tag] = {"launches_per_step": len(lst) / steps, "ms_per_step": ms / steps, "work_per_step": work / steps}
    return out


def committed_traffic(cfg_name, dtype, batch, seq):
    """Memory-side bytes per launch of the GEMM class from the committed PMC passes
    (profiles/r3_gemm_traffic.json, made by scripts/gpu_r3_profile.sh at the bench shape: the counters need
    passes of their own under rocprofv3, so the figure is read, not measured in this run); None if this run's
    shape is not the one that was profiled."""
    path = os.path.join(ROOT, "profiles", "r3_gemm_traffic.json")
    if not os.path.exists(path):
        return None
    tj = json.load(open(path))
    if (tj.get("config"), tj.get("dtype"), tj.get("batch"), tj.get("seq")) != (cfg_name, dtype, batch, seq):
        return None
    return tj


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
                    help="dropout probability of all three dropouts of the encoder layer (attention "
                         "probabilities inside the attention kernels, and the two residual dropouts); "
                         "configs/default.yaml:13 trains with 0.1, parity and the headline number use 0.0")
    ap.add_argument("--no-dropout-leg", action="store_true",
                    help="skip the second, shorter timing of the same step at dropout 0.1 (the 'dropout01' key)")
    ap.add_argument("--bucket-mb", type=int, default=64)
    ap.add_argument("--objective", default="reference", choices=["reference", "paper"],
                    help="reference: one scalar t per step, unweighted mean over masked positions; paper: "
                         "per-sequence t ~ U(eps,1), 1/t weights, B*L normaliser")
    ap.add_argument("--fp32-residual", action="store_true",
                    help="keep the residual stream in fp32 around the bf16 sublayers (DiffusionTransformer.fp32_residual)")
    ap.add_argument("--optimizer", default="hip", choices=["hip", "torch"],
                    help="hip: src.optim.FusedAdamW (one HIP pass, weight copies refreshed in it); torch: torch.optim.AdamW(fused=True)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    if os.environ.get("LDM_FAULT_TIMEOUT"):     # debugging aid: dump python stacks if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["LDM_FAULT_TIMEOUT"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("LDM_DP_SAME_GPU"):      # rehearsal of the multi-rank path on a one-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    # (LDM_DP_FORCE=1 under a launcher: one rank with the process group and the bucket machinery on -- the
    # rehearsal of the multi-rank path on a one-GPU box)
    if world > 1 or (os.environ.get("LDM_DP_FORCE") == "1" and "MASTER_ADDR" in os.environ):
        backend = os.environ.get("LDM_DP_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from src import ops
    from src.dp import GradBucketDataParallel, HipAccelerator
    from src.masking_strategy import create_masking_strategy
    from src.model import DiffusionLanguageModel
    from src.trainer import DiffusionLanguageModelTrainer
    from src.transformer import DiffusionTransformer, DiffusionTransformerConfig

    cfg = CONFIGS[args.config]
    torch.manual_seed(42)
    random.seed(42)     # the reference seeds both identically on every rank (main.py:39-40)
    tr = DiffusionTransformer(DiffusionTransformerConfig(dropout=args.dropout, layer_norm_eps=1e-12,
                                                         max_position_embeddings=max(512, args.seq), **cfg))
    tr.fp32_residual = bool(args.fp32_residual)
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
    trainer.objective = args.objective
    model.train()

    ids, att = synthetic_tokens(args.batch, args.seq, cfg["vocab_size"], 1234 + rank)
    batch = {"input_ids": ids.to(dev), "attention_mask": att.to(dev)}

    def sync():
        # device first: the gradient all-reduces of the last step (the data-parallel wrapper's own RCCL
        # communicator) have completed before the barrier's collective (torch.distributed's) starts
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # how many ranks does the collective see?  (an all-reduce of ones over the gradient backend)
    ranks_seen = 1
    if world > 1:
        one = torch.ones(1, device=dev) if backend == "nccl" else torch.ones(1)
        dist.all_reduce(one)
        ranks_seen = int(one.item())

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
    want_events = os.environ.get("LDM_BENCH_EVENTS", "1") != "0" and rank == 0
    if want_events:
        ops.KERNEL_EVENTS = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = read(trainer._step(batch))
    sync()
    elapsed = time.perf_counter() - t0
    events = ops.KERNEL_EVENTS or {}
    ops.KERNEL_EVENTS = None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # Second leg: the same step with the YAML's dropout 0.1 (configs/default.yaml:13) -- attention-probability
    # dropout inside the attention kernels and the two residual dropouts in the projection epilogues --, on
    # fewer steps, behind the headline timing and outside it.  Every rank runs it (the collectives match).
    drop_leg = None
    if args.dropout == 0.0 and not args.no_dropout_leg:
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.1
        k2 = max(1, min(args.steps, 5))
        for _ in range(2):
            last2 = read(trainer._step(batch))
        sync()
        t1 = time.perf_counter()
        for _ in range(k2):
            last2 = read(trainer._step(batch))
        sync()
        e2 = time.perf_counter() - t1
        if world > 1:
            t2 = torch.tensor([e2], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            e2 = float(t2.item())
        drop_leg = {"dropout": 0.1, "steps": k2, "warmup": 2, "ms_per_step": 1e3 * e2 / k2,
                    "value": args.batch * args.seq * world * k2 / e2, "unit": "tokens/s", "loss_last_step": last2}

    if rank == 0:
        tokens = args.batch * args.seq * world * args.steps
        tps = tokens / elapsed
        fpt = train_flops_per_token(cfg, args.seq)
        per = summarize_events(events, args.steps)
        by_kernel = []
        for tag, (bound, name) in CLASSES.items():
            if tag not in per:
                continue
            e = per[tag]
            sec = e["ms_per_step"] * 1e-3
            if bound == "mfma":
                ach, peak, unit = e["work_per_step"] / sec / 1e12, PEAK_BF16_TFLOPS, "TFLOP/s"
            else:
                ach, peak, unit = e["work_per_step"] / sec / 1e9, PEAK_HBM_GBS, "GB/s"
            by_kernel.append({"class": tag, "kernel": name, "bound": bound, "launches_per_step": e["launches_per_step"],
                              "ms_per_step": e["ms_per_step"], "work_per_step": e["work_per_step"],
                              "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak})
        # the class the step spends most of its time in: every role of the 256-row MFMA GEMM together
        gemm = [k for k in by_kernel if k["class"].startswith("gemm.")]
        g_ms = sum(k["ms_per_step"] for k in gemm)
        g_fl = sum(k["work_per_step"] for k in gemm)
        g_n = sum(k["launches_per_step"] for k in gemm)
        attn_fl = sum(k["work_per_step"] for k in by_kernel if k["class"].startswith("attn."))
        attn_ms = sum(k["ms_per_step"] for k in by_kernel if k["class"].startswith("attn."))
        head_rows = 1.0
        hf = [e for e in events.get("gemm.fwd", []) if e[3] is not None]
        if hf:
            head_rows = sum(float(e[3].item()) for e in hf) / (len(hf) * args.batch * args.seq)
        d_, V_ = cfg["d_model"], cfg["vocab_size"]
        # work actually executed: the three LM-head GEMMs run on the participating rows only
        fpt_exec = fpt - 6.0 * d_ * V_ * (1.0 - head_rows)
        achieved = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        tj = committed_traffic(args.config, args.dtype, args.batch, args.seq)
        out = {
            "metric": "masked-diffusion train tokens/sec (node)", "value": tps, "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"configs/default.yaml model, seq {args.seq}" if args.config == "default"
                       else f"{args.config} model, seq {args.seq}",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "seq_len": args.seq,
                       "parallelism": f"dp{world}", "dropout": args.dropout, "objective": args.objective,
                       "residual_stream": "fp32" if args.fp32_residual else args.dtype,
                       "optimizer": "AdamW(lr=5e-4, wd=0.1)", **cfg},
            "ranks_seen": ranks_seen,
            "grad_buckets": model.num_buckets if isinstance(model, GradBucketDataParallel) else 0,
            "dp_backend": backend,
            "loss_last_step": last,
            "train_flops_per_token": fpt,
            # fraction of the bf16 MFMA peak: with the flops executed here (LM head on the rows the loss
            # reads) -- the figure to quote -- and with the flops of the reference's step (LM head on
            # every position), which credits work that is not done
            "mfma_fraction_of_step": tps * fpt_exec / (world * PEAK_BF16_TFLOPS * 1e12),
            "mfma_fraction_algorithmic": tps * fpt / (world * PEAK_BF16_TFLOPS * 1e12),
            "executed_flops_per_token": fpt_exec, "lm_head_rows_fraction": head_rows,
            "mfma_fraction_attn_plus_gemm": ((g_fl + attn_fl) / ((g_ms + attn_ms) * 1e-3) / 1e12 / PEAK_BF16_TFLOPS
                                             if g_ms + attn_ms > 0 else None),
            "roofline": {"kernel": "the 256x256-tile MFMA GEMMs, gemm256_k (all variants) and gemm4w_k (the time-dominant kernel class of the step)",
                         "bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS,
                         "traffic": tj["gemm_bytes_per_launch"] if tj else None,
                         "traffic_source": ("profiles/r3_gemm_traffic.json: rocprofv3 --pmc passes of this command at this "
                                            "shape on the committed tree (scripts/gpu_r3_profile.sh), not counted in this run"
                                            if tj else None),
                         "flops_per_launch": g_fl / g_n if g_n else 0.0,
                         "avg_launch_ms": g_ms / g_n if g_n else 0.0,
                         "launches_per_step": g_n, "ms_per_step": g_ms,
                         "share_of_step": g_ms / (1e3 * elapsed / args.steps) if elapsed > 0 else None},
            "roofline_by_kernel": by_kernel,
            "dropout01": drop_leg,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args.seq)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        if isinstance(model, GradBucketDataParallel):
            model.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
