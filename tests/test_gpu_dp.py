"""The multi-rank training path on ONE GPU (a one-GPU box cannot run two RCCL ranks, so the two
ranks talk over gloo with host-staged buckets; everything else -- bucket assignment, hooks, launch
order, 1/W scaling -- is the code the RCCL path runs).

* same batch on both ranks + FusedAdamW taking over the 1 / world_size scaling: the parameters after
  three steps must equal a single-process run;
* DIFFERENT batches per rank: every rank's gradients after backward must equal the oracle's mean of
  the per-rank gradients (each rank's loss normalised by its own masked count -- what DDP computes
  for the reference, main.py:105-108 + src/trainer.py:136), dense head in fp32 and fused head in bf16;
* `python bench.py --gpus 2` typed as a plain command starts its own two ranks and reports
  ranks_seen = 2;
* RCCL itself with ONE rank: the C-ABI communicator and bucket all-reduce (ldm_comm_*,
  ldm_allreduce_bucket, ldm_allreduce_join), and the wrapper over the nccl backend taking that path."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run_steps(dev, distributed):
    import random
    import types
    from src.dp import GradBucketDataParallel, HipAccelerator
    from src.masking_strategy import create_masking_strategy
    from src.model import DiffusionLanguageModel
    from src.optim import FusedAdamW
    from src.trainer import DiffusionLanguageModelTrainer
    from src.transformer import DiffusionTransformer, DiffusionTransformerConfig
    torch.manual_seed(0)
    random.seed(0)
    tr = DiffusionTransformer(DiffusionTransformerConfig(vocab_size=500, d_model=64, nhead=2, num_layers=2,
                                                         dim_feedforward=128, dropout=0.0, layer_norm_eps=1e-12,
                                                         max_position_embeddings=64))
    tok = types.SimpleNamespace(mask_token_id=103, bos_token_id=101, eos_token_id=102, pad_token_id=0)
    model = DiffusionLanguageModel(tr, tok, create_masking_strategy("random", 101, 102, 0, 103), None)
    acc = HipAccelerator(mixed_precision="no", device=dev, bucket_bytes=64 << 10)
    model = acc.prepare(model)
    opt = acc.prepare(FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.1))
    if distributed:
        assert isinstance(model, GradBucketDataParallel) and model.defer_scale and opt.grad_scale == 0.5
        assert model.num_buckets >= 2
    trainer = DiffusionLanguageModelTrainer(model, opt, None, acc)
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(110, 500, (4, 32), generator=g)
    ids[:, 0], ids[:, -1] = 101, 102
    batch = {"input_ids": ids.to(dev), "attention_mask": torch.ones_like(ids).to(dev)}
    random.seed(3)
    torch.manual_seed(3)
    losses = [trainer.loss_value(trainer._step(batch)) for _ in range(3)]
    inner = model.module if distributed else model
    return losses, {n: p.detach().cpu().numpy() for n, p in inner.named_parameters()}


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    losses, params = _run_steps(torch.device("cuda", 0), True)
    q.put((rank, losses, params))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_fused_adamw_equal_single_process(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, losses, params = q.get(timeout=240)
        res[rank] = (losses, params)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref_losses, ref_params = _run_steps(dev, False)
    for r in range(2):
        assert np.allclose(res[r][0], ref_losses, rtol=1e-5), (res[r][0], ref_losses)
        # norm-relative per tensor: the split-K / scatter atomics make the last bits of a gradient
        # run-dependent, and Adam's first steps turn a near-zero gradient's sign into +-lr
        for n, v in ref_params.items():
            err = np.linalg.norm(res[r][1][n] - v) / max(np.linalg.norm(v), 1e-30)
            assert err < 1e-4, (r, n, err)


# --------------------------------------------------------------------------- different data per rank
class _InjectedDraw:
    def __init__(self, inner, draw):
        self.inner, self.draw = inner, draw

    def apply_random_mask(self, input_ids, mask_prob):
        return self.inner.apply_random_mask(input_ids, mask_prob, draw=self.draw)


def _dp_case(kind):
    """(oracle config, batch, seq, dtype, fused head?) of the two different-data cases."""
    from oracle import step_oracle as so
    if kind == "tiny_fp32_dense":
        return so.OracleConfig(vocab_size=500, d_model=64, nhead=2, num_layers=2, dim_feedforward=128), 3, 32, torch.float32, False
    # default width, one layer, T = 1024 rows per rank: the bf16 fused LM-head loss takes this shape
    return so.OracleConfig(vocab_size=30522, d_model=768, nhead=8, num_layers=1, dim_feedforward=1024), 2, 512, torch.bfloat16, True


def _rank_inputs(kind, rank):
    from oracle import step_oracle as so
    ocfg, B, S, _, _ = _dp_case(kind)
    ids, att = so.synthetic_batch(B, S, ocfg.vocab_size, seed=100 + rank, ragged=True, dirty=(rank == 1))
    draw = torch.rand(B, S, generator=torch.Generator().manual_seed(200 + rank))
    return ids, att, draw, 0.35 + 0.3 * rank       # a different t per rank as well


def _grad_worker(rank, world, port, kind, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import types
    import torch.distributed as dist
    from oracle import step_oracle as so
    from src import ops
    from src.dp import GradBucketDataParallel, HipAccelerator
    from src.loss import variational_lower_bound_loss
    from src.masking_strategy import create_masking_strategy
    from src.model import DiffusionLanguageModel
    from src.transformer import DiffusionTransformer, DiffusionTransformerConfig
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ocfg, B, S, dtype, fused = _dp_case(kind)
    params = so.init_params(ocfg, seed=11)
    tr = DiffusionTransformer(DiffusionTransformerConfig(
        vocab_size=ocfg.vocab_size, d_model=ocfg.d_model, nhead=ocfg.nhead, num_layers=ocfg.num_layers,
        dim_feedforward=ocfg.dim_feedforward, dropout=0.0, layer_norm_eps=1e-12, max_position_embeddings=512))
    tr.load_state_dict(params, strict=False)
    tok = types.SimpleNamespace(mask_token_id=103, bos_token_id=101, eos_token_id=102, pad_token_id=0)
    model = DiffusionLanguageModel(tr, tok, create_masking_strategy("random", 101, 102, 0, 103), None)
    acc = HipAccelerator(mixed_precision="bf16" if dtype == torch.bfloat16 else "no", device=dev,
                         bucket_bytes=(64 << 10) if ocfg.d_model == 64 else (32 << 20))
    model = acc.prepare(model)
    assert isinstance(model, GradBucketDataParallel) and model.num_buckets >= 2 and not model.defer_scale
    ids, att, draw, t = _rank_inputs(kind, rank)
    ids, att = ids.to(dev), att.to(dev)
    model.module.masking_strategy = _InjectedDraw(model.module.masking_strategy, draw.to(dev))
    model.train()
    used0 = ops.FUSED_HEAD_STATS["used"]
    if fused:
        loss = model.module.training_loss(ids, att, t=torch.tensor([t]))
    else:
        logits, masked, _ = model(input_ids=ids, attention_mask=att, t=torch.tensor([t]))
        loss = variational_lower_bound_loss(logits, ids, masked, 103, att)
    assert (ops.FUSED_HEAD_STATS["used"] == used0 + 1) == fused
    acc.backward(loss)
    torch.cuda.synchronize()
    # numpy, not tensors: a tensor in a Queue is handed over by file descriptor and the sender may be gone
    grads = {n: p.grad.detach().float().cpu().numpy() for n, p in model.module.transformer.named_parameters()}
    q.put((rank, float(loss.item()), grads))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["tiny_fp32_dense", "default_width_bf16_fused"])
def test_two_ranks_different_data_equal_oracle_mean(dev, kind):
    from conftest import rel_err
    from oracle import step_oracle as so
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 1500) + (7 if kind.startswith("tiny") else 0)
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    # the oracle runs while the ranks do: per-rank loss and gradients, then their mean
    ocfg, B, S, dtype, _ = _dp_case(kind)
    params = so.init_params(ocfg, seed=11)
    per_rank = []
    for r in range(2):
        ids, att, draw, t = _rank_inputs(kind, r)
        ol, og, _, _, _ = so.step_loss_and_grads(params, ocfg, ids, att, t, draw)
        per_rank.append((float(ol), og))
    mean = {n: (per_rank[0][1][n] + per_rank[1][1][n]) / 2 for n in per_rank[0][1]}
    res = {}
    for _ in range(2):
        rank, loss, grads = q.get(timeout=300)
        res[rank] = (loss, {n: torch.from_numpy(v) for n, v in grads.items()})
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for r in range(2):
        assert abs(res[r][0] - per_rank[r][0]) <= tol * abs(per_rank[r][0])     # each rank's OWN loss
        names = [n for n in mean if not n.endswith("k_proj.bias")]
        mine = torch.cat([res[r][1][n].flatten() for n in names])
        want = torch.cat([mean[n].flatten() for n in names])
        assert rel_err(mine, want) < tol, (r, "whole gradient")
        for n in names:
            assert rel_err(res[r][1][n], mean[n]) < tol, (r, n)
    # and both ranks hold the same reduced gradient
    for n in mean:
        assert torch.equal(res[0][1][n], res[1][1][n]), n


def test_bench_starts_its_own_ranks(dev):
    """`python bench.py --gpus 2 ...` with no launcher: one JSON line, n_gpus 2, ranks_seen 2."""
    import json
    import subprocess
    env = dict(os.environ, LDM_DP_BACKEND="gloo", LDM_DP_SAME_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "16", "--steps", "3",
                          "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["grad_buckets"] >= 2
    assert rec["config"]["global_batch"] == 32 and rec["value"] > 0
    # a child that fails makes the command fail
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "16", "--config",
                          "nonexistent"], env=env, capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0


# --------------------------------------------------------------------------- RCCL through the C-ABI
def _rccl_one_rank_worker(q):
    """ldm_comm_* + ldm_allreduce_bucket on a one-rank communicator: the sum over one rank is the input,
    so what is checked is that RCCL loads, the communicator comes up, the collective runs on the
    communication stream BEHIND the kernel that produced the bucket, and the join orders the reader."""
    sys.path.insert(0, ROOT)
    os.environ.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    import ctypes as C
    from src import _native as N
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    lib = N.lib()
    ident = C.create_string_buffer(128)
    N.check(lib.ldm_comm_unique_id(ident), "ldm_comm_unique_id")
    comm = C.c_void_p()
    N.check(lib.ldm_comm_init_rank(C.byref(comm), 1, ident.raw, 0), "ldm_comm_init_rank")
    n = C.c_int(0)
    N.check(lib.ldm_comm_size(comm, C.byref(n)), "ldm_comm_size")
    assert n.value == 1
    side = torch.cuda.Stream(device=dev)
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        big = torch.randn(64 << 20, device=dev)                 # a long producer kernel in front of the collective
        buf = torch.zeros(1 << 20, dtype=dt, device=dev)
        cur = torch.cuda.current_stream().cuda_stream
        big.mul_(1.0001)
        buf.copy_(torch.arange(1 << 20, device=dev) % 251)       # written on the compute stream
        want = buf.clone()
        N.check(lib.ldm_allreduce_bucket(comm, buf.data_ptr(), buf.numel(), N.dtype_code(dt), cur, side.cuda_stream),
                "ldm_allreduce_bucket")
        N.check(lib.ldm_allreduce_join(side.cuda_stream, cur), "ldm_allreduce_join")
        got = buf * 2                                             # a reader on the compute stream
        torch.cuda.synchronize()
        out[str(dt)] = bool(torch.equal(got, want * 2))
    # argument checks answer without touching RCCL
    assert lib.ldm_allreduce_bucket(comm, 8, 4, 7, None, None) == -4          # dtype
    assert lib.ldm_allreduce_bucket(None, 16, 4, 0, None, None) == -1
    assert lib.ldm_allreduce_bucket(comm, 8, 4, 0, None, None) == -2          # alignment
    N.check(lib.ldm_comm_destroy(comm), "ldm_comm_destroy")
    q.put(out)


def test_rccl_bucket_allreduce_through_the_c_abi_one_rank(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank_worker, args=(q,))
    p.start()
    out = q.get(timeout=170)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert out == {"torch.float32": True, "torch.bfloat16": True}, out


def _native_dp_worker(port, q):
    """The data-parallel wrapper over the RCCL backend with one rank (LDM_DP_FORCE=1): the buckets go
    through ldm_allreduce_bucket; three trainer steps must equal the plain single-process run."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", LDM_DP_FORCE="1")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import random
    import types
    from src.dp import GradBucketDataParallel, HipAccelerator
    from src.masking_strategy import create_masking_strategy
    from src.model import DiffusionLanguageModel
    from src.optim import FusedAdamW
    from src.trainer import DiffusionLanguageModelTrainer
    from src.transformer import DiffusionTransformer, DiffusionTransformerConfig
    res = {}
    for wire in (torch.float32, torch.bfloat16):
        torch.manual_seed(0)
        random.seed(0)
        tr = DiffusionTransformer(DiffusionTransformerConfig(vocab_size=500, d_model=64, nhead=2, num_layers=2,
                                                             dim_feedforward=128, dropout=0.0, layer_norm_eps=1e-12,
                                                             max_position_embeddings=64))
        tok = types.SimpleNamespace(mask_token_id=103, bos_token_id=101, eos_token_id=102, pad_token_id=0)
        model = DiffusionLanguageModel(tr, tok, create_masking_strategy("random", 101, 102, 0, 103), None)
        acc = HipAccelerator(mixed_precision="no", device=dev, bucket_bytes=64 << 10, grad_wire_dtype=wire)
        model = acc.prepare(model)
        assert isinstance(model, GradBucketDataParallel) and model._comm is not None and model.num_buckets >= 2
        opt = acc.prepare(FusedAdamW(model.parameters(), lr=1e-3, weight_decay=0.1))
        trainer = DiffusionLanguageModelTrainer(model, opt, None, acc)
        g = torch.Generator().manual_seed(7)
        ids = torch.randint(110, 500, (4, 32), generator=g)
        ids[:, 0], ids[:, -1] = 101, 102
        batch = {"input_ids": ids.to(dev), "attention_mask": torch.ones_like(ids).to(dev)}
        random.seed(3)
        torch.manual_seed(3)
        losses = [trainer.loss_value(trainer._step(batch)) for _ in range(3)]
        res[str(wire)] = (losses, {n: p.detach().cpu().numpy() for n, p in model.module.named_parameters()})
        model.close()
        assert model._comm is None
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_dp_wrapper_over_rccl_backend_uses_the_native_allreduce(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 2000)
    p = ctx.Process(target=_native_dp_worker, args=(port, q))
    p.start()
    res = q.get(timeout=170)
    p.join(timeout=60)
    assert p.exitcode == 0
    ref_losses, ref_params = _run_steps(dev, False)
    for wire, (losses, params) in res.items():
        tol = 1e-5 if wire == "torch.float32" else 2e-2
        assert np.allclose(losses, ref_losses, rtol=tol), (wire, losses, ref_losses)
        for n, v in ref_params.items():
            err = np.linalg.norm(params[n] - v) / max(np.linalg.norm(v), 1e-30)
            assert err < (1e-4 if wire == "torch.float32" else 2e-2), (wire, n, err)
