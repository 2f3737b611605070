"""Data-parallel gradient sync over 2 ranks on the CPU (gloo): the bucketed all-reduce must
leave every rank with the MEAN of the per-rank gradients (DDP semantics, each rank's loss
normalised by its own masked count), also when a parameter gets no gradient on some rank."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(16, 32)
        self.b = torch.nn.Linear(32, 8)
        self.unused = torch.nn.Linear(4, 4)
        self.tokenizer = "tok"

    def forward(self, x, use_extra=False):
        y = self.b(torch.tanh(self.a(x)))
        if use_extra:
            y = y + self.unused(x[:, :4]).sum(-1, keepdim=True)
        return y


def _local_grads(rank, use_extra):
    torch.manual_seed(0)
    net = _Net()
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(5 + rank, 16, generator=g)
    net(x, use_extra).pow(2).mean().backward()
    return {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in net.named_parameters()}


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from src.dp import GradBucketDataParallel
    torch.manual_seed(rank)          # different init per rank: the wrapper must broadcast rank 0's
    net = _Net()
    dp = GradBucketDataParallel(net, bucket_bytes=1024)   # tiny buckets -> several collectives
    assert dp.num_buckets >= 3 and dp.tokenizer == "tok"
    out = {}
    for step, use_extra in enumerate([rank == 0, True]):   # step 0: `unused` has a grad on rank 0 only
        for p in net.parameters():
            p.grad = None
        g = torch.Generator().manual_seed(100 + rank)
        x = torch.randn(5 + rank, 16, generator=g)
        dp(x, use_extra).pow(2).mean().backward()
        out[step] = {n: p.grad.clone().numpy() for n, p in net.named_parameters()}
    # deferred scaling: the buckets keep the SUM over ranks (an optimizer applies 1 / world_size)
    dp.defer_scale = True
    for p in net.parameters():
        p.grad = None
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(5 + rank, 16, generator=g)
    dp(x, True).pow(2).mean().backward()
    out[2] = {n: p.grad.clone().numpy() for n, p in net.named_parameters()}
    # numpy payloads: torch tensors would travel as shared-memory handles that die with this process
    q.put((rank, out, {n: p.detach().clone().numpy() for n, p in net.named_parameters()}))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, out, params = q.get(timeout=120)
        res[rank] = (out, params)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # parameters were broadcast from rank 0 (same values as a seed-0 model)
    torch.manual_seed(0)
    ref_net = _Net()
    for n, p in ref_net.named_parameters():
        assert torch.equal(torch.from_numpy(res[0][1][n]), p.detach()) and torch.equal(torch.from_numpy(res[1][1][n]), p.detach())
    for step, extras in enumerate([(True, False), (True, True)]):
        want = {}
        for r in range(2):
            for n, g in _local_grads(r, extras[r]).items():
                want[n] = want.get(n, 0) + g / 2
        for r in range(2):
            for n, g in res[r][0][step].items():
                assert torch.allclose(torch.from_numpy(g), want[n], rtol=1e-6, atol=1e-7), (step, r, n)
    # step 2 ran with defer_scale: gradients are the sum, i.e. twice the mean of step 1's inputs
    for r in range(2):
        for n, g in res[r][0][2].items():
            assert torch.allclose(torch.from_numpy(g), 2 * want[n], rtol=1e-6, atol=1e-7), ("deferred", r, n)


def _worker_bf16_wire(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from src.dp import GradBucketDataParallel
    torch.manual_seed(0)
    net = _Net()
    dp = GradBucketDataParallel(net, bucket_bytes=1024, wire_dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(5 + rank, 16, generator=g)
    dp(x, True).pow(2).mean().backward()
    q.put((rank, {n: p.grad.clone().numpy() for n, p in net.named_parameters()}))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_bf16_wire():
    """Gradient buckets exchanged in bf16 (half the bytes over the links): every rank ends with the same
    fp32 gradients, equal to the mean of the per-rank gradients up to one bf16 rounding per rank and one of
    the sum."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_bf16_wire, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, grads = q.get(timeout=120)
        res[rank] = grads
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = {}
    for r in range(2):
        for n, g in _local_grads(r, True).items():
            want[n] = want.get(n, 0) + g / 2
    for n in want:
        a, b = torch.from_numpy(res[0][n]), torch.from_numpy(res[1][n])
        assert torch.equal(a, b), n
        err = float((a - want[n]).norm() / max(float(want[n].norm()), 1e-30))
        assert err < 1e-2, (n, err)
