"""Same-box timing of the q/k/v projection: plain GEMM (+ separate RoPE pass) against the GEMM with the
rotary embedding in its epilogue.  python scripts/bench_qkv_rope.py [T] [D] [hd] [S]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from src import _native as N  # noqa: E402

T, D, hd, S = (int(a) for a in (sys.argv[1:5] + ["262144", "768", "96", "512"][len(sys.argv) - 1:]))
K = D
dev = torch.device("cuda:0")
lib = N.lib()
x = torch.randn(T, K, device=dev).bfloat16()
W = (torch.randn(3 * D, K, device=dev) / math.sqrt(K)).bfloat16()
b = torch.randn(3 * D, device=dev)
inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).double() / hd))
ang = torch.arange(S).double()[:, None] * inv[None, :]
ch, sh = ang.cos().float().to(dev).contiguous(), ang.sin().float().to(dev).contiguous()
step = torch.stack(((16 * inv).cos(), (16 * inv).sin())).float().contiguous().to(dev)
out = torch.empty(T, 3 * D, device=dev, dtype=torch.bfloat16)
st = N.stream()


def plain():
    N.check(lib.ldm_gemm(N.BF16, N.BF16, 0, 1, T, 3 * D, K, 1.0, x.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), 3 * D,
                         b.data_ptr(), None, 0, st), "gemm")


def rope_pass():
    N.check(lib.ldm_rope_qk(out.data_ptr(), N.BF16, T // S, S, D // hd, hd, 3 * D, ch.data_ptr(), sh.data_ptr(), 0.1, 0, st), "rope")


def fused():
    N.check(lib.ldm_gemm_qkv_rope(T, D, K, hd, S, x.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), ch.data_ptr(), sh.data_ptr(),
                                  step.data_ptr(), 0.1, out.data_ptr(), 3 * D, st), "qkv_rope")


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


fl = 2.0 * T * 3 * D * K
for rnd in range(3):
    tp, tr, tf = timeit(plain), timeit(rope_pass), timeit(fused)
    print(f"plain {tp*1e3:.0f} us ({fl/tp/1e9:.0f} TF/s)  rope pass {tr*1e3:.0f} us  fused {tf*1e3:.0f} us ({fl/tf/1e9:.0f} TF/s)")
