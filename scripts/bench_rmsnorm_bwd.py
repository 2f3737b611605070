"""Times ldm_rmsnorm_bwd (with the residual-stream gradient) at the step's shape; prints us and GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from src import _native as N
dev = torch.device("cuda:0")
T, d = int(os.environ.get("T", 262144)), int(os.environ.get("D", 768))
lib = N.lib()
x = torch.randn(T, d, device=dev).bfloat16(); dy = torch.randn_like(x); dres = torch.randn_like(x)
w = torch.randn(d, device=dev); rstd = torch.rand(T, device=dev) + 0.5
dx = torch.empty_like(x); dw = torch.empty(d, device=dev)
ws = torch.empty(lib.ldm_rmsnorm_bwd_ws_floats(T, d), device=dev)
def run():
    N.check(lib.ldm_rmsnorm_bwd(x.data_ptr(), w.data_ptr(), rstd.data_ptr(), dy.data_ptr(), dres.data_ptr(), T, d, 1, dx.data_ptr(), dw.data_ptr(), ws.data_ptr(), N.stream()), "bwd")
for _ in range(3): run()
torch.cuda.synchronize()
for rnd in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"rmsnorm_bwd T={T} d={d}: {us:.1f} us  {4.0 * T * d * 2 / us / 1e3:.0f} GB/s")
