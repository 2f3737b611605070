"""LM-head forward shape (rows x 30720 x 768, bf16, bias) through ldm_gemm: time per launch.
LDM_GEMM_4W=0|1 picks the kernel, LDM_GEMM_BAND the band height of the tile walk."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from src import _native as N
dev = torch.device("cuda:0")
M, Nn, K = int(os.environ.get("ROWS", 131072)), 30720, 768
a = torch.randn(M, K, device=dev).bfloat16()
b = (torch.randn(Nn, K, device=dev) / K ** 0.5).bfloat16()
bias = torch.randn(Nn, device=dev)
c = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
def run():
    N.check(N.lib().ldm_gemm(1, 1, 0, 1, M, Nn, K, 1.0, a.data_ptr(), K, b.data_ptr(), K, c.data_ptr(), Nn, N.ptr(bias), None, 0, N.stream()), "gemm")
for _ in range(3):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"4w={os.environ.get('LDM_GEMM_4W', '1')} band={os.environ.get('LDM_GEMM_BAND', 'default')}: {ms * 1e3:.0f} us, {2.0 * M * Nn * K / ms / 1e9:.0f} TF")
