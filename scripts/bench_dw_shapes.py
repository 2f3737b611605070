"""Weight-gradient GEMM (A [K, M], B [K, N], fp32 accumulate with row sums) over a list of shapes M,N,K;...
(environment SHAPES), through the C ABI; LDM_GEMM_4W_DW=0/1 selects the kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from src import _native as N

dev = torch.device("cuda:0")
shapes = [tuple(int(x) for x in s.split(",")) for s in os.environ.get("SHAPES", "2304,768,262144").split(";")]
reps = int(os.environ.get("REPS", 10))
_w = torch.randn(8192, 8192, device=dev).bfloat16()
for _ in range(30):
    torch.matmul(_w, _w)
torch.cuda.synchronize()
del _w
for M, Nn, K in shapes:
    a = torch.randn(K, M, device=dev).bfloat16()
    b = torch.randn(K, Nn, device=dev).bfloat16()
    c = torch.zeros(M, Nn, device=dev)
    rs = torch.zeros(M, device=dev)
    def run():
        N.check(N.lib().ldm_gemm_acc_rowsum(1, 0, M, Nn, K, 1.0, a.data_ptr(), M, b.data_ptr(), Nn, c.data_ptr(), Nn,
                                            rs.data_ptr(), N.stream()), "acc_rowsum")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"dw M={M:6d} N={Nn:6d} K={K:7d}  {ms*1e3:9.1f} us  {2.0*M*Nn*K/ms/1e9:8.1f} TF", flush=True)
