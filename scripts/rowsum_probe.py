import os, sys
sys.path.insert(0, "/root/repo")
import torch
from src import _native as N
dev = torch.device("cuda:0")
K = 262144
for M, Nn in [(2304, 768), (2048, 768), (2560, 768), (1792, 768), (2304, 1024), (2304, 512), (3072, 768), (1536,768)]:
    a = torch.randn(K, M, device=dev).bfloat16(); b = torch.randn(K, Nn, device=dev).bfloat16()
    c = torch.zeros(M, Nn, device=dev); rs = torch.zeros(M, device=dev)
    out = []
    for use in (0, 1):
        def run():
            if use: N.check(N.lib().ldm_gemm_acc_rowsum(1, 0, M, Nn, K, 1.0, a.data_ptr(), M, b.data_ptr(), Nn, c.data_ptr(), Nn, rs.data_ptr(), N.stream()), "x")
            else: N.check(N.lib().ldm_gemm_acc(1, 1, 0, M, Nn, K, 1.0, a.data_ptr(), M, b.data_ptr(), Nn, c.data_ptr(), Nn, N.stream()), "y")
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"M={M} N={Nn}: plain {out[0]:.1f} us, rowsum {out[1]:.1f} us  ({out[1]/out[0]-1:+.1%})", flush=True)
