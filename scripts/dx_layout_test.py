import os, sys
sys.path.insert(0, "/root/repo")
import torch
from src import _native as N
dev = torch.device("cuda:0")
T = 65536
for (Nn, K) in [(768, 2304), (768, 768), (768, 2048), (1024, 768), (768, 30528)]:
    M = T if K != 30528 else 32768
    for tb in (0, 1):
        a = torch.randn(M, K, device=dev).bfloat16()
        b = torch.randn((Nn, K) if tb else (K, Nn), device=dev).bfloat16()
        c = torch.zeros(M, Nn, device=dev, dtype=torch.bfloat16)
        def run():
            N.check(N.lib().ldm_gemm(1, 1, 0, tb, M, Nn, K, 1.0, a.data_ptr(), K, b.data_ptr(), b.stride(0), c.data_ptr(), Nn, None, None, 0, N.stream()), "gemm")
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"M={M} N={Nn} K={K} transB={tb}: {ms*1e3:8.1f} us {2.0*M*Nn*K/ms/1e9:7.1f} TF", flush=True)
