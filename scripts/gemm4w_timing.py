"""Per-item segment cycles of the four-wave GEMM (needs the -DLDM_GEMM_TIMING library: make timing)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
buf = torch.zeros(32, dtype=torch.int64, device=dev)
os.environ["LDM_GEMM4W_TIMING_BUF"] = str(buf.data_ptr())
os.environ["LDM_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                     "language-diffusion-model_amd", "lib", "libldm_hip_timing.so")
from src import _native as N
T = int(os.environ.get("T", 262144))
names = ["first data", "main loop", "bias req + next prologue", "bias/residual wait", "pack + 8 stores"]
for tag, M, Nn, K, bias, res in [("dx out (N=768,K=768)", T, 768, 768, False, False), ("dx qkv (N=768,K=2304)", T, 768, 2304, False, False),
                                 ("plain + bias N=2304,K=768", T, 2304, 768, True, False),
                                 ("out + bias + residual", T, 768, 768, True, True), ("down + bias + residual", T, 768, 1024, True, True),
                                 ("8192^3", 8192, 8192, 8192, False, False)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    b = (torch.randn(Nn, K, device=dev) / K ** 0.5).bfloat16()
    c = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
    bv = torch.randn(Nn, device=dev) if bias else None
    r = torch.randn(M, Nn, device=dev).bfloat16() if res else None
    def run():
        N.check(N.lib().ldm_gemm(1, 1, 0, 1, M, Nn, K, 1.0, a.data_ptr(), K, b.data_ptr(), K, c.data_ptr(), Nn, N.ptr(bv), N.ptr(r), Nn if res else 0,
                                 N.stream()), "gemm")
    run(); torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    t = buf.view(4, 8).cpu().double()
    us = e0.elapsed_time(e1) * 1e3
    print(f"{tag}: {us:.1f} us, {2.0 * M * Nn * K / us / 1e6:.0f} TF (instrumented)")
    for w in (0, 3):
        n = t[w, 5].item()
        if n == 0:
            print("  (the four-wave kernel did not run)")
            break
        tot = t[w, :5].sum().item()
        print(f"  wave {w}: items {int(n)}, {tot / n:.0f} per item ({tot / us / 1e3:.2f} GHz if the workgroup was busy throughout): "
              + "  ".join(f"{names[i]} {t[w, i].item() / n:.0f}" for i in range(5)))
        steps = K // 32
        print(f"          main loop per step: first 6 (deferred stores) {t[w, 6].item() / n / 6:.0f}, middle {steps - 10} "
              f"{(t[w, 1] - t[w, 6] - t[w, 7]).item() / n / max(steps - 10, 1):.0f}, last 4 {t[w, 7].item() / n / 4:.0f}")
    del a, b, c, r
