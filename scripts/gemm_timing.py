"""Per-segment cycle breakdown of the 256-row GEMM main loop (needs the -DLDM_GEMM_TIMING library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
buf = torch.zeros(64, dtype=torch.int64, device=dev)
os.environ["LDM_GEMM_TIMING_BUF"] = str(buf.data_ptr())
os.environ["LDM_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                     "language-diffusion-model_amd", "lib", "libldm_hip_timing.so")
from src import _native as N
T = int(os.environ.get("T", 32768))
names = ["reads+issue", "vmcnt wait", "barrier R", "mfma", "barrier M"]
for tag, ta, tb, M, Nn, K in [("fwd qkv", 0, 1, T, 2304, 768), ("fwd head", 0, 1, T, 30528, 768),
                              ("dx qkv", 0, 0, T, 768, 2304), ("dw qkv", 1, 0, 2304, 768, T),
                              ("dw head", 1, 0, 30528, 768, T)]:
    a = torch.randn((K, M) if ta else (M, K), device=dev).bfloat16()
    b = torch.randn((Nn, K) if tb else (K, Nn), device=dev).bfloat16()
    acc = tag.startswith("dw")
    c = torch.zeros(M, Nn, device=dev, dtype=torch.float32 if acc else torch.bfloat16)
    rs = torch.zeros(M, device=dev)
    def run():
        if acc and os.environ.get("ROWSUM"):
            N.check(N.lib().ldm_gemm_acc_rowsum(ta, tb, M, Nn, K, 1.0, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), Nn, rs.data_ptr(), N.stream()), "acc_rowsum")
        elif acc:
            N.check(N.lib().ldm_gemm_acc(1, ta, tb, M, Nn, K, 1.0, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), Nn, N.stream()), "acc")
        else:
            N.check(N.lib().ldm_gemm(1, 1, ta, tb, M, Nn, K, 1.0, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), Nn, None, None, 0, N.stream()), "gemm")
    run(); torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    t = buf.view(8, 8).cpu().double()
    us = e0.elapsed_time(e1) * 1e3
    cyc = t[0, :5].sum().item()
    print(f"{tag}: {us:.1f} us, {2.0 * M * Nn * K / us / 1e6:.0f} TF (instrumented); main-loop cycles of one workgroup {cyc:.0f} -> >= {cyc / us / 1e3:.2f} GHz")
    ntile = t[0, 5].item() / (K // 32)
    print(f"  per tile: main loop {t[0, :5].sum().item() / ntile:.0f}  epilogue+setup {t[0, 6].item() / max(ntile - 1, 1):.0f}  first-data wait {t[0, 7].item() / ntile:.0f}   (wave 4: {t[4, 6].item() / max(ntile - 1, 1):.0f} / {t[4, 7].item() / ntile:.0f})")
    if os.environ.get("TIMING_M"):   # library built with -DLDM_GEMM_TIMING_M: slots 6/7 = first 16 MFMAs / wait for the second half
        for w in (0, 4):
            n = t[w, 5].item()
            print(f"  wave {w}: segment M: first-half MFMAs {t[w, 6].item() / n:7.1f}  wait for second-half fragments {t[w, 7].item() / n:7.1f}")
    for w in (0, 4):
        n = t[w, 5].item()
        print(f"  wave {w}: " + "  ".join(f"{names[i]} {t[w, i].item() / n:7.1f}" for i in range(5)) + f"   total/phase {t[w, :5].sum().item() / n:7.1f}  phases {int(n)}")
