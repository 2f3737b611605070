"""Per-tile cycle breakdown (main loop / epilogue + set-up / first-data wait) of the 256-row GEMM's epilogue
variants at the training step's shapes (needs the -DLDM_GEMM_TIMING library: make timing)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
buf = torch.zeros(64, dtype=torch.int64, device=dev)
os.environ["LDM_GEMM_TIMING_BUF"] = str(buf.data_ptr())
os.environ["LDM_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                     "language-diffusion-model_amd", "lib", "libldm_hip_timing.so")
from src import _native as N
lib = N.lib()
T = int(os.environ.get("T", 262144))
d, F, hd, S = 768, 1024, 96, 512
st = N.stream()
x = torch.randn(T, d, device=dev).bfloat16()
xF = torch.randn(T, F, device=dev).bfloat16()
res = torch.randn(T, d, device=dev).bfloat16()
W = (torch.randn(d, d, device=dev) / math.sqrt(d)).bfloat16()
W3 = (torch.randn(d, F, device=dev) / math.sqrt(F)).bfloat16()
Wq = (torch.randn(3 * d, d, device=dev) / math.sqrt(d)).bfloat16()
W12 = (torch.randn(2 * F, d, device=dev) / math.sqrt(d)).bfloat16()
b768, b2304, b2048 = (torch.randn(n, device=dev) for n in (768, 2304, 2048))
out768 = torch.empty(T, d, device=dev, dtype=torch.bfloat16)
out2304 = torch.empty(T, 3 * d, device=dev, dtype=torch.bfloat16)
ag = torch.randn(T, 2 * F, device=dev).bfloat16()
dag = torch.empty_like(ag)
h = torch.empty(T, F, device=dev, dtype=torch.bfloat16)
inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).double() / hd))
ang = torch.arange(S).double()[:, None] * inv[None, :]
ch, sh = ang.cos().float().to(dev).contiguous(), ang.sin().float().to(dev).contiguous()
stp = torch.stack(((16 * inv).cos(), (16 * inv).sin())).float().contiguous().to(dev)
cases = [
    ("plain 768x768", 2.0 * T * d * d, lambda: lib.ldm_gemm(1, 1, 0, 1, T, d, d, 1.0, x.data_ptr(), d, W.data_ptr(), d, out768.data_ptr(), d, None, None, 0, st)),
    ("bias+residual 768x768 (RES 1)", 2.0 * T * d * d, lambda: lib.ldm_gemm(1, 1, 0, 1, T, d, d, 1.0, x.data_ptr(), d, W.data_ptr(), d, out768.data_ptr(), d, b768.data_ptr(), res.data_ptr(), d, st)),
    ("bias+residual K=1024 (RES 1)", 2.0 * T * d * F, lambda: lib.ldm_gemm(1, 1, 0, 1, T, d, F, 1.0, xF.data_ptr(), F, W3.data_ptr(), F, out768.data_ptr(), d, b768.data_ptr(), res.data_ptr(), d, st)),
    ("SwiGLU backward (RES 2)", 2.0 * T * F * d, lambda: lib.ldm_gemm_swiglu_bwd(T, F, d, x.data_ptr(), d, W3.data_ptr(), F, ag.data_ptr(), 2 * F, dag.data_ptr(), 2 * F, st)),
    ("SwiGLU forward (RES 3)", 2.0 * T * 2 * F * d, lambda: lib.ldm_gemm_swiglu_fwd(T, F, d, x.data_ptr(), d, W12.data_ptr(), d, b2048.data_ptr(), ag.data_ptr(), 2 * F, h.data_ptr(), F, st)),
    ("q/k/v + RoPE (RES 4)", 2.0 * T * 3 * d * d, lambda: lib.ldm_gemm_qkv_rope(T, d, d, hd, S, x.data_ptr(), d, Wq.data_ptr(), d, b2304.data_ptr(), ch.data_ptr(), sh.data_ptr(), stp.data_ptr(), 0.1, out2304.data_ptr(), 3 * d, st)),
]
for tag, fl, fn in cases:
    N.check(fn(), tag); torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); N.check(fn(), tag); e1.record(); torch.cuda.synchronize()
    t = buf.view(8, 8).cpu().double()
    us = e0.elapsed_time(e1) * 1e3
    K = 1024 if "K=1024" in tag else 768
    ntile = t[0, 5].item() / (K // 32)
    print(f"{tag}: {us:.1f} us, {fl / us / 1e6:.0f} TF (instrumented), tiles per workgroup {ntile:.0f}")
    if ntile == 0:      # (this variant's epilogue does not write the stamps out)
        continue
    for w in (0, 4):
        print(f"  wave {w}: per tile: main loop {t[w, :5].sum().item() / ntile:.0f}  epilogue+setup {t[w, 6].item() / max(ntile - 1, 1):.0f}  first-data wait {t[w, 7].item() / ntile:.0f}")
