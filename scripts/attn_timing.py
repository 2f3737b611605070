"""Per-segment cycle breakdown of the attention forward kernel (needs `make timing`)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
buf = torch.zeros(48, dtype=torch.int64, device=dev)
os.environ["LDM_ATTN_TIMING_BUF"] = str(buf.data_ptr())
os.environ["LDM_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                     "language-diffusion-model_amd", "lib", "libldm_hip_timing.so")
from src import _native as N
B, S, H, hd = (int(x) for x in os.environ.get("SHAPE", "128,512,8,96").split(","))
D = H * hd
qkv = torch.randn(B * S, 3 * D, device=dev).bfloat16()
o = torch.empty(B * S, D, device=dev, dtype=torch.bfloat16)
saved = torch.empty(N.lib().ldm_attn_saved_bytes(1, B, H, S, hd), dtype=torch.uint8, device=dev)
def run():
    N.check(N.lib().ldm_attn_fwd(1, qkv.data_ptr(), 3 * D, None, B, H, S, hd, o.data_ptr(), D, saved.data_ptr(), N.stream()), "fwd")
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
t = buf.cpu().double()
n = t[5].item()
names = ["wait + barrier + issue next tile", "S = K Q^T (12 MFMA)", "softmax", "O += V^T P (12 MFMA)", "  of the first: vmcnt wait"]
if os.environ.get("LDM_ATTN_DMA") == "0":
    names = ["issue next-tile loads", "S = K Q^T (12 MFMA)", "softmax", "O += V^T P (12 MFMA)", "barriers + tile store"]
print(f"fwd {e0.elapsed_time(e1) * 1e3:.1f} us (instrumented); per tile of one wave, cycles:")
for i, nm in enumerate(names):
    print(f"  {nm:28s} {t[i].item() / n:8.1f}")
print(f"  total {t[:4].sum().item() / n:8.1f}  tiles {int(n)}; whole kernel for that wave {t[6].item():.0f} cycles; workgroups per slot {B * H * (S // 128) / 512:.0f}")

print("per wave, cycles per tile: until past the barrier", [round(x / n) for x in t[8:12].tolist()], " of which vmcnt wait", [round(x / n) for x in t[12:16].tolist()])
print("a workgroup from the middle of the grid, cycles: entry -> first tile", int(t[16]), " tile loop", int(t[17]), " loop end -> stores issued", int(t[18]),
      " stores drained", int(t[19]), "; lifetime", t[20].item() / 100, "us; kernel time / rounds =", e0.elapsed_time(e1) * 1e3 / (B * H * ((S + 127) // 128) / 512), "us")
go = torch.randn(B * S, D, device=dev).bfloat16()
dqkv = torch.empty_like(qkv)
ws = torch.empty(N.lib().ldm_attn_bwd_ws_bytes(1, B, H, S, hd), dtype=torch.uint8, device=dev)
def bwd():
    N.check(N.lib().ldm_attn_bwd(1, qkv.data_ptr(), 3 * D, None, B, H, S, hd, o.data_ptr(), go.data_ptr(), D, saved.data_ptr(),
                                 dqkv.data_ptr(), ws.data_ptr(), N.stream()), "bwd")
bwd(); torch.cuda.synchronize()
e0.record(); bwd(); e1.record(); torch.cuda.synchronize()
t = buf.cpu().double()
rounds = B * H * ((S + 127) // 128) / 512
print(f"bwd {e0.elapsed_time(e1) * 1e3:.1f} us (instrumented), {e0.elapsed_time(e1) * 1e3 / rounds:.2f} us per round of workgroups (both kernels)")
for nm, b0 in (("dQ", 24), ("dK/dV", 32)):
    print(f"  {nm}: a workgroup from the middle of the grid, cycles: entry -> first tile {int(t[b0])}  tile loop {int(t[b0 + 1])}"
          f"  loop end -> stores issued {int(t[b0 + 2])}  stores drained {int(t[b0 + 3])}; lifetime {t[b0 + 4].item() / 100:.2f} us;"
          f" of the tile loop: waiting for the LDS-DMA {int(t[b0 + 5])}, at the barrier {int(t[b0 + 6])}")
