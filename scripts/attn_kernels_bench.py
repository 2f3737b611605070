"""Times the three attention kernels alone (default-model shape) with events; prints us per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from src import _native as N
dev = torch.device("cuda:0")
B, S, H, hd = (int(x) for x in os.environ.get("SHAPE", "128,512,8,96").split(","))
D = H * hd
torch.manual_seed(0)
qkv = torch.randn(B * S, 3 * D, device=dev).bfloat16()
go = torch.randn(B * S, D, device=dev).bfloat16()
o = torch.empty(B * S, D, device=dev, dtype=torch.bfloat16)
lib = N.lib()
saved = torch.empty(lib.ldm_attn_saved_bytes(1, B, H, S, hd), dtype=torch.uint8, device=dev)
dqkv = torch.empty_like(qkv)
ws = torch.empty(lib.ldm_attn_bwd_ws_bytes(1, B, H, S, hd), dtype=torch.uint8, device=dev)
UNROPE = bool(os.environ.get("UNROPE"))     # the step's form: rotary backward inside the attention backward's stores
MASK = bool(os.environ.get("MASK"))         # the step's form: an all-ones key-validity mask instead of none
if UNROPE:
    ang = torch.arange(S, device=dev, dtype=torch.float32)[:, None] * (10000.0 ** (-torch.arange(0, hd, 2, device=dev, dtype=torch.float32) / hd))[None, :]
    cos_t, sin_t = ang.cos().contiguous(), ang.sin().contiguous()
kv = torch.ones(B, S, dtype=torch.uint8, device=dev) if MASK else None
def fwd():
    N.check(lib.ldm_attn_fwd(1, qkv.data_ptr(), 3 * D, N.ptr(kv), B, H, S, hd, o.data_ptr(), D, saved.data_ptr(), N.stream()), "fwd")
def bwd():
    if UNROPE:
        N.check(lib.ldm_attn_bwd_unrope(1, qkv.data_ptr(), 3 * D, N.ptr(kv), B, H, S, hd, o.data_ptr(), go.data_ptr(), D, saved.data_ptr(),
                                        dqkv.data_ptr(), ws.data_ptr(), cos_t.data_ptr(), sin_t.data_ptr(), 1.0 / hd ** 0.5, N.stream()), "bwd")
    else:
        N.check(lib.ldm_attn_bwd(1, qkv.data_ptr(), 3 * D, N.ptr(kv), B, H, S, hd, o.data_ptr(), go.data_ptr(), D, saved.data_ptr(),
                                 dqkv.data_ptr(), ws.data_ptr(), N.stream()), "bwd")
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
fl = 4.0 * B * H * S * S * hd
tf, tb = timeit(fwd), timeit(bwd)
print(f"LDM_ATTN_DMA={os.environ.get('LDM_ATTN_DMA', '1')} fwd {tf:.1f} us ({fl / tf / 1e6:.0f} TF)  bwd {tb:.1f} us ({2.5 * fl / tb / 1e6:.0f} TF)")
if os.environ.get("TORCH"):     # calibration only: torch's scaled_dot_product_attention on the same shape
    import torch.nn.functional as Fn
    q, k, v = (torch.randn(B, H, S, hd, device=dev).bfloat16().requires_grad_(True) for _ in range(3))
    g4 = torch.randn(B, H, S, hd, device=dev).bfloat16()
    def sf():
        return Fn.scaled_dot_product_attention(q, k, v)
    out = sf()
    def sb():
        out.backward(g4, retain_graph=True)
    try:
        t1, t2 = timeit(sf), timeit(sb)
        print(f"torch sdpa fwd {t1:.1f} us ({fl / t1 / 1e6:.0f} TF)  bwd {t2:.1f} us ({2.5 * fl / t2 / 1e6:.0f} TF)")
    except Exception as e:  # noqa: BLE001
        print("torch sdpa failed:", e)
