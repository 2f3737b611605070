"""GEMM micro-benchmark over the training step's shapes (bf16 MFMA kernel through the C ABI)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from src import _native as N

dev = torch.device("cuda:0")
T = int(os.environ.get("T", 32768))
reps = int(os.environ.get("REPS", 20))
only = os.environ.get("ONLY", "")
d, F, V = 768, 1024, 30528
shapes = []
for n, k in [(3 * d, d), (d, d), (2 * F, d), (d, F), (V, d)]:
    shapes.append(("fwd", 0, 1, T, n, k))
    shapes.append(("dx", 0, 0, T, k, n))
    shapes.append(("dw", 1, 0, n, k, T))
if os.environ.get("CUBE"):      # calibration: square problems (A [M,K], B [N,K], both K-contiguous), e.g. CUBE=4096,8192
    shapes = [("fwd", 0, 1, int(n), int(n), int(n)) for n in os.environ["CUBE"].split(",")]
    shapes += [("dx", 0, 0, int(n), int(n), int(n)) for n in os.environ["CUBE"].split(",")]
    shapes += [("dw", 1, 0, int(n), int(n), int(n)) for n in os.environ["CUBE"].split(",")]
torch.manual_seed(0)
# clocks and power state settle over the first tens of milliseconds of load: warm up before the first shape
_w = torch.randn(8192, 8192, device=dev).bfloat16()
for _ in range(30):
    torch.matmul(_w, _w)
torch.cuda.synchronize()
del _w
for name, ta, tb, M, Nn, K in shapes:
    if only and name not in only.split(","):
        continue
    a = torch.randn((K, M) if ta else (M, K), device=dev).bfloat16()
    b = torch.randn((Nn, K) if tb else (K, Nn), device=dev).bfloat16()
    acc = name == "dw"
    use_res = bool(os.environ.get("RES")) and name == "fwd"
    bias = torch.randn(Nn, device=dev) if use_res else None
    res = torch.randn(M, Nn, device=dev).bfloat16() if use_res else None
    c = torch.zeros(M, Nn, device=dev, dtype=torch.float32 if acc else torch.bfloat16)
    rs = torch.zeros(M, device=dev, dtype=torch.float32)
    def run():
        if acc and os.environ.get("ROWSUM"):
            N.check(N.lib().ldm_gemm_acc_rowsum(ta, tb, M, Nn, K, 1.0, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), Nn, rs.data_ptr(), N.stream()), "acc_rowsum")
        elif acc:
            N.check(N.lib().ldm_gemm_acc(1, ta, tb, M, Nn, K, 1.0, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), Nn, N.stream()), "acc")
        else:
            N.check(N.lib().ldm_gemm(1, 1, ta, tb, M, Nn, K, 1.0, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), Nn, N.ptr(bias), N.ptr(res), Nn if use_res else 0, N.stream()), "gemm")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    line = f"{name:4s} M={M:6d} N={Nn:6d} K={K:6d}  {ms*1e3:9.1f} us  {2.0*M*Nn*K/ms/1e9:8.1f} TF"
    if os.environ.get("TORCH"):     # calibration only: the vendor library GEMM of the same shape (bf16 output)
        at = a.t() if ta else a
        bt = b.t() if tb else b
        for _ in range(3):
            torch.matmul(at, bt)
        e0.record()
        for _ in range(reps):
            torch.matmul(at, bt)
        e1.record()
        torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / reps
        line += f"   | torch.matmul {ms2*1e3:9.1f} us  {2.0*M*Nn*K/ms2/1e9:8.1f} TF"
    print(line, flush=True)
