"""Times ldm_gemm_swiglu_bwd against the two-call form (ldm_gemm + ldm_swiglu_bwd) at the step's shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from src import _native as N
dev = torch.device("cuda:0")
T, F, K = int(os.environ.get("T", 262144)), 1024, 768
lib = N.lib()
dy = torch.randn(T, K, device=dev).bfloat16()
w = torch.randn(K, F, device=dev).bfloat16()
ag = torch.randn(T, 2 * F, device=dev).bfloat16()
dag = torch.empty_like(ag)
dh = torch.empty(T, F, device=dev, dtype=torch.bfloat16)
def fused():
    N.check(lib.ldm_gemm_swiglu_bwd(T, F, K, dy.data_ptr(), K, w.data_ptr(), F, ag.data_ptr(), 2 * F, dag.data_ptr(), 2 * F, N.stream()), "f")
def two():
    N.check(lib.ldm_gemm(1, 1, 0, 0, T, F, K, 1.0, dy.data_ptr(), K, w.data_ptr(), F, dh.data_ptr(), F, None, None, 0, N.stream()), "g")
    N.check(lib.ldm_swiglu_bwd(ag.data_ptr(), dh.data_ptr(), T, F, 1, dag.data_ptr(), N.stream()), "s")
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
print(f"T={T}: fused {timeit(fused):.1f} us   gemm + swiglu_bwd {timeit(two):.1f} us")
