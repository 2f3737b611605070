"""Build the helper first: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o scripts/micro/libcu_hog.so scripts/micro/cu_hog.hip

How much a compute kernel slows down while another stream holds a few CUs (what an RCCL
all-reduce overlapped with backward does): times the training step's GEMM shapes and the attention
kernels alone and with `HOG` workgroups spinning on a side stream."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from src import _native as N
dev = torch.device("cuda:0")
hog = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libcu_hog.so"))
hog.cu_hog.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
T, d = 65536, 768
lib = N.lib()
a = torch.randn(T, d, device=dev).bfloat16()
w = torch.randn(2304, d, device=dev).bfloat16()
c = torch.empty(T, 2304, device=dev, dtype=torch.bfloat16)
g = torch.randn(T, 2304, device=dev).bfloat16()
dw = torch.zeros(2304, d, device=dev)
def fwd():
    N.check(lib.ldm_gemm(1, 1, 0, 1, T, 2304, d, 1.0, a.data_ptr(), d, w.data_ptr(), d, c.data_ptr(), 2304, None, None, 0, N.stream()), "g")
def dwf():
    N.check(lib.ldm_gemm_acc(1, 1, 0, 2304, d, T, 1.0, g.data_ptr(), 2304, a.data_ptr(), d, dw.data_ptr(), d, N.stream()), "dw")
def timed(fn, nhog, reps=10):
    fn(); torch.cuda.synchronize()
    if nhog:
        with torch.cuda.stream(side):
            assert hog.cu_hog(nhog, 20000.0, sink.data_ptr(), side.cuda_stream) == 0     # 20 ms
        torch.cuda._sleep(200000)   # let the hog land first
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for name, fn in (("qkv fwd gemm256", fwd), ("qkv dW gemm256 split-K", dwf)):
    base = timed(fn, 0)
    line = f"{name:26s} alone {base:7.1f} us"
    for nh in (8, 16, 32, 64):
        t = timed(fn, nh)
        line += f" | {nh} hogs {t:7.1f} ({t / base:4.2f}x)"
    print(line, flush=True)
