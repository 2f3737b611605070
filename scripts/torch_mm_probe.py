"""Calibration only: which vendor-library kernels torch.matmul picks for square and step-shaped bf16 problems
(run under rocprofv3 --kernel-trace; the kernel names carry the library's tile configuration)."""
import torch
dev = torch.device("cuda:0")
torch.manual_seed(0)
for M, N, K in [(4096, 4096, 4096), (8192, 8192, 8192), (262144, 2304, 768), (262144, 768, 768), (262144, 2048, 768), (262144, 768, 1024)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    for _ in range(3):
        torch.matmul(a, w.t())          # NT: both operands K-contiguous
    torch.cuda.synchronize()
    if M > 8192:
        dy = torch.randn(M, N, device=dev).bfloat16()
        for _ in range(3):
            torch.matmul(dy, w)         # NN: input gradient
            torch.matmul(dy.t(), a)     # TN: weight gradient
        torch.cuda.synchronize()
