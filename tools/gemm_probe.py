#!/usr/bin/env python
"""Launch each forward/backward GEMM shape of cfg #2 a few times (for rocprofv3 --pmc runs)."""
import sys
import torch
sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H, _lib as L

M = 8 * 2356
bf = torch.bfloat16
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for name, N, K in [("qkv", 2304, 768), ("out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
    A = torch.randn(M, K, device="cuda").to(bf)
    W = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
    dY = torch.randn(M, N, device="cuda").to(bf)
    bias = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, dtype=bf, device="cuda")
    dX = torch.empty(M, K, dtype=bf, device="cuda")
    slabs = torch.empty(4, N, K, device="cuda")
    for _ in range(iters):
        H.gemm(A, W, M, N, K, out=out, epilogue=L.EPI_BIAS, bias=bias)
        H.gemm(dY, W, M, K, N, b_kstrided=True, out=dX)
        H.gemm(dY, A, N, K, M, a_kstrided=True, b_kstrided=True, lda=N, ldb=K, split_k=4, out=slabs)
torch.cuda.synchronize()
print("done")
