#!/usr/bin/env python
"""Launch the dominant kernel (fc1 forward GEMM, bias+quick_gelu epilogue, cfg #2 shape) a few times -- for PMC runs --
next to a calibration kernel with exactly known traffic (xp_cast fp32 -> bf16 of 64 Mi elements: 256 MiB read,
128 MiB written, streaming 16 B / lane)."""
import sys
import torch
sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H, _lib as L
M, D, Dff = 8 * 2356, 768, 3072
bf = torch.bfloat16
A = torch.randn(M, D, device="cuda").to(bf); W = (torch.randn(Dff, D, device="cuda") * 0.02).to(bf)
bias = torch.zeros(Dff, device="cuda"); out = torch.empty(M, Dff, dtype=bf, device="cuda"); aux = torch.empty_like(out)
src = torch.randn(64 * 1024 * 1024, device="cuda"); dst = torch.empty(64 * 1024 * 1024, dtype=bf, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    H.cast(src, bf, out=dst)
    H.gemm(A, W, M, Dff, D, out=out, epilogue=L.EPI_BIAS_GELU, bias=bias, aux=aux)
torch.cuda.synchronize()
print("done")
