#!/usr/bin/env python
"""Launch the three operand variants of the 256x256 GEMM family at BASELINE cfg #2 shapes a few times -- for the PMC passes
of tools/pmc_gemm256.sh -- next to a calibration kernel of exactly known traffic (xp_cast fp32 -> bf16 of 64 Mi elements:
256 MiB read, 128 MiB written, 16 B / lane streaming):

  NT  fc1 forward   [18848 x 768] x [3072 x 768]^T  +bias +quick_gelu, two bf16 outputs   (gemm256_kernel<false,false>)
  NS  dpre = dx3.W2 [18848 x 768] x [768 x 3072]    *quick_gelu'(pre), fused column sums  (gemm256_kernel<false,true>)
  SS  dW1 = dpre^T.h2, split-K 3 into fp32 slabs                                         (gemm256_kernel<true,true>)
"""
import sys

import torch

sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H, _lib as L  # noqa: E402
from xpretrain_amd.functional import _wgrad  # noqa: E402

M, D, Dff = 8 * 2356, 768, 3072
bf = torch.bfloat16
dev = "cuda"
A = torch.randn(M, D, device=dev).to(bf)
W1 = (torch.randn(Dff, D, device=dev) * 0.02).to(bf)
W2 = (torch.randn(D, Dff, device=dev) * 0.02).to(bf)
bias = torch.zeros(Dff, device=dev)
out = torch.empty(M, Dff, dtype=bf, device=dev)
aux = torch.empty_like(out)
dx3 = (torch.randn(M, D, device=dev) * 1e-3).to(bf)
dpre = torch.empty(M, Dff, dtype=bf, device=dev)
src = torch.randn(64 * 1024 * 1024, device=dev)
dst = torch.empty(64 * 1024 * 1024, dtype=bf, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    H.cast(src, bf, out=dst)
    H.gemm(A, W1, M, Dff, D, out=out, epilogue=L.EPI_BIAS_GELU, bias=bias, aux=aux)
    d = H.DeferredReduce(torch.device(dev))
    H.gemm(dx3, W2, M, Dff, D, b_kstrided=True, epilogue=L.EPI_GELU_BWD, resid=aux, out=dpre, colsum_defer=d)
    d.flush()
    _wgrad(dpre, A, M, Dff, D, slack=True)        # (as the layer backward issues it: xp_gemm_auto_split_slack)
torch.cuda.synchronize()
print("done")
