#!/usr/bin/env python
"""Cycle totals of one workgroup of the 256x256 GEMM family (xp_debug_set_gemm_trace): main loop and epilogue."""
import os
import sys
import ctypes as C
os.environ["XPRETRAIN_GEMM256"] = "2"
import torch
sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H, _lib as L

M = 8 * 2356
bf = torch.bfloat16
res = {}
for name, N, K in [("out", 768, 768), ("fc1", 3072, 768), ("k1536", 768, 1536), ("fc2", 768, 3072)]:
    A = torch.randn(M, K, device="cuda").to(bf)
    W = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
    bias = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, dtype=bf, device="cuda")
    for _ in range(3):
        H.gemm(A, W, M, N, K, out=out, epilogue=L.EPI_BIAS, bias=bias)
    buf = torch.zeros(256, dtype=torch.int64, device="cuda")
    L.lib().xp_debug_set_gemm_trace(C.c_void_p(buf.data_ptr()))
    H.gemm(A, W, M, N, K, out=out, epilogue=L.EPI_BIAS, bias=bias)
    torch.cuda.synchronize()
    L.lib().xp_debug_set_gemm_trace(C.c_void_p(0))
    t = buf.cpu().tolist()
    res[name] = (t[1], t[2] - t[0], t[3] - t[2])
    print(f"== {name}: nk={t[1]} loop={t[2]-t[0]} ticks epilogue={t[3]-t[2]} ticks")
(n1, l1, _), (n2, l2, _) = res["fc1"], res["fc2"]
per = (l2 - l1) / (n2 - n1)
print(f"per k-tile {per:.1f} ticks; prologue+drain {l1 - n1 * per:.1f} ticks (s_memtime: 100 MHz constant clock -> x24 for 2.4 GHz cycles)")
