#!/usr/bin/env python
"""Cycle-stamp trace of one workgroup's main loop in the 128x128 GEMM (xp_debug_set_gemm_trace)."""
import sys
import ctypes as C
import torch
sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H, _lib as L

M = 8 * 2356
bf = torch.bfloat16
for name, N, K in [("fc1", 3072, 768), ("fc2", 768, 3072)]:
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
    nk = t[1]
    print(f"== {name}: nk={nk} loop={t[2]-t[0]} cyc epilogue={t[3]-t[2]} cyc  (s_memtime ticks = shader cycles, profiles/r03a_clock_probe.txt)")
    import os
    g256 = os.environ.get("XPRETRAIN_GEMM256", "0") != "0"
    print(" kt  vmwait  barrier   issue  compute   total" if g256 else " kt   issue  compute  vmwait  barrier   total")
    prev = None
    for kt in range(min(nk, 24)):
        s = t[8 + kt * 5: 8 + kt * 5 + 5]
        print(f"{kt:3d} {s[1]-s[0]:7d} {s[2]-s[1]:8d} {s[3]-s[2]:7d} {s[4]-s[3]:8d} {s[4]-s[0]:7d}")
