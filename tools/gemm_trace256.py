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
    # barrier stamps of k-tile nk/2 (TRACE build of the NT kernel): per wave group, per phase [arrive b1, leave b1, arrive b2, leave b2]
    for grp, base in (("leading wave 0", 16), ("lagging wave 4", 32)):
        st = t[base:base + 8]
        if all(st):
            seg = [st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5], st[7] - st[6]]
            print(f"   {name} {grp}: wait@b1 {seg[0]} | MFMA seg(32) {seg[1]} | wait@b2 {seg[2]} | read seg(ph1) {seg[3]} | wait@b1 {seg[4]} | "
                  f"MFMA seg(ph1) {seg[5]} | wait@b2 {seg[6]} | sum {st[7] - st[0]}")
    # the same launch timed from outside (HIP events, 20 back-to-back launches): the difference to the traced workgroup's own
    # lifetime x rounds is the per-launch fixed cost (dispatch, cold start, kernel-boundary L2 write-back, tail)
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(20):
        H.gemm(A, W, M, N, K, out=out, epilogue=L.EPI_BIAS, bias=bias)
    en.record(); torch.cuda.synchronize()
    us = st.elapsed_time(en) / 20 * 1e3
    th = int(os.environ.get('XPRETRAIN_GEMM256_MT1', '0') or 0)
    th = {3: 224, 4: 256}.get(th, 224 if M == 18848 else 256)      # tile height the library picks (xp_gemm_tile_rows)
    tiles = ((M + th - 1) // th) * ((N + 255) // 256)
    rounds = -(-tiles // 256)
    life = t[3] - t[0]
    print(f"== {name}: nk={t[1]} loop={t[2]-t[0]} ticks epilogue={t[3]-t[2]} ticks | workgroup lifetime {life} shader cycles x {rounds} "
          f"round(s) ({tiles} tiles) in {us:.1f} us per launch (HIP events) -> >= {life * rounds / us / 1e3:.2f} GHz shader clock")
(n1, l1, _), (n2, l2, _) = res["fc1"], res["fc2"]
per = (l2 - l1) / (n2 - n1)
print(f"per k-tile {per:.1f} ticks; prologue+drain {l1 - n1 * per:.1f} ticks (s_memtime tick = one shader cycle: ticks / measured "
      "time is the shader clock the kernel actually ran at; tools/clock_probe.hip pins 1 tick = 1 shader cycle with an MFMA ruler)")
