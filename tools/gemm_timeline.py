#!/usr/bin/env python
"""Per-tile timeline of one launch of the 8-wave 256x256 forward kernel: every workgroup stamps (100 MHz constant-rate clock) its
start, the end of its main loop and the end of its epilogue (after its stores are acknowledged) + its hardware id.  Shows whether
the epilogues of a round coincide chip-wide (one HBM write burst) or are spread over the other tiles' main loops.

    python tools/gemm_timeline.py [fc1|out|qkv|fc2 ...]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H, _lib as L  # noqa: E402

M = 8 * 2356
bf = torch.bfloat16
SH = {"qkv": (2304, 768, dict(epilogue=L.EPI_BIAS_QSCALE, scale=0.125, scale_cols=768)), "out": (768, 768, dict(epilogue=L.EPI_BIAS_RESID)),
      "fc1": (3072, 768, dict(epilogue=L.EPI_BIAS_GELU)), "fc2": (768, 3072, dict(epilogue=L.EPI_BIAS_RESID))}
for name in (sys.argv[1:] or ["fc1", "out"]):
    N, K, kw = SH[name]
    kw = dict(kw)
    A = torch.randn(M, K, device="cuda").to(bf)
    W = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
    bias = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, dtype=bf, device="cuda")
    if kw["epilogue"] == L.EPI_BIAS_RESID:
        kw["resid"] = torch.randn(M, N, device="cuda").to(bf)
    if kw["epilogue"] == L.EPI_BIAS_GELU:
        kw["aux"] = torch.empty(M, N, dtype=bf, device="cuda")
    for _ in range(300):                                  # sustained clocks (tools/power_probe.py)
        H.gemm(A, W, M, N, K, out=out, bias=bias, **kw)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    buf = torch.zeros(64 + 4 * tiles, dtype=torch.int64, device="cuda")
    buf[4] = 0x7ace
    L.lib().xp_debug_set_gemm_trace(C.c_void_p(buf.data_ptr()))
    H.gemm(A, W, M, N, K, out=out, bias=bias, **kw)
    torch.cuda.synchronize()
    L.lib().xp_debug_set_gemm_trace(C.c_void_p(0))
    t = buf.cpu()[64:].view(tiles, 4).tolist()
    t0 = min(r[0] for r in t)
    rows = sorted(((r[0] - t0) / 100.0, (r[1] - t0) / 100.0, (r[2] - t0) / 100.0, r[3]) for r in t)   # us
    end = max(r[2] for r in rows)
    print(f"== {name}: {tiles} tiles, launch span {end:.1f} us (first start -> last epilogue end)")
    loop = sorted(r[1] - r[0] for r in rows); epi = sorted(r[2] - r[1] for r in rows)
    q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]
    print(f"   main loop us: min {loop[0]:.1f} p10 {q(loop,.1):.1f} median {q(loop,.5):.1f} p90 {q(loop,.9):.1f} max {loop[-1]:.1f}")
    print(f"   epilogue  us: min {epi[0]:.1f} p10 {q(epi,.1):.1f} median {q(epi,.5):.1f} p90 {q(epi,.9):.1f} max {epi[-1]:.1f}")
    # occupancy histogram: how many tiles are in their epilogue / main loop per 2-us bin
    nb = int(end / 2) + 1
    in_epi = [0] * nb; in_loop = [0] * nb
    for s, m, e, _ in rows:
        for b in range(nb):
            c = b * 2 + 1.0
            if s <= c < m: in_loop[b] += 1
            elif m <= c < e: in_epi[b] += 1
    print("   t(us) loop/epi: " + " ".join(f"{b*2}:{in_loop[b]}/{in_epi[b]}" for b in range(nb)))
    starts = [r[0] for r in rows]
    print("   tile starts (us), every 16th: " + " ".join(f"{starts[i]:.1f}" for i in range(0, len(starts), max(1, len(starts) // 56))))
