#!/usr/bin/env python
"""Is the attention forward bound by the scatter of its 128-byte K/V rows (one head of a token = 128 B at a 4608-B pitch in
qkv[B*S, 3*H*64])?  Same problem count and arithmetic with the heads folded into the batch (B*H samples of ONE head: row pitch 384 B)."""
import sys
import torch
sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


M, N, Lp = 4, 12, 196
S = M + N * Lp
for B, Hh in ((8, 12), (96, 1), (24, 4)):
    qkv = torch.randn(B * S, 3 * Hh * 64, device="cuda").to(torch.bfloat16)
    us = timeit(lambda: H.attn_fwd(qkv, B, S, Hh, size=(M, N, Lp)))
    out, stats = H.attn_fwd(qkv, B, S, Hh, size=(M, N, Lp))
    dout = torch.randn_like(out)
    usb = timeit(lambda: H.attn_bwd(qkv, out, dout, stats, B, S, Hh, size=(M, N, Lp), q_scale=0.125))
    print(f"B={B:3d} H={Hh:2d} (row pitch {3 * Hh * 128:5d} B): fwd {us:6.1f} us  bwd {usb:6.1f} us   [{B * Hh * N} problems, {4 * B * S * Hh * 128 / 1e6:.1f} MB q/k/v/o]")
