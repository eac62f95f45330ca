#!/usr/bin/env python
"""Cycle stamps of one persistent workgroup of the fused attention backward at cfg #2 (xp_debug_set_attn_trace), waves 0 and 7:
per problem -- barrier wait (X landed), setup, phase A steps, phase A epilogue, barrier wait (Y landed), phase B steps, epilogue."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H, _lib as L  # noqa: E402

B, Hh, M, N, Lp = 8, 12, 4, 12, 196
S = M + N * Lp
colsum = len(sys.argv) > 1
qkv = torch.randn(B * S, 3 * Hh * 64, device="cuda").to(torch.bfloat16)
out, stats = H.attn_fwd(qkv, B, S, Hh, size=(M, N, Lp))
dout = torch.randn_like(out)


def once():
    d = H.DeferredReduce(qkv.device) if colsum else None
    return H.attn_bwd(qkv, out, dout, stats, B, S, Hh, size=(M, N, Lp), q_scale=0.125, colsum_defer=d)


for _ in range(300):
    once()
buf = torch.zeros(128, dtype=torch.int64, device="cuda")
L.lib().xp_debug_set_attn_trace(C.c_void_p(buf.data_ptr()))
once()
torch.cuda.synchronize()
L.lib().xp_debug_set_attn_trace(C.c_void_p(0))
t = buf.cpu().tolist()
names = ["wait X", "setup", "A steps", "A epilogue", "wait Y", "B setup+steps", "B epilogue"]
for name, base in (("wave 0", 0), ("wave 7", 64)):
    for it in range(8):
        r = t[base + it * 8: base + it * 8 + 8]
        if not r[7]:
            break
        print(f"{name} problem {it}: " + " | ".join(f"{n} {b - a:6d}" for n, a, b in zip(names, r[:-1], r[1:])) + f" | total {r[7] - r[0]:6d}")
