#!/usr/bin/env python
"""Cycle stamps of one persistent attention-forward workgroup at cfg #2 (xp_debug_set_attn_trace): per problem, for wave 0 (two query
tiles) and wave 7 (one): arrival at the barrier, barrier + DMA wait, prefetch issue, compute + stores."""
import sys
import ctypes as C
import torch
sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H, _lib as L

B, Hh, M, N, Lp = 8, 12, 4, 12, 196
S = M + N * Lp
qkv = torch.randn(B * S, 3 * Hh * 64, device="cuda").to(torch.bfloat16)
for _ in range(3):
    H.attn_fwd(qkv, B, S, Hh, size=(M, N, Lp))
buf = torch.zeros(128, dtype=torch.int64, device="cuda")
L.lib().xp_debug_set_attn_trace(C.c_void_p(buf.data_ptr()))
H.attn_fwd(qkv, B, S, Hh, size=(M, N, Lp))
torch.cuda.synchronize()
L.lib().xp_debug_set_attn_trace(C.c_void_p(0))
t = buf.cpu().tolist()
for name, base in (("wave 0 (2 tiles)", 0), ("wave 7 (1 tile)", 64)):
    t0 = t[base]
    for it in range(8):
        a, b, c, d = t[base + it * 4: base + it * 4 + 4]
        if not d:
            break
        print(f"{name} problem {it}: at barrier +{a - t0:6d} | waited {b - a:5d} (DMA landed, all waves here) | prefetch issue {c - b:4d} | "
              f"compute + stores {d - c:5d} | problem total {d - a:5d}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    H.attn_fwd(qkv, B, S, Hh, size=(M, N, Lp))
e1.record(); torch.cuda.synchronize()
print(f"attn fwd (+ merge): {e0.elapsed_time(e1)/20*1e3:.1f} us per call; {B*Hh*N} problems")
