#!/usr/bin/env python
"""Cycle stamps of one attention-forward workgroup at cfg #2 (xp_debug_set_attn_trace) + kernel time."""
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
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
L.lib().xp_debug_set_attn_trace(C.c_void_p(buf.data_ptr()))
H.attn_fwd(qkv, B, S, Hh, size=(M, N, Lp))
torch.cuda.synchronize()
L.lib().xp_debug_set_attn_trace(C.c_void_p(0))
t = buf.cpu().tolist()
print(f"attn fwd WG: loads issued +{t[1]-t[0]}, landed (barrier) +{t[2]-t[1]}, loop +{t[3]-t[2]} cycles; total {t[3]-t[0]}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    H.attn_fwd(qkv, B, S, Hh, size=(M, N, Lp))
e1.record(); torch.cuda.synchronize()
print(f"attn fwd: {e0.elapsed_time(e1)/20*1e3:.1f} us per call; {B*Hh*N} workgroups")
