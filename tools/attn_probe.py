#!/usr/bin/env python
"""Launch the attention forward / backward kernels at cfg #2 a few times -- for PMC runs (tools/pmc.sh)."""
import sys
import torch
sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H
B, Hh, M, N, Lp = 8, 12, 4, 12, 196
S = M + N * Lp
qkv = torch.randn(B * S, 3 * Hh * 64, device="cuda").to(torch.bfloat16)
out, stats = H.attn_fwd(qkv, B, S, Hh, size=(M, N, Lp))
dout = torch.randn_like(out)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    H.attn_fwd(qkv, B, S, Hh, size=(M, N, Lp))
    H.attn_bwd(qkv, out, dout, stats, B, S, Hh, size=(M, N, Lp), q_scale=0.125)
torch.cuda.synchronize()
print("done")
