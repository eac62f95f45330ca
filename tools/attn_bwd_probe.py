#!/usr/bin/env python
"""The attention backward of one video-tower layer at BASELINE cfg #2 (B=8, H=12, (M,N,L)=(4,12,196)), N launches of the fused kernel
and of the dQ / dKV pair (XPRETRAIN_DEBUG=attn_bwd_split) -- the workload of the rocprofv3 kernel-trace / PMC passes of round 6
(tools/profile.sh, tools/pmc.sh):  python tools/attn_bwd_probe.py [iters] [fused|split|both] [colsum]"""
import os
import sys

import torch

sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
which = sys.argv[2] if len(sys.argv) > 2 else "both"
colsum = len(sys.argv) > 3
B, Hh, M, N, Lp = 8, 12, 4, 12, 196
S = M + N * Lp
torch.manual_seed(0)
qkv = torch.randn(B * S, 3 * Hh * 64, device="cuda").to(torch.bfloat16)
out, stats = H.attn_fwd(qkv, B, S, Hh, size=(M, N, Lp))
dout = torch.randn_like(out)


def run(tag):
    d = H.DeferredReduce(qkv.device) if colsum else None
    def once():
        r = H.attn_bwd(qkv, out, dout, stats, B, S, Hh, size=(M, N, Lp), q_scale=0.125, colsum_defer=d)
        if d is not None:
            d.segs.clear(); d._keep.clear(); d._names.clear()
        return r
    for _ in range(300):
        once()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        once()
    en.record()
    torch.cuda.synchronize()
    print(f"attn bwd {tag:5s} colsum={int(colsum)}: {st.elapsed_time(en) / iters * 1e3:7.1f} us per call (all launches of the call)")


if which in ("fused", "both"):
    os.environ.pop("XPRETRAIN_DEBUG", None)
    run("fused")
if which in ("split", "both"):
    os.environ["XPRETRAIN_DEBUG"] = "attn_bwd_split"
    run("split")
