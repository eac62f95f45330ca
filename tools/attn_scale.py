import sys, torch
sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H
Hh, M, N, Lp = 12, 4, 12, 196
S = M + N * Lp
for B in (1, 2, 3, 4, 6, 8, 16):
    qkv = torch.randn(B * S, 3 * Hh * 64, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        H.attn_fwd(qkv, B, S, Hh, size=(M, N, Lp))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        H.attn_fwd(qkv, B, S, Hh, size=(M, N, Lp))
    e1.record(); torch.cuda.synchronize()
    print(f"B={B:2d}: {B*Hh*N*2:5d} WGs ({B*Hh*N*2/256:.2f}/CU)  {e0.elapsed_time(e1)/20*1e3:7.1f} us per call (incl. merge kernel)")
