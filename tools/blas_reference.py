import torch, time
M=18848
def t(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(it): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/it*1e3
for N,K in [(2304,768),(768,768),(3072,768),(768,3072)]:
    A=torch.randn(M,K,device='cuda',dtype=torch.bfloat16); W=torch.randn(N,K,device='cuda',dtype=torch.bfloat16)*0.02
    b=torch.zeros(N,device='cuda',dtype=torch.bfloat16)
    us=t(lambda: torch.nn.functional.linear(A,W,b)); print(f"hipBLASLt linear fwd N={N} K={K}: {us:.1f} us {2*M*N*K/us/1e6:.0f} TF")
    dY=torch.randn(M,N,device='cuda',dtype=torch.bfloat16)
    us=t(lambda: dY@W); print(f"   dX: {us:.1f} us {2*M*N*K/us/1e6:.0f} TF")
    us=t(lambda: dY.t()@A); print(f"   dW: {us:.1f} us {2*M*N*K/us/1e6:.0f} TF")
