import os, sys, torch
sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H
geom=(4,2,49); B,Hh=1,1
M,N,Lp=geom; S=M+N*Lp
torch.manual_seed(3)
qkv=(torch.randn(B*S,3*Hh*64,device="cuda")*0.7).to(torch.bfloat16)
out,stats=H.attn_fwd(qkv,B,S,Hh,size=geom)
dout=torch.randn_like(out)
os.environ["XPRETRAIN_DEBUG"]="attn_bwd_split"
want=H.attn_bwd(qkv,out,dout,stats,B,S,Hh,size=geom,q_scale=0.125).float()
for flags in ("", "b5_own_ptr", "b5_dma_old", "b5_own_ptr,b5_dma_old"):
    os.environ["XPRETRAIN_DEBUG"]=flags
    got=H.attn_bwd(qkv,out,dout,stats,B,S,Hh,size=geom,q_scale=0.125).float()
    print(repr(flags), "max err", (got-want).abs().max().item(), "want max", want.abs().max().item())
