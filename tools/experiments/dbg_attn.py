import os, sys, torch
sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H
geom=(4,2,49); B,Hh=1,1
M,N,Lp=geom; S=M+N*Lp
torch.manual_seed(3)
qkv=(torch.randn(B*S,3*Hh*64,device="cuda")*0.7).to(torch.bfloat16)
out,stats=H.attn_fwd(qkv,B,S,Hh,size=geom)
dout=torch.randn_like(out)
got=H.attn_bwd(qkv,out,dout,stats,B,S,Hh,size=geom,q_scale=0.125).float()
os.environ["XPRETRAIN_DEBUG"]="attn_bwd_split"
want=H.attn_bwd(qkv,out,dout,stats,B,S,Hh,size=geom,q_scale=0.125).float()
for j,nm in enumerate("qkv"):
    a=got.view(S,3,64)[:,j]; b=want.view(S,3,64)[:,j]
    print(nm,"got absmax",a.abs().max().item(),"want absmax",b.abs().max().item())
    err=(a-b).abs().amax(1)
    print(" rows err:", [f"{e:.2g}" for e in err[:12].tolist()], "...", [f"{e:.2g}" for e in err[50:58].tolist()])
    print(" row5 got", a[5,:6].tolist()); print(" row5 want", b[5,:6].tolist())
# hypotheses
q,k,v=[t.float() for t in qkv.view(S,3,64).unbind(1)]
st=stats.view(S,2)
def model(hyp, n):
    rows=list(range(M))+list(range(M+n*Lp, M+(n+1)*Lp))
    Q,K,V,dO,O=q[rows],k[rows],v[rows],dout.float()[rows],out.float()[rows]
    m,lg=st[rows,0],st[rows,1]
    Sx=Q@K.T
    if hyp=="S0": Sx=Sx*0
    c=-(m+lg)
    if hyp=="c2": c=2*c
    P=torch.exp(Sx+c[:,None])
    if n!=0: P[:M,:M]=0
    dP=dO@V.T; dl=(dO*O).sum(1)
    dS=P*(dP-dl[:,None])
    return rows,(dS@K)*0.125
for hyp in ("ok","S0","c2"):
    rows,dq=model(hyp,0)
    a=got.view(S,3,64)[rows,0]
    print(hyp,"dq frame0 rows4.. maxabs diff",(a[4:]-dq[4:]).abs().max().item(),"scale",dq[4:].abs().max().item())
