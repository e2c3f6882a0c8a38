import sys, os
sys.path.insert(0, os.getcwd())
import torch
import medvill_amd as mv
from medvill_amd import hip_ops as ops
dev="cuda"; B,A,dh=64,12,64
def bench(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
for (N,S,fam) in ((36,473,"full"),(36,473,"s2s"),(36,473,"noncross"),(100,665,"s2s")):
    L=N+S+3; H=A*dh
    n_ids=torch.randint((S+1)//2+1, S+2, (B,))
    mask=mv.data.build_mask(fam,N,S,n_ids,dev)
    qkv=(torch.randn(B*L,3*H,device=dev)).to(torch.bfloat16); dctx=torch.randn(B*L,H,device=dev).to(torch.bfloat16)
    bits=torch.zeros((B,L,(L+31)//32),dtype=torch.int32,device=dev); ti=torch.zeros((B,(L+63)//64,(L+63)//64),dtype=torch.uint8,device=dev)
    ops.mask_pack(mask,bits,ti)
    ctx=torch.empty(B*L,H,device=dev,dtype=torch.bfloat16); lse=torch.empty(B,A,L,device=dev); dqkv=torch.empty_like(qkv); delta=torch.empty(B,A,L,device=dev)
    fl=4.0*B*A*L*L*dh
    for p in (0.0,0.1):
        tf=bench(lambda: ops.attn_fwd(qkv,bits,ti,ctx,lse,B,L,A,dh,p_drop=p,drop_key=12345))
        tb=bench(lambda: ops.attn_bwd(qkv,ctx,dctx,lse,bits,ti,dqkv,delta,B,L,A,dh,p_drop=p,drop_key=12345))
        print(f"L={L} {fam:9s} p={p}: fwd {tf:7.1f} us ({fl/tf/1e6:6.0f} TF/s dense)  bwd {tb:7.1f} us ({2.5*fl/tb/1e6:6.0f} TF/s dense)", flush=True)
