import sys, os
sys.path.insert(0, os.getcwd())
import torch
from medvill_amd import hip_ops as ops
from medvill_amd._lib import *
dev="cuda"; M=32768; H=768; I=3072
def bench1(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
def t(x): return (torch.randn(*x,device=dev)*0.5).to(torch.bfloat16)
b3=torch.randn(3*H,device=dev); bI=torch.randn(I,device=dev); bh=torch.randn(H,device=dev)
ws=torch.empty(16*I*H,device=dev)
for pa in (64,0,64,0):
  for pc in (pa,):
    x=t((M,H+pa)); Wq=t((3*H,H)); W1=t((I,H)); W2=t((H,I)); xi=t((M,I+pa))
    o3=torch.empty(M,3*H+pc,device=dev,dtype=torch.bfloat16); oI=torch.empty(M,I+pc,device=dev,dtype=torch.bfloat16); oI2=torch.empty_like(oI)
    oH=torch.empty(M,H+pc,device=dev,dtype=torch.bfloat16); gW=torch.empty(I,H,device=dev)
    r={}
    r["qkv"]=bench1(lambda: ops.gemm(x,Wq,o3,M=M,N=3*H,K=H,lda=H+pa,ldc=3*H+pc,bias=b3,epi=EPI_BIAS))
    r["ffn1+gelu"]=bench1(lambda: ops.gemm(x,W1,oI,M=M,N=I,K=H,lda=H+pa,ldc=I+pc,ldc2=I+pc,bias=bI,epi=EPI_BIAS_GELU,c2=oI2))
    r["ffn2"]=bench1(lambda: ops.gemm(xi,W2,oH,M=M,N=H,K=I,lda=I+pa,ldc=H+pc,bias=bh,epi=EPI_BIAS))
    r["da(NN)"]=bench1(lambda: ops.gemm(xi,W1,oH,tb=True,M=M,N=H,K=I,lda=I+pa,ldc=H+pc))
    r["dz(NN)"]=bench1(lambda: ops.gemm(x,W2,oI,tb=True,M=M,N=I,K=H,lda=H+pa,ldc=I+pc,epi=EPI_DGELU,r=oI2,ldr=I+pc))
    r["dW1(TN)"]=bench1(lambda: ops.gemm(xi,x,gW,ta=True,tb=True,M=I,N=H,K=M,lda=I+pa,ldb=H+pa,splitk=0,ws=ws))
    print(f"pad A={pa} C={pc}: "+"  ".join(f"{k} {v:.0f}" for k,v in r.items()), flush=True)
