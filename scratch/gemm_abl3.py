import sys, os
sys.path.insert(0, os.getcwd())
import torch
from medvill_amd import hip_ops as ops
from medvill_amd._lib import *
dev="cuda"; H=768; I=3072; M=32768
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
W1=(torch.randn(I,H,device=dev)*0.5).to(torch.bfloat16); b1=torch.randn(I,device=dev)
W3=(torch.randn(3*H,H,device=dev)*0.5).to(torch.bfloat16); b3=torch.randn(3*H,device=dev)
W2=(torch.randn(H,I,device=dev)*0.5).to(torch.bfloat16); bh=torch.randn(H,device=dev)
x=(torch.randn(M,H,device=dev)*0.5).to(torch.bfloat16); o=torch.empty(M,I,device=dev,dtype=torch.bfloat16); o2=torch.empty_like(o)
xi=(torch.randn(M,I,device=dev)*0.5).to(torch.bfloat16); oh=torch.empty(M,H,device=dev); r=torch.randn(M,H,device=dev)
o3=torch.empty(M,3*H,device=dev,dtype=torch.bfloat16)
for rep in range(2):
  for nj in (14,24,1):
    res=[]
    for dbg in (0,16):
        ops.set_gemm_variant((2 if nj!=1 else 1) | (dbg<<8), nj if nj!=1 else 0)
        res.append(bench(lambda: ops.gemm(x,W1,o,M=M,N=I,K=H,bias=b1,epi=EPI_BIAS)))
        res.append(bench(lambda: ops.gemm(x,W1,o,M=M,N=I,K=H,bias=b1,epi=EPI_BIAS_GELU,c2=o2)))
        res.append(bench(lambda: ops.gemm(x,W3,o3,M=M,N=3*H,K=H,bias=b3,epi=EPI_BIAS)))
        res.append(bench(lambda: ops.gemm(xi,W2,oh,M=M,N=H,K=I,bias=bh,epi=EPI_BIAS_RES,r=r)))
    print(f"variant {nj:2d}: nt stores: ffn1 {res[0]:6.1f} gelu {res[1]:6.1f} qkv {res[2]:6.1f} ffn2 {res[3]:6.1f} | plain stores: ffn1 {res[4]:6.1f} gelu {res[5]:6.1f} qkv {res[6]:6.1f} ffn2 {res[7]:6.1f}", flush=True)
ops.set_gemm_variant(0,0)
