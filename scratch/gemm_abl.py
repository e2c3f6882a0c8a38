import sys, os
sys.path.insert(0, os.getcwd())
import torch
from medvill_amd import hip_ops as ops
from medvill_amd._lib import *
dev="cuda"; M=32768; H=768; I=3072
x=(torch.randn(M,H,device=dev)*0.5).to(torch.bfloat16); W1=(torch.randn(I,H,device=dev)*0.5).to(torch.bfloat16)
xi=(torch.randn(M,I,device=dev)*0.5).to(torch.bfloat16); W2=(torch.randn(H,I,device=dev)*0.5).to(torch.bfloat16)
b1=torch.randn(I,device=dev); bh=torch.randn(H,device=dev)
oI=torch.empty(M,I,device=dev,dtype=torch.bfloat16); oH=torch.empty(M,H,device=dev,dtype=torch.bfloat16)
def bench(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
for nj in (14,):
  for dbg,name in ((0,"full"),(128+(1<<8),"desync 3.4us"),(128+(2<<8),"desync 6.8us"),(128+(3<<8),"desync 10us"),(128+(4<<8),"desync 13.6us"),(0,"full again")):
    ops.set_gemm_variant(2 | (dbg<<8), nj)
    t1=bench(lambda: ops.gemm(x,W1,oI,M=M,N=I,K=H,bias=b1,epi=EPI_BIAS))
    t2=bench(lambda: ops.gemm(xi,W2,oH,M=M,N=H,K=I,bias=bh,epi=EPI_BIAS))
    t3=bench(lambda: ops.gemm(x,W2,oI,tb=True,M=M,N=I,K=H))
    print(f"nj={nj} {name:22s} ffn1(K=768) {t1:7.1f} us   ffn2(K=3072) {t2:7.1f} us   NN dz(K=768) {t3:7.1f} us", flush=True)
ops.set_gemm_variant(0,0)
