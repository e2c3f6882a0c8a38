import sys, os
sys.path.insert(0, os.getcwd())
import torch
from medvill_amd import hip_ops as ops
from medvill_amd._lib import *
dev="cuda"; H=768; I=3072; M=25483
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
bf=torch.bfloat16
dy=(torch.randn(M,H,device=dev)*0.5).to(bf); W2t=(torch.randn(I,H,device=dev)*0.5).to(bf)
z=torch.randn(M,I,device=dev).to(bf); dz=torch.empty(M,I,device=dev,dtype=bf)
xi=(torch.randn(M,I,device=dev)*0.5).to(bf); W2=(torch.randn(H,I,device=dev)*0.5).to(bf); bh=torch.randn(H,device=dev); r=torch.randn(M,H,device=dev).to(bf); oh=torch.empty(M,H,device=dev)
for rep in range(3):
    res=[]
    for dbg in (0,16):
        ops.set_gemm_variant(0 | (dbg<<8), 0)
        res.append(bench(lambda: ops.gemm(dy,W2t,dz,M=M,N=I,K=H,epi=EPI_MUL,r=z)))
        res.append(bench(lambda: ops.gemm(xi,W2,oh,M=M,N=H,K=I,bias=bh,epi=EPI_BIAS_RES,r=r)))
    print(f"nt loads: dz {res[0]:6.1f} ffn2 {res[1]:6.1f} | plain loads: dz {res[2]:6.1f} ffn2 {res[3]:6.1f}", flush=True)
ops.set_gemm_variant(0,0)
