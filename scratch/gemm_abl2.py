import sys, os
sys.path.insert(0, os.getcwd())
import torch
from medvill_amd import hip_ops as ops
from medvill_amd._lib import *
dev="cuda"; H=768; I=3072
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
W1=(torch.randn(I,H,device=dev)*0.5).to(torch.bfloat16); b1=torch.randn(I,device=dev)
for rounds in (1,2,3,6,12):
    M = 256*rounds*256//12
    M = (M+255)//256*256
    x=(torch.randn(M,H,device=dev)*0.5).to(torch.bfloat16); o=torch.empty(M,I,device=dev,dtype=torch.bfloat16)
    units = (M//256)*12
    res=[]
    for nj in (14,24):
        for dbg in (0,1):
            ops.set_gemm_variant(2 | (dbg<<8), nj)
            res.append(bench(lambda: ops.gemm(x,W1,o,M=M,N=I,K=H,bias=b1,epi=EPI_BIAS)))
    print(f"M={M:6d} units={units:5d} ({units/256:.2f} rounds): ring {res[0]:7.1f} (no store {res[1]:7.1f})   pring {res[2]:7.1f} (no store {res[3]:7.1f})   per round ring {res[0]/units*256:6.1f} pring {res[2]/units*256:6.1f}", flush=True)
ops.set_gemm_variant(0,0)
