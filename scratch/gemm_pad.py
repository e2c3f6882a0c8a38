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
W2=(torch.randn(H,I,device=dev)*0.5).to(torch.bfloat16); bh=torch.randn(H,device=dev)
oH=torch.empty(M,H,device=dev,dtype=torch.bfloat16)
for pad in (0, 8, 64, 128, 256, 512):
    xi=(torch.randn(M,I+pad,device=dev)*0.5).to(torch.bfloat16)
    W2p=(torch.randn(H,I+pad,device=dev)*0.5).to(torch.bfloat16)
    res=[]
    for force,nj in ((1,0),(2,14),(2,15)):
        ops.set_gemm_variant(force,nj)
        res.append(bench1(lambda: ops.gemm(xi,W2p,oH,M=M,N=H,K=I,lda=I+pad,ldb=I+pad,bias=bh,epi=EPI_BIAS)))
    print(f"ffn2 NT 32768x768x3072 lda=ldb={I+pad}: old128 {res[0]:.0f} us  256k64 {res[1]:.0f}  192k64 {res[2]:.0f}", flush=True)
ops.set_gemm_variant(0,0)
