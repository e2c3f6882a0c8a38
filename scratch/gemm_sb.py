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
bf=torch.bfloat16
for M in (25483, 32768):
    x=(torch.randn(M,H,device=dev)*0.5).to(bf); Wo=(torch.randn(H,H,device=dev)*0.5).to(bf); bh=torch.randn(H,device=dev)
    xi=(torch.randn(M,I,device=dev)*0.5).to(bf); W2=(torch.randn(H,I,device=dev)*0.5).to(bf); r=torch.randn(M,H,device=dev).to(bf)
    W1=(torch.randn(I,H,device=dev)*0.5).to(bf); x3=(torch.randn(M,3*H,device=dev)*0.5).to(bf); Wq=(torch.randn(3*H,H,device=dev)*0.5).to(bf)
    of=torch.empty(M,H,device=dev); ob=torch.empty(M,H,device=dev,dtype=bf)
    cases = {
      "out  NT": lambda: ops.gemm(x,Wo,of,M=M,N=H,K=H,bias=bh,epi=EPI_BIAS_RES,r=r),
      "ffn2 NT": lambda: ops.gemm(xi,W2,of,M=M,N=H,K=I,bias=bh,epi=EPI_BIAS_RES,r=r),
      "da   NN": lambda: ops.gemm(xi,W1,ob,tb=True,M=M,N=H,K=I,epi=EPI_RES,r=r),
      "dctx NN": lambda: ops.gemm(x,Wo,ob,tb=True,M=M,N=H,K=H),
      "dx   NN": lambda: ops.gemm(x3,Wq,ob,tb=True,M=M,N=H,K=3*H,epi=EPI_RES,r=r),
    }
    for name, fn in cases.items():
        t=[]
        for nj in (0, 31, 0, 31):
            ops.set_gemm_variant(0, nj); t.append(bench(fn))
        ops.set_gemm_variant(0,0)
        print(f"M={M} {name}: two stages {t[2]:6.1f} us   one stage, 3 blocks/CU {t[3]:6.1f} us", flush=True)
