import sys, os
sys.path.insert(0, os.getcwd())
import torch
from medvill_amd import hip_ops as ops
from medvill_amd._lib import *
dev="cuda"; H=768; I=3072; M=int(sys.argv[1]) if len(sys.argv)>1 else 32768
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
bf=torch.bfloat16
dy=(torch.randn(M,H,device=dev)*0.5).to(bf); W2=(torch.randn(H,I,device=dev)*0.5).to(bf); W2t=W2.t().contiguous()
z=torch.randn(M,I,device=dev).to(bf); dz=torch.empty(M,I,device=dev,dtype=bf); dz2=torch.empty_like(dz)
W1=(torch.randn(I,H,device=dev)*0.5).to(bf); W1t=W1.t().contiguous(); da=torch.empty(M,H,device=dev,dtype=bf); da2=torch.empty_like(da); r=torch.randn(M,H,device=dev).to(bf)
Wq=(torch.randn(3*H,H,device=dev)*0.5).to(bf); Wqt=Wq.t().contiguous(); dqkv=torch.randn(M,3*H,device=dev).to(bf); dx=torch.empty(M,H,device=dev,dtype=bf)
for rep in range(2):
    t1=bench(lambda: ops.gemm(dy,W2,dz,tb=True,M=M,N=I,K=H,epi=EPI_MUL,r=z))
    t2=bench(lambda: ops.gemm(dy,W2t,dz2,M=M,N=I,K=H,epi=EPI_MUL,r=z))
    t3=bench(lambda: ops.gemm(dz,W1,da,tb=True,M=M,N=H,K=I,epi=EPI_RES,r=r))
    t4=bench(lambda: ops.gemm(dz,W1t,da2,M=M,N=H,K=I,epi=EPI_RES,r=r))
    t5=bench(lambda: ops.gemm(dqkv,Wq,dx,tb=True,M=M,N=H,K=3*H,epi=EPI_RES,r=r))
    t6=bench(lambda: ops.gemm(dqkv,Wqt,dx,M=M,N=H,K=3*H,epi=EPI_RES,r=r))
    ops.set_gemm_variant(2,14)
    t7=bench(lambda: ops.gemm(dz,W1t,da2,M=M,N=H,K=I,epi=EPI_RES,r=r))
    t8=bench(lambda: ops.gemm(dqkv,Wqt,dx,M=M,N=H,K=3*H,epi=EPI_RES,r=r))
    ops.set_gemm_variant(0,0)
    print(f"dz: NN {t1:6.1f}  NT(W2^T) {t2:6.1f} | da: NN {t3:6.1f} NT {t4:6.1f} ring {t7:6.1f} | dx: NN {t5:6.1f} NT {t6:6.1f} ring {t8:6.1f}   maxdiff dz {float((dz.float()-dz2.float()).abs().max()):.3g}", flush=True)
