"""Where a 256x256 ring GEMM's time goes: the kernel's debug bits (mv_set_gemm_variant(force | dbg << 8, variant): 1 = no epilogue,
2 = no operand loads, 4 = no fragment reads / MFMAs) on the step's shapes.  usage: python profiles/tools/gemm_ablate.py [rows]"""
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medvill_amd import hip_ops as ops
from medvill_amd._lib import EPI_BIAS, EPI_BIAS_GELU_D
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25483
H, I = 768, 3072
f16 = torch.float16
x, xi = (torch.randn(M, H, device=dev) * 0.5).to(f16), (torch.randn(M, I, device=dev) * 0.5).to(f16)
W1 = (torch.randn(I, H, device=dev) * 0.02).to(f16)
b1 = torch.randn(I, device=dev)
oI, oH = torch.empty(M, I, device=dev, dtype=f16), torch.empty(M, H, device=dev, dtype=f16)
gW1 = torch.empty(I, H, device=dev)
ws = torch.empty(32 * I * H, device=dev)
oI2 = torch.empty(M, I, device=dev, dtype=f16)
CASES = [
    ("NT ffn1 +bias+gelu+gelu' (K=768)", lambda: ops.gemm(x, W1, oI, M=M, N=I, K=H, bias=b1, epi=EPI_BIAS_GELU_D, c2=oI2)),
    ("NT ffn1 +bias (K=768)", lambda: ops.gemm(x, W1, oI, M=M, N=I, K=H, bias=b1, epi=EPI_BIAS)),
    ("TN dW1 3072x768 (K=rows, 7 slabs)", lambda: ops.gemm(xi, x, gW1, ta=True, tb=True, M=I, N=H, K=M, lda=I, ldb=H, splitk=0, ws=ws)),
    ("NN dx(ffn1) (K=3072)", lambda: ops.gemm(xi, W1, oH, tb=True, M=M, N=H, K=I, ldb=H)),
]
ARMS = [("whole kernel", 0), ("stores folded onto 256 rows (no HBM writes)", 16), ("no epilogue", 1), ("no epilogue, no loads", 3), ("no epilogue, no MFMA", 5), ("no epilogue, neither", 7)]


def t_of(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, fn in CASES:
    out = []
    for arm, dbg in ARMS:
        ops.set_gemm_variant(2 | (dbg << 8), 14)
        fn()
        torch.cuda.synchronize()
        out.append((arm, statistics.median(t_of(fn) for _ in range(4))))
    ops.set_gemm_variant(0, 0)
    print(f"{name:36s} " + " | ".join(f"{a}: {t:6.1f} us" for a, t in out), flush=True)
