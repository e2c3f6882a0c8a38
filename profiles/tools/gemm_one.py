"""The dominant kernel alone, for rocprofv3 passes: the FFN-up projection of one layer at the bench's padded row count
([32768,768] x [3072,768]^T + bias, GELU and GELU' epilogue, f16 operands, the three outputs the training step writes).
usage: python3 profiles/tools/gemm_one.py [reps] [rows]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medvill_amd import hip_ops as ops
from medvill_amd._lib import EPI_BIAS_GELU_D
dev = "cuda"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
M = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
H, I = 768, 3072
x = (torch.randn(M, H, device=dev) * 0.5).to(torch.float16)
w = (torch.randn(I, H, device=dev) * 0.5).to(torch.float16)
b = torch.randn(I, device=dev)
o, d = torch.empty(M, I, device=dev, dtype=torch.float16), torch.empty(M, I, device=dev, dtype=torch.float16)
ob = torch.empty(M, I, device=dev, dtype=torch.bfloat16)
for _ in range(reps):
    ops.gemm(x, w, o, M=M, N=I, K=H, bias=b, epi=EPI_BIAS_GELU_D, c2=d, c3=ob)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.gemm(x, w, o, M=M, N=I, K=H, bias=b, epi=EPI_BIAS_GELU_D, c2=d, c3=ob)
e1.record()
e1.synchronize()
print(f"ffn-up {M}x{I}x{H}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch (HIP events)")
