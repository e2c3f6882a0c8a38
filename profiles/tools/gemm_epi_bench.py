"""What the FFN-up GEMM's epilogue costs, by part: the same 256x256x64 ring GEMM (25,483 x 3072 x 768, f16 operands) with
{bias} / {bias + GELU + GELU'} and one, two or three 16-bit outputs.  usage: python profiles/tools/gemm_epi_bench.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medvill_amd import hip_ops as ops
from medvill_amd._lib import EPI_BIAS, EPI_BIAS_GELU_D, EPI_NONE
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25483
H, I = 768, 3072
f16, b16 = torch.float16, torch.bfloat16


def bench1(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


x = (torch.randn(M, H, device=dev) * 0.5).to(f16)
W1 = (torch.randn(I, H, device=dev) * 0.05).to(f16)
b1 = torch.randn(I, device=dev)
o1, o2 = torch.empty(M, I, device=dev, dtype=f16), torch.empty(M, I, device=dev, dtype=f16)
o3 = torch.empty(M, I, device=dev, dtype=b16)
o32 = torch.empty(M, I, device=dev)
cases = [
    ("no epilogue, f16 out", lambda: ops.gemm(x, W1, o1, M=M, N=I, K=H, epi=EPI_NONE)),
    ("bias, f16 out", lambda: ops.gemm(x, W1, o1, M=M, N=I, K=H, bias=b1, epi=EPI_BIAS)),
    ("bias, f16 + bf16 out", lambda: ops.gemm(x, W1, o1, M=M, N=I, K=H, bias=b1, epi=EPI_BIAS, c3=o3)),
    ("bias, f32 out", lambda: ops.gemm(x, W1, o32, M=M, N=I, K=H, bias=b1, epi=EPI_BIAS)),
    ("bias+gelu+gelu', f16 x2 out", lambda: ops.gemm(x, W1, o1, M=M, N=I, K=H, bias=b1, epi=EPI_BIAS_GELU_D, c2=o2)),
    ("bias+gelu+gelu', f16 x2 + bf16 out", lambda: ops.gemm(x, W1, o1, M=M, N=I, K=H, bias=b1, epi=EPI_BIAS_GELU_D, c2=o2, c3=o3)),
]
fl = 2.0 * M * I * H
for rnd in range(2):
    for name, fn in cases:
        ms = bench1(fn)
        print(f"{name:40s} {ms * 1e3:7.1f} us {fl / ms / 1e9:5.0f} TF/s", flush=True)
