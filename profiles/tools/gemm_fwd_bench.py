"""Stand-alone timing of the forward GEMMs of one encoder layer (y = x.W^T forms) at the bench's packed row count, per kernel
variant.  HIP events, 20 reps after 3 warm-ups, random operands.  usage: python profiles/tools/gemm_fwd_bench.py [rows] [f16|bf16]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medvill_amd import hip_ops as ops
from medvill_amd._lib import EPI_BIAS, EPI_BIAS_GELU_D, EPI_BIAS_RES
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25483
fdt = torch.bfloat16 if "bf16" in sys.argv[2:] else torch.float16
H, I = 768, 3072


def rnd(*s, dt=fdt):
    return (torch.randn(*s, device=dev) * 0.5).to(dt)


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


x, xi = rnd(M, H), rnd(M, I)
W1, W2, Wq, Wo, E = rnd(I, H), rnd(H, I), rnd(3 * H, H), rnd(H, H), rnd(30522, H)
b1, bh, b3, bv = (torch.randn(n, device=dev) for n in (I, H, 3 * H, 30522))
oI, oI2 = torch.empty(M, I, device=dev, dtype=fdt), torch.empty(M, I, device=dev, dtype=fdt)
oIb = torch.empty(M, I, device=dev, dtype=torch.bfloat16) if fdt == torch.float16 else None
o3 = torch.empty(M, 3 * H, device=dev, dtype=fdt)
o3b = torch.empty(M, 3 * H, device=dev, dtype=torch.bfloat16) if fdt == torch.float16 else None
oHf = torch.empty(M, H, device=dev)
R = 3300
t = rnd(R, H)
lg = torch.empty(R, 30528, device=dev)
cases = [
    ("qkv   %dx2304x768 bias (+bf16 copy)" % M, lambda: ops.gemm(x, Wq, o3, M=M, N=3 * H, K=H, bias=b3, epi=EPI_BIAS, c3=o3b), 2.0 * M * 3 * H * H),
    ("ffn1  %dx3072x768 bias+gelu+gelu' (+bf16 copy)" % M, lambda: ops.gemm(x, W1, oI, M=M, N=I, K=H, bias=b1, epi=EPI_BIAS_GELU_D, c2=oI2, c3=oIb), 2.0 * M * I * H),
    ("ffn1  %dx3072x768 bias only" % M, lambda: ops.gemm(x, W1, oI, M=M, N=I, K=H, bias=b1, epi=EPI_BIAS), 2.0 * M * I * H),
    ("out   %dx768x768 bias+res f32" % M, lambda: ops.gemm(x, Wo, oHf, M=M, N=H, K=H, bias=bh, epi=EPI_BIAS_RES, r=x), 2.0 * M * H * H),
    ("ffn2  %dx768x3072 bias+res f32" % M, lambda: ops.gemm(xi, W2, oHf, M=M, N=H, K=I, bias=bh, epi=EPI_BIAS_RES, r=x), 2.0 * M * I * H),
    ("dec   %dx30522x768 bias f32" % R, lambda: ops.gemm(t, E, lg, M=R, N=30522, K=H, ldc=30528, bias=bv, epi=EPI_BIAS), 2.0 * R * 30522 * H),
]
variants = (("auto", 0, 0), ("ring", 2, 14), ("128", 1, 0))
for name, fn, fl in cases:
    out = []
    for vn, force, nj in variants:
        ops.set_gemm_variant(force, nj)
        ms = bench1(fn)
        out.append(f"{vn}: {ms * 1e3:6.1f} us {fl / ms / 1e9:5.0f} TF/s")
    ops.set_gemm_variant(0, 0)
    print(f"{name:52s} | " + " | ".join(out), flush=True)
