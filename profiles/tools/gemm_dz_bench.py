"""dz = dpre2 . W2 (NN form, multiply-by-gelu' epilogue) of one layer at the packed row count: 128x128 kernel against the 256-row
ring kernel, with and without the elementwise operand, to price the epilogue.  usage: python profiles/tools/gemm_dz_bench.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medvill_amd import hip_ops as ops
from medvill_amd._lib import EPI_BIAS_RES, EPI_MUL, EPI_NONE, EPI_RES
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25483
H, I = 768, 3072
b16, f16 = torch.bfloat16, torch.float16


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


dy = (torch.randn(M, H, device=dev) * 0.5).to(b16)
W2 = (torch.randn(H, I, device=dev) * 0.05).to(b16)          # [out=768][in=3072]: contraction-major for dz = dy . W2
W2t = W2.t().contiguous()                                     # [3072][768]: the NT form's operand
gd = torch.rand(M, I, device=dev).to(f16)
dz = torch.empty(M, I, device=dev, dtype=b16)
dzi = (torch.randn(M, I, device=dev) * 0.5).to(b16)
W1 = (torch.randn(I, H, device=dev) * 0.05).to(b16)          # da = dz . W1 : [M,3072] x [3072,768]
da = torch.empty(M, H, device=dev, dtype=b16)
dq3 = (torch.randn(M, 3 * H, device=dev) * 0.5).to(b16)
Wq = (torch.randn(3 * H, H, device=dev) * 0.05).to(b16)
xi16 = (torch.randn(M, I, device=dev) * 0.5).to(f16)
W2f = (torch.randn(H, I, device=dev) * 0.05).to(f16)
a16 = (torch.randn(M, H, device=dev) * 0.5).to(f16)
pre16 = torch.empty(M, H, device=dev, dtype=f16)
bh = torch.randn(H, device=dev)
cases = [
    ("dz NN  (W2 as stored)  no epilogue", lambda: ops.gemm(dy, W2, dz, tb=True, M=M, N=I, K=H), 2.0 * M * I * H),
    ("dz NN  (W2 as stored)  x gelu'", lambda: ops.gemm(dy, W2, dz, tb=True, M=M, N=I, K=H, epi=EPI_MUL, r=gd), 2.0 * M * I * H),
    ("dz NT  (W2^T copy)     no epilogue", lambda: ops.gemm(dy, W2t, dz, M=M, N=I, K=H), 2.0 * M * I * H),
    ("dz NT  (W2^T copy)     x gelu'", lambda: ops.gemm(dy, W2t, dz, M=M, N=I, K=H, epi=EPI_MUL, r=gd), 2.0 * M * I * H),
    ("da NN  (W1 as stored)  + residual", lambda: ops.gemm(dzi, W1, da, tb=True, M=M, N=H, K=I, epi=EPI_RES, r=dy), 2.0 * M * I * H),
    ("da NN  (W1 as stored)  no epilogue", lambda: ops.gemm(dzi, W1, da, tb=True, M=M, N=H, K=I), 2.0 * M * I * H),
    ("dx NN  (Wqkv)          + residual", lambda: ops.gemm(dq3, Wq, da, tb=True, M=M, N=H, K=3 * H, epi=EPI_RES, r=dy), 2.0 * M * 3 * H * H),
    ("dx NN  (Wqkv)          no epilogue", lambda: ops.gemm(dq3, Wq, da, tb=True, M=M, N=H, K=3 * H), 2.0 * M * 3 * H * H),
    ("ffn2 NT f16 bias+res+drop -> f16", lambda: ops.gemm(xi16, W2f, pre16, M=M, N=H, K=I, bias=bh, epi=EPI_BIAS_RES, r=a16, p_drop=0.1, drop_key=7), 2.0 * M * I * H),
    ("ffn2 NT f16 no epilogue   -> f16", lambda: ops.gemm(xi16, W2f, pre16, M=M, N=H, K=I), 2.0 * M * I * H),
]
for name, fn, fl in cases:
    out = []
    for vn, force, nj in (("auto", 0, 0), ("ring", 2, 14), ("128", 1, 0)):
        ops.set_gemm_variant(force, nj)
        ms = bench1(fn)
        out.append(f"{vn}: {ms * 1e3:6.1f} us {fl / ms / 1e9:5.0f} TF/s")
    ops.set_gemm_variant(0, 0)
    print(f"{name:40s} | " + " | ".join(out), flush=True)
