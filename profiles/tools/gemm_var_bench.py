"""A/B of 256-row GEMM kernel variants on the step's shapes (f16 operands, packed row count), HIP events, interleaved rounds.
variants: (2,14) ring 256x256x64 x2 stages | (2,24) persistent ring | (2,2) 256x128x32 x3 stages, 4 waves, two blocks per CU | (1,0) 128x128
usage: python profiles/tools/gemm_var_bench.py [rows]"""
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medvill_amd import hip_ops as ops
from medvill_amd._lib import EPI_BIAS, EPI_BIAS_GELU_D, EPI_BIAS_RES, EPI_MUL, EPI_NONE
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25483
H, I = 768, 3072
f16 = torch.float16


def rnd(*s, sc=0.5):
    return (torch.randn(*s, device=dev) * sc).to(f16)


x, xi, x3 = rnd(M, H), rnd(M, I), rnd(M, 3 * H)
W1, Wq, W2t = rnd(I, H, sc=0.02), rnd(3 * H, H, sc=0.02), rnd(I, H, sc=0.02)
b1, b3 = torch.randn(I, device=dev), torch.randn(3 * H, device=dev)
oI, oI2, o3 = torch.empty(M, I, device=dev, dtype=f16), torch.empty(M, I, device=dev, dtype=f16), torch.empty(M, 3 * H, device=dev, dtype=f16)
gW1, gW2, gWq, gWo = (torch.empty(s, device=dev) for s in ((I, H), (H, I), (3 * H, H), (H, H)))
ws = torch.empty(32 * I * H, device=dev)
alpha = torch.tensor([1.0 / 32768], device=dev)
cpart = torch.empty(2 * ((M + 255) // 256), I, device=dev)
gb1 = torch.zeros(I, device=dev)
oH, Wo = torch.empty(M, H, device=dev, dtype=f16), rnd(H, H, sc=0.02)
W2, bH = rnd(H, I, sc=0.02), torch.randn(H, device=dev)
CASES = [
    ("NT ffn1 +bias+gelu+gelu'", lambda: ops.gemm(x, W1, oI, M=M, N=I, K=H, bias=b1, epi=EPI_BIAS_GELU_D, c2=oI2), 2.0 * M * I * H),
    ("NT ffn1 +bias", lambda: ops.gemm(x, W1, oI, M=M, N=I, K=H, bias=b1, epi=EPI_BIAS), 2.0 * M * I * H),
    ("NT qkv  +bias", lambda: ops.gemm(x, Wq, o3, M=M, N=3 * H, K=H, bias=b3, epi=EPI_BIAS), 2.0 * M * 3 * H * H),
    ("NT dz   x gelu'", lambda: ops.gemm(x, W2t, oI, M=M, N=I, K=H, epi=EPI_MUL, r=oI2), 2.0 * M * I * H),
    ("NT dz   x gelu' + colsums", lambda: ops.gemm(x, W2t, oI, M=M, N=I, K=H, epi=EPI_MUL, r=oI2, colsum_part=cpart), 2.0 * M * I * H),
    ("fold dz partials", lambda: ops.colsum_partials(cpart, cpart.shape[0], I, I, gb1), 1.0),
    ("colsum(dz)", lambda: ops.colsum(oI, I, M, I, gb1), 1.0),
    ("TN dW1  3072x768", lambda: ops.gemm(xi, x, gW1, ta=True, tb=True, M=I, N=H, K=M, lda=I, ldb=H, splitk=0, ws=ws, alpha=alpha), 2.0 * M * I * H),
    ("TN dW2  768x3072", lambda: ops.gemm(x, xi, gW2, ta=True, tb=True, M=H, N=I, K=M, lda=H, ldb=I, splitk=0, ws=ws, alpha=alpha), 2.0 * M * I * H),
    ("TN dWqkv 2304x768", lambda: ops.gemm(x3, x, gWq, ta=True, tb=True, M=3 * H, N=H, K=M, lda=3 * H, ldb=H, splitk=0, ws=ws, alpha=alpha), 2.0 * M * 3 * H * H),
    ("NT ffn2 +bias+res K=3072 N=768", lambda: ops.gemm(xi, W2, oH, M=M, N=H, K=I, bias=bH, epi=EPI_BIAS_RES, r=x, p_drop=0.1, drop_key=7), 2.0 * M * I * H),
    ("NT Wo +bias+res K=768 N=768", lambda: ops.gemm(x, Wo, oH, M=M, N=H, K=H, bias=bH, epi=EPI_BIAS_RES, r=x, p_drop=0.1, drop_key=7), 2.0 * M * H * H),
    ("NN dx(qkv) K=2304", lambda: ops.gemm(x3, Wq, oH, tb=True, M=M, N=H, K=3 * H, ldb=H), 2.0 * M * 3 * H * H),
    ("NN dx(ffn1) K=3072", lambda: ops.gemm(xi, W1, oH, tb=True, M=M, N=H, K=I, ldb=H), 2.0 * M * I * H),
    ("NN dx(Wo) K=768", lambda: ops.gemm(x, Wo, oH, tb=True, M=M, N=H, K=H, ldb=H), 2.0 * M * H * H),
    ("TN dWo  768x768", lambda: ops.gemm(x, x, gWo, ta=True, tb=True, M=H, N=H, K=M, lda=H, ldb=H, splitk=0, ws=ws, alpha=alpha), 2.0 * M * H * H),
]
VARS = [("auto", (0, 0)), ("ring256x256", (2, 14)), ("pring", (2, 24)), ("ring 32-deep x4", (2, 4)), ("256x128 2/CU", (2, 2)), ("128x128", (1, 0))]


def t_of(fn, reps=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, fn, fl in CASES:
    res = {v: [] for v, _ in VARS}
    for v, (f, nj) in VARS:
        ops.set_gemm_variant(f, nj)
        try:
            fn()
        except RuntimeError as e:
            res[v] = None
    torch.cuda.synchronize()
    for r in range(4):
        for v, (f, nj) in VARS:
            if res[v] is None:
                continue
            ops.set_gemm_variant(f, nj)
            res[v].append(t_of(fn))
    ops.set_gemm_variant(0, 0)
    print(f"{name:26s} " + " | ".join(f"{v}: " + (f"{statistics.median(t):6.1f} us {fl / statistics.median(t) / 1e6:5.0f} TF/s" if t else "   n/a") for v, t in res.items()), flush=True)
