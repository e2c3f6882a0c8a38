"""Stand-alone TF/s of the 256x256x64 ring GEMM (y = x.W^T, no epilogue work, 16-bit output) over a range of shapes: separates
what the main loop can do (long contractions, square shapes) from what the model's K = 768 shapes cost in prologue / epilogue /
tile quantisation.  usage: python profiles/tools/gemm_shapes.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medvill_amd import hip_ops as ops
dev = "cuda"


def bench1(fn, reps=10):
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


for (M, N, K) in ((4096, 4096, 4096), (8192, 8192, 8192), (8192, 8192, 768), (25600, 3072, 768), (25600, 3072, 3072), (25600, 3072, 12288),
                  (25600, 768, 3072), (65536, 3072, 768), (4096, 4096, 768), (4096, 4096, 1536)):
    out = []
    for dt in (torch.bfloat16, torch.float16):
        a = (torch.randn(M, K, device=dev) * 0.5).to(dt)
        b = (torch.randn(N, K, device=dev) * 0.5).to(dt)
        c = torch.empty(M, N, device=dev, dtype=dt)
        for vn, force, nj in (("ring", 2, 14), ("128", 1, 0)):
            ops.set_gemm_variant(force, nj)
            ms = bench1(lambda: ops.gemm(a, b, c, M=M, N=N, K=K))
            out.append(f"{str(dt)[6:]:8s} {vn}: {ms * 1e3:8.1f} us {2.0 * M * N * K / ms / 1e9:5.0f} TF/s")
        ops.set_gemm_variant(0, 0)
    print(f"{M:6d}x{N:5d}x{K:5d} | " + " | ".join(out), flush=True)
