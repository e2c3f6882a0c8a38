"""LayerNorm backward at the step's row count (f16 in / out), interleaved rounds of the kernel variants.  usage: python profiles/tools/ln_bwd_bench.py [rows]"""
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medvill_amd import hip_ops as ops
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25483
H = 768
f16 = torch.float16
x, dy = torch.randn(M, H, device=dev).to(f16), torch.randn(M, H, device=dev).to(f16)
g = torch.ones(H, device=dev)
mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
dx, dxd = torch.empty_like(dy), torch.empty_like(dy)
dg, db, cs = (torch.zeros(H, device=dev) for _ in range(3))
us = torch.tensor([1.0 / 32768], device=dev)


def t_of(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


CASES = [("with dropout output + colsum", lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dx, dg, db, cs, M, H, dx_drop=dxd, p_drop=0.1, drop_key=5, unscale=us), 10),
         ("plain", lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dx, dg, db, None, M, H, unscale=us), 8)]
VARS = [("old: one row at a time, 4 waves x 1024 blocks", 1), ("prefetch, 4 x 1024", 2), ("prefetch, 4 x 512", 2 | (512 << 8)), ("prefetch, 8 x 512 (default)", 0), ("prefetch, 8 x 256", 256 << 8), ("prefetch, 16 x 256", 3), ("prefetch, 16 x 128", 3 | (128 << 8))]
for name, fn, bpe in CASES:
    res = {v: [] for v, _ in VARS}
    for r in range(5):
        for v, code in VARS:
            ops.set_rowops_variant(code)
            fn()
            res[v].append(t_of(fn))
    ops.set_rowops_variant(0)
    print(name + ": " + " | ".join(f"{v}: {statistics.median(t):5.1f} us ({M * H * bpe / statistics.median(t) / 1e6:4.2f} TB/s)" for v, t in res.items()), flush=True)
