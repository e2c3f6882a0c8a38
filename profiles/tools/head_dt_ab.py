"""Experiment: the MLM head's input gradient dt[R, 768] = dlogits[R, V] . E[V, 768] (K = V = 30,522 is not a multiple of 8, so the product runs
on the 128 x 128 kernel with 4 K-slabs).  Measured here at K = 30,520 (the aligned part; the 2-column remainder would be a second tiny call):
128 x 128 kernel (auto) against the 256-row ring kernel (forced), both with their automatic split-K.   usage: python profiles/tools/head_dt_ab.py"""
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medvill_amd import hip_ops as ops

dev = torch.device("cuda:0")
R, H, V, Vp = 3400, 768, 30522, 30528
dl = (torch.randn(R, Vp, device=dev) * 0.01).to(torch.float16)
E = (torch.randn(V, H, device=dev) * 0.02).to(torch.float16)
out = torch.empty(R, H, dtype=torch.float32, device=dev)
ws = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)
for K in (V, 30520):
    res = {}
    for name, (force, nj) in (("auto", (0, 0)), ("ring 256x256 forced", (2, 14)), ("ring 320x256 forced", (2, 10))):
        ops.set_gemm_variant(force, nj)
        try:
            ts = []
            for r in range(5):
                for _ in range(5):
                    ops.gemm(dl, E, out, tb=True, M=R, N=H, K=K, lda=Vp, ldb=H, splitk=0, ws=ws)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ops.gemm(dl, E, out, tb=True, M=R, N=H, K=K, lda=Vp, ldb=H, splitk=0, ws=ws)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) / 20 * 1e3)
            res[name] = f"{statistics.median(ts):6.1f} us"
        except Exception as ex:      # (K % 8 != 0 is refused by the ring kernels)
            res[name] = "refused"
    ops.set_gemm_variant(0, 0)
    print(f"K = {K}: " + "  |  ".join(f"{n}: {v}" for n, v in res.items()), flush=True)
