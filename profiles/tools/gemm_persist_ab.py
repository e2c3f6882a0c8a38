"""Experiment: the persistent form of the 256 x 256 ring kernel (one block per CU walking tiles; a tile's stores drain under the next tile's MFMAs)
for y = x.W^T with f16 operands, against the one-tile-per-block kernel the product uses.  One process, interleaved rounds, HIP events.
usage: python profiles/tools/gemm_persist_ab.py"""
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import dominant
from medvill_amd import hip_ops as ops

ARMS = [("product", (0, 0)), ("256-row ring forced", (2, 14)), ("persistent ring", (2, 24))]
for case in sys.argv[1:] or ("ffn1", "qkv"):
    fn, meta = dominant.make_case(case)
    res = {n: [] for n, _ in ARMS}
    for r in range(6):
        for n, (force, nj) in ARMS:
            ops.set_gemm_variant(force, nj)
            for _ in range(10):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                fn()
            e1.record()
            e1.synchronize()
            if r:
                res[n].append(e0.elapsed_time(e1) / 40 * 1e3)
    ops.set_gemm_variant(0, 0)
    print(f"{case:6s} " + "  |  ".join(f"{n}: {statistics.median(v):6.1f} us" for n, v in res.items()), flush=True)
