"""Experiment: row tiles per group of the ring GEMMs' tile rasterisation (debug bits 32 / 64 / 128 of gemm_dbg: 4 / 16 / 12 instead of 8), i.e. the
shape of the window of tiles resident on one XCD (8 x 4 by default: 8 A panels + 4 B panels behind its 4 MiB L2), at the step's shapes.
One process, interleaved rounds, HIP events.      usage: python profiles/tools/gemm_group_ab.py"""
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import dominant
from medvill_amd import hip_ops as ops

ARMS = [("GM 8 (product)", 0), ("GM 4", 32), ("GM 12", 128), ("GM 16", 64)]
for case in ("ffn1", "qkv", "dz", "ffn2", "da", "dxqkv", "wo"):
    fn, meta = dominant.make_case(case)
    res = {n: [] for n, _ in ARMS}
    for r in range(6):
        for n, bits in ARMS:
            ops.set_gemm_variant(bits << 8, 0)       # force = 0 (auto), dbg = bits
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
