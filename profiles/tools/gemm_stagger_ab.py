"""Experiment: staggered start of half of the first round's blocks in the ring GEMMs (debug bits 32 / 64 of gemm_dbg: 2 / 4 / 6 x ~8,100
cycles), on the epilogue-heavy wide products of a layer at the step's shapes.  One process, interleaved rounds, HIP events.
usage: python profiles/tools/gemm_stagger_ab.py"""
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import dominant
from medvill_amd import hip_ops as ops

ARMS = [("lockstep (product)", 0), ("stagger ~16k cycles", 32), ("stagger ~32k cycles", 64), ("stagger ~49k cycles", 96)]
for case in ("ffn1", "qkv", "dz", "ffn2", "da"):
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
