"""Weight-gradient GEMMs of a layer on the two kernels that can run them: the persistent 256x256 kernel (default, split-K slabs that fill 256
CUs) against the 128x128 kernel at three blocks per CU (forced with mv_set_gemm_variant(1, 0); its own split-K choice).  HIP events, same process.
usage: python profiles/tools/dw_kernel_sweep.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from medvill_amd import hip_ops as ops
from dominant import make_case


def t(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


CASES = sys.argv[1:] or ["dw", "dw2", "dwqkv", "dwo", "wo", "dctx", "ffn2", "da", "dxqkv"]
for case in CASES:
    fn, meta = make_case(case)
    res = []
    for name, force, nj in (("default", 0, 0), ("128x128 kernel", 1, 0), ("256-row ring kernel", 2, 0), ("256x128 two blocks per CU", 2, 2), ("256x192 ring", 2, 13)):
        ops.set_gemm_variant(force, nj)
        try:
            us = t(fn)
            res.append(f"{name}: {us:6.1f} us {meta['flop'] / us / 1e6:5.0f} TF/s")
        except RuntimeError as e:
            res.append(f"{name}: refused")
        ops.set_gemm_variant(0, 0)
    print(f"{case:6s} | " + " | ".join(res), flush=True)
