"""Attention forward / backward times at the bench shape (and the other mask families) with whatever library MV_LIB_PATH names: for
same-box A/Bs of two builds (run once per library in one gpurun call).  HIP events, median of 7 rounds x 20 launches.
usage: [MV_LIB_PATH=...] python profiles/tools/attn_time.py"""
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import medvill_amd as mv
from medvill_amd import hip_ops as ops
dev = "cuda"
B, A, dh = 64, 12, 64


def timed(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (N, S, fam, ragged) in ((36, 473, "full", True), (36, 473, "full", False), (36, 473, "bar", False), (36, 473, "s2s", True), (100, 665, "s2s", True)):
    L, H = N + S + 3, A * dh
    g = torch.Generator().manual_seed(1)
    n_ids = torch.randint((S + 1) // 2 + 1, S + 2, (B,), generator=g) if ragged else torch.full((B,), S + 1)
    desc = mv.data.MaskDesc.make(fam, N, S, n_ids, dev)
    pack = desc.packable() and ragged
    bits = torch.zeros((B, L, (L + 31) // 32), dtype=torch.int32, device=dev)
    ti = torch.zeros((B, (L + 63) // 64, (L + 63) // 64), dtype=torch.uint8, device=dev)
    ops.mask_build(desc.desc, B, L, bits, ti)
    cu, M = None, B * L
    if pack:
        cu, _, _ = ops.pack_plan(desc.desc, B, L)
        M = int(cu[-1])
    f16 = torch.float16
    qkv = torch.randn(M, 3 * H, device=dev).to(f16)
    dctx = torch.randn(M, H, device=dev).to(f16)
    ctx = torch.empty(M, H, device=dev, dtype=f16)
    lse, delta = torch.empty(B, A, L, device=dev), torch.empty(B, A, L, device=dev)
    dqkv = torch.empty_like(qkv)
    db = torch.empty(ops.dropbits_numel(B, L, A), dtype=torch.int32, device=dev)
    ops.attn_dropmask(0.1, 12345, B, L, A, db, cu=cu)
    fwd = lambda: ops.attn_fwd(qkv, bits, ti, ctx, lse, B, L, A, dh, p_drop=0.1, cu=cu, total_rows=M, dropbits=db)
    bwd = lambda: ops.attn_bwd(qkv, ctx, dctx, lse, bits, ti, dqkv, delta, B, L, A, dh, p_drop=0.1, cu=cu, total_rows=M, dropbits=db)
    fwd(); bwd()
    tf, tb = [], []
    for r in range(7):
        tf.append(timed(fwd))
        tb.append(timed(bwd))
    print(f"L={L} {fam:9s} {'packed' if pack else 'padded'}: fwd {statistics.median(tf):6.1f} us  bwd {statistics.median(tb):6.1f} us   checksum {float(dqkv.float().abs().sum()):.6e}", flush=True)
