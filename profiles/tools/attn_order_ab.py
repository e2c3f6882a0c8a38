"""A/B of the attention kernels' block order (att_block in csrc/mv_attn.hip; knob attn_order of the debug library) in ONE process:
order 0 = row block slowest (rounds 3-4), order 1 = the row blocks of a (sample, head) pair adjacent on one XCD (round 5).
Bench shape B = 64, A = 12, dh = 64, f16 operands, dropout 0.1; interleaved rounds, median of HIP-event times.
usage: python profiles/tools/attn_order_ab.py [rounds]"""
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import medvill_amd as mv
from medvill_amd import hip_ops as ops
dev = "cuda"
B, A, dh = 64, 12, 64
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7


def timed(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


CASES = ((36, 473, "full", True), (36, 473, "full", False), (36, 473, "bar", False), (36, 473, "noncross", False), (36, 473, "s2s", True),
         (100, 665, "s2s", True))
for (N, S, fam, ragged) in CASES:
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
    res = {(o, k): [] for o in (0, 1) for k in ("fwd", "bwd")}
    outs = {}
    for r in range(rounds + 1):
        for o in (0, 1):
            ops.set_attn_order(o)
            fwd(); bwd()
            if r == 0:
                torch.cuda.synchronize()
                outs[o] = (ctx.clone(), dqkv.clone())
                continue
            res[(o, "fwd")].append(timed(fwd))
            res[(o, "bwd")].append(timed(bwd))
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    med = {k: statistics.median(v) for k, v in res.items()}
    print(f"L={L} {fam:9s} {'packed' if pack else 'padded'} rows {M:6d}: fwd {med[(0, 'fwd')]:6.1f} -> {med[(1, 'fwd')]:6.1f} us   "
          f"bwd {med[(0, 'bwd')]:6.1f} -> {med[(1, 'bwd')]:6.1f} us   (order 0 -> order 1; min fwd {min(res[(0, 'fwd')]):.1f} / {min(res[(1, 'fwd')]):.1f}, "
          f"bwd {min(res[(0, 'bwd')]):.1f} / {min(res[(1, 'bwd')]):.1f}); outputs bit-identical: {same}", flush=True)
ops.set_attn_order(1)
