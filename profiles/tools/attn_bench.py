"""Stand-alone timing of the attention kernels at the bench shape (B=64, A=12, dh=64; ragged valid lengths like the bench
batches, packed rows), dropout off / on.  HIP events, 10 reps after 2 warm-ups.  usage: python profiles/tools/attn_bench.py [f16]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import medvill_amd as mv
from medvill_amd import hip_ops as ops
dev = "cuda"
B, A, dh = 64, 12, 64
fdt = torch.float16 if "f16" in sys.argv[1:] else torch.bfloat16


def bench(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


CASES = ((36, 473, "full"), (36, 473, "s2s"), (36, 473, "noncross"), (100, 665, "s2s"))
if "full512" in sys.argv[1:]:
    CASES = CASES[:1]
for (N, S, fam) in CASES:
    L = N + S + 3
    H = A * dh
    g = torch.Generator().manual_seed(1)
    n_ids = torch.randint((S + 1) // 2 + 1, S + 2, (B,), generator=g)
    desc = mv.data.MaskDesc.make(fam, N, S, n_ids, dev)
    pack = desc.packable()
    bits = torch.zeros((B, L, (L + 31) // 32), dtype=torch.int32, device=dev)
    ti = torch.zeros((B, (L + 63) // 64, (L + 63) // 64), dtype=torch.uint8, device=dev)
    ops.mask_build(desc.desc, B, L, bits, ti)
    cu = None
    M = B * L
    if pack:
        cu, rowmap, inv = ops.pack_plan(desc.desc, B, L)
        M = int(cu[-1])
    vl = (N + 2 + n_ids).double() if pack else torch.full((B,), float(L)).double()
    gdt = torch.float16 if "f16g" in sys.argv[1:] else torch.bfloat16          # encoding of the backward's operands
    qkv = torch.randn(M, 3 * H, device=dev).to(gdt)
    qkv_f = qkv.to(fdt)
    dctx = torch.randn(M, H, device=dev).to(gdt)
    ctx = torch.empty(M, H, device=dev, dtype=fdt)
    ctx_b = torch.empty(M, H, device=dev, dtype=torch.bfloat16) if (fdt == torch.float16 and gdt != fdt) else None
    lse = torch.empty(B, A, L, device=dev)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B, A, L, device=dev)
    fl = 4.0 * A * dh * float((vl * vl).sum())          # forward FLOPs on the rows that exist (dense within them)
    for p in (0.0, 0.1):
        db = None
        extra = ""
        if p > 0:
            db = torch.empty(ops.dropbits_numel(B, L, A), dtype=torch.int32, device=dev)
            tgs = []
            for pl in (8, 12, 16):
                ops.set_attn_planes(pl)
                tgs.append(bench(lambda: ops.attn_dropmask(p, 12345, B, L, A, db, cu=cu)))
            extra = f"  | keep-bits generator, 8 / 12 / 16 bits per uniform: {tgs[0]:5.1f} / {tgs[1]:5.1f} / {tgs[2]:5.1f} us"
        tf = bench(lambda: ops.attn_fwd(qkv_f, bits, ti, ctx, lse, B, L, A, dh, p_drop=p, cu=cu, total_rows=M, ctx_bf16=ctx_b, dropbits=db))
        cb = ctx_b if ctx_b is not None else ctx
        tb = bench(lambda: ops.attn_bwd(qkv, cb, dctx, lse, bits, ti, dqkv, delta, B, L, A, dh, p_drop=p, cu=cu, total_rows=M, dropbits=db))
        print(f"L={L} {fam:9s} rows {M:6d} p={p}: fwd {tf:7.1f} us ({fl / tf / 1e6:6.0f} TF/s)  bwd {tb:7.1f} us ({2.5 * fl / tb / 1e6:6.0f} TF/s)"
              + extra, flush=True)
