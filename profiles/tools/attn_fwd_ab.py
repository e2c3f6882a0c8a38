"""A/B of the attention FORWARD kernels in one process (knob attn_fwd of the debug library): 0 = one 32-query sub-tile per wave, 128-query
blocks, three blocks per CU (rounds 1-4) | 1 = two sub-tiles per wave over shared K / V fragments, 256-query blocks, two blocks per CU.
Bench shape B = 64, A = 12, dh = 64, f16, dropout 0.1 and 0; interleaved rounds, median HIP-event times; outputs compared.
usage: python profiles/tools/attn_fwd_ab.py [rounds]"""
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
ARMS = [int(x) for x in os.environ.get("ARMS", "0,1").split(",")]


def timed(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


CASES = ((36, 473, "full", True, None), (36, 473, "full", False, None), (36, 473, "bar", False, None), (36, 473, "noncross", False, None),
         (36, 473, "s2s", True, None), (100, 665, "s2s", True, None), (36, 473, "full", True, 60))
for (N, S, fam, ragged, qcap) in CASES:
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
    qlim = torch.full((B,), qcap, dtype=torch.int32, device=dev) if qcap else None      # the last layer's form: few queries, every row a key
    f16 = torch.float16
    qkv = torch.randn(M, 3 * H, device=dev).to(f16)
    lse = torch.empty(B, A, L, device=dev)
    db = torch.empty(ops.dropbits_numel(B, L, A), dtype=torch.int32, device=dev)
    ops.attn_dropmask(0.1, 12345, B, L, A, db, cu=cu)
    for p in (0.1, 0.0):
        ctx = torch.zeros(M, H, device=dev, dtype=f16)
        fwd = lambda: ops.attn_fwd(qkv, bits, ti, ctx, lse, B, L, A, dh, p_drop=p, cu=cu, total_rows=M, dropbits=db if p > 0 else None, qlim=qlim)
        res = {o: [] for o in ARMS}
        outs = {}
        for r in range(rounds + 1):
            for o in ARMS:
                ops.set_attn_fwd(o)
                if r == 0:
                    ctx.zero_(); lse.zero_()
                    fwd()
                    torch.cuda.synchronize()
                    outs[o] = (ctx.clone(), lse.clone())
                    continue
                fwd()
                res[o].append(timed(fwd))
        ref = outs[ARMS[0]]
        dif = " ".join(f"[{o}: |dctx| {float((outs[o][0].float() - ref[0].float()).abs().max()):.2e} |dlse| {float((outs[o][1] - ref[1]).abs().max()):.2e}]" for o in ARMS[1:])
        print(f"L={L} {fam:9s} {'packed' if pack else 'padded'} rows {M:6d}{' qlim ' + str(qcap) if qcap else ''} p={p}: "
              + "  ".join(f"fwd[{o}] {statistics.median(res[o]):6.1f} us (min {min(res[o]):.1f})" for o in ARMS) + "  " + dif, flush=True)
ops.set_attn_fwd(0)
