"""The last layer's attention with every row a query against the consumed rows only (qlim), and what the row reordering costs.
usage: python profiles/tools/tailq_bench.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import medvill_amd as mv
from medvill_amd import hip_ops as ops
dev = "cuda"
cfg = mv.ModelConfig()
B, N, S, A, dh = 64, 36, 473, 12, 64
H, Lq = A * dh, N + S + 3
batch = mv.data.synthetic_batch(cfg.vocab_size, B, N, S, "full", seed=1234, device=dev)
desc = batch["attn_desc"]
cu, rowmap, inv = ops.pack_plan(desc.desc, B, Lq)
M = int(cu[-1])
bits = torch.zeros((B, Lq, (Lq + 31) // 32), dtype=torch.int32, device=dev)
tinfo = torch.zeros((B, (Lq + 63) // 64, (Lq + 63) // 64), dtype=torch.uint8, device=dev)
ops.mask_build(desc.desc, B, Lq, bits, tinfo)
lab = inv.index_select(0, batch["label_rows"].to(dev).long())
sel = torch.cat([lab.to(torch.int32), cu[:B].to(torch.int32)]).contiguous()
perm, newpos, qlim, sel_new = ops.tail_perm(cu, B, Lq, sel, M)
print(f"rows {M}, consumed {sel.numel()}, queries per sample: mean {float(qlim.float().mean()):.1f} max {int(qlim.max())}")
f16 = torch.float16
qkv = (torch.randn(M, 3 * H, device=dev) * 0.5).to(f16)
dctx = (torch.randn(M, H, device=dev) * 0.1).to(f16)
ctx, dq = torch.zeros(M, H, device=dev, dtype=f16), torch.zeros(M, 3 * H, device=dev, dtype=f16)
lse, delta = torch.zeros(B, A, Lq, device=dev), torch.zeros(B, A, Lq, device=dev)
db = torch.zeros((ops.dropbits_numel(B, Lq, A),), dtype=torch.int32, device=dev)
ops.attn_dropmask(0.1, 5, B, Lq, A, db, cu=cu)
x, xp = torch.randn(M, H, device=dev).to(f16), torch.empty(M, H, device=dev, dtype=f16)


def t_of(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, ql in (("every row a query", None), ("consumed rows only", qlim)):
    tf = t_of(lambda: ops.attn_fwd(qkv, bits, tinfo, ctx, lse, B, Lq, A, dh, p_drop=0.1, cu=cu, total_rows=M, dropbits=db, qlim=ql))
    tb = t_of(lambda: ops.attn_bwd(qkv, ctx, dctx, lse, bits, tinfo, dq, delta, B, Lq, A, dh, p_drop=0.1, cu=cu, total_rows=M, dropbits=db, qlim=ql))
    print(f"{name:24s} fwd {tf:6.1f} us  bwd {tb:6.1f} us")
print(f"tail_perm {t_of(lambda: ops.tail_perm(cu, B, Lq, sel, M)):6.1f} us | gather [M,768] {t_of(lambda: ops.gather_rows(x, H, perm, M, H, xp, H)):6.1f} us | "
      f"scatter [M,768] {t_of(lambda: ops.scatter_rows(x, H, perm, M, H, xp, H)):6.1f} us")
for v in (0, 32, 64, 128, 256, 512):
    ql = torch.full((B,), v, dtype=torch.int32, device=dev)
    tf = t_of(lambda: ops.attn_fwd(qkv, bits, tinfo, ctx, lse, B, Lq, A, dh, p_drop=0.1, cu=cu, total_rows=M, dropbits=db, qlim=ql))
    tb = t_of(lambda: ops.attn_bwd(qkv, ctx, dctx, lse, bits, tinfo, dq, delta, B, Lq, A, dh, p_drop=0.1, cu=cu, total_rows=M, dropbits=db, qlim=ql))
    print(f"qlim = {v:3d} for every sample: fwd {tf:6.1f} us  bwd {tb:6.1f} us")
