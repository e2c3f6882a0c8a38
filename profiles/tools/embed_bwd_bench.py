"""embed_bwd alone at the bench shape: how much of it is atomic contention on hot embedding rows ([MASK] appears ~2,600 times per
batch; every position row is hit by every sample)?  Variants: the bench batch / all token ids distinct / no word-table scatter.
usage: python profiles/tools/embed_bwd_bench.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import medvill_amd as mv
from medvill_amd import hip_ops as ops
from medvill_amd._lib import MV_BF16
dev = "cuda"
cfg = mv.ModelConfig()
B, N, S = 64, 36, 473
H, V, T = cfg.hidden, cfg.vocab_size, S
batch = mv.data.synthetic_batch(V, B, N, S, "full", seed=1234, device=dev)
desc = batch["attn_desc"]
L = N + S + 3
cu, rowmap, inv = ops.pack_plan(desc.desc, B, L)
M = int(cu[-1])


def bench(fn, reps=20):
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


dx0 = (torch.randn(M, H, device=dev) * 0.01).to(torch.bfloat16)
pre = torch.randn(M, H, device=dev)
mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
g = torch.ones(H, device=dev)
dE, dP, dTy = torch.zeros(V, H, device=dev), torch.zeros(cfg.max_pos, H, device=dev), torch.zeros(2, H, device=dev)
dg, db = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
dimg = torch.zeros(B * N, H, device=dev, dtype=torch.bfloat16)
i64 = torch.int64
cls_tok, sep_tok = batch["cls_tok"].to(dev, i64).view(-1), batch["sep_tok"].to(dev, i64).view(-1)
seg, ipos = batch["segment"].to(dev, i64).contiguous(), batch["img_pos"].to(dev, i64).contiguous()


def run(txt):
    ops.embed_bwd(MV_BF16, dx0, pre, mean, rstd, g, cls_tok, txt, seg, ipos, sep_tok, dE, dP, dTy, dg, db, dimg, B, N, T, H, V, cfg.max_pos,
                  pad_token_id=0, p_drop=0.1, drop_key=5, rowmap=rowmap, n_rows=M)


txt = batch["input_txt"].to(dev, i64).contiguous()
uniq = (torch.arange(B * T, device=dev) % (V - 1000) + 1000).view(B, T).contiguous()
pad = torch.zeros_like(txt)
vals, cnt = torch.unique(txt, return_counts=True)
print(f"rows {M}; most frequent token ids in the bench batch: {[(int(v), int(c)) for c, v in sorted(zip(cnt.tolist(), vals.tolist()), reverse=True)[:4]]}")
print(f"bench batch                      {bench(lambda: run(txt)):7.1f} us")
print(f"all token ids distinct           {bench(lambda: run(uniq)):7.1f} us")
print(f"all [PAD] (no word-table adds)   {bench(lambda: run(pad)):7.1f} us")
