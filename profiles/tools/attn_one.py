"""One forward + one backward attention launch set at the bench shape (packed rows, full mask, dropout 0.1) for rocprofv3 --pmc
runs: python3 profiles/tools/attn_one.py [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import medvill_amd as mv
from medvill_amd import hip_ops as ops
dev = "cuda"
B, A, dh, N, S = 64, 12, 64, 36, 473
L, H = N + S + 3, A * dh
g = torch.Generator().manual_seed(1)
n_ids = torch.randint((S + 1) // 2 + 1, S + 2, (B,), generator=g)
desc = mv.data.MaskDesc.make("full", N, S, n_ids, dev)
bits = torch.zeros((B, L, (L + 31) // 32), dtype=torch.int32, device=dev)
ti = torch.zeros((B, (L + 63) // 64, (L + 63) // 64), dtype=torch.uint8, device=dev)
ops.mask_build(desc.desc, B, L, bits, ti)
cu, rowmap, inv = ops.pack_plan(desc.desc, B, L)
M = int(cu[-1])
qkv = torch.randn(M, 3 * H, device=dev).to(torch.bfloat16)
qkv_f = qkv.to(torch.float16)
dctx = torch.randn(M, H, device=dev).to(torch.bfloat16)
ctx = torch.empty(M, H, device=dev, dtype=torch.float16)
ctx_b = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, A, L, device=dev)
dqkv = torch.empty_like(qkv)
delta = torch.empty(B, A, L, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    ops.attn_fwd(qkv_f, bits, ti, ctx, lse, B, L, A, dh, p_drop=0.0, cu=cu, total_rows=M, ctx_bf16=ctx_b)
    ops.attn_bwd(qkv, ctx_b, dctx, lse, bits, ti, dqkv, delta, B, L, A, dh, p_drop=0.0, cu=cu, total_rows=M)
torch.cuda.synchronize()
print("rows", M)
