"""EXPERIMENT: does running two half-batches of the encoder forward on two HIP streams fill the tile-quantisation tails (and
overlap the HBM-bound LayerNorms with the MFMA-bound GEMMs) better than one full batch on one stream?  12 layers of
[qkv GEMM, attention, out GEMM + residual, LN, FFN-up + GELU + GELU', FFN-down + residual, LN] at the bench shape (B=64, ragged
packed rows), f16 operands with the bf16 copies the training step writes.  usage: python profiles/tools/two_stream_fwd.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import medvill_amd as mv
from medvill_amd import hip_ops as ops
from medvill_amd._lib import EPI_BIAS, EPI_BIAS_GELU_D, EPI_BIAS_RES
dev = "cuda"
A, dh, H, I, N, S = 12, 64, 768, 3072, 36, 473
L = N + S + 3
f16, b16 = torch.float16, torch.bfloat16


def rnd(*s, dt=f16):
    return (torch.randn(*s, device=dev) * 0.5).to(dt)


Wq, Wo, W1, W2 = rnd(3 * H, H), rnd(H, H), rnd(I, H), rnd(H, I)
bq, bo, b1, b2 = (torch.randn(n, device=dev) for n in (3 * H, H, I, H))
g, be = torch.ones(H, device=dev), torch.zeros(H, device=dev)


class Part:
    def __init__(self, n_ids):
        B = self.B = n_ids.numel()
        desc = mv.data.MaskDesc.make("full", N, S, n_ids, dev)
        self.bits = torch.zeros((B, L, (L + 31) // 32), dtype=torch.int32, device=dev)
        self.ti = torch.zeros((B, (L + 63) // 64, (L + 63) // 64), dtype=torch.uint8, device=dev)
        ops.mask_build(desc.desc, B, L, self.bits, self.ti)
        self.cu, _, _ = ops.pack_plan(desc.desc, B, L)
        M = self.M = int(self.cu[-1])
        e = lambda n, dt: torch.empty(M, n, device=dev, dtype=dt)
        self.x, self.xb = rnd(M, H), e(H, b16)
        self.qkv, self.qkvb = e(3 * H, f16), e(3 * H, b16)
        self.ctx, self.ctxb = e(H, f16), e(H, b16)
        self.pre, self.a, self.ab = e(H, torch.float32), e(H, f16), e(H, b16)
        self.i, self.ib, self.gd = e(I, f16), e(I, b16), e(I, f16)
        self.lse = torch.empty(B, A, L, device=dev)
        self.mean, self.rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)

    def layer(self, l):
        M = self.M
        ops.gemm(self.x, Wq, self.qkv, M=M, N=3 * H, K=H, bias=bq, epi=EPI_BIAS, c3=self.qkvb)
        ops.attn_fwd(self.qkv, self.bits, self.ti, self.ctx, self.lse, self.B, L, A, dh, p_drop=0.0, cu=self.cu, total_rows=M,
                     ctx_bf16=self.ctxb)
        ops.gemm(self.ctx, Wo, self.pre, M=M, N=H, K=H, bias=bo, epi=EPI_BIAS_RES, r=self.x, p_drop=0.1, drop_key=5 + l)
        ops.layernorm_fwd(self.pre, g, be, self.a, self.mean, self.rstd, M, H, 1e-12, y_bf16=self.ab)
        ops.gemm(self.a, W1, self.i, M=M, N=I, K=H, bias=b1, epi=EPI_BIAS_GELU_D, c2=self.gd, c3=self.ib)
        ops.gemm(self.i, W2, self.pre, M=M, N=H, K=I, bias=b2, epi=EPI_BIAS_RES, r=self.a, p_drop=0.1, drop_key=9 + l)
        ops.layernorm_fwd(self.pre, g, be, self.x, self.mean, self.rstd, M, H, 1e-12, y_bf16=self.xb)


gen = torch.Generator().manual_seed(1)
n_ids = torch.randint((S + 1) // 2 + 1, S + 2, (64,), generator=gen)
full = Part(n_ids)
halves = [Part(n_ids[:32]), Part(n_ids[32:])]
quarters = [Part(n_ids[i:i + 16]) for i in range(0, 64, 16)]
streams = [torch.cuda.Stream() for _ in range(4)]


def run_one():
    for l in range(12):
        full.layer(l)


def run_split(parts):
    cur = torch.cuda.current_stream()
    for s in streams[:len(parts)]:
        s.wait_stream(cur)
    for l in range(12):
        for p, s in zip(parts, streams):
            with torch.cuda.stream(s):
                p.layer(l)
    for s in streams[:len(parts)]:
        cur.wait_stream(s)


def run_serial(parts):
    for l in range(12):
        for p in parts:
            p.layer(l)


def bench(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"rows: full {full.M}, halves {[p.M for p in halves]}, quarters {[p.M for p in quarters]}")
for rnd_ in range(3):
    print(f"round {rnd_}: one stream, full batch {bench(run_one):.3f} ms | two halves on two streams {bench(lambda: run_split(halves)):.3f} ms | "
          f"two halves on one stream {bench(lambda: run_serial(halves)):.3f} ms | four quarters on four streams {bench(lambda: run_split(quarters)):.3f} ms", flush=True)
