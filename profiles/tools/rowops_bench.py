import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import medvill_amd as mv
from medvill_amd import hip_ops as ops
dev = "cuda"; M, H, I = 32768, 768, 3072
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
bf = torch.bfloat16
x = torch.randn(M, H, device=dev); dy = torch.randn(M, H, device=dev).to(bf)
g = torch.ones(H, device=dev); b = torch.zeros(H, device=dev)
y = torch.empty(M, H, device=dev, dtype=bf); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
dx = torch.empty_like(dy); dxd = torch.empty_like(dy); dg = torch.zeros(H, device=dev); db = torch.zeros(H, device=dev); cs = torch.zeros(H, device=dev)
t = bench(lambda: ops.layernorm_fwd(x, g, b, y, mean, rstd, M, H, 1e-12)); print(f"ln_fwd f32->bf16  {t:7.1f} us  {(M*H*6)/t/1e6:6.2f} TB/s")
t = bench(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dx, dg, db, cs, M, H, dx_drop=dxd, p_drop=0.1, drop_key=5)); print(f"ln_bwd +drop       {t:7.1f} us  {(M*H*10)/t/1e6:6.2f} TB/s")
t = bench(lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dx, dg, db, cs, M, H)); print(f"ln_bwd            {t:7.1f} us  {(M*H*8)/t/1e6:6.2f} TB/s")
for N in (2304, 3072, 768):
    z = torch.randn(M, N, device=dev).to(bf); o = torch.zeros(N, device=dev)
    t = bench(lambda: ops.colsum(z, N, M, N, o)); print(f"colsum N={N:5d}     {t:7.1f} us  {(M*N*2)/t/1e6:6.2f} TB/s")
P = 111_680_060
p = torch.randn(P, device=dev); gr = torch.randn(P, device=dev); m = torch.zeros(P, device=dev); v = torch.zeros(P, device=dev); sh = torch.empty(P, device=dev, dtype=bf)
t = bench(lambda: ops.adamw_step(p, gr, m, v, sh, P, 1e-5, 0.9, 0.999, 1e-6, 0.0, 1), reps=5); print(f"adamw             {t:7.1f} us  {(P*30)/t/1e6:6.2f} TB/s")
# split-K weight gradient: TN GEMM with workspace
for (Mo, No, nm) in ((768, 3072, "dW2"), (2304, 768, "dWqkv"), (768, 768, "dWo")):
    a = torch.randn(M, Mo, device=dev).to(bf); bb = torch.randn(M, No, device=dev).to(bf); c = torch.zeros(Mo, No, device=dev)
    ws = torch.empty(16 * Mo * No, device=dev)
    t0 = bench(lambda: ops.gemm(a, bb, c, ta=True, tb=True, M=Mo, N=No, K=M, lda=Mo, ldb=No, splitk=0, ws=ws))
    print(f"TN {nm:6s} auto splitk {t0:7.1f} us {2.0*M*Mo*No/t0/1e6:6.0f} TF/s")
# cross-entropy over the labelled rows' logits (fused loss + argmax + gradient), R x V f32 in, bf16 gradient out
R, V = 3300, 30522
Vp = (V + 7) // 8 * 8
logits = torch.randn(R, Vp, device=dev); labels = torch.randint(0, V, (R,), device=dev, dtype=torch.int32)
out = torch.zeros(3, device=dev); dl = torch.empty(R, Vp, device=dev, dtype=bf)
t = bench(lambda: ops.ce_fwd_bwd(logits, Vp, labels, R, V, out, dl, Vp, grad_scale=1.0 / R)); print(f"ce fwd+bwd R={R}    {t:7.1f} us  {(R*V*6)/t/1e6:6.2f} TB/s")
