import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import medvill_amd as mv
from medvill_amd import hip_ops as ops
from medvill_amd._lib import *
dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
def bench1(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
def bench(name, fn, flops, reps=20):
    out = []
    for force, nj in ((0, 0), (2, 24), (2, 14), (1, 0)):
        ops.set_gemm_variant(force, nj)
        out.append(bench1(fn, reps))
    ops.set_gemm_variant(0, 0)
    ms = out[0]
    print(f"{name:38s} auto {ms*1e3:7.1f} us {flops/ms/1e9:6.0f} TF/s | p256:{out[1]*1e3:5.0f} 256k64:{out[2]*1e3:5.0f} old128:{out[3]*1e3:5.0f}", flush=True)
    return ms
H, I = 768, 3072
x = rnd(M, H); xi = rnd(M, I); x3 = rnd(M, 3*H)
W1 = rnd(I, H); W2 = rnd(H, I); Wq = rnd(3*H, H); Wo = rnd(H, H)
b1 = torch.randn(I, device=dev); bh = torch.randn(H, device=dev); b3 = torch.randn(3*H, device=dev)
oI = torch.empty(M, I, device=dev, dtype=torch.bfloat16); oI2 = torch.empty_like(oI)
oH = torch.empty(M, H, device=dev, dtype=torch.bfloat16); oHf = torch.empty(M, H, device=dev)
o3 = torch.empty(M, 3*H, device=dev, dtype=torch.bfloat16)
gW1 = torch.empty(I, H, device=dev); gW2 = torch.empty(H, I, device=dev); gWq = torch.empty(3*H, H, device=dev); gWo = torch.empty(H, H, device=dev)
ws = torch.empty(16 * I * H, device=dev)
tot = 0
tot += bench("NT qkv   32768x2304x768  bias", lambda: ops.gemm(x, Wq, o3, M=M, N=3*H, K=H, bias=b3, epi=EPI_BIAS), 2*M*3*H*H)
tot += bench("NT out   32768x768x768   bias+res f32", lambda: ops.gemm(x, Wo, oHf, M=M, N=H, K=H, bias=bh, epi=EPI_BIAS_RES, r=x), 2*M*H*H)
tot += bench("NT ffn1  32768x3072x768  bias+gelu+d", lambda: ops.gemm(x, W1, oI, M=M, N=I, K=H, bias=b1, epi=EPI_BIAS_GELU_D, c2=oI2), 2*M*I*H)
tot += bench("NT ffn1  32768x3072x768  bias only", lambda: ops.gemm(x, W1, oI, M=M, N=I, K=H, bias=b1, epi=EPI_BIAS), 2*M*I*H)
tot += bench("NT ffn2  32768x768x3072  bias+res f32", lambda: ops.gemm(xi, W2, oHf, M=M, N=H, K=I, bias=bh, epi=EPI_BIAS_RES, r=x), 2*M*I*H)
tot += bench("NN dz    32768x3072x768  mul", lambda: ops.gemm(x, W2, oI, tb=True, M=M, N=I, K=H, epi=EPI_MUL, r=oI2), 2*M*I*H)
tot += bench("NN da    32768x768x3072  res", lambda: ops.gemm(xi, W1, oH, tb=True, M=M, N=H, K=I, epi=EPI_RES, r=x), 2*M*I*H)
tot += bench("NN dctx  32768x768x768", lambda: ops.gemm(x, Wo, oH, tb=True, M=M, N=H, K=H), 2*M*H*H)
tot += bench("NN dx    32768x768x2304  res", lambda: ops.gemm(x3, Wq, oH, tb=True, M=M, N=H, K=3*H, epi=EPI_RES, r=x), 2*M*3*H*H)
tot += bench("TN dW2   768x3072x32768  sk5", lambda: ops.gemm(x, xi, gW2, ta=True, tb=True, M=H, N=I, K=M, lda=H, ldb=I, splitk=0, ws=ws), 2*M*I*H)
tot += bench("TN dW1   3072x768x32768  sk5", lambda: ops.gemm(xi, x, gW1, ta=True, tb=True, M=I, N=H, K=M, lda=I, ldb=H, splitk=0, ws=ws), 2*M*I*H)
tot += bench("TN dWqkv 2304x768x32768  sk7", lambda: ops.gemm(x3, x, gWq, ta=True, tb=True, M=3*H, N=H, K=M, lda=3*H, ldb=H, splitk=0, ws=ws), 2*M*3*H*H)
tot += bench("TN dWo   768x768x32768   sk16", lambda: ops.gemm(x, x, gWo, ta=True, tb=True, M=H, N=H, K=M, lda=H, ldb=H, splitk=0, ws=ws), 2*M*H*H)
print("sum per layer (excl. bias-only variant) ms:", tot)
