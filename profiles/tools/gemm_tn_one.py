"""The weight-gradient GEMM alone, for rocprofv3 passes: dW1 = dz^T . a of one layer at the packed row count ([25483,3072]^T x
[25483,768], contraction over the rows, split-K partials + reduction).  usage: python3 profiles/tools/gemm_tn_one.py [reps] [rows] [nt]
(nt: the same FLOPs as a row-major y = x.W^T GEMM with a 25483-deep contraction, for comparison)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medvill_amd import hip_ops as ops
dev = "cuda"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
M = int(sys.argv[2]) if len(sys.argv) > 2 else 25483
nt = "nt" in sys.argv[3:]
H, I = 768, 3072
b16 = torch.bfloat16
Mk = (M + 7) // 8 * 8
dz = (torch.randn(M, I, device=dev) * 0.5).to(b16)
a = (torch.randn(M, H, device=dev) * 0.5).to(b16)
dzT = (torch.randn(I, Mk, device=dev) * 0.5).to(b16)
aT = (torch.randn(H, Mk, device=dev) * 0.5).to(b16)
g = torch.zeros(I, H, device=dev)
ws = torch.empty(16 * I * H, device=dev)


def run():
    if nt:
        ops.gemm(dzT, aT, g, M=I, N=H, K=Mk, splitk=0, ws=ws)
    else:
        ops.gemm(dz, a, g, ta=True, tb=True, M=I, N=H, K=M, lda=I, ldb=H, splitk=0, ws=ws)


if "ring" in sys.argv[3:]:
    ops.set_gemm_variant(2, 24)
for _ in range(reps):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
e1.synchronize()
t = e0.elapsed_time(e1) / reps
print(f"dW1 {'NT' if nt else 'TN'} {I}x{H}x{M}: {t * 1e3:.1f} us per launch incl. split-K reduction = {2.0 * M * I * H / t / 1e9:.0f} TFLOP/s")
