"""Cross-entropy over the labelled rows' logits (loss + argmax + gradient in one pass) at the step's shape.  usage: python profiles/tools/ce_bench.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from medvill_amd import hip_ops as ops
dev = "cuda"
R, V = 3400, 30522
Vp = (V + 7) // 8 * 8
logits = torch.randn(R, Vp, device=dev) * 3
labels = torch.randint(0, V, (R,), device=dev, dtype=torch.int32)
out = torch.zeros(3, device=dev)
dl = torch.empty(R, Vp, device=dev, dtype=torch.float16)
ls = torch.tensor([32768.0], device=dev)
fn = lambda: ops.ce_fwd_bwd(logits, Vp, labels, R, V, out, dl, Vp, grad_scale=1.0 / R, loss_scale_dev=ls)
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fn()
e1.record()
e1.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e3
print(f"ce fwd+bwd R={R} V={V} f32 -> f16: {t:7.1f} us  {(R * Vp * 6) / t / 1e6:5.2f} TB/s")
