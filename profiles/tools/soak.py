#!/usr/bin/env python3
"""Soak: N fused training steps at the benchmark configuration (BERT-base, L = 512, B = 64, dropout 0.1) over a cycle of synthetic batches with
different lengths and labels, lr 5e-5.  Reports every 100 steps: MLM / ITM loss and accuracy of the block, the dynamic loss scale, steps
applied / skipped (f16 gradient operands), and at the end: parameters finite, wall time per step.       usage: soak.py [steps] [batches]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import medvill_amd as mv  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
cfg = mv.ModelConfig()
model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=dev)
model.reset_parameters(seed=7)
model.train()
step = mv.TrainStep(model, lr=5e-5)
fams = ["full", "mixed", "s2s", "full"]
batches = []
for i in range(nb):
    b = mv.data.synthetic_batch(cfg.vocab_size, 64, 36, 473, fams[i % len(fams)], seed=1000 + i, device=dev)
    b["attn_mask"] = None
    batches.append(b)
eng = model.engine
acc = torch.zeros(6, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(steps):
    acc += step(batches[s % nb], train=True)
    if (s + 1) % 100 == 0:
        a = acc.cpu()
        sc = eng.scaler.cpu() if eng.scaler is not None else None
        print(f"step {s + 1:5d}: mlm loss {a[0] / max(a[1], 1):7.4f} acc {a[2] / max(a[1], 1):6.4f} | itm loss {a[3] / a[4]:6.4f} acc {a[5] / a[4]:6.4f}"
              + ("" if sc is None else f" | loss scale {sc[0]:9.0f}, steps applied {int(sc[4])}, skipped {int(sc[5])}"), flush=True)
        acc.zero_()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps * 1e3
eng.wait_optimizer()
finite = bool(torch.isfinite(eng.flat_p).all())
print(f"{steps} steps over {nb} batches: {dt:.2f} ms per step (incl. a read-back every 100 steps); parameters finite: {finite}; "
      f"|p| max {float(eng.flat_p.abs().max()):.3f}")
sys.exit(0 if finite else 1)
