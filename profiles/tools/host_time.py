"""How far ahead of the GPU is the host?  Enqueue time (Python + ctypes + launch calls) per training step against the GPU's
time per step at the bench shape.  usage: python profiles/tools/host_time.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import medvill_amd as mv

dev = torch.device("cuda", 0)
cfg = mv.ModelConfig()
torch.manual_seed(1)
model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=dev)
model.train()
step = mv.TrainStep(model, lr=1e-5, overlap_optimizer=True)
batches = [mv.data.synthetic_batch(cfg.vocab_size, 64, 36, 473, "full", seed=10 + i, device=dev) for i in range(4)]
for i in range(5):
    step(batches[i % 4])
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for i in range(n):
    step(batches[i % 4])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step; GPU {1e3 * (t2 - t0) / n:.2f} ms/step; host idle at the end {1e3 * (t2 - t1):.1f} ms")
