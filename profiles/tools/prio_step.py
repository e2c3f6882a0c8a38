"""Whole-step A/B of stream priorities and of the weight-gradient kernel form (persistent / one tile per block), one process.
The main chain runs on a stream of the given priority; the engine's side stream gets the other one.  usage: python profiles/tools/prio_step.py"""
import os
import statistics
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import medvill_amd as mv
from medvill_amd import hip_ops as ops

dev = torch.device("cuda", 0)
print("priority range (least, greatest):", torch.cuda.Stream.priority_range())
cfg = mv.ModelConfig()
torch.manual_seed(1234)
model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=dev)
model.train()
step = mv.TrainStep(model, lr=1e-5, overlap_optimizer=True)
B, N, S = 64, 36, 473
batches = [mv.data.synthetic_batch(cfg.vocab_size, B, N, S, "full", seed=1234 + 1000 * i, device=dev) for i in range(4)]
eng = model.engine


def timed(stream, n=8):
    with torch.cuda.stream(stream):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            step(batches[i % 4])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3


lo, hi = torch.cuda.Stream.priority_range()
ARMS = []
for name, main_p, side_p, dw in (("main default stream, side normal, persistent dW (as shipped)", None, 0, 0),
                                 ("main HIGH, side normal, persistent dW", hi, 0, 0),
                                 ("main HIGH, side LOW, persistent dW", hi, lo, 0),
                                 ("main HIGH, side LOW, one-tile-per-block dW", hi, lo, 14),
                                 ("main default, side normal, one-tile-per-block dW", None, 0, 14)):
    ARMS.append((name, main_p, side_p, dw))
res = {a[0]: [] for a in ARMS}
streams = {}
for r in range(4):
    for name, main_p, side_p, dw in ARMS:
        ms = torch.cuda.current_stream() if main_p is None else streams.setdefault(("m", main_p), torch.cuda.Stream(device=dev, priority=main_p))
        torch.cuda.synchronize()
        eng._side = streams.setdefault(("s", side_p), torch.cuda.Stream(device=dev, priority=side_p))
        ops.set_gemm_variant(0, dw)
        with torch.cuda.stream(ms):
            step(batches[0])
        res[name].append(timed(ms))
ops.set_gemm_variant(0, 0)
for n, v in res.items():
    print(f"{n:64s} median {statistics.median(v):.2f} ms  (min {min(v):.2f}, max {max(v):.2f})")
