#!/usr/bin/env python3
"""Where a step of the model-API (drop-in) loop goes: forward / loss / backward / optimizer, each closed with a device synchronisation
(GPU time of the phase incl. its launch latency), beside the unsynchronised loop and the host-only time of each phase (time until the
call returns).      usage: dropin_phases.py [B] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import medvill_amd as mv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
cfg = mv.ModelConfig()
N, S = 36, 473
model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=dev)
model.reset_parameters(seed=1)
model.train()
model.lazy_logits = True
eng = model.engine
bl = mv.data.synthetic_batch(cfg.vocab_size, B, N, S, "full", seed=998, device=dev)
sync = torch.cuda.synchronize


def phases(t, synced, acc):
    def mark(name, t0):
        ret = time.perf_counter()
        if synced:
            sync()
        end = time.perf_counter()
        acc.setdefault(name, [0.0, 0.0])
        acc[name][0] += (ret - t0) * 1e3
        acc[name][1] += (end - t0) * 1e3
        return time.perf_counter()
    model.zero_grad()                 # optim.zero_grad() of train_origin.py:129 (.grad = None)
    t0 = time.perf_counter()
    mlm, itm = model(bl["cls_tok"], bl["input_txt"], bl["attn_mask"], bl["segment"], (bl["img_feats"], bl["img_pos"]), bl["sep_tok"])
    t0 = mark("forward", t0)
    loss = mv.losses.mlm_itm_loss(mlm, itm, bl["txt_labels"], bl["is_aligned"])
    t0 = mark("loss", t0)
    loss.backward()
    t0 = mark("backward", t0)
    eng.adamw_step(t, lr=1e-5)
    t0 = mark("optimizer", t0)


only_free = len(sys.argv) > 3 and sys.argv[3] == "free"        # (for a kernel trace of the free-running loop alone)
for synced in ((False,) if only_free else (False, True)):
    acc = {}
    phases(1, synced, {})
    phases(2, synced, {})
    sync()
    w0 = time.perf_counter()
    for t in range(n):
        phases(t + 3, synced, acc)
    sync()
    tot = (time.perf_counter() - w0) / n * 1e3
    print(f"B={B} {'synchronised after every phase' if synced else 'free-running'}: {tot:.2f} ms per step")
    for k, (h, e) in acc.items():
        print(f"    {k:10s} call returns after {h / n:7.2f} ms" + (f", device done after {e / n:7.2f} ms" if synced else ""))
if only_free:
    raise SystemExit(0)
# the fused step on the same batch, for scale
step = mv.TrainStep(model, lr=1e-5)
batch = dict(bl)
batch["attn_mask"] = None
for _ in range(3):
    step(batch, train=True)
sync()
w0 = time.perf_counter()
for _ in range(n):
    step(batch, train=True)
sync()
print(f"B={B} TrainStep on the same batch: {(time.perf_counter() - w0) / n * 1e3:.2f} ms per step")
