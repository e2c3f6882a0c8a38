"""Soak run of the fused training step at the bench shape (BERT-base, B=64, L=512, ragged full masks, dropout on): N steps over 8
rotating synthetic batches with a real learning rate; prints the losses every 50 steps and fails on a non-finite value or a loss
that does not fall.  Races in the hand-ordered LDS-DMA rings would show up here as sporadic garbage.  usage: soak.py [steps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import medvill_amd as mv

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda", 0)
dist_on = os.environ.get("MV_DP_FORCE") == "1" and "RANK" in os.environ       # one-rank RCCL group, every collective issued
if dist_on:
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", device_id=dev)
cfg = mv.ModelConfig()
torch.manual_seed(7)
model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=dev)
model.train()
step = mv.TrainStep(model, lr=5e-5, distributed=dist_on, overlap_optimizer=True)
B, N, S = 64, 36, 473
batches = [mv.data.synthetic_batch(cfg.vocab_size, B, N, S, "full" if i % 2 == 0 else "mixed", seed=100 + i, device=dev) for i in range(8)]
hist = []
t0 = time.perf_counter()
for i in range(steps):
    st = step(batches[i % 8])
    if i % 50 == 49 or i == 0:
        s = st.double().cpu()
        mlm, itm = float(s[0] / s[1]), float(s[3] / s[4])
        hist.append(mlm)
        print(f"step {i + 1:5d}: mlm {mlm:.4f} (acc {float(s[2] / s[1]):.3f})  itm {itm:.4f}  [{(time.perf_counter() - t0):.1f} s]", flush=True)
        assert all(map(lambda v: v == v and abs(v) < 1e4, (mlm, itm))), "non-finite loss"
step.sync()
p = model.engine.flat_p
assert bool(torch.isfinite(p).all()), "non-finite parameter"
assert hist[-1] < hist[0] - 1.0, f"loss did not fall: {hist}"
print("ok: parameters finite, mlm loss", hist[0], "->", hist[-1], "(RCCL path)" if dist_on else "")
if dist_on:
    torch.distributed.destroy_process_group()
