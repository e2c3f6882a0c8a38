"""Is the host ahead of the GPU?  Per step: when the step call RETURNS on the host against when the GPU finishes that step.
A per-step lead near zero means something blocks the launch thread until the device has caught up.  usage: python profiles/tools/host_lead.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import medvill_amd as mv

dev = torch.device("cuda", 0)
dist_on = os.environ.get("MV_DP_FORCE") == "1"           # one-rank RCCL group, every collective issued (MV_DP_FORCE=1 python host_lead.py)
if dist_on:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", device_id=dev)
cfg = mv.ModelConfig()
torch.manual_seed(1234)
model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=dev)
model.train()
step = mv.TrainStep(model, lr=1e-5, overlap_optimizer=True, distributed=dist_on)
B, N, S = 64, 36, 473
batches = [mv.data.synthetic_batch(cfg.vocab_size, B, N, S, "full", seed=1234 + 1000 * i, device=dev) for i in range(4)]
for i in range(4):
    step(batches[i % 4])
torch.cuda.synchronize()
n = 12
e0 = torch.cuda.Event(enable_timing=True)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
e0.record()
t0 = time.perf_counter()
ret = []
for i in range(n):
    step(batches[i % 4])
    evs[i].record()
    ret.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
for i in range(n):
    g = e0.elapsed_time(evs[i])
    print(f"step {i:2d}: host returned at {ret[i]:7.2f} ms, GPU finished at {g:7.2f} ms, lead {g - ret[i]:7.2f} ms")
