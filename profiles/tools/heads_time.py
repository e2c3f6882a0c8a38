"""MLM head forward (transform + LayerNorm + tied decoder) and the loss kernel at the step's shape, f32 against f16 logits.
usage: python profiles/tools/heads_time.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import medvill_amd as mv
from medvill_amd import hip_ops as ops
dev = torch.device("cuda", 0)
cfg = mv.ModelConfig()
torch.manual_seed(1234)
model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=dev)
model.train()
step = mv.TrainStep(model, lr=1e-5, overlap_optimizer=True)
b = mv.data.synthetic_batch(cfg.vocab_size, 64, 36, 473, "full", seed=1234, device=dev)
step(b); step(b)
torch.cuda.synchronize()
eng = model.engine
S = eng.S
R = S["n_lab"]
xr, xr_b = S["hidden_f"][:R], S["hidden"][:R]
V = cfg.vocab_size
lab = b["label_ids"].to(dev)
def t_of(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for flag in (False, True):
    logits = eng._mlm_forward(xr, xr_b, R, "ht_", logits16=flag)
    Vp = logits.shape[1]
    dl = torch.empty((R, Vp), dtype=eng.adt, device=dev)
    stats = torch.zeros(6, device=dev)
    tf = t_of(lambda: eng._mlm_forward(xr, xr_b, R, "ht_", logits16=flag))
    tc = t_of(lambda: ops.ce_fwd_bwd(logits, Vp, lab, R, V, stats[0:3], dl, Vp, grad_scale=1.0 / R, loss_scale_dev=eng.loss_scale_dev))
    print(f"logits16={flag}: R={R} logits dtype {logits.dtype}: transform+LN+decoder {tf:.1f} us, CE {tc:.1f} us")
