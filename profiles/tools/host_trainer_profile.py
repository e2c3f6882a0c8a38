"""Where a CXRBERT_Trainer step over HOST batches (the reference Dataset's 9-tuples, int64 [B,L,L] masks) spends its time: wall time of
the batch preparation (mask recognition, pinned uploads) and of the step call (kernel launches), against the device-resident step.
usage: python profiles/tools/host_trainer_profile.py [family]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import contextlib
import torch
import medvill_amd as mv
from types import SimpleNamespace
fam = sys.argv[1] if len(sys.argv) > 1 else "full"
dev = torch.device("cuda", 0)
cfg = mv.ModelConfig()
B, N, S = 64, 36, 473
args_t = SimpleNamespace(with_cuda=True, weight_load=False, bert_model="bert-base-scratch", lr=1e-5, log_freq=10, mlm_task=True, itm_task=True,
                         cuda_devices=None, dropout_prob=0.1)
host = []
for i in range(2):
    b = mv.data.synthetic_batch(cfg.vocab_size, B, N, S, fam, seed=4321 + i, device="cpu")
    host.append((b["cls_tok"], b["input_txt"], b["txt_labels"], b["attn_mask"], (b["img_feats"], b["img_pos"]), b["segment"], b["is_aligned"],
                 b["sep_tok"], torch.zeros(B)))
with contextlib.redirect_stdout(sys.stderr):
    tr = mv.CXRBERT_Trainer(args_t, host, None, config=cfg)
    tr.model.train()
    tr._run_epoch(host * 2, 0, True)
    torch.cuda.synchronize()
    acc = {"to_batch": 0.0, "recognise": 0.0, "step": 0.0}
    o_tb, o_rec, o_step = tr._to_batch, tr._recognise_masks, tr.step

    def wrap(name, f):
        def g(*a, **k):
            t0 = time.perf_counter()
            r = f(*a, **k)
            acc[name] += time.perf_counter() - t0
            return r
        return g
    tr._to_batch, tr._recognise_masks = wrap("to_batch", o_tb), wrap("recognise", o_rec)
    class _Timed:
        def __init__(self, inner):
            self.inner = inner

        def __call__(self, *a, **k):
            t0 = time.perf_counter()
            r = self.inner(*a, **k)
            acc["step"] += time.perf_counter() - t0
            return r

        def __getattr__(self, n):
            return getattr(self.inner, n)
    tr.step = _Timed(o_step)
    n = 12
    t0 = time.perf_counter()
    tr._run_epoch(host * (n // 2), 0, True)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / n * 1e3
# the same process, device-resident batches straight into the step (what bench.py's headline times)
res = [mv.data.synthetic_batch(cfg.vocab_size, B, N, S, fam, seed=4321 + i, device=dev) for i in range(2)]
for i in range(4):
    o_step(res[i % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
th = 0.0
for i in range(n):
    t1 = time.perf_counter()
    o_step(res[i % 2])
    th += time.perf_counter() - t1
torch.cuda.synchronize()
print(f"resident batches, same process: {(time.perf_counter() - t0) / n * 1e3:.1f} ms per step, host wall in the step call {th / n * 1e3:.1f} ms")
print(f"{fam}: {tot:.1f} ms per step over host batches; host wall per step: batch preparation {acc['to_batch'] / n * 1e3:.1f} ms "
      f"(of it mask recognition {acc['recognise'] / n * 1e3:.1f}), step call (launches) {acc['step'] / n * 1e3:.1f} ms")
