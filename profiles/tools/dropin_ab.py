#!/usr/bin/env python3
"""A/B of the model-API (drop-in) loop of train_origin.py:95-131 at BERT-base, L = 512:
    mlm, itm = model(...dense [B,L,L] mask on the device...); loss = mlm_itm_loss(mlm, itm, labels, aligned); loss.backward(); optimizer step
under model.lazy_logits = True, with the device mask recognition and the gradient views switched on / off, and with the engine's fused AdamW
against torch.optim.AdamW(fused=True) on the Parameters.      usage: dropin_ab.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import medvill_amd as mv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
cfg = mv.ModelConfig()
N, S = 36, 473
model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=dev)
model.reset_parameters(seed=1)
model.train()
model.lazy_logits = True
eng = model.engine
for B in (64, 16):
    bl = mv.data.synthetic_batch(cfg.vocab_size, B, N, S, "full", seed=998, device=dev)
    for rec, views, opt in ((False, False, "engine"), (True, False, "engine"), (False, True, "engine"), (True, True, "engine"), (True, True, "torch"),
                            (True, True, "engine+labels"), (True, True, "torch+labels"), (True, True, "mvopt"), (True, True, "mvopt+labels"),
                            (True, True, "mvovl"), (True, True, "mvovl+labels")):
        model.recognise_masks, model.grad_views = rec, views
        with_labels = opt.endswith("+labels")           # forward(..., txt_labels=labels): last layer on the consumed rows only
        opt = opt.split("+")[0]
        model.zero_grad()
        topt = torch.optim.AdamW(model.parameters(), lr=1e-5, fused=True) if opt == "torch" else None
        if opt in ("mvopt", "mvovl"):       # medvill_amd.optim.AdamW (the engine's kernel behind the torch optimizer protocol; mvovl: overlap=True)
            topt = mv.optim.AdamW(model.parameters(), lr=1e-5, overlap=(opt == "mvovl"))

        def one(t):
            mlm, itm = model(bl["cls_tok"], bl["input_txt"], bl["attn_mask"], bl["segment"], (bl["img_feats"], bl["img_pos"]), bl["sep_tok"],
                             **({"txt_labels": bl["txt_labels"]} if with_labels else {}))
            model.zero_grad()             # optim.zero_grad() of train_origin.py:129 (.grad = None)
            mv.losses.mlm_itm_loss(mlm, itm, bl["txt_labels"], bl["is_aligned"]).backward()
            if topt is not None:
                topt.step()
            else:
                eng.adamw_step(t, lr=1e-5)
        one(1)
        one(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(n):
            one(t + 3)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"B={B:3d} recognise_masks={int(rec)} grad_views={int(views)} optimizer={opt:6s} labels_in_forward={int(with_labels)}: {ms:7.2f} ms per step "
              f"({B / ms * 1e3:7.0f} pairs/s), packed={eng.S['cu'] is not None}", flush=True)
        del topt
