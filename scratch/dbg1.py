import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import medvill_amd as mv
from test_model_gpu import load_case, make_model, fwd
G = "tests/golden"
for name in ("c1_full", "c1v1k_full"):
    z, meta, cfg, P, b = load_case(G, name)
    for dtype in (torch.float32, torch.bfloat16):
        model = make_model(cfg, P, dtype)
        mlm, itm = fwd(model, b)
        ce_m, ce_i = torch.nn.CrossEntropyLoss(ignore_index=-100), torch.nn.CrossEntropyLoss()
        loss = ce_m(mlm.transpose(1, 2), b["txt_labels"].cuda()) + ce_i(itm, b["is_aligned"].cuda())
        loss.backward()
        names = [str(n) for n in z["grad_names"]]
        grads = dict(model.named_parameters())
        for k in ("enc.pooler.dense.weight", "enc.pooler.dense.bias", "itm.linear.weight"):
            i = names.index(k)
            g = grads[k].grad.float().cpu()
            got = g.reshape(-1)[torch.from_numpy(z["grad_idx"][i])].numpy()
            print(name, dtype, k, "norm", float(g.norm()), "ref", float(z["grad_norms"][i]))
            print("  got", np.array2string(got[:8], precision=5))
            print("  ref", np.array2string(z["grad_vals"][i][:8], precision=5))
