import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_model_gpu as T
gd = "/root/repo/tests/golden"
for name in ("base_s2s", "base_full"):
    z, meta, cfg, P, b = T.load_case(gd, name)
    cols = torch.from_numpy(z["cols"].astype(np.int64))
    model = T.make_model(cfg, P, torch.bfloat16)
    with torch.no_grad():
        mlm, itm = T.fwd(model, b)
    mlm = mlm.float().cpu()
    d = np.abs(mlm[..., cols].numpy() - z["logits_cols"])
    am = mlm.argmax(-1).numpy()
    mism = np.argwhere(am != z["argmax"])
    print(name, os.environ.get("MV_LIB_PATH", "default")[-24:], "max-abs", d.max(), "mean", d.mean(), "argmax agree", (am == z["argmax"]).mean(), "mismatches", len(mism))
    for (bb, i) in mism[:6]:
        row = mlm[bb, i]
        top2 = torch.topk(row, 2)
        print("   pos", bb, i, "top2 gap", float(top2.values[0] - top2.values[1]), "ref argmax", int(z["argmax"][bb, i]), "ours", int(am[bb, i]), "logit(ref idx)-max", float(row[int(z["argmax"][bb, i])] - top2.values[0]))
