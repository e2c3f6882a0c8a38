"""How the 16-bit path copes with outlier channels: logits / loss / gradient deviation from the engine's own
exact fp32 path for a residual-stream outlier of a given size and LayerNorm gains x30, with the encoder's LayerNorm inputs
stored f16 or fp32.  usage: python profiles/tools/outlier_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import medvill_amd as mv
from oracle import cxrbert_oracle as O
from oracle import synth

DEV = "cuda"
cfg = O.OracleConfig(vocab_size=2048, hidden=768, layers=2, heads=12, intermediate=3072, max_pos=256)
cd = dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
          intermediate_size=cfg.intermediate, max_position_embeddings=cfg.max_pos, layer_norm_eps=cfg.ln_eps)
B, N, S = 2, 16, 100
b = {k: torch.from_numpy(v) for k, v in synth.make_batch(cfg, B, N, S, "s2s", seed=13).items()}


def fwd(model):
    return model(b["cls_tok"].to(DEV), b["input_txt"].to(DEV), b["attn_mask"].to(DEV), b["segment"].to(DEV),
                 (b["img_feats"].to(DEV), b["img_pos"].to(DEV)), b["sep_tok"].to(DEV))


for outlier in (0.0, 30.0, 100.0, 300.0, 1000.0):
    P = O.make_params(cfg, seed=13)
    if outlier:
        P["enc.txt_embeddings.LayerNorm.bias"][37] = outlier
        for l in range(cfg.layers):
            for ln in ("attention.output.LayerNorm.weight", "output.LayerNorm.weight"):
                P[f"enc.encoder.layer.{l}.{ln}"][[5, 111, 300]] *= 30.0
    m32 = mv.CXRBERT(cd, None, dtype=torch.float32, device=DEV)
    m32.load_state_dict(P)
    m32.eval()
    with torch.no_grad():
        l32, _ = fwd(m32)
    s32 = mv.TrainStep(m32, lr=0.0)(dict(b), train=True).cpu()
    g32 = m32.engine.flat_g.clone()
    for gop in ("f16", "bf16"):
        for ln16 in ("1", "0"):
            os.environ["MV_LN_IN_16"] = ln16
            m = mv.CXRBERT(cd, None, dtype=torch.bfloat16, device=DEV, grad_operand=gop)
            m.load_state_dict(P)
            m.eval()
            with torch.no_grad():
                l16, _ = fwd(m)
            s16 = mv.TrainStep(m, lr=0.0)(dict(b), train=True).cpu()
            g16 = m.engine.flat_g
            worst = max(((float((m.engine.g[k] - m32.engine.g[k]).norm() / (m32.engine.g[k].norm() + 1e-12)), k) for k in m.engine.g), key=lambda t: t[0])
            print(f"outlier {outlier:6.0f} grad {gop:4s} ln_in_16={ln16}: logits max-abs {float((l16.float() - l32).abs().max()):.3e} (|logit| max {float(l32.abs().max()):.2f})"
                  f"  mlm loss {float(s16[0] / s16[1]):.4f} vs {float(s32[0] / s32[1]):.4f}  grad rel-L2 {float((g16 - g32).norm() / g32.norm()):.3e}"
                  f"  finite {bool(torch.isfinite(g16).all())}  worst tensor {worst[0]:.2e} {worst[1]}", flush=True)
