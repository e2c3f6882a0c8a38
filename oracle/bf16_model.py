"""CPU model of WHERE the product's bf16 path rounds, for error attribution.

TEST INFRASTRUCTURE ONLY (see ``oracle/cxrbert_oracle.py`` header): never imported by the product.

The fp32 oracle is the reference; this file re-runs the same arithmetic with ``bf16(x)`` inserted at the
points where the HIP engine stores or feeds bf16 (engine.py: GEMM operands, stored activations, the
residual operand, the embedding tables), each point behind a switch, so the contribution of every rounding
site to the final logit error can be measured on the CPU (B=1, BERT-base, L=512 takes seconds) before a
kernel is changed.  ``python -m oracle.bf16_model`` prints the attribution table that is committed as
profiles/r02_bf16_error.txt.
"""
from __future__ import annotations

import math
import sys

import numpy as np
import torch
import torch.nn.functional as F

from . import cxrbert_oracle as O
from . import synth

SITES = ("w", "tables", "feats", "imgproj", "x_op", "x_res", "qkv", "p", "ctx", "a_op", "a_res", "act", "head_x", "t", "decoder_w")
# not a product rounding site (the LayerNorm inputs stay fp32); switchable to price storing them in 16 bits
EXTRA_SITES = ("pre",)


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def h16(t):
    return t.to(torch.float16).to(torch.float32)


def split2(t):
    """hi + lo bf16 split (what a two-pass MFMA sees): exact to ~2^-17 relative."""
    hi = bf(t)
    return hi + bf(t - hi)


class Rounder:
    """on: sites rounded to bf16; split: sites kept as hi+lo bf16 pairs; half: sites rounded to f16."""

    def __init__(self, on, split=(), half=()):
        self.on, self.split, self.half = set(on), set(split), set(half)

    def __call__(self, site, t):
        if site in self.split:
            return split2(t)
        if site in self.half:
            return h16(t)
        return bf(t) if site in self.on else t


def forward(P, cfg, b, R: Rounder):
    """Same arithmetic as cxrbert_oracle.forward (eval mode), with the product's rounding sites."""
    e = "enc.txt_embeddings."
    E, Pos, Ty = (R("tables", P[e + k]) for k in ("word_embeddings.weight", "position_embeddings.weight", "token_type_embeddings.weight"))
    g, bb = P[e + "LayerNorm.weight"], P[e + "LayerNorm.bias"]
    T = b["input_txt"].shape[1]
    ln = lambda x: O.layer_norm(x, g, bb, cfg.ln_eps)
    imgp = R("imgproj", F.linear(R("feats", b["img_feats"]), R("w", P["enc.img_embeddings.img_embeddings.weight"]),
                                 P["enc.img_embeddings.img_embeddings.bias"]))
    cls_o = ln(E[b["cls_tok"]] + Ty[torch.zeros_like(b["cls_tok"])] + Pos[:1][None])
    sep_o = ln(E[b["sep_tok"]] + Ty[torch.zeros_like(b["sep_tok"])] + Pos[:1][None])
    img_o = ln(imgp + Pos[b["img_pos"]] + Ty[0][None, None])
    txt_o = ln(E[b["input_txt"]] + Ty[b["segment"]] + Pos[:T][None])
    x = torch.cat([cls_o, img_o, sep_o, txt_o], dim=1)
    add = O.extend_mask(b["attn_mask"])
    B, L, H = x.shape
    A, dh = cfg.heads, cfg.hidden // cfg.heads
    hd = lambda t: t.view(B, L, A, dh).permute(0, 2, 1, 3)
    for l in range(cfg.layers):
        p = f"enc.encoder.layer.{l}."
        W = lambda n: R("w", P[p + n + ".weight"])
        Bi = lambda n: P[p + n + ".bias"]
        xo, xr = R("x_op", x), R("x_res", x)
        q = hd(R("qkv", F.linear(xo, W("attention.self.query"), Bi("attention.self.query"))))
        k = hd(R("qkv", F.linear(xo, W("attention.self.key"), Bi("attention.self.key"))))
        v = hd(R("qkv", F.linear(xo, W("attention.self.value"), Bi("attention.self.value"))))
        s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh) + add
        pr = torch.softmax(s, dim=-1)
        # the kernel rounds exp(s - m) to bf16 for the PV MFMA and divides by the UNROUNDED row sum afterwards
        c = R("ctx", torch.matmul(R("p", pr), v).permute(0, 2, 1, 3).reshape(B, L, H))
        a = O.layer_norm(R("pre", F.linear(c, W("attention.output.dense"), Bi("attention.output.dense")) + xr),
                         P[p + "attention.output.LayerNorm.weight"], P[p + "attention.output.LayerNorm.bias"], cfg.ln_eps)
        ao, ar = R("a_op", a), R("a_res", a)
        i = R("act", O.gelu_erf(F.linear(ao, W("intermediate.dense"), Bi("intermediate.dense"))))
        x = O.layer_norm(R("pre", F.linear(i, W("output.dense"), Bi("output.dense")) + ar),
                         P[p + "output.LayerNorm.weight"], P[p + "output.LayerNorm.bias"], cfg.ln_eps)
    pooled = torch.tanh(F.linear(R("x_op", x[:, 0]), R("w", P["enc.pooler.dense.weight"]), P["enc.pooler.dense.bias"]))
    t = O.gelu_erf(F.linear(R("head_x", x), R("w", P["mlm.predictions.transform.dense.weight"]),
                            P["mlm.predictions.transform.dense.bias"]))
    t = O.layer_norm(t, P["mlm.predictions.transform.LayerNorm.weight"], P["mlm.predictions.transform.LayerNorm.bias"], cfg.head_ln_eps)
    mlm = F.linear(R("t", t), R("decoder_w", P[e + "word_embeddings.weight"])) + P["mlm.predictions.bias"]
    itm = F.linear(R("x_op", pooled), R("w", P["itm.linear.weight"]), P["itm.linear.bias"])
    return mlm, itm, x


def case(cfg_name="base", B=1, N=36, S=473, family="s2s", seed=21):
    cfg = O.CONFIGS[cfg_name]
    P = O.make_params(cfg, seed=seed)
    b = {k: torch.from_numpy(v) for k, v in synth.make_batch(cfg, B, N, S, family, seed=seed).items()}
    return cfg, P, b


def stats(d):
    d = d.abs().flatten()
    k = max(1, d.numel() // 1000)
    return float(d.max()), float(d.mean()), float(d.topk(k).values.min())


def main(argv):
    torch.set_num_threads(8)
    cfg, P, b = case()
    with torch.no_grad():
        ref, ref_itm, ref_x = forward(P, cfg, b, Rounder(()))
        rows = []

        def run(name, on, split=(), half=()):
            mlm, itm, x = forward(P, cfg, b, Rounder(on, split, half))
            mx, mean, p999 = stats(mlm - ref)
            hx = float((x - ref_x).abs().max())
            rows.append((name, mx, mean, p999, hx, float((itm - ref_itm).abs().max())))
            print(f"{name:58s} logits max {mx:.3e} mean {mean:.3e} p99.9 {p999:.3e} | hidden max {hx:.3e} | itm {rows[-1][5]:.1e}", flush=True)

        allsites = set(SITES)
        run("all product rounding sites (round-1 engine)", allsites)
        for s in SITES:
            run(f"only '{s}'", {s})
        run("all, residual operands fp32 (x_res, a_res off)", allsites - {"x_res", "a_res"})
        run("all, res fp32 + tables fp32", allsites - {"x_res", "a_res", "tables"})
        run("all, res fp32 + head split (head_x,t,decoder_w hi+lo)", allsites - {"x_res", "a_res"}, {"head_x", "t", "decoder_w"})
        run("all, res fp32 + tables fp32 + head split", allsites - {"x_res", "a_res", "tables"}, {"head_x", "t", "decoder_w"})
        run("encoder exact, head bf16 only", {"head_x", "t", "decoder_w", "w"} - {"w"} | {"head_x", "t", "decoder_w"})
        run("encoder GEMM operands only (w, x_op, a_op, act, ctx)", {"w", "x_op", "a_op", "act", "ctx"})
        run("encoder: activations split, weights bf16", allsites - {"x_res", "a_res"}, {"x_op", "a_op", "act", "ctx", "qkv", "p", "head_x", "t"})
        run("encoder: weights split, activations bf16", allsites - {"x_res", "a_res"}, {"w", "decoder_w", "tables"})
        run("f16 forward operands, f16 residual operand (round-2 default)", (), (), allsites)
        run("f16 forward operands, fp32 residual operand", (), (), allsites - {"x_res", "a_res"})
        if "pre" in argv:
            run("f16 forward operands + f16 LayerNorm inputs (not adopted)", (), (), allsites | {"pre"})
            run("only 'pre' in f16", (), (), {"pre"})
    print(f"logit std {float(ref.std()):.3f} abs-max {float(ref.abs().max()):.2f}")


if __name__ == "__main__":
    main(sys.argv[1:])
