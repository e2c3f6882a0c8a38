"""CPU oracle for the MedViLL / CXRBERT pretraining hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The shipped path (``multi-modality-self-supervision_amd``) never imports it and
fails loudly when its HIP extension is missing.

What this file is: a plain-PyTorch fp32 restatement of the reference
algorithm, written from the reference's behaviour, not
copied from it.  Each function cites the reference file:line it follows
(paths relative to the upstream repo root):

* sequence assembly / embeddings ....... models/cxrbert_origin.py:22-35, 114-125
* mask extension (additive -10000) ..... models/cxrbert_origin.py:75-85
* encoder layer arithmetic ............. third-party HF ``transformers`` BertLayer
  (un-vendored, version unpinned: import paths imply 2.x <= v < 4.0; 5.15.0 is
  what is installed here).  Readable in-tree spec of the same math:
  Downstream_task/report_generation_and_vqa/sc/pytorch_pretrained_bert/model.py:261-432
  (with LayerNorm eps 1e-12 instead of that copy's 1e-5).
* pooler ............................... models/cxrbert_origin.py:130
* MLM head (tied decoder, LN eps 1e-5) . models/cxrbert_origin.py:189-238
* ITM head ............................. models/cxrbert_origin.py:164-173
* losses ............................... models/train_origin.py:62-63, 120-126
* step metrics ......................... models/train_origin.py:133-146
* HF AdamW (<=4.x semantics, restated) . models/train_origin.py:15, 60 (call site)

Parity pin: the reference ships no tests or golden vectors, so
this restatement is pinned against the reference itself, imported in the build
container through ``oracle/gen_golden.py`` (shims of SURVEY.md Appendix E).
The vectors it produced are committed under ``tests/golden/`` and
``tests/test_oracle_golden.py`` checks this file against them on every run.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, asdict

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- config
@dataclass
class OracleConfig:
    vocab_size: int = 30522
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    max_pos: int = 512
    type_vocab: int = 2
    img_hidden: int = 2048
    ln_eps: float = 1e-12        # HF bert-base config (embeddings + encoder LayerNorms)
    head_ln_eps: float = 1e-5    # cxrbert_origin.py:189-202,214 (TF-style LN of the MLM transform)
    img_position: bool = True    # args.img_postion (sic), cxrbert_origin.py:27-31: False -> image rows add no position embedding

    def to_dict(self):
        d = asdict(self)
        if d["img_position"]:    # the default is left out, so the metadata of the earlier fixtures regenerates byte for byte
            del d["img_position"]
        return d


CONFIGS = {
    # BASELINE.json configs[0]: 2-layer/2-head/hidden128, seq64 (16 regions + 48 text), batch 4
    "c1": OracleConfig(hidden=128, layers=2, heads=2, intermediate=512, max_pos=512),
    # BASELINE.json configs[1..3]: BERT-base, L=512 (36 regions + 476)
    "base": OracleConfig(),
    # BASELINE.json configs[4]: BERT-base, L=768 (100 regions + 668); text block 666 > 512 positions
    "base768": OracleConfig(max_pos=768),
}


# --------------------------------------------------------------------------- parameters
def param_shapes(cfg: OracleConfig) -> "OrderedDict[str, tuple]":
    """Canonical (alias-free) parameter list with the reference's state-dict names
    (SURVEY.md Appendix C).  Aliases (img_embeddings.position_embeddings ...,
    mlm.predictions.decoder.weight) are the same tensors and are not listed."""
    H, I, V = cfg.hidden, cfg.intermediate, cfg.vocab_size
    s = OrderedDict()
    e = "enc.txt_embeddings."
    s[e + "word_embeddings.weight"] = (V, H)
    s[e + "position_embeddings.weight"] = (cfg.max_pos, H)
    s[e + "token_type_embeddings.weight"] = (cfg.type_vocab, H)
    s[e + "LayerNorm.weight"] = (H,)
    s[e + "LayerNorm.bias"] = (H,)
    s["enc.img_embeddings.img_embeddings.weight"] = (H, cfg.img_hidden)
    s["enc.img_embeddings.img_embeddings.bias"] = (H,)
    for l in range(cfg.layers):
        p = f"enc.encoder.layer.{l}."
        for n in ("query", "key", "value"):
            s[p + f"attention.self.{n}.weight"] = (H, H)
            s[p + f"attention.self.{n}.bias"] = (H,)
        s[p + "attention.output.dense.weight"] = (H, H)
        s[p + "attention.output.dense.bias"] = (H,)
        s[p + "attention.output.LayerNorm.weight"] = (H,)
        s[p + "attention.output.LayerNorm.bias"] = (H,)
        s[p + "intermediate.dense.weight"] = (I, H)
        s[p + "intermediate.dense.bias"] = (I,)
        s[p + "output.dense.weight"] = (H, I)
        s[p + "output.dense.bias"] = (H,)
        s[p + "output.LayerNorm.weight"] = (H,)
        s[p + "output.LayerNorm.bias"] = (H,)
    s["enc.pooler.dense.weight"] = (H, H)
    s["enc.pooler.dense.bias"] = (H,)
    s["mlm.predictions.bias"] = (V,)
    s["mlm.predictions.transform.dense.weight"] = (H, H)
    s["mlm.predictions.transform.dense.bias"] = (H,)
    s["mlm.predictions.transform.LayerNorm.weight"] = (H,)
    s["mlm.predictions.transform.LayerNorm.bias"] = (H,)
    s["itm.linear.weight"] = (2, H)
    s["itm.linear.bias"] = (2,)
    return s


def num_params(cfg: OracleConfig) -> int:
    return sum(int(np.prod(v)) for v in param_shapes(cfg).values())


def splitmix_uniform(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n floats in [-1, 1) from a counter-based splitmix64 stream: exactly
    reproducible everywhere (pure uint64 arithmetic, 24 random bits per value)."""
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    top = (z >> np.uint64(40)).astype(np.float64)          # 24 bits
    return (top / float(1 << 23) - 1.0).astype(np.float32)


def make_params(cfg: OracleConfig, seed: int = 7, scale: float = 0.0346) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic fp32 parameters (uniform, std ~= 0.02 like BERT's N(0,0.02) init;
    LayerNorm gains near 1, all biases non-zero so that every term is exercised)."""
    out = OrderedDict()
    off = 0
    for k, (name, shp) in enumerate(param_shapes(cfg).items()):
        n = int(np.prod(shp))
        u = splitmix_uniform(seed * 1000003 + k, n)
        if name.endswith("LayerNorm.weight"):
            w = 1.0 + 0.1 * u
        elif name.endswith("bias"):
            w = 0.05 * u
        else:
            w = scale * u
        out[name] = torch.from_numpy(w.reshape(shp).copy())
        off += n
    return out


def expand_aliases(params: "dict[str, torch.Tensor]") -> "OrderedDict[str, torch.Tensor]":
    """Full reference state-dict (with the aliased names) from the canonical one."""
    sd = OrderedDict(params)
    e = "enc.txt_embeddings."
    i = "enc.img_embeddings."
    sd[i + "position_embeddings.weight"] = params[e + "position_embeddings.weight"]
    sd[i + "token_type_embeddings.weight"] = params[e + "token_type_embeddings.weight"]
    sd[i + "LayerNorm.weight"] = params[e + "LayerNorm.weight"]
    sd[i + "LayerNorm.bias"] = params[e + "LayerNorm.bias"]
    sd["mlm.predictions.decoder.weight"] = params[e + "word_embeddings.weight"]
    return sd


# --------------------------------------------------------------------------- arithmetic
PAD_TOKEN_ID = 0


def layer_norm(x, g, b, eps):
    """(x-mean)/sqrt(var_biased+eps)*g+b -- HF LayerNorm and the TF-style BertLayerNorm of
    cxrbert_origin.py:198-202 are the same formula (eps inside the sqrt)."""
    u = x.mean(-1, keepdim=True)
    s = (x - u).pow(2).mean(-1, keepdim=True)
    return (x - u) / torch.sqrt(s + eps) * g + b


def gelu_erf(x):
    """cxrbert_origin.py:176-181."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def extend_mask(attn_mask: torch.Tensor) -> torch.Tensor:
    """cxrbert_origin.py:75-85: [B,L,L] -> [B,1,L,L], [B,L] -> [B,1,1,L]; cast to fp16,
    (1-m)*-10000 (both 0 and -10000 are fp16-exact), then used additively in fp32."""
    if attn_mask.dim() == 2:
        m = attn_mask[:, None, None, :]
    elif attn_mask.dim() == 3:
        m = attn_mask[:, None, :, :]
    else:
        raise NotImplementedError
    m = m.to(torch.float16)
    return ((1.0 - m) * -10000.0).to(torch.float32)


def _drop(t, p_drop, training, masks, key):
    """Dropout as the reference applies it (nn.Dropout, p = 0.1 at every site in train mode).  `masks` lets a
    test inject explicit keep*scale tensors (the HIP kernels' counter-based masks) instead of torch's RNG."""
    if masks is not None:
        return t * masks[key]
    return F.dropout(t, p_drop, training) if (training and p_drop > 0) else t


def embed(P, cfg, cls_tok, input_txt, segment, img_feats, img_pos, sep_tok, p_drop=0.0, training=False, masks=None):
    """cxrbert_origin.py:114-125 + :22-35 + HF BertEmbeddings.  One shared set of
    position / type tables and one LayerNorm for all four calls (cxrbert_origin.py:17-20).
    cls and sep use position 0 and type 0; text positions restart at 0; image
    positions are the CNN feature-map indices."""
    e = "enc.txt_embeddings."
    E, Pos, Ty = P[e + "word_embeddings.weight"], P[e + "position_embeddings.weight"], P[e + "token_type_embeddings.weight"]
    g, b = P[e + "LayerNorm.weight"], P[e + "LayerNorm.bias"]
    Wi, bi = P["enc.img_embeddings.img_embeddings.weight"], P["enc.img_embeddings.img_embeddings.bias"]
    T = input_txt.shape[1]

    def ln_drop(x):
        x = layer_norm(x, g, b, cfg.ln_eps)
        return x if masks is not None else _drop(x, p_drop, training, None, None)

    def word(ids):
        # HF BertEmbeddings builds nn.Embedding(vocab, hidden, padding_idx=pad_token_id=0): the LOOK-UP gradient of
        # row 0 ([PAD]) is always zero (row 0 still gets the tied-decoder gradient, cxrbert_origin.py:231)
        return F.embedding(ids, E, padding_idx=PAD_TOKEN_ID)

    cls_o = ln_drop(word(cls_tok) + Ty[torch.zeros_like(cls_tok)] + Pos[:1][None])
    sep_o = ln_drop(word(sep_tok) + Ty[torch.zeros_like(sep_tok)] + Pos[:1][None])
    if cfg.img_position:         # cxrbert_origin.py:27-31
        img_o = ln_drop(F.linear(img_feats, Wi, bi) + Pos[img_pos] + Ty[0][None, None])
    else:
        img_o = ln_drop(F.linear(img_feats, Wi, bi) + Ty[0][None, None])
    txt_o = ln_drop(word(input_txt) + Ty[segment] + Pos[:T][None])
    x = torch.cat([cls_o, img_o, sep_o, txt_o], dim=1)
    return x * masks["emb"] if masks is not None else x


def encoder_layer(P, cfg, l, x, add_mask, p_drop=0.0, training=False, masks=None):
    """One BertLayer."""
    p = f"enc.encoder.layer.{l}."
    B, L, H = x.shape
    A, dh = cfg.heads, cfg.hidden // cfg.heads

    def drop(t, site=None):
        return _drop(t, p_drop, training, masks, (site, l))

    def heads(t):
        return t.view(B, L, A, dh).permute(0, 2, 1, 3)

    q = heads(F.linear(x, P[p + "attention.self.query.weight"], P[p + "attention.self.query.bias"]))
    k = heads(F.linear(x, P[p + "attention.self.key.weight"], P[p + "attention.self.key.bias"]))
    v = heads(F.linear(x, P[p + "attention.self.value.weight"], P[p + "attention.self.value.bias"]))
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh) + add_mask
    pr = drop(torch.softmax(s, dim=-1), "attn")
    c = torch.matmul(pr, v).permute(0, 2, 1, 3).reshape(B, L, H)
    a = layer_norm(drop(F.linear(c, P[p + "attention.output.dense.weight"], P[p + "attention.output.dense.bias"]), "out1") + x,
                   P[p + "attention.output.LayerNorm.weight"], P[p + "attention.output.LayerNorm.bias"], cfg.ln_eps)
    i = gelu_erf(F.linear(a, P[p + "intermediate.dense.weight"], P[p + "intermediate.dense.bias"]))
    o = layer_norm(drop(F.linear(i, P[p + "output.dense.weight"], P[p + "output.dense.bias"]), "out2") + a,
                   P[p + "output.LayerNorm.weight"], P[p + "output.LayerNorm.bias"], cfg.ln_eps)
    return o


def encode(P, cfg, cls_tok, input_txt, attn_mask, segment, img_feats, img_pos, sep_tok, p_drop=0.0, training=False, masks=None):
    """CXRBertEncoder.forward else-branch (cxrbert_origin.py:114-130): returns
    (last_hidden [B,L,H], pooled [B,H])."""
    add = extend_mask(attn_mask)
    x = embed(P, cfg, cls_tok, input_txt, segment, img_feats, img_pos, sep_tok, p_drop, training, masks)
    for l in range(cfg.layers):
        x = encoder_layer(P, cfg, l, x, add, p_drop, training, masks)
    pooled = torch.tanh(F.linear(x[:, 0], P["enc.pooler.dense.weight"], P["enc.pooler.dense.bias"]))
    return x, pooled


def mlm_transform(P, cfg, x):
    """BertPredictionHeadTransform (cxrbert_origin.py:205-218)."""
    t = gelu_erf(F.linear(x, P["mlm.predictions.transform.dense.weight"], P["mlm.predictions.transform.dense.bias"]))
    return layer_norm(t, P["mlm.predictions.transform.LayerNorm.weight"],
                      P["mlm.predictions.transform.LayerNorm.bias"], cfg.head_ln_eps)


def heads(P, cfg, x, pooled):
    """BertLMPredictionHead with the decoder tied to the word-embedding table
    (cxrbert_origin.py:221-238, tie at :141,:231) over ALL L positions, and
    ImageTextMatching (cxrbert_origin.py:164-173)."""
    t = mlm_transform(P, cfg, x)
    mlm = F.linear(t, P["enc.txt_embeddings.word_embeddings.weight"]) + P["mlm.predictions.bias"]
    itm = F.linear(pooled, P["itm.linear.weight"], P["itm.linear.bias"])
    return mlm, itm


def forward(P, cfg, cls_tok, input_txt, attn_mask, segment, img_feats, img_pos, sep_tok, p_drop=0.0, training=False, masks=None):
    """CXRBERT.forward (cxrbert_origin.py:144-149) -> (mlm_logits [B,L,V], itm_logits [B,2])."""
    x, pooled = encode(P, cfg, cls_tok, input_txt, attn_mask, segment, img_feats, img_pos, sep_tok, p_drop, training, masks)
    return heads(P, cfg, x, pooled)


def losses(mlm, itm, txt_labels, is_aligned):
    """train_origin.py:62-63,120-126: CE(mlm.transpose(1,2), labels; ignore_index=-100)
    = mean over labelled tokens of the whole batch; CE(itm, is_aligned) = mean over B."""
    mlm_loss = F.cross_entropy(mlm.transpose(1, 2), txt_labels, ignore_index=-100)
    itm_loss = F.cross_entropy(itm, is_aligned)
    return mlm_loss, itm_loss


def step_metrics(mlm, itm, txt_labels, is_aligned):
    """train_origin.py:133-146: ITM correct count; MLM correct / labelled counts."""
    itm_correct = int(itm.argmax(-1).eq(is_aligned).sum())
    lab = txt_labels != -100
    mlm_correct = int((mlm.argmax(-1).eq(txt_labels) & lab).sum())
    return itm_correct, mlm_correct, int(lab.sum())


# --------------------------------------------------------------------------- optimizer
def hf_adamw_step(p, g, m, v, t, lr=1e-5, b1=0.9, b2=0.999, eps=1e-6, wd=0.0, correct_bias=True):
    """HF ``transformers.optimization.AdamW`` (<=4.x; removed from 5.x so restated from its
    published algorithm).  ``eps`` is added to sqrt(v) BEFORE the bias
    correction is folded into the step size (differs from torch.optim.AdamW); weight decay
    is decoupled and applied after the update with plain lr.  Effective reference
    hyper-parameters (train_origin.py:60 passes only lr): lr=1e-5, betas=(0.9,0.999),
    eps=1e-6, wd=0, correct_bias=True.  In-place on p, m, v; t is the 1-based step."""
    m.mul_(b1).add_(g, alpha=1.0 - b1)
    v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
    denom = v.sqrt().add_(eps)
    step = lr
    if correct_bias:
        step = step * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    p.addcdiv_(m, denom, value=-step)
    if wd > 0.0:
        p.add_(p, alpha=-lr * wd)
    return p


def train_step(P, M, Vv, t, cfg, batch, lr=1e-5, p_drop=0.0, training=True):
    """The loop body of train_origin.py:95-131 on the oracle: forward, both CE
    losses, backward (autograd over the restatement), HF AdamW.  P: dict of leaf
    tensors with requires_grad; M, Vv: Adam moment dicts.  Returns python floats."""
    for w in P.values():
        w.grad = None
    mlm, itm = forward(P, cfg, batch["cls_tok"], batch["input_txt"], batch["attn_mask"], batch["segment"],
                       batch["img_feats"], batch["img_pos"], batch["sep_tok"], p_drop, training)
    mlm_loss, itm_loss = losses(mlm, itm, batch["txt_labels"], batch["is_aligned"])
    loss = itm_loss + mlm_loss
    loss.backward()
    with torch.no_grad():
        for k, w in P.items():
            hf_adamw_step(w, w.grad, M[k], Vv[k], t, lr=lr)
    return float(loss), float(mlm_loss), float(itm_loss)


# --------------------------------------------------------------------------- FLOPs
def flops_fwd_per_sample(cfg: OracleConfig, L: int, N: int) -> float:
    H, I, V, D = cfg.hidden, cfg.intermediate, cfg.vocab_size, cfg.img_hidden
    return (2.0 * N * D * H + cfg.layers * (L * (8.0 * H * H + 4.0 * H * I) + 4.0 * L * L * H)
            + 2.0 * H * H + L * (2.0 * H * H + 2.0 * H * V) + 4.0 * H)
