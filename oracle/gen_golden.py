"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (it reads
/root/reference, which does not exist on the GPU box); the .npz files it writes
are what travels.  Nothing from the reference is copied: the reference modules
are imported in place, through the shims of SURVEY.md Appendix E
(stub torchvision / wandb / fuzzywuzzy, alias the legacy ``transformers.modeling_*``
module paths onto the installed transformers 5.x, offline ``BertConfig``,
identity ``Tensor.cuda``, pass-through region encoder, legacy-tuple encoder
wrapper, fake tokenizer / PIL for the Dataset).

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz

Fixtures:
  masks.npz      packed-bit [L,L] masks + ids/labels/segment from CXRDataset.__getitem__
                 for every mask family over a small (N,S,len) grid, and random_word KATs
  c1_<fam>.npz   config C1 (2L/2H/128, L=64, B=4): hidden, pooled, ITM logits, MLM-logit
                 summaries, both losses, per-parameter gradient norms + sampled entries
  c1v1k_full.npz same model with V=1024: full logits
  c1v1k_nopos.npz  the same with args.img_postion = False (no position embedding on the image rows), full logits + gradients
  base_s2s.npz   BERT-base L=512 B=1: losses, ITM logits, logit summaries
  base_full.npz  BERT-base L=512 B=2 ragged, bidirectional (BASELINE config 2's family) + gradients of every parameter
  base_full_b4.npz   the same at B=4 (the benchmarked path's form: the samples pack), + the LABELLED rows' logits (1,009 sampled columns of
                 each, the logit at the label; logsumexp / arg-max / maximum over all columns for every position)
  base_noncross.npz  BERT-base L=512 B=1, non-cross modality mask (config 4; n2 = 38 is not tile-aligned)
  base768_s2s.npz    BERT-base L=768 (100 regions + 665 text, max_position_embeddings 768) B=1, seq2seq (config 5)

    python oracle/gen_golden.py --only base_full,base_noncross     # regenerate a subset
  adamw.npz      3-step HF-AdamW known-answer test computed with python floats
  state_manifest.json  key -> [shape, dtype] of the reference CXRBERT.state_dict() at BERT-base (`--only manifest`)
"""
from __future__ import annotations

import importlib.util
import json
import math
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("MEDVILL_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import cxrbert_oracle as O      # noqa: E402
from oracle import data_oracle as D         # noqa: E402
from oracle import synth                    # noqa: E402


# --------------------------------------------------------------------------- shims (Appendix E)
def install_shims():
    import transformers
    from transformers.models.bert import modeling_bert as mb

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    tv = stub("torchvision")
    tv.models = stub("torchvision.models")
    tv.transforms = stub("torchvision.transforms")
    stub("wandb", init=lambda *a, **k: None, watch=lambda *a, **k: None, log=lambda *a, **k: None)
    fz = stub("fuzzywuzzy")
    fz.fuzz = stub("fuzzywuzzy.fuzz", token_sort_ratio=lambda a, b: 100 if a == b else 0)

    class OfflineBertConfig(transformers.BertConfig):
        _local = {}

        @classmethod
        def from_pretrained(cls, name, *a, **k):
            c = transformers.BertConfig(**cls._local)
            c._attn_implementation = "eager"
            return c

    stub("transformers.modeling_bert", BertConfig=OfflineBertConfig, BertModel=mb.BertModel,
         BertPreTrainedModel=mb.BertPreTrainedModel)
    stub("transformers.modeling_auto", AutoModel=transformers.AutoModel, AutoConfig=transformers.AutoConfig)
    stub("transformers.modeling_albert", AlbertModel=transformers.AlbertModel)
    stub("transformers.tokenization_albert", AlbertTokenizer=object)
    torch.Tensor.cuda = lambda self, *a, **k: self

    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    image = load("models.image", os.path.join(REF, "models", "image.py"))

    class PassThroughRegions(torch.nn.Module):
        def __init__(self, args):
            super().__init__()

        def forward(self, x):
            return x  # (feats[B,N,2048], pos[B,N]) handed in as ``input_img``

    image.ImageEncoder_cnn = PassThroughRegions
    cx = load("models.cxrbert_origin", os.path.join(REF, "models", "cxrbert_origin.py"))
    return cx, OfflineBertConfig


def build_reference_model(cx, CfgCls, cfg: O.OracleConfig, N: int, params, manifest=None):
    CfgCls._local = dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
                         num_attention_heads=cfg.heads, intermediate_size=cfg.intermediate,
                         max_position_embeddings=cfg.max_pos, hidden_dropout_prob=0.1,
                         attention_probs_dropout_prob=0.1, layer_norm_eps=cfg.ln_eps)
    config = CfgCls.from_pretrained("bert-base-uncased")
    args = types.SimpleNamespace(bert_model="bert-base-scratch", img_hidden_sz=cfg.img_hidden,
                                 embedding_size=cfg.hidden, hidden_size=cfg.hidden, dropout_prob=0.1,
                                 img_postion=cfg.img_position, img_encoder="random-pixel", img_size=512,
                                 num_image_embeds=N, disturbing_mask=False, vocab_size=cfg.vocab_size)
    model = cx.CXRBERT(config, args)
    if manifest is not None:      # the reference's own state_dict(), before anything here touches the module tree
        manifest.update({k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()})
    inner = model.enc.encoder

    class LegacyTuple(torch.nn.Module):
        def __init__(self, enc):
            super().__init__()
            self.enc = enc

        def forward(self, x, mask, output_hidden_states=False, output_attentions=False):
            out = self.enc(x, attention_mask=mask)
            return (out.last_hidden_state, None)

    model.enc.encoder = LegacyTuple(inner)
    sd = model.state_dict()
    full = O.expand_aliases(params)
    remap = {}
    for k in sd:
        kk = k.replace("enc.encoder.enc.", "enc.encoder.")
        if kk in full:
            remap[k] = full[kk].clone()
        elif "position_ids" in kk or "token_type_ids" in kk:
            remap[k] = sd[k]
        else:
            raise KeyError(f"reference state-dict key without oracle counterpart: {k}")
    missing = set(full) - {k.replace("enc.encoder.enc.", "enc.encoder.") for k in sd}
    assert not missing, missing
    model.load_state_dict(remap)
    # tie check (cxrbert_origin.py:141,231)
    assert model.mlm.predictions.decoder.weight.data_ptr() == model.enc.txt_embeddings.word_embeddings.weight.data_ptr()
    model.eval()
    return model


def ref_named_grads(model):
    out = {}
    for k, p in model.named_parameters():
        kk = k.replace("enc.encoder.enc.", "enc.encoder.")
        out[kk] = p.grad.detach().clone() if p.grad is not None else None
    return out


def logits_summary(mlm: torch.Tensor, cols: np.ndarray):
    lse = torch.logsumexp(mlm.double(), dim=-1).float().numpy()
    sub = mlm[..., torch.from_numpy(cols)].numpy()
    amax = mlm.argmax(-1).numpy().astype(np.int32)
    vmax = mlm.max(-1).values.numpy()
    return dict(lse=lse, cols=cols.astype(np.int32), logits_cols=sub, argmax=amax, maxval=vmax)


def grad_summary(grads: dict, shapes, seed=99):
    names, norms, idx, vals = [], [], [], []
    for k, shp in shapes.items():
        g = grads[k]
        n = g.numel()
        ii = (np.abs(O.splitmix_uniform(seed + len(names), 16)) * n).astype(np.int64) % n
        names.append(k)
        norms.append(float(g.double().norm()))
        idx.append(ii)
        vals.append(g.reshape(-1)[torch.from_numpy(ii)].numpy())
    return dict(grad_names=np.array(names), grad_norms=np.array(norms, np.float64),
                grad_idx=np.stack(idx), grad_vals=np.stack(vals))


def run_case(cx, CfgCls, cfg, B, N, S, family, seed, with_grads, full_logits=False, store_hidden=True, ncols=64, lab_ncols=0):
    params = O.make_params(cfg, seed=seed)
    batch = synth.make_batch(cfg, B, N, S, family, seed=seed)
    model = build_reference_model(cx, CfgCls, cfg, N, params)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    img = (tb["img_feats"], tb["img_pos"])
    with torch.no_grad():
        hidden, pooled, _ = model.enc(tb["cls_tok"], tb["input_txt"], tb["attn_mask"], tb["segment"], img, tb["sep_tok"])
    mlm, itm = model(tb["cls_tok"], tb["input_txt"], tb["attn_mask"], tb["segment"], img, tb["sep_tok"])
    ce_m = torch.nn.CrossEntropyLoss(ignore_index=-100)
    ce_i = torch.nn.CrossEntropyLoss()
    mlm_loss = ce_m(mlm.transpose(1, 2), tb["txt_labels"])
    itm_loss = ce_i(itm, tb["is_aligned"])
    out = dict(meta=np.array(json.dumps(dict(cfg=cfg.to_dict(), B=B, N=N, S=S, family=family, seed=seed))),
               pooled=pooled.numpy(), itm=itm.detach().numpy(),
               mlm_loss=np.float64(mlm_loss.item()), itm_loss=np.float64(itm_loss.item()))
    for k in ("cls_tok", "input_txt", "segment", "img_pos", "sep_tok", "txt_labels", "is_aligned", "n_ids"):
        out["in_" + k] = batch[k]
    out["in_mask_bits"] = D.pack_bits(batch["attn_mask"])
    if store_hidden:
        out["hidden"] = hidden.numpy()
    cols = (np.abs(O.splitmix_uniform(4242, ncols)) * cfg.vocab_size).astype(np.int64) % cfg.vocab_size
    out.update(logits_summary(mlm.detach(), np.unique(cols)))
    if full_logits:
        out["mlm"] = mlm.detach().numpy()
    if lab_ncols:
        # The LABELLED rows of the MLM logits, entry by entry, for the fused step's labelled-rows-only head (train_origin.py:120-126 only
        # ever reads those rows: CrossEntropyLoss(ignore_index=-100)).  All 30,522 columns of ~200 rows would be 26 MB of incompressible
        # f32; stored instead: `lab_ncols` sampled columns of every labelled row + the logit at the label + (above, for every
        # position) the row's logsumexp over ALL columns, its arg-max and its maximum.
        lab = tb["txt_labels"].reshape(-1)
        rows = torch.nonzero(lab != -100).reshape(-1)                    # flat b * L + i, row-major: the order TrainStep uses
        lcols = np.unique((np.abs(O.splitmix_uniform(5151, lab_ncols)) * cfg.vocab_size).astype(np.int64) % cfg.vocab_size)
        flat = mlm.detach().reshape(-1, cfg.vocab_size)[rows]
        out["lab_rows"] = rows.numpy().astype(np.int32)
        out["lab_ids"] = lab[rows].numpy().astype(np.int32)
        out["lab_cols"] = lcols.astype(np.int32)
        out["lab_logits_cols"] = flat[:, torch.from_numpy(lcols)].numpy()
        out["lab_logit_at_label"] = flat.gather(1, lab[rows].view(-1, 1)).reshape(-1).numpy()
    if with_grads:
        (itm_loss + mlm_loss).backward()
        g = ref_named_grads(model)
        out.update(grad_summary(g, O.param_shapes(cfg)))
        # full gradient rows of the special tokens of the tied word-embedding matrix ([PAD] row: look-up part is zero)
        out["dE_special_rows"] = g["enc.txt_embeddings.word_embeddings.weight"][[0, 100, 101, 102, 103]].numpy()
    return out


def gen_masks(out_path):
    """Drive the reference CXRDataset.__getitem__ under shims and record what it builds."""
    import io
    spec = importlib.util.spec_from_file_location("ref_dataset_origin", os.path.join(REF, "data", "dataset_origin.py"))
    # PIL is importable here; patch Image.open afterwards
    ds_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ds_mod)

    words = [f"w{i}" for i in range(2000)]
    vocab = {"[PAD]": 0, "[UNK]": 100, "[CLS]": 101, "[SEP]": 102, "[MASK]": 103}
    for i, w in enumerate(words):
        vocab[w] = 1000 + i
    vocab_full = dict(vocab)
    # reference uses len(vocab) as the randrange bound: pad to 30522 entries
    for i in range(D.VOCAB - len(vocab_full)):
        vocab_full[f"[unused{i}]"] = 40000 + i

    class FakeTok:
        def __init__(self):
            self.vocab = vocab_full

        @classmethod
        def from_pretrained(cls, *a, **k):
            return cls()

    ds_mod.BertTokenizer = FakeTok
    ds_mod.Image.open = lambda p: types.SimpleNamespace(convert=lambda mode: "img")
    ds_mod.CXRDataset.disturbing_mask = False     # Appendix D item 1

    rec = {}
    cases = []
    tmp = "/tmp/_medvill_golden.jsonl"
    grid = [(4, 8), (16, 45), (3, 29), (36, 60)]
    fam_flags = {
        "full": dict(Mixed=False, BAR_attn=False, disturbing_mask=False, attn_1d=False),
        "s2s": dict(Mixed=True, s2s_prob=1.0, bi_prob=0.0, BAR_attn=False, disturbing_mask=False, attn_1d=False),
        "bar": dict(Mixed=False, BAR_attn=True, disturbing_mask=False, attn_1d=False),
        "noncross": dict(Mixed=False, BAR_attn=False, disturbing_mask=True, attn_1d=False),
        "1d": dict(Mixed=False, BAR_attn=False, disturbing_mask=False, attn_1d=True),
    }
    ci = 0
    for (N, S) in grid:
        for n_txt in sorted({1, 2, S // 2, S - 1, S, S + 5}):
            text = " ".join(words[(7 * k + n_txt) % 2000] for k in range(n_txt))
            with open(tmp, "w") as f:
                f.write(json.dumps({"id": "a", "split": "train", "label": "L0", "text": text, "img": "x.jpg"}) + "\n")
                f.write(json.dumps({"id": "b", "split": "train", "label": "L1", "text": text + " w5", "img": "y.jpg"}) + "\n")
            for fam, flags in fam_flags.items():
                args = types.SimpleNamespace(max_seq_len=512, num_image_embeds=N, seq_len=S, bert_model="bert-base-scratch",
                                             img_channel=3, **{**dict(s2s_prob=0.0, bi_prob=1.0), **flags})
                ds = ds_mod.CXRDataset(tmp, str.split, lambda im: torch.zeros(1), args)
                random.seed(1000 + ci)
                item = None
                while item is None:          # random_pair_sampling may pick the negative branch; both fine
                    try:
                        item = ds[0]
                    except TypeError:        # Appendix D item 8 (None after 300 tries)
                        item = None
                cls_tok, ids, labels, mask, image, segment, is_aligned, sep_tok, itm_prob = item
                n_ids = int((ids != 0).sum())
                m = mask.numpy()
                tag = f"{ci:03d}"
                rec[f"bits_{tag}"] = D.pack_bits(m)
                rec[f"ids_{tag}"] = ids.numpy()
                rec[f"labels_{tag}"] = labels.numpy()
                rec[f"segment_{tag}"] = segment.numpy()
                cases.append(dict(tag=tag, family=fam, N=N, S=S, n_ids=n_ids, cls=int(cls_tok), sep=int(sep_tok),
                                  mask_ndim=int(m.ndim)))
                ci += 1
    # random_word known-answer tests straight from the reference method
    ds = ds_mod.CXRDataset(tmp, str.split, lambda im: torch.zeros(1), args)
    rw = []
    for k in range(24):
        n = [1, 2, 3, 5, 17, 40, 100, 300][k % 8]
        toks = [int(1000 + (13 * i + 7 * k) % 2000) for i in range(n)]
        random.seed(500 + k)
        t_out, lab = ds.random_word(list(toks))
        rec[f"rw_in_{k:02d}"] = np.array(toks, np.int64)
        rec[f"rw_tok_{k:02d}"] = np.array(t_out, np.int64)
        rec[f"rw_lab_{k:02d}"] = np.array(lab, np.int64)
        rw.append(dict(k=k, seed=500 + k))
    rec["cases"] = np.array(json.dumps(cases))
    rec["rw_cases"] = np.array(json.dumps(rw))
    np.savez_compressed(out_path, **rec)
    os.remove(tmp)
    print("masks:", len(cases), "cases ->", out_path)


def gen_manifest(cx, CfgCls, out_path):
    """state_manifest.json: key -> [shape, dtype] of the REAL reference CXRBERT.state_dict() at BERT-base (what train_origin.py:254-266
    writes into pytorch_model.bin and :28-34 loads back), so that tests can show a reference-written checkpoint loads here without the
    reference at hand.  The CNN trunk's keys (`enc.img_encoder.model.*`, torchvision ResNet-50 children 0-7: models/image.py:46-53)
    cannot come from the import -- torchvision is in neither the reference tree nor this image -- and are listed separately as
    'unpinned' from torchvision's published module names."""
    import transformers
    cfg = O.CONFIGS["base"]
    man = {}
    build_reference_model(cx, CfgCls, cfg, 36, O.make_params(cfg, seed=1), manifest=man)
    blocks = {4: 3, 5: 4, 6: 6, 7: 3}
    cnn = {"enc.img_encoder.model.0.weight": [64, 3, 7, 7]}

    def bn(prefix, c):
        for n_ in ("weight", "bias", "running_mean", "running_var"):
            cnn[f"{prefix}.{n_}"] = [c]
        cnn[f"{prefix}.num_batches_tracked"] = []
    bn("enc.img_encoder.model.1", 64)
    cin = 64
    for child, nb in blocks.items():
        w = 64 * 2 ** (child - 4)
        for i in range(nb):
            pre = f"enc.img_encoder.model.{child}.{i}"
            cnn[f"{pre}.conv1.weight"] = [w, cin, 1, 1]; bn(f"{pre}.bn1", w)
            cnn[f"{pre}.conv2.weight"] = [w, w, 3, 3]; bn(f"{pre}.bn2", w)
            cnn[f"{pre}.conv3.weight"] = [4 * w, w, 1, 1]; bn(f"{pre}.bn3", 4 * w)
            if i == 0:
                cnn[f"{pre}.downsample.0.weight"] = [4 * w, cin, 1, 1]; bn(f"{pre}.downsample.1", 4 * w)
            cin = 4 * w
    doc = dict(source="CXRBERT(config, args).state_dict() of /root/reference imported under the shims of oracle/gen_golden.py",
               transformers_version=transformers.__version__, torch_version=torch.__version__,
               config=dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                           intermediate_size=cfg.intermediate, max_position_embeddings=cfg.max_pos),
               keys=man,
               older_transformers_extra_keys={"enc.txt_embeddings.position_ids": [[1, cfg.max_pos], "int64"],
                                              "note": "persistent buffer in transformers 3.x-4.30 (the versions the reference's import paths "
                                                      "imply); newer versions register it non-persistent, so the import above does not list it"},
               img_encoder_keys_unpinned=dict(note="torchvision.models.resnet50 children()[:-2] under nn.Sequential (models/image.py:46-53): names "
                                                   "from torchvision's published module layout, NOT produced by the import (torchvision absent)",
                                              keys=cnn))
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print("manifest:", len(man), "reference keys +", len(cnn), "unpinned CNN keys ->", out_path)


def gen_adamw(out_path):
    """3 steps on 16 elements with python floats (independent of torch): HF AdamW <=4.x."""
    lr, b1, b2, eps, wd = 1e-3, 0.9, 0.999, 1e-6, 0.01
    u = O.splitmix_uniform(31337, 16 * 4).astype(np.float64)
    p = list(u[:16])
    gs = [list(u[16 * (s + 1):16 * (s + 2)] * (0.5 + s)) for s in range(3)]
    m = [0.0] * 16
    v = [0.0] * 16
    p0 = list(p)
    traj = []
    for t in range(1, 4):
        g = gs[t - 1]
        for i in range(16):
            m[i] = b1 * m[i] + (1 - b1) * g[i]
            v[i] = b2 * v[i] + (1 - b2) * g[i] * g[i]
            denom = math.sqrt(v[i]) + eps
            step = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
            p[i] = p[i] - step * m[i] / denom
            p[i] = p[i] - lr * wd * p[i]
        traj.append(list(p))
    np.savez_compressed(out_path, p0=np.array(p0), grads=np.array(gs), p=np.array(traj), m=np.array(m), v=np.array(v),
                        hyper=np.array([lr, b1, b2, eps, wd]))
    print("adamw ->", out_path)


def main(argv=()):
    only = None
    if "--only" in argv:
        only = set(argv[list(argv).index("--only") + 1].split(","))
    want = lambda name: only is None or name in only
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cx, CfgCls = install_shims()
    if want("masks"):
        gen_masks(os.path.join(OUT, "masks.npz"))
    if want("adamw"):
        gen_adamw(os.path.join(OUT, "adamw.npz"))
    if want("manifest"):
        gen_manifest(cx, CfgCls, os.path.join(OUT, "state_manifest.json"))
    c1 = O.CONFIGS["c1"]
    for fam in ("full", "s2s", "bar", "noncross", "1d"):
        if not want(f"c1_{fam}"):
            continue
        r = run_case(cx, CfgCls, c1, B=4, N=16, S=45, family=fam, seed=11, with_grads=(fam in ("full", "s2s")))
        np.savez_compressed(os.path.join(OUT, f"c1_{fam}.npz"), **r)
        print("c1", fam, "mlm_loss", float(r["mlm_loss"]), "itm_loss", float(r["itm_loss"]))
    c1v = O.OracleConfig(**{**c1.to_dict(), "vocab_size": 1024})
    if want("c1v1k_full"):
        r = run_case(cx, CfgCls, c1v, B=4, N=16, S=45, family="full", seed=12, with_grads=True, full_logits=True)
        np.savez_compressed(os.path.join(OUT, "c1v1k_full.npz"), **r)
    # args.img_postion = False (cxrbert_origin.py:27-31): the image rows add no position embedding
    if want("c1v1k_nopos"):
        c1n = O.OracleConfig(**{**c1v.to_dict(), "img_position": False})
        r = run_case(cx, CfgCls, c1n, B=4, N=16, S=45, family="full", seed=14, with_grads=True, full_logits=True)
        np.savez_compressed(os.path.join(OUT, "c1v1k_nopos.npz"), **r)
    # odd, non-tile-aligned geometry (L=37) to pin ragged handling
    if want("c1v1k_bar_ragged"):
        r = run_case(cx, CfgCls, c1v, B=3, N=5, S=29, family="bar", seed=13, with_grads=True, full_logits=True)
        np.savez_compressed(os.path.join(OUT, "c1v1k_bar_ragged.npz"), **r)
    base = O.CONFIGS["base"]
    if want("base_s2s"):
        r = run_case(cx, CfgCls, base, B=1, N=36, S=473, family="s2s", seed=21, with_grads=False, store_hidden=False)
        np.savez_compressed(os.path.join(OUT, "base_s2s.npz"), **r)
        print("base s2s", float(r["mlm_loss"]), float(r["itm_loss"]))
    # BASELINE.json configs 2, 4 and 5 at their own scale (256 logit columns per position)
    if want("base_full"):
        r = run_case(cx, CfgCls, base, B=2, N=36, S=473, family="full", seed=22, with_grads=True, store_hidden=False, ncols=256)
        np.savez_compressed(os.path.join(OUT, "base_full.npz"), **r)
        print("base full", float(r["mlm_loss"]), float(r["itm_loss"]))
    # the benchmarked path's shape: B = 4 ragged bidirectional samples (they pack), gradients of every parameter and the labelled
    # rows' logits for the labelled-rows-only head
    if want("base_full_b4"):
        r = run_case(cx, CfgCls, base, B=4, N=36, S=473, family="full", seed=26, with_grads=True, store_hidden=False, ncols=32, lab_ncols=1024)
        np.savez_compressed(os.path.join(OUT, "base_full_b4.npz"), **r)
        print("base full b4", float(r["mlm_loss"]), float(r["itm_loss"]), "labelled rows", int(r["lab_rows"].shape[0]))
    if want("base_noncross"):
        r = run_case(cx, CfgCls, base, B=1, N=36, S=473, family="noncross", seed=23, with_grads=False, store_hidden=False, ncols=256)
        np.savez_compressed(os.path.join(OUT, "base_noncross.npz"), **r)
        print("base noncross", float(r["mlm_loss"]), float(r["itm_loss"]))
    if want("base768_s2s"):
        b768 = O.CONFIGS["base768"]
        r = run_case(cx, CfgCls, b768, B=1, N=100, S=665, family="s2s", seed=24, with_grads=False, store_hidden=False, ncols=256)
        np.savez_compressed(os.path.join(OUT, "base768_s2s.npz"), **r)
        print("base768 s2s", float(r["mlm_loss"]), float(r["itm_loss"]))


if __name__ == "__main__":
    main(sys.argv[1:])
