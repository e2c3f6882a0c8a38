"""Checkpoint key maps between the pretraining model (models/train_origin.py:254-266, HF `save_pretrained` layout,
keys as in SURVEY.md Appendix C) and the reference's downstream programs, which rename the keys when they recover a
pretrained model:

  * fine-tuning for report generation / VQA (Downstream_task/report_generation_and_vqa/sc/finetune.py:338-339):
        key.replace('enc.', '').replace('mlm.', 'cls.')
  * decoding (Downstream_task/report_generation_and_vqa/sc/generation_decode.py:385-388), applied to an
    already fine-tuned state dict: prefix the encoder parts with `bert.`.

These are pure renames (no arithmetic); they let checkpoints written by this library's `CXRBERT.save_pretrained`
feed those scripts, and the inverse maps let their checkpoints come back.

`from_hf_bert_keys` is the map the reference gets implicitly when it builds its encoder from a PRETRAINED text BERT
(models/cxrbert_origin.py:41-59: `bert = BertModel.from_pretrained(args.bert_model)`, then `self.txt_embeddings = bert.embeddings`,
`self.encoder = bert.encoder`, `self.pooler = bert.pooler`).
"""
from __future__ import annotations

from collections import OrderedDict


def to_finetune_keys(state_dict):
    """finetune.py:338-339."""
    return OrderedDict((k.replace("enc.", "").replace("mlm.", "cls."), v) for k, v in state_dict.items())


def from_finetune_keys(state_dict):
    """Inverse of `to_finetune_keys` for the tensors of the pretraining path."""
    out = OrderedDict()
    for k, v in state_dict.items():
        if k.startswith("cls."):
            out["mlm." + k[4:]] = v
        elif k.startswith(("txt_embeddings.", "img_embeddings.", "img_encoder.", "encoder.", "pooler.")):
            out["enc." + k] = v
        else:
            out[k] = v
    return out


def to_decode_keys(state_dict):
    """generation_decode.py:385-388 (on a fine-tune-style state dict)."""
    out = OrderedDict()
    for k, v in state_dict.items():
        k2 = (k.replace("txt_embeddings", "bert.txt_embeddings").replace("img_embeddings", "bert.img_embeddings")
              .replace("img_encoder.model", "bert.img_encoder.model").replace("encoder.layer", "bert.encoder.layer")
              .replace("pooler", "bert.pooler"))
        k2 = k2.replace("bert.img_embeddings.bert.img_embeddings", "bert.img_embeddings.img_embeddings")
        out[k2] = v
    return out


def from_hf_bert_keys(state_dict):
    """A HF `BertModel` / `BertForPreTraining` state dict (`[bert.]embeddings.* / encoder.layer.* / pooler.*`, TF-era `LayerNorm.gamma /
    beta` accepted) -> the CXRBERT keys those modules live under (cxrbert_origin.py:56-57,72-73): `enc.txt_embeddings.*`,
    `enc.encoder.*`, `enc.pooler.*`.  Everything else of such a checkpoint (its `cls.*` pretraining heads, `position_ids`) is dropped:
    the reference keeps nothing of it either -- its MLM / ITM heads and the image projection start from their own initialisers."""
    out = OrderedDict()
    for k, v in state_dict.items():
        if k.startswith("bert."):
            k = k[5:]
        k = k.replace("LayerNorm.gamma", "LayerNorm.weight").replace("LayerNorm.beta", "LayerNorm.bias")
        if "position_ids" in k or "token_type_ids" in k:
            continue
        if k.startswith("embeddings."):
            out["enc.txt_embeddings." + k[len("embeddings."):]] = v
        elif k.startswith("encoder."):
            out["enc.encoder." + k[len("encoder."):]] = v
        elif k.startswith("pooler."):
            out["enc.pooler." + k[len("pooler."):]] = v
    return out
