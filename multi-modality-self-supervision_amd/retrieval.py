"""Mirror of the reference's retrieval head (Downstream_task/Retrieval/retrieval.py:12-32): ITM-style matching on
top of CXRBERT.enc + .itm with 1-D attention masks (the `attn_mask.dim() == 2` branch, cxrbert_origin.py:76-77).
Encoder-only inference / fine-tuning through the same HIP kernels."""
from __future__ import annotations

import torch
import torch.nn as nn

from .cxrbert import CXRBERT


class CXRBertForRetrieval(nn.Module):
    def __init__(self, config, args=None, **kw):
        super().__init__()
        self.bert = CXRBERT(config, args, **kw)
        self.enc, self.itm = self.bert.enc, self.bert.itm

    @classmethod
    def from_pretrained(cls, path, args=None, **kw):
        m = cls.__new__(cls)
        nn.Module.__init__(m)
        m.bert = CXRBERT.from_pretrained(path, args=args, **kw)
        m.enc, m.itm = m.bert.enc, m.bert.itm
        return m

    def forward(self, cls_tok, input_txt, attn_mask, segment, input_img, sep_tok):
        """-> ITM logits [B,2] (retrieval.py:26-31: `_, cls, _ = self.enc(...); return self.itm(cls)` -- that literal form works
        too, `enc` and `itm` are callable; this is the same arithmetic as ONE autograd node, without the hand-over tensors)."""
        return self.bert._itm_only(cls_tok, input_txt, attn_mask, segment, input_img, sep_tok)

    @torch.no_grad()
    def score(self, cls_tok, input_txt, attn_mask, segment, input_img, sep_tok):
        """P(aligned) per pair, as full_dset_retrieval.py:461-510 ranks candidates.  When `attn_mask` is a
        `data.MaskDesc` of a family whose padding is invisible (the retrieval scripts' 1-D masks are), the encoder runs
        on the valid rows only (inference form of the padding removal)."""
        from .data import MaskDesc
        eng = self.bert.engine
        if isinstance(attn_mask, MaskDesc) and eng.is16 and attn_mask.packable():
            feats, pos = self.bert._regions(input_img)
            prev = (eng.training, eng.keep_acts)
            eng.training, eng.keep_acts = False, False
            try:
                eng.encoder_forward(cls_tok, input_txt, attn_mask, segment, feats, pos, sep_tok, pack=True)
                logits = eng._itm_forward().clone()
            finally:                # sticky engine state: a later direct Engine user must find what it left
                eng.training, eng.keep_acts = prev
        else:
            logits = self.forward(cls_tok, input_txt, attn_mask, segment, input_img, sep_tok)
        return torch.softmax(logits.float(), dim=-1)[:, 1]
