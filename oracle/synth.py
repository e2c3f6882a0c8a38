"""Deterministic synthetic batches for the oracle, the golden generator and the CPU baseline.

TEST INFRASTRUCTURE ONLY (see ``oracle/cxrbert_oracle.py`` header).

Follows the synthetic-input recipe of SURVEY.md §8(d): region features ~ unit
variance, region positions = sort(sample of M=256 without replacement) shared by
the batch (models/image.py:63-68), per-sample text length ~ U{ceil(S/2)..S}, token
ids uniform over the non-special vocabulary, MLM corruption exactly as
``random_word`` (data/dataset_origin.py:183-209), [SEP] appended, pads, labels,
segment (dataset_origin.py:105-135), is_aligned ~ Bernoulli(0.5), mask family
per config (dataset_origin.py:138-176).  Everything is driven by splitmix64 /
``random.Random`` so the same batch is reproduced bit-for-bit anywhere.
"""
from __future__ import annotations

import random

import numpy as np

from . import data_oracle as D
from .cxrbert_oracle import splitmix_uniform


def make_batch(cfg, B: int, N: int, S: int, family: str, seed: int, M: int = 256) -> dict:
    V = cfg.vocab_size
    lo = 1000 if V > 2000 else 200
    T = S + 1
    L = S + N + 3
    rng = random.Random(seed)
    feats = (splitmix_uniform(seed * 7919 + 1, B * N * cfg.img_hidden) * np.float32(1.7320508)).reshape(B, N, cfg.img_hidden)
    perm = np.argsort(splitmix_uniform(seed * 7919 + 2, M), kind="stable")
    pos = np.sort(perm[:N]).astype(np.int64)
    img_pos = np.broadcast_to(pos[None], (B, N)).copy()
    ids = np.zeros((B, T), np.int64)
    labels = np.zeros((B, L), np.int64)
    seg = np.zeros((B, T), np.int64)
    n_ids = np.zeros((B,), np.int64)
    masks = []
    for b in range(B):
        n_txt = rng.randint((S + 1) // 2, S)
        toks = [rng.randrange(lo, V) for _ in range(n_txt)]
        t2, lab = D.random_word(toks, rng, vocab_len=V)
        i_, l_, s_, n_ = D.assemble_sample(t2, lab, N, S)
        ids[b], labels[b], seg[b], n_ids[b] = i_, l_, s_, n_
        fam = family
        if family == "mixed":       # dataset_origin.py:152-155: per-sample choice, weights [bi, s2s] = [.25,.75]
            fam = rng.choices(["full", "s2s"], weights=[0.25, 0.75])[0]
        masks.append(D.build_mask(fam, N, S, int(n_)))
    is_aligned = np.array([1 if rng.random() > 0.5 else 0 for _ in range(B)], np.int64)
    return dict(cls_tok=np.full((B, 1), D.CLS, np.int64), input_txt=ids, attn_mask=np.stack(masks),
                segment=seg, img_feats=feats.astype(np.float32), img_pos=img_pos,
                sep_tok=np.full((B, 1), D.SEP, np.int64), txt_labels=labels, is_aligned=is_aligned, n_ids=n_ids)
