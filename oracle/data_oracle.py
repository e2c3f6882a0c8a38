"""CPU oracle for the sample / mask / label contract of the pretraining path.

TEST INFRASTRUCTURE ONLY (see ``oracle/cxrbert_oracle.py`` header for who may import it).

Restates, in numpy, the integer side of the reference's Dataset so that the
on-device mask packer / synthetic-batch generator of the product can be checked
BIT-EXACTLY.  References (relative to the upstream repo root):

* attention-mask families .......... data/dataset_origin.py:138-176
* label / id / pad / segment layout . data/dataset_origin.py:102-135
* MLM corruption (random_word) ...... data/dataset_origin.py:183-209
* region sampling ................... models/image.py:60-69

``build_mask`` follows the reference's *construction* (tensor fills / tril copy);
``mask_predicate`` is the closed form of SURVEY.md Appendix B.  The golden test
checks both against packed matrices produced by the reference builder itself
(``tests/golden/masks.npz``).
"""
from __future__ import annotations

import numpy as np

PAD, UNK, CLS, SEP, MASK = 0, 100, 101, 102, 103
VOCAB = 30522
FAMILIES = ("full", "s2s", "bar", "noncross", "1d")


def build_mask(family: str, N: int, S: int, n_ids: int) -> np.ndarray:
    """Construct the attention-mask matrix the way dataset_origin.py does.

    N = num_image_embeds, S = seq_len, n_ids = number of text ids INCLUDING the
    text [SEP] and BEFORE padding (len(input_ids) at dataset_origin.py:113).
    Returns int64 [L,L] (or [L] for '1d'), L = S + N + 3 (dataset_origin.py:37)."""
    L = S + N + 3
    n2 = N + 2
    T = S + 1
    attn_1d = np.concatenate([np.ones(n2, np.int64), np.ones(n_ids, np.int64),
                              np.full(T - n_ids, PAD, np.int64)])          # :113-126 (pad ids == 0)
    assert attn_1d.shape[0] == L
    full = np.broadcast_to(attn_1d[None, :], (L, L)).copy()                 # :140-141
    ext = np.zeros((L, L), np.int64)                                         # :143
    # `second_end` is computed from len(input_ids) AFTER padding (:122,144) => T, not n_ids:
    # the tril covers the whole text block, pads included.
    st, en = n2, n2 + T
    ext[:, :n2] = 1                                                          # :145
    ext[st:en, st:en] = np.tril(np.ones((T, T), np.int64))                   # :146-147
    if family == "full":
        return full
    if family == "s2s":
        return ext
    if family == "bar":
        ext = ext.copy()
        ext[:n2, :] = 1                                                      # :159
        return ext
    if family == "noncross":
        m = np.zeros((L, L), np.int64)                                       # :164-166
        m[:n2, :n2] = 1
        m[n2:, n2:] = 1
        return m
    if family == "1d":
        return attn_1d                                                       # :170-172
    raise ValueError(family)


def mask_predicate(family: str, N: int, S: int, n_ids: int) -> np.ndarray:
    """Closed-form predicates of SURVEY.md Appendix B (M[i,j]=1: query i may attend key j)."""
    L = S + N + 3
    n2 = N + 2
    vl = n2 + n_ids
    i = np.arange(L)[:, None]
    j = np.arange(L)[None, :]
    if family == "full":
        m = np.broadcast_to(j < vl, (L, L))
    elif family == "s2s":
        m = (j < n2) | ((i >= n2) & (j >= n2) & (j <= i))
    elif family == "bar":
        m = (i < n2) | (j < n2) | (j <= i)
    elif family == "noncross":
        m = (i < n2) == (j < n2)
    elif family == "1d":
        return (np.arange(L) < vl).astype(np.int64)
    else:
        raise ValueError(family)
    return m.astype(np.int64)


def pack_bits(m: np.ndarray) -> np.ndarray:
    """[...,L] 0/1 -> uint32 words, bit (j & 31) of word (j >> 5) = m[..., j] (little-endian
    in the word).  This is the layout the HIP attention kernels consume."""
    L = m.shape[-1]
    W = (L + 31) // 32
    pad = W * 32 - L
    mm = np.concatenate([m != 0, np.zeros(m.shape[:-1] + (pad,), bool)], axis=-1)
    mm = mm.reshape(m.shape[:-1] + (W, 32)).astype(np.uint64)
    w = (mm << np.arange(32, dtype=np.uint64)).sum(-1)
    return w.astype(np.uint32)


def random_word(tokens, rng, vocab_len=VOCAB):
    """dataset_origin.py:183-209 with an injectable uniform source ``rng.random()`` /
    ``rng.randrange(n)`` (python ``random`` API): 15 % selected; of those 80 % -> [MASK],
    10 % -> random id, 10 % kept; label = original id else -100; force >= 1 label."""
    tokens = list(tokens)
    labels = []
    for i, tok in enumerate(tokens):
        p = rng.random()
        if p < 0.15:
            p /= 0.15
            if p < 0.8:
                tokens[i] = MASK
            elif p < 0.9:
                tokens[i] = rng.randrange(vocab_len)
            labels.append(tok)
        else:
            labels.append(-100)
    if all(o == -100 for o in labels):
        labels[0] = tokens[0]
        tokens[0] = MASK
    return tokens, labels


def assemble_sample(ids_corrupted, labels, N: int, S: int):
    """dataset_origin.py:105-135: append [SEP]; labels = [-100]*(N+2) + txt + [-100];
    pad ids with [PAD]=0 and labels with -100 to T=S+1; segment = 1 over T."""
    ids = list(ids_corrupted) + [SEP]
    lab_t = list(labels) + [-100]
    n_ids = len(ids)
    T = S + 1
    ids = ids + [PAD] * (T - n_ids)
    lab_t = lab_t + [-100] * (T - n_ids)
    lab = [-100] * (N + 2) + lab_t
    seg = [1] * T
    return np.array(ids, np.int64), np.array(lab, np.int64), np.array(seg, np.int64), n_ids


def sample_regions(M: int, N: int, gen) -> np.ndarray:
    """models/image.py:63-65: sort(randperm(M)[:N]); shared by the whole batch."""
    import torch
    idx = torch.randperm(M, generator=gen)[:N]
    return torch.sort(idx)[0].numpy()
