"""Batch-side contract of the pretraining path: attention-mask families, label layout and a
synthetic batch generator that produces the reference Dataset's 9-tuple contents directly on
the device (no PIL / tokenizer / DataLoader workers on the benchmark path).

Mirrors (paths relative to the upstream repo):
  mask families ............. data/dataset_origin.py:138-176  (closed forms: SURVEY Appendix B)
  label / pad / segment ..... data/dataset_origin.py:105-135
  MLM corruption ............ data/dataset_origin.py:183-209 (15 % / 80-10-10, >= 1 label)
  region sampling ........... models/image.py:63-68 (sorted sample of M positions, shared by the batch)
"""
from __future__ import annotations

import torch

PAD, UNK, CLS, SEP, MASK = 0, 100, 101, 102, 103
FAMILIES = ("full", "s2s", "bar", "noncross", "1d")


FAMILY_ID = {"full": 0, "s2s": 1, "bar": 2, "noncross": 3, "1d": 4}


class MaskDesc:
    """Per-sample mask descriptors {family, n2, vl} (int32 [B,3]): what the attention kernels need instead of the
    reference's materialised int64 [B,L,L] matrices.  Accepted wherever `attn_mask` is (CXRBERT.forward, TrainStep)."""

    def __init__(self, desc: torch.Tensor, L: int, host: torch.Tensor = None):
        self.desc, self.L = desc.to(torch.int32), int(L)
        self._host = host                # CPU copy (kept when the descriptors were built on the host: no sync later)

    def host_desc(self) -> torch.Tensor:
        if self._host is None:
            self._host = self.desc.cpu()
        return self._host

    def packable(self) -> bool:
        """True when no valid query can see a position after the text [SEP] (full / seq2seq / 1-D families)."""
        f = self.host_desc()[:, 0]
        return bool(((f == 0) | (f == 1) | (f == 4)).all())

    @classmethod
    def make(cls, family, N: int, S: int, n_ids, device="cpu"):
        """family: one name for the whole batch or a per-sample sequence of names (Mixed)."""
        n_ids = torch.as_tensor(n_ids, dtype=torch.int32).view(-1)
        B = n_ids.numel()
        fam = [family] * B if isinstance(family, str) else list(family)
        d = torch.empty((B, 3), dtype=torch.int32)
        d[:, 0] = torch.tensor([FAMILY_ID[f] for f in fam], dtype=torch.int32)
        d[:, 1] = N + 2
        d[:, 2] = N + 2 + n_ids
        return cls(d.to(device), S + N + 3, host=d)

    def dim(self):
        return 3

    def __getitem__(self, idx):          # batch slicing, like a [B, ...] tensor
        return MaskDesc(self.desc[idx].reshape(-1, 3), self.L, None if self._host is None else self._host[idx].reshape(-1, 3))

    def __len__(self):
        return int(self.desc.shape[0])

    def to(self, device, *a, **k):
        return MaskDesc(self.desc.to(device), self.L, self._host)


def descriptors_from_dense(mask: torch.Tensor, input_ids: torch.Tensor, N: int):
    """HYPOTHESIS {family, n2, vl} per sample for a materialised reference mask (dataset_origin.py:138-176), from two rows and a column
    per sample -- the torch restatement, on whatever device the tensors live on, of the trainer's host recogniser
    (CXRBERT_Trainer._recognise_masks: same probes, same precedence).  Nothing is read back: -> (desc int32 [B,3], ok bool []) with
    `ok` false when some sample matches no family; the caller confirms the hypothesis entry by entry (CXRBERT._mask_descriptors compares
    the mask words of mv_mask_pack(mask) and mv_mask_build(desc)) before anything runs on it.  None when the shapes rule it out."""
    if not torch.is_tensor(mask) or mask.dtype != torch.int64 or mask.dim() not in (2, 3):
        return None
    B, L = mask.shape[0], mask.shape[-1]
    S, n2 = L - N - 3, N + 2
    if S < 1 or tuple(input_ids.shape) != (B, S + 1):
        return None
    dev = mask.device
    ids = input_ids.to(dev)
    T = S + 1
    t = torch.arange(T, device=dev).view(1, T)
    # valid length from the LAST non-zero id (random_word, dataset_origin.py:183-209, may put id 0 at a labelled in-text position)
    last = torch.where(ids != 0, t, torch.full_like(t, -1)).max(dim=1).values
    vl = n2 + last + 1
    j = torch.arange(L, device=dev).view(1, L)
    full_row = (j < vl.view(B, 1)).to(torch.int64)
    if mask.dim() == 2:
        fam = torch.full((B,), FAMILY_ID["1d"], dtype=torch.int64, device=dev)
        ok = (mask == full_row).all()
    else:
        r0, rl, cl = mask[:, 0, :], mask[:, L - 1, :], mask[:, :, L - 1]
        img_row, txt_row = (j < n2).to(torch.int64), (j >= n2).to(torch.int64)
        bar_col = ((j < n2) | (j == L - 1)).to(torch.int64)
        eq = lambda a, b: (a == b).all(dim=1)
        one = lambda a: (a == 1).all(dim=1)
        # (the last column tells a full-length BAR sample -- whose probe ROWS are all ones too -- from a full one)
        is_full = eq(r0, full_row) & eq(rl, full_row) & eq(cl, full_row[:, L - 1:L].expand(B, L))
        is_s2s = eq(r0, img_row) & one(rl)
        is_bar = one(r0) & one(rl) & eq(cl, bar_col)
        is_non = eq(r0, img_row) & eq(rl, txt_row)
        fam = torch.full((B,), -1, dtype=torch.int64, device=dev)
        for cond, name in ((is_non, "noncross"), (is_bar, "bar"), (is_s2s, "s2s"), (is_full, "full")):     # later entries win
            fam = torch.where(cond, torch.full_like(fam, FAMILY_ID[name]), fam)
        ok = (fam >= 0).all()
    desc = torch.stack([fam.clamp(min=0), torch.full_like(fam, n2), vl], dim=1).to(torch.int32)
    return desc, ok


def build_mask(family: str, N: int, S: int, n_ids, device="cpu") -> torch.Tensor:
    """int64 [B,L,L] (or [B,L] for '1d') for per-sample text lengths n_ids (incl. the text [SEP])."""
    n_ids = torch.as_tensor(n_ids, device=device, dtype=torch.int64).view(-1)
    B = n_ids.numel()
    L = S + N + 3
    n2 = N + 2
    vl = (n2 + n_ids).view(B, 1, 1)
    i = torch.arange(L, device=device).view(1, L, 1)
    j = torch.arange(L, device=device).view(1, 1, L)
    if family == "full":
        m = (j < vl).expand(B, L, L)
    elif family == "s2s":
        m = ((j < n2) | ((i >= n2) & (j >= n2) & (j <= i))).expand(B, L, L)
    elif family == "bar":
        m = ((i < n2) | (j < n2) | (j <= i)).expand(B, L, L)
    elif family == "noncross":
        m = ((i < n2) == (j < n2)).expand(B, L, L)
    elif family == "1d":
        return (torch.arange(L, device=device).view(1, L) < vl.view(B, 1)).to(torch.int64)
    else:
        raise ValueError(family)
    return m.to(torch.int64).contiguous()


def mixed_mask(N: int, S: int, n_ids, choose_s2s, device="cpu"):
    """`Mixed` (dataset_origin.py:152-155): per-sample choice between full and s2s."""
    full = build_mask("full", N, S, n_ids, device)
    s2s = build_mask("s2s", N, S, n_ids, device)
    sel = torch.as_tensor(choose_s2s, device=device, dtype=torch.bool).view(-1, 1, 1)
    return torch.where(sel, s2s, full)


def assemble_batch(ids: torch.Tensor, lengths: torch.Tensor, N: int, vocab: int, key: int, family="full", draws=None) -> dict:
    """Device-side sample assembly from RAW token ids (HIP kernels mv_mlm_draws / mv_mlm_corrupt): the MLM corruption
    of random_word (dataset_origin.py:183-209), [SEP]/[PAD]/label/segment layout (:105-135), mask descriptors and the
    labelled-row index.  ids int64 [B,S] on the GPU, valid for t < lengths[b].  `draws` = (u f32 [B,S], rnd int32 [B,S])
    overrides the hash-generated random sources.  One host sync (the label count)."""
    from . import hip_ops as ops
    B, S = ids.shape
    dev = ids.device
    u, rnd = draws if draws is not None else ops.mlm_draws(key, B, S, vocab, dev)
    fams = [family] * B if isinstance(family, str) else list(family)
    fam_t = torch.tensor([FAMILY_ID[f] for f in fams], dtype=torch.int32).to(dev)
    out = ops.mlm_corrupt(ids, lengths.to(torch.int32), u, rnd, N, family=fam_t)
    n = int(out["n_labels"].item())
    return dict(input_txt=out["input_txt"], segment=out["segment"], txt_labels=out["txt_labels"], n_ids=out["n_ids"].to(torch.int64),
                label_rows=out["label_rows"][:n], label_ids=out["label_ids"][:n], attn_desc=MaskDesc(out["desc"], S + N + 3))


def corrupt_tokens(ids: torch.Tensor, lengths: torch.Tensor, vocab: int, gen: torch.Generator):
    """Host-side (torch) vectorised random_word for CPU-only tooling and tests; the GPU path is `assemble_batch`.
    ids [B,S] (valid for t < lengths[b]).  Returns (ids', labels) with labels = original id where selected
    else -100; guarantees >= 1 label per sample."""
    B, S = ids.shape
    dev = ids.device
    t = torch.arange(S, device=dev).view(1, S)
    valid = t < lengths.view(B, 1)
    u = torch.rand((B, S), generator=gen, device=dev)
    sel = (u < 0.15) & valid
    none = ~sel.any(dim=1)
    sel[none, 0] = True                       # "at least one mask": position 0 becomes [MASK]
    u2 = u / 0.15
    rnd = torch.randint(0, vocab, (B, S), generator=gen, device=dev)
    out = ids.clone()
    to_mask = sel & ((u2 < 0.8) | none.view(B, 1))
    to_rand = sel & (u2 >= 0.8) & (u2 < 0.9) & ~none.view(B, 1)
    out[to_mask] = MASK
    out[to_rand] = rnd[to_rand]
    labels = torch.where(sel, ids, torch.full_like(ids, -100))
    return out, labels


def synthetic_batch(vocab: int, B: int, N: int, S: int, family: str, seed: int, device="cuda", img_hidden: int = 2048,
                    M_regions: int = 256, feat_dtype=torch.float32, lengths=None) -> dict:
    """One mini-batch in the reference's batch protocol (dataset_origin.py:181) plus the
    labelled-row index the fused MLM head consumes.  family: full|s2s|bar|noncross|1d|mixed."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    T, L = S + 1, S + N + 3
    lo = 1000 if vocab > 2000 else 200
    drawn = torch.randint((S + 1) // 2, S + 1, (B,), generator=gen, device=dev)      # len ~ U{ceil(S/2)..S}
    lengths = drawn if lengths is None else torch.as_tensor(lengths, dtype=torch.int64).clamp(1, S).to(dev)
    ids = torch.randint(lo, vocab, (B, S), generator=gen, device=dev)
    n_ids = lengths + 1
    if dev.type == "cuda":
        asm = assemble_batch(ids, lengths, N, vocab, key=0x5EED0000 + seed)     # family descriptors are filled in below
        txt, labels, segment = asm["input_txt"], asm["txt_labels"], asm["segment"]
        rows, lids = asm["label_rows"], asm["label_ids"]
    else:
        ids_c, lab = corrupt_tokens(ids, lengths, vocab, gen)
        t = torch.arange(T, device=dev).view(1, T)
        txt = torch.zeros((B, T), dtype=torch.int64, device=dev)
        txt[:, :S] = torch.where(t[:, :S] < lengths.view(B, 1), ids_c, torch.zeros_like(ids_c))
        txt[torch.arange(B, device=dev), lengths] = SEP
        labels = torch.full((B, L), -100, dtype=torch.int64, device=dev)
        labels[:, N + 2:N + 2 + S] = torch.where(t[:, :S] < lengths.view(B, 1), lab, torch.full_like(lab, -100))
        segment = torch.ones((B, T), dtype=torch.int64, device=dev)
        rows, lids = label_index(labels)
    perm = torch.randperm(M_regions, generator=gen, device=dev)[:N]
    pos = torch.sort(perm)[0].view(1, N).expand(B, N).contiguous()
    feats = torch.randn((B, N, img_hidden), generator=gen, device=dev, dtype=torch.float32).to(feat_dtype)
    is_aligned = (torch.rand((B,), generator=gen, device=dev) > 0.5).to(torch.int64)
    if family == "mixed":
        choose = torch.rand((B,), generator=gen, device=dev) < 0.75
        mask = mixed_mask(N, S, n_ids, choose, dev)
        fams = ["s2s" if c else "full" for c in choose.tolist()]
    else:
        mask = build_mask(family, N, S, n_ids, dev)
        fams = family
    desc = MaskDesc.make(fams, N, S, n_ids.cpu(), dev)
    return dict(cls_tok=torch.full((B, 1), CLS, dtype=torch.int64, device=dev), input_txt=txt, attn_mask=mask, segment=segment,
                img_feats=feats, img_pos=pos, sep_tok=torch.full((B, 1), SEP, dtype=torch.int64, device=dev), txt_labels=labels,
                is_aligned=is_aligned, n_ids=n_ids, label_rows=rows, label_ids=lids, attn_desc=desc)


def label_index(txt_labels: torch.Tensor):
    """(rows int32 [R], ids int32 [R]) of the positions with a label (!= -100), row-major."""
    flat = txt_labels.reshape(-1)
    rows = torch.nonzero(flat != -100).view(-1)
    return rows.to(torch.int32), flat[rows].to(torch.int32)
