"""Host-side logic of the product (CPU only): parameter layout / state-dict contract, mask and
batch builders against the oracle (bit-exact), DP bucket map."""
import os

import numpy as np
import pytest
import torch

import medvill_amd as mv
from medvill_amd.dist import bucket_ranges
from medvill_amd.engine import ALIASES
from oracle import cxrbert_oracle as O
from oracle import data_oracle as D

TINY = dict(vocab_size=1024, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
            max_position_embeddings=128)


def test_param_layout_matches_reference_names_and_count():
    for name in ("c1", "base"):
        oc = O.CONFIGS[name]
        cfg = mv.ModelConfig(vocab_size=oc.vocab_size, hidden=oc.hidden, layers=oc.layers, heads=oc.heads,
                             intermediate=oc.intermediate, max_pos=oc.max_pos)
        lay, n_flat = mv.param_layout(cfg)
        shapes = O.param_shapes(oc)
        assert set(lay) == set(shapes)
        assert all(lay[k][1] == tuple(shapes[k]) for k in shapes)
        assert sum(int(np.prod(s)) for _, s in lay.values()) == O.num_params(oc)
        assert all(off % 64 == 0 for off, _ in lay.values())            # 16-byte aligned in fp32 and bf16
        ends = sorted((off, off + int(np.prod(s))) for off, s in lay.values())
        assert all(a[1] <= b[0] for a, b in zip(ends, ends[1:])) and ends[-1][1] <= n_flat
        H = cfg.hidden
        for l in range(cfg.layers):                                     # fused QKV views must be contiguous
            p = f"enc.encoder.layer.{l}.attention.self."
            assert lay[p + "key.weight"][0] == lay[p + "query.weight"][0] + H * H
            assert lay[p + "value.weight"][0] == lay[p + "query.weight"][0] + 2 * H * H
            assert lay[p + "key.bias"][0] == lay[p + "query.bias"][0] + H
            assert lay[p + "value.bias"][0] == lay[p + "query.bias"][0] + 2 * H


def test_state_dict_contract_and_checkpoint_roundtrip(tmp_path):
    m = mv.CXRBERT(TINY, None, device="cpu")
    sd = m.state_dict()
    for alias, canon in ALIASES.items():                                # cxrbert_origin.py:17-20,141,231
        assert torch.equal(sd[alias], sd[canon])
    names = [n for n, _ in m.named_parameters()]
    assert "mlm.predictions.transform.dense.weight" in names and "itm.linear.bias" in names
    assert sum(p.numel() for p in m.parameters()) == sum(int(np.prod(s)) for _, s in m.engine.layout.values())
    m.save_pretrained(str(tmp_path / "ckpt"))
    assert os.path.exists(tmp_path / "ckpt" / "config.json") and os.path.exists(tmp_path / "ckpt" / "pytorch_model.bin")
    m2 = mv.CXRBERT.from_pretrained(str(tmp_path / "ckpt"), device="cpu")
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    # reference checkpoints may carry position_ids / a ResNet trunk: accepted and ignored
    sd["enc.txt_embeddings.position_ids"] = torch.arange(4)
    sd["enc.img_encoder.model.0.weight"] = torch.zeros(1)
    assert m2.load_state_dict(sd, strict=True).unexpected_keys == []
    with pytest.raises(RuntimeError):
        m2.load_state_dict({"bogus": torch.zeros(1)}, strict=True)


def test_reference_error_behaviour():
    m = mv.CXRBERT(TINY, None, device="cpu")
    z = torch.zeros((2, 1), dtype=torch.int64)
    with pytest.raises(NotImplementedError):                              # cxrbert_origin.py:80-81
        m(z, torch.zeros((2, 5), dtype=torch.int64), torch.zeros((2, 3, 3, 3), dtype=torch.int64),
          torch.ones((2, 5), dtype=torch.int64), (torch.zeros(2, 1, 2048), torch.zeros((2, 1), dtype=torch.int64)), z)
    with pytest.raises(TypeError):
        m(z, torch.zeros((2, 5), dtype=torch.int64), torch.zeros((2, 8, 8), dtype=torch.int64),
          torch.ones((2, 5), dtype=torch.int64), torch.zeros(2, 3, 8, 8), z)   # raw pixels need an img_encoder


@pytest.mark.parametrize("fam", D.FAMILIES)
def test_mask_builder_bit_exact_against_oracle(fam):
    for N, S in ((4, 8), (16, 45), (3, 29), (36, 60)):
        n_ids = [1, 2, S // 2 + 1, S, S + 1]
        got = mv.data.build_mask(fam, N, S, n_ids).numpy()
        for b, n in enumerate(n_ids):
            assert np.array_equal(got[b], D.build_mask(fam, N, S, n)), (fam, N, S, n)


def test_synthetic_batch_follows_the_dataset_contract():
    V, B, N, S = 30522, 16, 36, 61
    b = mv.data.synthetic_batch(V, B, N, S, "mixed", seed=5, device="cpu")
    L, T = S + N + 3, S + 1
    assert b["input_txt"].shape == (B, T) and b["txt_labels"].shape == (B, L) and b["attn_mask"].shape == (B, L, L)
    assert (b["segment"] == 1).all() and (b["cls_tok"] == 101).all() and (b["sep_tok"] == 102).all()
    for i in range(B):
        n = int(b["n_ids"][i])
        ids, lab = b["input_txt"][i], b["txt_labels"][i]
        assert ids[n - 1] == 102 and (ids[n:] == 0).all() and (ids[:n] != 0).all()
        assert (lab[:N + 2] == -100).all() and (lab[N + 2 + n - 1:] == -100).all() and (lab != -100).sum() >= 1
        sel = lab[N + 2:N + 2 + n - 1] != -100
        kept = ids[:n - 1][~sel]
        assert ((kept >= 1000) & (kept < V)).all()                      # unselected tokens are untouched
        m = b["attn_mask"][i].numpy()
        assert any(np.array_equal(m, D.build_mask(f, N, S, n)) for f in ("full", "s2s"))
    pos = b["img_pos"]
    assert (pos[0] == pos[-1]).all() and (pos[0][1:] > pos[0][:-1]).all() and pos.max() < 256   # image.py:63-68
    d = b["attn_desc"]                                                   # {family, n2, vl} per sample
    assert d.L == L and (d.desc[:, 1] == N + 2).all() and torch.equal(d.desc[:, 2].long(), N + 2 + b["n_ids"])
    for i in range(B):
        fam = {0: "full", 1: "s2s"}[int(d.desc[i, 0])]
        assert np.array_equal(b["attn_mask"][i].numpy(), D.build_mask(fam, N, S, int(b["n_ids"][i])))
    rows, ids = mv.data.label_index(b["txt_labels"])
    assert torch.equal(rows, b["label_rows"]) and torch.equal(ids, b["label_ids"])
    frac = float((b["txt_labels"] != -100).sum()) / float((b["n_ids"] - 1).sum())
    assert 0.08 < frac < 0.25                                            # ~15 % of the text tokens


def test_dp_buckets_tile_the_flat_buffer():
    cfg = mv.ModelConfig()
    lay, n = mv.param_layout(cfg)
    r = bucket_ranges(lay, n, cfg.layers)
    order = ["embeddings"] + [f"layer{l}" for l in range(cfg.layers)] + ["heads"]
    assert r[order[0]][0] == 0 and r[order[-1]][1] == n
    assert all(r[a][1] == r[b][0] for a, b in zip(order, order[1:]))
    assert r["layer0"][1] - r["layer0"][0] == r["layer5"][1] - r["layer5"][0]


def test_downstream_checkpoint_key_maps():
    """the renames finetune.py:338-339 / generation_decode.py:385-388 apply, and their inverse."""
    m = mv.CXRBERT(TINY, None, device="cpu")
    sd = m.state_dict()
    ft = mv.checkpoint.to_finetune_keys(sd)
    assert "txt_embeddings.word_embeddings.weight" in ft and "cls.predictions.decoder.weight" in ft
    assert "encoder.layer.0.attention.self.query.weight" in ft and "itm.linear.weight" in ft
    assert not any(k.startswith(("enc.", "mlm.")) for k in ft)
    back = mv.checkpoint.from_finetune_keys(ft)
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    dec = mv.checkpoint.to_decode_keys(ft)
    assert "bert.txt_embeddings.word_embeddings.weight" in dec and "bert.encoder.layer.1.output.dense.bias" in dec
    assert "bert.img_embeddings.img_embeddings.weight" in dec and "bert.pooler.dense.weight" in dec
    assert "cls.predictions.bias" in dec


def test_mask_descriptors_know_when_padding_is_invisible():
    """data.MaskDesc.packable: padding removal is offered for the full / seq2seq / 1-D families only, per batch."""
    n_ids = torch.tensor([5, 9, 3])
    for fam, ok in (("full", True), ("s2s", True), ("1d", True), ("bar", False), ("noncross", False)):
        d = mv.data.MaskDesc.make(fam, 4, 10, n_ids)
        assert d.packable() is ok and d.host_desc().tolist() == [[mv.data.FAMILY_ID[fam], 6, 6 + int(n)] for n in n_ids]
    mixed = mv.data.MaskDesc.make(["full", "s2s", "bar"], 4, 10, n_ids)
    assert not mixed.packable() and mixed[:2].packable() and len(mixed[1:]) == 2
    # the closed forms behind the claim: in the packable families no valid query row has a visible key at or after vl
    for fam in ("full", "s2s", "1d"):
        for n in (1, 4, 11):
            m = D.mask_predicate(fam, 4, 10, n)
            vl = 6 + n
            rows = m[None, :] if m.ndim == 1 else m[:vl]
            assert not rows[:, vl:].any(), (fam, n)
    for fam in ("bar", "noncross"):
        m = D.mask_predicate(fam, 4, 10, 4)
        assert m[:10, 10:].any()                          # valid queries do see padding there


def test_region_encoder_container_matches_torchvision_layout():
    """ImageEncoder_cnn holds exactly the parameters / buffers of torchvision's resnet50 children()[:-2] under the
    reference's `model.<idx>` names (models/image.py:50-52), all frozen (cxrbert_origin.py:66-70 unfreezes nothing)."""
    enc = mv.ImageEncoder_cnn(num_image_embeds=9)
    sd = enc.state_dict()
    convs = [k for k in sd if k.endswith(".weight") and sd[k].dim() == 4]
    assert len(convs) == 53 and sum(v.numel() for k, v in sd.items() if sd[k].dim() in (1, 4) and "running" not in k) == 23508032
    assert sd["model.0.weight"].shape == (64, 3, 7, 7) and sd["model.4.0.downsample.0.weight"].shape == (256, 64, 1, 1)
    assert sd["model.5.0.conv2.weight"].shape == (128, 128, 3, 3) and enc.model[5][0].conv2.stride == (2, 2)
    assert sd["model.7.2.bn3.running_var"].shape == (2048,) and "model.1.num_batches_tracked" in sd
    assert not any(p.requires_grad for p in enc.parameters())
    from oracle import resnet_oracle as R
    y = R.trunk({k: v.float() for k, v in sd.items()}, torch.zeros(1, 3, 64, 64), training=False)
    assert y.shape == (1, 2048, 2, 2)
