"""The oracle (oracle/*.py, a CPU restatement) against the golden vectors that
oracle/gen_golden.py produced by running the REFERENCE itself (imported from
/root/reference in the build container).  Pins the oracle; CPU only."""
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import cxrbert_oracle as O
from oracle import data_oracle as D
from oracle import synth


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    meta = json.loads(str(z["meta"])) if "meta" in z else None
    return z, meta


def _oracle_inputs(z, meta):
    cfg = O.OracleConfig(**meta["cfg"])
    batch = synth.make_batch(cfg, meta["B"], meta["N"], meta["S"], meta["family"], seed=meta["seed"])
    # the committed integer inputs are authoritative; the generator must reproduce them bit-exactly
    for k in ("cls_tok", "input_txt", "segment", "img_pos", "sep_tok", "txt_labels", "is_aligned", "n_ids"):
        assert np.array_equal(batch[k], z["in_" + k]), k
    assert np.array_equal(D.pack_bits(batch["attn_mask"]), z["in_mask_bits"])
    P = O.make_params(cfg, seed=meta["seed"])
    return cfg, P, {k: torch.from_numpy(v) for k, v in batch.items()}


# ------------------------------------------------------------------ integer side: bit-exact
def test_mask_families_bit_exact(golden_dir):
    z = np.load(os.path.join(golden_dir, "masks.npz"))
    cases = json.loads(str(z["cases"]))
    assert len(cases) == 120
    fams = set()
    for c in cases:
        fams.add(c["family"])
        want = z["bits_" + c["tag"]]
        built = D.build_mask(c["family"], c["N"], c["S"], c["n_ids"])
        closed = D.mask_predicate(c["family"], c["N"], c["S"], c["n_ids"])
        assert built.ndim == c["mask_ndim"]
        assert np.array_equal(D.pack_bits(built), want), c
        assert np.array_equal(built, closed), c
        assert c["cls"] == D.CLS and c["sep"] == D.SEP
    assert fams == set(D.FAMILIES)


def test_label_id_segment_layout(golden_dir):
    z = np.load(os.path.join(golden_dir, "masks.npz"))
    for c in json.loads(str(z["cases"])):
        ids, lab, seg = z["ids_" + c["tag"]], z["labels_" + c["tag"]], z["segment_" + c["tag"]]
        N, S, n = c["N"], c["S"], c["n_ids"]
        assert ids.shape == (S + 1,) and lab.shape == (S + N + 3,) and seg.shape == (S + 1,)
        assert ids[n - 1] == D.SEP and (ids[n:] == D.PAD).all() and (seg == 1).all()
        assert (lab[:N + 2] == -100).all() and (lab[N + 2 + n - 1:] == -100).all()
        # our assembler reproduces the layout from the (already corrupted) ids + text labels
        i2, l2, s2, n2 = D.assemble_sample(ids[:n - 1], lab[N + 2:N + 2 + n - 1], N, S)
        assert np.array_equal(i2, ids) and np.array_equal(l2, lab) and np.array_equal(s2, seg) and n2 == n


def test_random_word_known_answers(golden_dir):
    z = np.load(os.path.join(golden_dir, "masks.npz"))
    for c in json.loads(str(z["rw_cases"])):
        k = c["k"]
        random.seed(c["seed"])
        tok, lab = D.random_word(list(z[f"rw_in_{k:02d}"]), random)
        assert tok == list(z[f"rw_tok_{k:02d}"]) and lab == list(z[f"rw_lab_{k:02d}"])
        assert any(l != -100 for l in lab)


def test_pack_bits_layout():
    m = np.zeros((3, 70), np.int64)
    m[0, 0] = m[1, 33] = m[2, 69] = 1
    w = D.pack_bits(m)
    assert w.shape == (3, 3) and w.dtype == np.uint32
    assert w[0, 0] == 1 and w[1, 1] == 2 and w[2, 2] == (1 << 5)


# ------------------------------------------------------------------ float side: forward, losses, grads
FWD_TOL = 2e-5   # fp32 CPU vs fp32 CPU, different op order only


@pytest.mark.parametrize("name", ["c1_full", "c1_s2s", "c1_bar", "c1_noncross", "c1_1d", "c1v1k_full",
                                  "c1v1k_bar_ragged", "c1v1k_nopos"])
def test_forward_matches_reference(golden_dir, name):
    z, meta = _load(golden_dir, name + ".npz")
    cfg, P, b = _oracle_inputs(z, meta)
    with torch.no_grad():
        hid, pooled = O.encode(P, cfg, b["cls_tok"], b["input_txt"], b["attn_mask"], b["segment"], b["img_feats"],
                               b["img_pos"], b["sep_tok"])
        mlm, itm = O.heads(P, cfg, hid, pooled)
        ml, il = O.losses(mlm, itm, b["txt_labels"], b["is_aligned"])
    assert np.abs(hid.numpy() - z["hidden"]).max() < FWD_TOL
    assert np.abs(pooled.numpy() - z["pooled"]).max() < FWD_TOL
    assert np.abs(itm.numpy() - z["itm"]).max() < FWD_TOL
    cols = torch.from_numpy(z["cols"].astype(np.int64))
    assert np.abs(mlm[..., cols].numpy() - z["logits_cols"]).max() < FWD_TOL
    assert np.abs(torch.logsumexp(mlm, -1).numpy() - z["lse"]).max() < 1e-4
    if "mlm" in z:
        assert np.abs(mlm.numpy() - z["mlm"]).max() < FWD_TOL
    assert abs(float(ml) - float(z["mlm_loss"])) < 1e-5 and abs(float(il) - float(z["itm_loss"])) < 1e-5


@pytest.mark.parametrize("name", ["c1_full", "c1_s2s", "c1v1k_full", "c1v1k_bar_ragged", "c1v1k_nopos"])
def test_gradients_match_reference(golden_dir, name):
    z, meta = _load(golden_dir, name + ".npz")
    cfg, P, b = _oracle_inputs(z, meta)
    for w in P.values():
        w.requires_grad_(True)
    mlm, itm = O.forward(P, cfg, b["cls_tok"], b["input_txt"], b["attn_mask"], b["segment"], b["img_feats"],
                         b["img_pos"], b["sep_tok"])
    ml, il = O.losses(mlm, itm, b["txt_labels"], b["is_aligned"])
    (ml + il).backward()
    names = [str(n) for n in z["grad_names"]]
    assert names == list(P.keys())
    for i, k in enumerate(names):
        g = P[k].grad
        ref_norm = float(z["grad_norms"][i])
        assert abs(float(g.double().norm()) - ref_norm) <= 1e-4 * max(ref_norm, 1e-6) + 1e-7, k
        got = g.reshape(-1)[torch.from_numpy(z["grad_idx"][i])].numpy()
        assert np.abs(got - z["grad_vals"][i]).max() <= 1e-5 * max(1.0, np.abs(z["grad_vals"][i]).max()) + 1e-7, k
    # [PAD]/[UNK]/[CLS]/[SEP]/[MASK] rows of the tied matrix (the [PAD] row has no look-up gradient: padding_idx)
    dE = P["enc.txt_embeddings.word_embeddings.weight"].grad[[0, 100, 101, 102, 103]].numpy()
    assert np.abs(dE - z["dE_special_rows"]).max() <= 2e-6 * max(1.0, np.abs(z["dE_special_rows"]).max())


@pytest.mark.parametrize("name", ["base_s2s", "base_full", "base_full_b4", "base_noncross", "base768_s2s"])
def test_bert_base_matches_reference(golden_dir, name):
    """BERT-base at the scale of BASELINE.json configs 3 / 2 / 4 / 5 (L = 512 seq2seq, bidirectional B=2 and B=4 ragged,
    non-cross; L = 768 seq2seq with max_position_embeddings 768)."""
    z, meta = _load(golden_dir, name + ".npz")
    cfg, P, b = _oracle_inputs(z, meta)
    with_grads = "grad_names" in z
    if with_grads:
        for w in P.values():
            w.requires_grad_(True)
    with torch.set_grad_enabled(with_grads):
        mlm, itm = O.forward(P, cfg, b["cls_tok"], b["input_txt"], b["attn_mask"], b["segment"], b["img_feats"],
                             b["img_pos"], b["sep_tok"])
        ml, il = O.losses(mlm, itm, b["txt_labels"], b["is_aligned"])
    cols = torch.from_numpy(z["cols"].astype(np.int64))
    assert np.abs(mlm.detach()[..., cols].numpy() - z["logits_cols"]).max() < 1e-4
    assert np.abs(itm.detach().numpy() - z["itm"]).max() < 1e-4
    assert abs(float(ml) - float(z["mlm_loss"])) < 1e-4 and abs(float(il) - float(z["itm_loss"])) < 1e-4
    if "lab_rows" in z:          # the labelled rows' logits (what train_origin.py:120-126's CrossEntropyLoss(ignore_index=-100) reads)
        rows = torch.from_numpy(z["lab_rows"].astype(np.int64))
        assert np.array_equal(rows.numpy(), np.nonzero(b["txt_labels"].reshape(-1).numpy() != -100)[0])
        flat = mlm.detach().reshape(-1, cfg.vocab_size)[rows]
        assert np.abs(flat[:, torch.from_numpy(z["lab_cols"].astype(np.int64))].numpy() - z["lab_logits_cols"]).max() < 1e-4
        assert np.abs(flat.gather(1, torch.from_numpy(z["lab_ids"].astype(np.int64)).view(-1, 1)).reshape(-1).numpy() - z["lab_logit_at_label"]).max() < 1e-4
    if with_grads:
        (ml + il).backward()
        for i, k in enumerate(str(n) for n in z["grad_names"]):
            ref_norm = float(z["grad_norms"][i])
            assert abs(float(P[k].grad.double().norm()) - ref_norm) <= 2e-4 * max(ref_norm, 1e-6) + 1e-7, k
            got = P[k].grad.reshape(-1)[torch.from_numpy(z["grad_idx"][i])].numpy()
            assert np.abs(got - z["grad_vals"][i]).max() <= 2e-5 * max(1.0, np.abs(z["grad_vals"][i]).max()) + 1e-7, k


# ------------------------------------------------------------------ optimizer KAT
def test_hf_adamw_known_answer(golden_dir):
    z = np.load(os.path.join(golden_dir, "adamw.npz"))
    lr, b1, b2, eps, wd = [float(x) for x in z["hyper"]]
    p = torch.from_numpy(z["p0"].copy())            # float64: compare the algorithm, not fp32 rounding
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for t in range(1, 4):
        O.hf_adamw_step(p, torch.from_numpy(z["grads"][t - 1]), m, v, t, lr=lr, b1=b1, b2=b2, eps=eps, wd=wd)
        assert np.abs(p.numpy() - z["p"][t - 1]).max() < 1e-14
    assert np.abs(m.numpy() - z["m"]).max() < 1e-15 and np.abs(v.numpy() - z["v"]).max() < 1e-15


def test_param_count_and_flops():
    assert O.num_params(O.CONFIGS["base"]) == 111_680_060
    assert O.num_params(O.CONFIGS["c1"]) == 4_695_740
    assert abs(3 * O.flops_fwd_per_sample(O.CONFIGS["base"], 512, 36) / 1e9 - 364.0760) < 1e-3
    assert abs(3 * O.flops_fwd_per_sample(O.CONFIGS["base768"], 768, 100) / 1e9 - 568.2895) < 1e-3
    assert abs(3 * O.flops_fwd_per_sample(O.CONFIGS["c1"], 64, 16) / 1e9 - 1.6954) < 1e-3


# ------------------------------------------------------------------ region encoder
def test_resnet_oracle_against_torchvision_fixture_when_one_exists(golden_dir):
    """oracle/resnet_oracle.py restates torchvision's ResNet-50 trunk (models/image.py:46-53); torchvision is in neither the reference
    tree nor this image, so the restatement is UNPINNED until tests/golden/resnet50.npz -- written by oracle/gen_resnet_golden.py
    wherever torchvision is installed -- exists.  With the fixture: eval- and train-mode feature maps and the updated running
    statistics against torchvision's own."""
    path = os.path.join(golden_dir, "resnet50.npz")
    if not os.path.exists(path):
        pytest.skip("parity unpinned: no torchvision here and no tests/golden/resnet50.npz (python oracle/gen_resnet_golden.py where torchvision exists)")
    import medvill_amd as mv
    from oracle import gen_resnet_golden as G
    from oracle import resnet_oracle as R
    z = np.load(path)
    enc = mv.ImageEncoder_cnn(num_image_embeds=4)
    sd = G.fill_state({k[len("model."):]: v for k, v in enc.state_dict().items()}, seed=int(z["w_seed"]))
    assert sorted(sd) == [str(k) for k in z["keys"]]
    sd = {"model." + k: v.clone() for k, v in sd.items()}
    x = G.make_input(int(z["in_seed"]))
    assert np.abs(R.trunk(sd, x, training=False).numpy() - z["out_eval"]).max() < 1e-4 * max(1.0, np.abs(z["out_eval"]).max())
    assert np.abs(R.trunk(sd, x, training=True).numpy() - z["out_train"]).max() < 1e-3 * max(1.0, np.abs(z["out_train"]).max())
    for p_ in G.PROBES:
        assert np.abs(sd[f"model.{p_}.running_mean"].numpy() - z[f"rm_{p_}"]).max() < 1e-4
        assert np.abs(sd[f"model.{p_}.running_var"].numpy() - z[f"rv_{p_}"]).max() < 1e-4
