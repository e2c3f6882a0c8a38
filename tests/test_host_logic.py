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


def _manifest():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "state_manifest.json")) as f:
        return json.load(f)


def test_state_dict_keys_and_shapes_equal_the_reference_manifest():
    """tests/golden/state_manifest.json is the key -> shape list of the REAL reference CXRBERT.state_dict() at BERT-base
    (oracle/gen_golden.py --only manifest imports the reference to write it).  This model's state_dict() must have exactly those keys
    and shapes: what the reference's trainer saves (train_origin.py:254-266) is what loads here, and the other way round."""
    man = _manifest()
    c = man["config"]
    cfg = mv.ModelConfig(vocab_size=c["vocab_size"], hidden=c["hidden_size"], layers=c["num_hidden_layers"], heads=c["num_attention_heads"],
                         intermediate=c["intermediate_size"], max_pos=c["max_position_embeddings"])
    lay, _ = mv.param_layout(cfg)
    ours = {k: list(s) for k, (_, s) in lay.items()}
    for alias, canon in ALIASES.items():
        ours[alias] = ours[canon]
    ref = {k: v[0] for k, v in man["keys"].items()}
    assert set(ours) == set(ref), (sorted(set(ours) - set(ref))[:5], sorted(set(ref) - set(ours))[:5])
    assert all(ours[k] == ref[k] for k in ref)
    # the CNN trunk's keys (unpinned: torchvision's published names) are the ones ImageEncoder_cnn holds
    enc = mv.ImageEncoder_cnn(num_image_embeds=9)
    cnn = {"enc.img_encoder." + k: list(v.shape) for k, v in enc.state_dict().items()}
    assert cnn == man["img_encoder_keys_unpinned"]["keys"]


def test_a_checkpoint_written_with_the_reference_keys_loads(tmp_path):
    """A pytorch_model.bin holding EVERY key of the reference manifest at a small geometry -- the tied / shared tensors stored under all
    their names, the `position_ids` buffer older transformers versions save, the CNN trunk -- loads strictly (nothing missing, nothing
    unexpected), lands in the right parameters, and a checkpoint saved from here has the manifest's key set again."""
    import json
    man = _manifest()
    c = man["config"]
    dims = {c["vocab_size"]: 1024, c["hidden_size"]: 128, c["intermediate_size"]: 512, c["max_position_embeddings"]: 128}
    g = torch.Generator().manual_seed(5)
    sd = {}
    for k, (shape, dt) in man["keys"].items():
        l = k.split("layer.")[1].split(".")[0] if "layer." in k else None
        if l is not None and int(l) >= 2:
            continue                                          # TINY has two layers
        sd[k] = torch.randn([dims.get(d, d) for d in shape], generator=g)
    for alias, canon in ALIASES.items():                      # the reference stores one tensor under several names
        sd[alias] = sd[canon]
    sd["enc.txt_embeddings.position_ids"] = torch.arange(128).view(1, -1)
    enc = mv.ImageEncoder_cnn(num_image_embeds=9)
    cnn_sd = {k: torch.randn(v.shape, generator=g) if v.dtype.is_floating_point else v.clone() for k, v in enc.state_dict().items()}
    sd.update({"enc.img_encoder." + k: v for k, v in cnn_sd.items()})
    d = tmp_path / "ref_ckpt"
    os.makedirs(d)
    torch.save(sd, d / "pytorch_model.bin")
    with open(d / "config.json", "w") as f:
        json.dump(TINY, f)
    m = mv.CXRBERT.from_pretrained(str(d), device="cpu", img_encoder=enc)
    r = m.load_state_dict(sd, strict=True)
    assert r.missing_keys == [] and r.unexpected_keys == []
    out = m.state_dict()
    for k in sd:
        if "position_ids" in k:
            continue
        assert torch.equal(out[k].float().cpu(), sd[k].float()), k
    assert {k for k in out if not k.startswith("enc.img_encoder.")} == \
        {k for k in man["keys"] if "layer." not in k or int(k.split("layer.")[1].split(".")[0]) < 2}


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


# ------------------------------------------------------------------ host-side every-entry mask check (mv_mask_verify_host)
def _golden_mask_cases():
    import json
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "masks.npz"))
    return z, json.loads(str(z["cases"]))


def _unpack_bits(words, L):
    """inverse of oracle.data_oracle.pack_bits: uint32 [..., W] -> 0/1 [..., L]"""
    b = (words[..., :, None] >> np.arange(32, dtype=np.uint32)) & 1
    return b.reshape(words.shape[:-1] + (-1,))[..., :L]


def test_host_mask_check_accepts_every_reference_built_matrix_and_finds_any_flipped_entry():
    """mv_mask_verify_host is the host twin of mv_mask_build: the reference Dataset's own matrices (tests/golden/masks.npz, built by
    CXRDataset.__getitem__ for every family over a grid of geometries) agree with the closed form of their descriptors in every
    entry, and flipping ANY single entry is reported with its exact index."""
    from medvill_amd import hip_ops as ops
    z, cases = _golden_mask_cases()
    rng = np.random.default_rng(0)
    seen = set()
    for c in cases:
        L = c["N"] + c["S"] + 3
        bits = z[f"bits_{c['tag']}"]
        m = _unpack_bits(bits, L).astype(np.int64)
        assert m.ndim == c["mask_ndim"]
        desc = mv.data.MaskDesc.make(c["family"], c["N"], c["S"], [c["n_ids"]]).host_desc()
        mt = torch.from_numpy(np.ascontiguousarray(m[None]))
        assert ops.mask_verify_host(mt, desc, threads=1) == -1, c
        idx = tuple(int(rng.integers(0, s)) for s in mt.shape)
        bad = mt.clone()
        bad[idx] ^= 1
        assert ops.mask_verify_host(bad, desc, threads=2) == int(np.ravel_multi_index(idx, mt.shape)), (c, idx)
        seen.add(c["family"])
    assert seen == {"full", "s2s", "bar", "noncross", "1d"}
    # a wrong descriptor (valid length off by one) is a mismatch too; descriptors out of range are refused
    c = next(c for c in cases if c["family"] == "full" and c["n_ids"] < c["S"])
    L = c["N"] + c["S"] + 3
    mt = torch.from_numpy(_unpack_bits(z[f"bits_{c['tag']}"], L).astype(np.int64)[None].copy())
    d = mv.data.MaskDesc.make("full", c["N"], c["S"], [c["n_ids"] + 1]).host_desc()
    assert ops.mask_verify_host(mt, d) == c["N"] + 2 + c["n_ids"]               # first row, first column past the true valid length
    d2 = d.clone()
    d2[0, 0] = 9
    with pytest.raises(RuntimeError):
        ops.mask_verify_host(mt, d2)


def test_trainer_checks_every_entry_of_every_batch_one_batch_ahead():
    """the DEFAULT policy proves the derived descriptors on every entry of every batch.  Row 300 of sample 17 of
    batch 5 (B = 64, L = 512: the benchmark geometry) is flipped -- the probes still recognise the family, the check names the entry,
    and only that batch falls back to the matrix.  Host logic only: the trainer object is built without its model."""
    from types import SimpleNamespace
    from medvill_amd.trainer import CXRBERT_Trainer
    B, N, S = 64, 36, 473
    L = N + S + 3
    tr = object.__new__(CXRBERT_Trainer)
    tr.args = SimpleNamespace(num_image_embeds=N)
    tr._init_mask_policy(tr.args)
    assert tr.verify_masks == "full"
    b = mv.data.synthetic_batch(30522, B, N, S, "full", seed=5, device="cpu")
    good = b["attn_mask"]
    bad = good.clone()
    bad[17, 300, 200] ^= 1
    feats = torch.zeros((B, N, 8))

    def tup(m):
        return (b["cls_tok"], b["input_txt"], b["txt_labels"], m, (feats, b["img_pos"]), b["segment"], b["is_aligned"], b["sep_tok"])
    loader = [tup(bad if i == 5 else good) for i in range(7)]
    results = []
    for i, tk in enumerate(tr._prefetch(loader, True)):
        assert tk["desc"] is not None and tk["check"] is not None          # recognised, and every batch gets the every-entry check
        results.append(tk["check"].result())
    assert results == [-1] * 5 + [(17 * L + 300) * L + 200, -1]
    # the opt-in "sampled" policy checks every entry of the first two batches and every 64th only (random probe rows otherwise);
    # training and evaluation batches are counted separately (eval batches must not shift the training schedule)
    tr2 = object.__new__(CXRBERT_Trainer)
    tr2.args = SimpleNamespace(num_image_embeds=N, verify_masks="sampled")
    tr2._init_mask_policy(tr2.args)
    sched = [tr2._full_check_now(True) for _ in range(130)]
    assert [i + 1 for i, f in enumerate(sched) if f] == [1, 2, 64, 128]
    assert [tr2._full_check_now(False) for _ in range(3)] == [True, True, False]
    assert tr2._full_check_now(True) is False and tr2._mask_batches == {True: 131, False: 3}
    with pytest.raises(ValueError):
        tr2._init_mask_policy(SimpleNamespace(verify_masks="sometimes"))


def test_numa_binding_reads_sysfs_and_never_raises(tmp_path, monkeypatch):
    """dist.bind_to_gpu_numa: rank -> PCI address from the KFD topology (GPU agents in enumeration order), -> NUMA node -> CPU list.
    Driven here over a fake sysfs tree; on a host without the files it reports why and binds nothing."""
    from medvill_amd import dist as D2
    root = tmp_path
    for i, (simd, loc) in enumerate([(0, 0), (0, 0), (1024, 0x0500), (1024, 0x8500)]):          # two CPU agents, two GPUs
        d = root / "sys/class/kfd/kfd/topology/nodes" / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nlocation_id {loc}\ndomain 0\n")
    for bdf, node in (("0000:05:00.0", 0), ("0000:85:00.0", 1)):
        d = root / "sys/bus/pci/devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
    allowed = sorted(os.sched_getaffinity(0))
    half = max(1, len(allowed) // 2)
    lists = {0: allowed[:half], 1: allowed[half:] or allowed[:1]}
    for node, cpus in lists.items():
        d = root / f"sys/devices/system/node/node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(str(c) for c in cpus) + "\n")
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    assert D2.gpu_numa_node(0, str(root)) == (0, "0000:05:00.0") and D2.gpu_numa_node(1, str(root)) == (1, "0000:85:00.0")
    assert D2.gpu_numa_node(2, str(root)) == (None, None)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")
    assert D2.gpu_numa_node(0, str(root)) == (1, "0000:85:00.0")
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    try:
        info = D2.bind_to_gpu_numa(1, str(root))
        assert info["bound"] and info["numa_node"] == 1 and set(os.sched_getaffinity(0)) == set(lists[1])
    finally:
        os.sched_setaffinity(0, allowed)
    assert D2.bind_to_gpu_numa(0, str(tmp_path / "nothing_here"))["bound"] is False          # no sysfs: says why, binds nothing
    monkeypatch.setenv("MV_NUMA_BIND", "0")
    assert D2.bind_to_gpu_numa(0, str(root)) == {"local_rank": 0, "bound": False, "why": "MV_NUMA_BIND=0"}
    env = D2.rank_environment({"x": 1})
    assert len(env) == 1 and env[0]["rank"] == 0 and env[0]["x"] == 1 and "GPU_MAX_HW_QUEUES" in env[0]


def test_mlm_itm_loss_on_plain_tensors_is_the_two_cross_entropies():
    """medvill_amd.losses.mlm_itm_loss with ordinary logits: exactly train_origin.py:120-126 (CrossEntropyLoss(ignore_index=-100) on the
    transposed MLM logits + CrossEntropyLoss() on the ITM logits; a task switched off contributes nothing)."""
    g = torch.Generator().manual_seed(4)
    mlm, itm = torch.randn(3, 7, 19, generator=g, requires_grad=True), torch.randn(3, 2, generator=g, requires_grad=True)
    labels = torch.full((3, 7), -100)
    labels[0, 2], labels[1, 5], labels[2, 0] = 4, 18, 0
    aligned = torch.tensor([1, 0, 1])
    ce_m, ce_i = torch.nn.CrossEntropyLoss(ignore_index=-100), torch.nn.CrossEntropyLoss()
    want = ce_i(itm, aligned) + ce_m(mlm.transpose(1, 2), labels)
    got = mv.losses.mlm_itm_loss(mlm, itm, labels, aligned)
    assert torch.allclose(got, want, atol=1e-6)
    got.backward()
    assert mlm.grad is not None and float(mlm.grad[0, 0].abs().sum()) == 0.0 and float(mlm.grad[0, 2].abs().sum()) > 0
    assert torch.allclose(mv.losses.mlm_itm_loss(mlm, itm, labels, aligned, mlm_task=False), ce_i(itm, aligned), atol=1e-6)
    assert torch.allclose(mv.losses.mlm_itm_loss(mlm, itm, labels, aligned, itm_task=False), ce_m(mlm.transpose(1, 2), labels), atol=1e-6)


def test_torch_mask_recogniser_agrees_with_the_trainers_host_recogniser():
    """data.descriptors_from_dense (torch, any device: the model API's recogniser for a `.to(device)`-ed mask, cxrbert.CXRBERT._mask_descriptors)
    states the same hypothesis as CXRBERT_Trainer._recognise_masks (numpy, host) -- on the reference Dataset's own matrices
    (tests/golden/masks.npz), on mixed batches at the benchmark geometry, and on matrices outside the families."""
    from types import SimpleNamespace
    from medvill_amd.trainer import CXRBERT_Trainer
    tr = object.__new__(CXRBERT_Trainer)
    tr.args = SimpleNamespace()
    tr._init_mask_policy(tr.args)
    z, cases = _golden_mask_cases()
    for c in cases:
        N, S = c["N"], c["S"]
        L = N + S + 3
        m = torch.from_numpy(np.ascontiguousarray(_unpack_bits(z[f"bits_{c['tag']}"], L).astype(np.int64)[None]))
        ids = torch.zeros((1, S + 1), dtype=torch.int64)
        ids[0, :c["n_ids"]] = 7                                  # n_ids counts the text [SEP]
        got = mv.data.descriptors_from_dense(m, ids, N)
        want = mv.data.MaskDesc.make(c["family"], N, S, [c["n_ids"]]).host_desc()
        assert got is not None and bool(got[1]) and torch.equal(got[0], want), c
        host = tr._recognise_masks(m, ids, N)
        assert host is not None and torch.equal(host.host_desc(), got[0]), c
    B, N, S = 64, 36, 473
    for fam in ("mixed", "full", "s2s", "bar", "noncross", "1d"):
        b = mv.data.synthetic_batch(30522, B, N, S, fam, seed=11, device="cpu")
        desc, ok = mv.data.descriptors_from_dense(b["attn_mask"], b["input_txt"], N)
        assert bool(ok) and torch.equal(desc, b["attn_desc"].host_desc()), fam
    b = mv.data.synthetic_batch(30522, 4, N, S, "full", seed=12, device="cpu")
    odd = b["attn_mask"].clone()
    odd[2, 0, 5] = 0                                             # a probe row that fits no family: no hypothesis for the batch
    assert not bool(mv.data.descriptors_from_dense(odd, b["input_txt"], N)[1]) and tr._recognise_masks(odd, b["input_txt"], N) is None
    inner = b["attn_mask"].clone()
    inner[1, 300, 200] ^= 1                                      # off the probes: still a hypothesis -- the every-entry check is what rejects it
    assert bool(mv.data.descriptors_from_dense(inner, b["input_txt"], N)[1])
    assert mv.data.descriptors_from_dense(b["attn_mask"], b["input_txt"][:, :-1], N) is None          # shapes rule it out
    assert mv.data.descriptors_from_dense(b["attn_mask"].to(torch.int32), b["input_txt"], N) is None


def test_flat_adamw_takes_the_whole_parameter_set_of_one_model_and_tracks_foreign_writes():
    """medvill_amd.optim.AdamW (train_origin.py:60's optimizer line on the flat buffers), host logic only: anything but ALL parameters of ONE
    CXRBERT is refused; the version-counter bookkeeping that lets a forward skip the 16-bit weight refresh notices in-place writes by others."""
    cfg = mv.ModelConfig(vocab_size=64, hidden=64, layers=1, heads=2, intermediate=64, max_pos=32, img_hidden=8)
    model = mv.CXRBERT(cfg, None, dtype=torch.float32, device="cpu")
    params = list(model.parameters())
    with pytest.raises(ValueError, match="whole flat parameter buffer"):
        mv.optim.AdamW(params[:-1], lr=1e-3)
    with pytest.raises(ValueError, match="not the Parameters"):
        mv.optim.AdamW(torch.nn.Linear(2, 2).parameters(), lr=1e-3)
    with pytest.raises(ValueError, match="single parameter group"):
        mv.optim.AdamW([{"params": params}], lr=1e-3)
    with pytest.raises(ValueError, match="do not belong"):
        mv.optim.AdamW(params + list(torch.nn.Linear(2, 2).parameters()), lr=1e-3)
    opt = mv.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
    assert len(opt.param_groups) == 1 and len(opt.param_groups[0]["params"]) == len(params) and opt.param_groups[0]["eps"] == 1e-6
    assert opt.step() is None and opt._t == 0                     # no gradients anywhere: nothing happens (and no kernel is needed)
    assert model._params_dirty()                                  # nobody has vouched for the 16-bit copies yet
    model._opt_versions = sum(p._version for p in model._plist)   # (what step() records after its kernel has written them)
    assert not model._params_dirty()
    with torch.no_grad():
        params[3].mul_(1.0)
    assert model._params_dirty()
    params[0].grad = torch.zeros_like(params[0])
    with pytest.raises(RuntimeError, match="some Parameters have a gradient"):
        opt.step()
