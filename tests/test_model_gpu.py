"""GPU parity of the whole hot path (HIP engine behind the CXRBERT / TrainStep interface) against
(a) the golden vectors produced by the reference itself and (b) the CPU oracle on the same
seeded inputs.  Tolerances are the ones BASELINE.json's north_star states: logits within 1e-3
(fp32 path) / 1e-2 (bf16 path) of the reference; mask handling bit-exact (test_kernels_gpu)."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import medvill_amd as mv                                   # noqa: E402
from oracle import cxrbert_oracle as O                      # noqa: E402
from oracle import synth                                    # noqa: E402

DEV = "cuda"
FP32_TOL = 1e-3      # north_star: logits within 1e-3 fp32
BF16_TOL = 1e-2      # north_star: logits within 1e-2 bf16


def cfg_dict(c: O.OracleConfig):
    return dict(vocab_size=c.vocab_size, hidden_size=c.hidden, num_hidden_layers=c.layers, num_attention_heads=c.heads,
                intermediate_size=c.intermediate, max_position_embeddings=c.max_pos, layer_norm_eps=c.ln_eps)


def load_case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    cfg = O.OracleConfig(**meta["cfg"])
    batch = synth.make_batch(cfg, meta["B"], meta["N"], meta["S"], meta["family"], seed=meta["seed"])
    P = O.make_params(cfg, seed=meta["seed"])
    return z, meta, cfg, P, {k: torch.from_numpy(v) for k, v in batch.items()}


def make_model(cfg, P, dtype, fwd_operand=None, grad_operand=None):
    # the two fields ImageBertEmbeddings reads from the reference's argparse namespace (cxrbert_origin.py:19,27-31)
    args = None if cfg.img_position else SimpleNamespace(img_postion=False, dropout_prob=0.1)
    m = mv.CXRBERT(cfg_dict(cfg), args, dtype=dtype, device=DEV, fwd_operand=fwd_operand, grad_operand=grad_operand)
    m.load_state_dict(P, strict=True)
    m.eval()            # parity runs with dropout off, like the golden vectors (the reference in .eval())
    return m


def fwd(model, b):
    return model(b["cls_tok"].to(DEV), b["input_txt"].to(DEV), b["attn_mask"].to(DEV), b["segment"].to(DEV),
                 (b["img_feats"].to(DEV), b["img_pos"].to(DEV)), b["sep_tok"].to(DEV))


CASES = ["c1_full", "c1_s2s", "c1_bar", "c1_noncross", "c1_1d", "c1v1k_full", "c1v1k_bar_ragged", "c1v1k_nopos"]


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dtype,tol", [(torch.float32, FP32_TOL), (torch.bfloat16, BF16_TOL)])
def test_forward_matches_reference_golden(golden_dir, name, dtype, tol):
    z, meta, cfg, P, b = load_case(golden_dir, name)
    model = make_model(cfg, P, dtype)
    with torch.no_grad():
        hid, pooled, _ = model.enc(b["cls_tok"].to(DEV), b["input_txt"].to(DEV), b["attn_mask"].to(DEV), b["segment"].to(DEV),
                                   (b["img_feats"].to(DEV), b["img_pos"].to(DEV)), b["sep_tok"].to(DEV))
        mlm, itm = fwd(model, b)
    mlm, itm = mlm.float().cpu(), itm.float().cpu()
    assert mlm.shape == (meta["B"], meta["N"] + meta["S"] + 3, cfg.vocab_size) and itm.shape == (meta["B"], 2)
    cols = torch.from_numpy(z["cols"].astype(np.int64))
    e_log = float(np.abs(mlm[..., cols].numpy() - z["logits_cols"]).max())
    e_itm = float(np.abs(itm.numpy() - z["itm"]).max())
    e_hid = float(np.abs(hid.float().cpu().numpy() - z["hidden"]).max())
    print(f"{name} {dtype}: |dlogits|max={e_log:.2e} |ditm|={e_itm:.2e} |dhidden|={e_hid:.2e}")
    assert e_log < tol and e_itm < tol
    assert e_hid < tol                 # hidden states are O(3); the 16-bit path returns them in the f16 encoding
    if "mlm" in z:
        assert float(np.abs(mlm.numpy() - z["mlm"]).max()) < tol
    assert float(np.abs(torch.logsumexp(mlm, -1).numpy() - z["lse"]).max()) < tol
    # losses through the reference's own criterion calls (train_origin.py:120-126)
    ml, il = O.losses(mlm, itm, b["txt_labels"], b["is_aligned"])
    assert abs(float(ml) - float(z["mlm_loss"])) < tol and abs(float(il) - float(z["itm_loss"])) < tol


@pytest.mark.parametrize("name", ["c1_full", "c1_s2s", "c1v1k_full", "c1v1k_bar_ragged", "c1v1k_nopos"])
@pytest.mark.parametrize("dtype,rtol", [(torch.float32, 1e-3), (torch.bfloat16, 4e-2)])
def test_dropin_backward_matches_reference_gradients(golden_dir, name, dtype, rtol):
    """loss.backward() through CXRBERT.forward, exactly as train_origin.py:106-130 does."""
    z, meta, cfg, P, b = load_case(golden_dir, name)
    model = make_model(cfg, P, dtype)
    mlm, itm = fwd(model, b)
    ce_m, ce_i = torch.nn.CrossEntropyLoss(ignore_index=-100), torch.nn.CrossEntropyLoss()
    loss = ce_m(mlm.transpose(1, 2), b["txt_labels"].to(DEV)) + ce_i(itm, b["is_aligned"].to(DEV))
    loss.backward()
    names = [str(n) for n in z["grad_names"]]
    grads = dict(model.named_parameters())
    worst = 0.0
    gmax = float(z["grad_norms"].max())
    for i, k in enumerate(names):
        g = grads[k].grad.float().cpu()
        ref_norm = float(z["grad_norms"][i])
        got = g.reshape(-1)[torch.from_numpy(z["grad_idx"][i])].numpy()
        # gradients that are analytically zero (e.g. the key bias: softmax is shift-invariant) are pure rounding
        # noise in the reference too (norm ~1e-8): compare those against an absolute floor tied to the largest gradient
        # In bf16 the same holds for cancellation-dominated tensors: with random-init weights `pooled` is almost
        # the same vector for every sample, so d(itm/pooler) = sum_b ditm[b] (x) pooled[b] cancels to ~1% of its terms
        # and is decided by the low bits that bf16 STORAGE of `pooled` drops (c1_full: |g| = 9e-3 vs 0.9 in c1v1k_full).
        floor = (1e-5 if dtype == torch.float32 else 3e-2) * gmax
        efloor = floor if dtype == torch.float32 else 8 * floor / np.sqrt(g.numel())
        scale = max(ref_norm / np.sqrt(g.numel()), float(np.abs(z["grad_vals"][i]).max()), efloor)
        e1 = abs(float(g.double().norm()) - ref_norm) / max(ref_norm, floor)
        e2 = float(np.abs(got - z["grad_vals"][i]).max()) / scale
        worst = max(worst, e1, e2 * 0.25)
        assert e1 < rtol, (k, e1, ref_norm)
        assert e2 < 4 * rtol + 1e-6, (k, e2)
    dE = grads["enc.txt_embeddings.word_embeddings.weight"].grad[[0, 100, 101, 102, 103]].float().cpu().numpy()
    ref = z["dE_special_rows"]
    assert np.abs(dE - ref).max() < (1e-4 if dtype == torch.float32 else 5e-2) * max(np.abs(ref).max(), 1e-6)
    print(f"{name} {dtype}: worst grad deviation {worst:.2e}")


@pytest.mark.parametrize("dtype,rtol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
def test_lazy_logits_loss_equals_the_two_cross_entropies(golden_dir, dtype, rtol):
    """`model.lazy_logits = True` + `medvill_amd.losses.mlm_itm_loss(mlm, itm, txt_labels, is_aligned)` in place of
    the two CrossEntropyLoss calls of train_origin.py:120-126.  Same loss value as the reference's golden, same gradient of every parameter
    as the literal path (full logits -> torch cross-entropy -> backward), an upstream factor is honoured, the accuracy counters of
    train_origin.py:133-146 come out of `.stats`, and with mask DESCRIPTORS the forward runs on packed rows."""
    z, meta, cfg, P, b = load_case(golden_dir, "c1v1k_full")
    model = make_model(cfg, P, dtype)
    labels, aligned = b["txt_labels"].to(DEV), b["is_aligned"].to(DEV)
    tol = FP32_TOL if dtype == torch.float32 else BF16_TOL
    mlm, itm = fwd(model, b)
    ref = mv.losses.mlm_itm_loss(mlm, itm, labels, aligned)              # plain tensors: exactly the two torch cross-entropies
    assert abs(float(ref) - float(z["mlm_loss"]) - float(z["itm_loss"])) < 2 * tol
    ref.backward()
    g_ref = {n: p.grad.clone() for n, p in model.named_parameters()}
    n_correct = int(((mlm.argmax(-1) == labels) & (labels != -100)).sum())
    gmax = max(float(g.abs().max()) for g in g_ref.values())
    for how in ("matrix", "descriptors", "recognised"):
        # "recognised": the Dataset's matrix as the reference loop passes it (.to(device)); the lazy forward proves it equal to a closed
        # form on the device and runs on descriptors (16-bit path: packed rows)
        desc = how != "matrix"
        model.zero_grad()
        model.lazy_logits, model.recognise_masks = True, how == "recognised"
        bb = dict(b)
        if how == "descriptors":
            bb["attn_mask"] = mv.data.MaskDesc.make("full", meta["N"], meta["S"], b["n_ids"], DEV)
        lz, itm2 = fwd(model, bb)
        if how == "recognised":
            assert (model.n_masks_seen, model.n_masks_recognised) == ((1, 1) if dtype != torch.float32 else (0, 0))
        assert isinstance(lz, mv.cxrbert.LazyLogits) and lz.shape == tuple(mlm.shape) and not itm2.requires_grad
        assert float((itm2 - itm.detach()).abs().max()) < tol
        loss = mv.losses.mlm_itm_loss(lz, itm2, labels, aligned)
        assert abs(float(loss) - float(ref)) < (1e-5 if dtype == torch.float32 else 5e-3)
        assert (model.engine.S["cu"] is not None) == (desc and dtype != torch.float32)
        st = lz.stats.cpu()
        assert int(st[1]) == int((labels != -100).sum()) and int(st[4]) == meta["B"]
        if dtype == torch.float32:
            assert int(st[2]) == n_correct
        (2.0 * loss).backward()                                         # an upstream factor reaches every gradient
        for n, p in model.named_parameters():
            scale = max(float(g_ref[n].abs().max()), 1e-3 * gmax)
            assert float((p.grad - 2.0 * g_ref[n]).abs().max()) <= 2.0 * rtol * scale, (desc, n)
        if not desc or dtype == torch.float32:
            assert float((lz.materialize().float() - mlm.detach().float()).abs().max()) < tol
        else:
            with pytest.raises(RuntimeError, match="packed rows"):
                lz.materialize()
    model.zero_grad()
    model.recognise_masks = True
    lz, itm2 = fwd(model, b)                                            # (still lazy)
    with pytest.raises(RuntimeError, match="mlm_itm_loss"):
        lz.tok.sum().backward()                                         # a backward that never went through the loss is refused


def test_lazy_forward_proves_a_device_matrix_before_it_runs_on_descriptors(golden_dir):
    """The reference loop hands the model the Dataset's [B, L, L] matrix after `.to(device)` (train_origin.py:95-104).  Under lazy logits the
    model derives {family, n2, vl} on the device and runs on them ONLY when the matrix equals the closed form in every entry (mask words of
    mv_mask_pack against mv_mask_build): a recognised batch reproduces the descriptor run; one flipped entry off the probe lines
    keeps the matrix (and reproduces the un-recognising run); a label at a padded position is refused."""
    z, meta, cfg, P, b = load_case(golden_dir, "c1v1k_full")
    model = make_model(cfg, P, torch.bfloat16)
    model.lazy_logits = True
    labels, aligned = b["txt_labels"].to(DEV), b["is_aligned"].to(DEV)
    dense = b["attn_mask"].to(DEV)

    def run(mask, recognise=True, lab=labels):
        model.zero_grad()
        model.recognise_masks = recognise
        bb = dict(b)
        bb["attn_mask"] = mask
        lz, itm = fwd(model, bb)
        packed = model.engine.S["cu"] is not None
        loss = mv.losses.mlm_itm_loss(lz, itm, lab, aligned)
        loss.backward()
        return packed, float(loss), model.engine.flat_g.clone()
    p0, l0, g0 = run(mv.data.MaskDesc.make("full", meta["N"], meta["S"], b["n_ids"], DEV))
    p1, l1, g1 = run(dense)
    same = lambda a_, b_: float((a_ - b_).norm() / b_.norm()) < 1e-5         # (atomic sums: not bit-reproducible from run to run)
    assert p0 and p1 and abs(l0 - l1) < 1e-5 and same(g0, g1) and (model.n_masks_seen, model.n_masks_recognised) == (1, 1)
    p2, l2, g2 = run(dense, recognise=False)
    assert not p2 and abs(l2 - l1) < 5e-3 and float((g2 - g1).norm() / g2.norm()) < 2e-2 and model.n_masks_seen == 1
    bad = dense.clone()
    L = dense.shape[-1]
    bad[1, L // 2, 3] ^= 1                           # neither probe row (0, L-1) nor the probe column (L-1)
    p3, l3, g3 = run(bad)
    p4, l4, g4 = run(bad, recognise=False)
    assert not p3 and not p4 and abs(l3 - l4) < 1e-5 and same(g3, g4) and (model.n_masks_seen, model.n_masks_recognised) == (2, 1)
    n_ids = [int(v) for v in b["n_ids"]]
    short = min(range(len(n_ids)), key=lambda i: n_ids[i])
    if n_ids[short] < meta["S"] + 1:
        lab2 = labels.clone()
        lab2[short, L - 1] = 5                       # a label on a padded row: exists only when every row runs
        with pytest.raises(ValueError, match="padded position"):
            run(dense, lab=lab2)
        assert run(dense, recognise=False, lab=lab2)[0] is False
    model.zero_grad()


@pytest.mark.parametrize("dtype,rtol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
def test_lazy_forward_given_the_labels_runs_its_last_layer_on_the_consumed_rows(golden_dir, dtype, rtol):
    """forward(..., txt_labels=labels) under lazy logits (an optional keyword beyond the reference's signature): the last layer's per-row work
    runs on the labelled rows + first rows only, like the fused step; loss and every gradient equal the literal path's; the loss refuses other
    labels than the forward's; without lazy logits the keyword is refused."""
    z, meta, cfg, P, b = load_case(golden_dir, "c1v1k_full")
    model = make_model(cfg, P, dtype)
    labels, aligned = b["txt_labels"].to(DEV), b["is_aligned"].to(DEV)
    args = (b["cls_tok"].to(DEV), b["input_txt"].to(DEV), b["attn_mask"].to(DEV), b["segment"].to(DEV),
            (b["img_feats"].to(DEV), b["img_pos"].to(DEV)), b["sep_tok"].to(DEV))
    with pytest.raises(ValueError, match="lazy_logits"):
        model(*args, txt_labels=labels)
    mlm, itm = model(*args)
    ref = mv.losses.mlm_itm_loss(mlm, itm, labels, aligned)
    ref.backward()
    g_ref = {n: p.grad.clone() for n, p in model.named_parameters()}
    gmax = max(float(g.abs().max()) for g in g_ref.values())
    model.zero_grad()
    model.lazy_logits = True
    lz, itm2 = model(*args, txt_labels=labels)
    S = model.engine.S
    assert S["sel"] is not None and S["n_lab"] == int((labels != -100).sum()) and (S["cu"] is not None) == (dtype != torch.float32)
    assert float((itm2 - itm.detach()).abs().max()) < (FP32_TOL if dtype == torch.float32 else BF16_TOL)
    other = labels.clone()
    other[other != -100] = 7
    with pytest.raises(ValueError, match="differ"):
        mv.losses.mlm_itm_loss(lz, itm2, other, aligned)
    loss = mv.losses.mlm_itm_loss(lz, itm2, labels.clone(), aligned)          # (an equal tensor is accepted, not only the same object)
    assert abs(float(loss) - float(ref)) < (1e-5 if dtype == torch.float32 else 5e-3)
    loss.backward()
    for n, p in model.named_parameters():
        scale = max(float(g_ref[n].abs().max()), 1e-3 * gmax)
        assert float((p.grad - g_ref[n]).abs().max()) <= rtol * scale, n
    with pytest.raises(RuntimeError, match="labelled rows only"):
        lz.materialize()
    model.zero_grad()
    lz, itm3 = model(*args)                                                    # the next forward without labels is an ordinary lazy one
    assert model.engine.S["sel"] is None and model._lazy_rows is None
    mv.losses.mlm_itm_loss(lz, itm3, labels, aligned).backward()
    # no zero_grad: the next lazy step adds to the gradients held through the .grad views (the loss zeroes the flat buffer for its head's
    # gradients -- the held values are set aside first)
    lz, itm3 = model(*args)
    mv.losses.mlm_itm_loss(lz, itm3, labels, aligned).backward()
    for n, p in model.named_parameters():
        scale = max(float(g_ref[n].abs().max()), 1e-3 * gmax)
        assert float((p.grad - 2.0 * g_ref[n]).abs().max()) <= 2.0 * rtol * scale, n
    # grad_in_loss = False: the head runs for the value in the loss and again, with its gradient, in the backward -- same result
    model.zero_grad()
    model.grad_in_loss = False
    lz, itm3 = model(*args)
    mv.losses.mlm_itm_loss(lz, itm3, labels, aligned).backward()
    for n, p in model.named_parameters():
        scale = max(float(g_ref[n].abs().max()), 1e-3 * gmax)
        assert float((p.grad - g_ref[n]).abs().max()) <= rtol * scale, n
    model.zero_grad()


@pytest.mark.parametrize("family", ["full", "s2s", "mixed", "bar", "noncross", "1d"])
@pytest.mark.parametrize("labels_in_forward", [False, True])
def test_model_api_loop_on_every_mask_family_of_the_dataset(family, labels_in_forward):
    """The reference's loop (train_origin.py:95-131) with the model swapped and lazy logits, over every mask family the Dataset builds
    (dataset_origin.py:138-176), ragged lengths incl. a full-length and a one-token sample: the device matrix is recognised in all of them
    (a full-length BAR sample is not mistaken for a full one), the padding-invisible families run on packed rows, and loss + gradients equal
    those of the same loop with the recognition switched off (every row, the matrix as shipped)."""
    cfg = mv.ModelConfig(hidden=128, heads=2, intermediate=512, layers=2, vocab_size=1024, max_pos=512, dropout=0.0)
    B, N, S = 5, 6, 120
    lens = [1, S, S] + [None] * (B - 3)
    drawn = mv.data.synthetic_batch(cfg.vocab_size, B, N, S, family, seed=41, device="cpu")["n_ids"] - 1
    lens = [int(drawn[i]) if v is None else v for i, v in enumerate(lens)]
    b = mv.data.synthetic_batch(cfg.vocab_size, B, N, S, family, seed=41, device=DEV, lengths=lens)
    model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=DEV)
    model.reset_parameters(seed=6)
    model.train()                                           # (dropout 0.0 in the config: deterministic)
    model.lazy_logits = True
    out = []
    for rec in (False, True):
        model.zero_grad()
        model.recognise_masks = rec
        kw = {"txt_labels": b["txt_labels"]} if labels_in_forward else {}
        mlm, itm = model(b["cls_tok"], b["input_txt"], b["attn_mask"], b["segment"], (b["img_feats"], b["img_pos"]), b["sep_tok"], **kw)
        packed = model.engine.S["cu"] is not None
        loss = mv.losses.mlm_itm_loss(mlm, itm, b["txt_labels"], b["is_aligned"])
        loss.backward()
        out.append((packed, float(loss.detach()), model.engine.flat_g.clone(), mlm.stats.clone()))
    (p0, l0, g0, s0), (p1, l1, g1, s1) = out
    assert model.n_masks_seen == 1 and model.n_masks_recognised == 1
    assert not p0 and p1 == (family in ("full", "s2s", "mixed", "1d"))
    assert torch.equal(s0[[1, 2, 4, 5]], s1[[1, 2, 4, 5]])                  # label / sample counts, correct predictions
    assert abs(l0 - l1) < 2e-3 and float((g0 - g1).norm() / g0.norm()) < 3e-3
    model.zero_grad()


def test_drop_in_gradients_are_views_of_the_flat_buffer_with_autograds_semantics(golden_dir):
    """loss.backward() on the model API (train_origin.py:129-131): with .grad = None (after optimizer.zero_grad()) every Parameter's .grad
    becomes a VIEW of the engine's flat gradient buffer (no copy, nothing for autograd to accumulate); a second backward without zero_grad
    adds to it like autograd does; a foreign tensor in .grad is accumulated into, not replaced; grad_views = False restores the copies;
    torch optimizers step on the views."""
    z, meta, cfg, P, b = load_case(golden_dir, "c1v1k_full")
    model = make_model(cfg, P, torch.float32)
    labels, aligned = b["txt_labels"].to(DEV), b["is_aligned"].to(DEV)
    eng = model.engine

    def backward(scale=1.0):
        mlm, itm = fwd(model, b)
        (scale * mv.losses.mlm_itm_loss(mlm, itm, labels, aligned)).backward()

    def close(a, w):          # (sums over rows may be taken in a different order from one run to the next)
        return float((a - w).abs().max()) <= 1e-5 * max(float(w.abs().max()), 1e-6)
    model.grad_views = False
    backward()
    ref = {n: p.grad.clone() for n, p in model.named_parameters()}
    assert all(p.grad.data_ptr() != eng.g[n].data_ptr() for n, p in model.named_parameters())
    model.grad_views = True
    model.zero_grad()
    backward()
    for n, p in model.named_parameters():
        assert p.grad.data_ptr() == eng.g[n].data_ptr() and close(p.grad, ref[n]), n
    backward(0.5)                                    # no zero_grad in between: accumulation
    for n, p in model.named_parameters():
        assert p.grad.data_ptr() == eng.g[n].data_ptr()
        assert float((p.grad - 1.5 * ref[n]).abs().max()) <= 1e-5 * max(float(ref[n].abs().max()), 1e-6), n
    for p in model.parameters():
        p.grad.zero_()                               # optimizer.zero_grad(set_to_none=False)
    backward()
    assert all(close(p.grad, ref[n]) for n, p in model.named_parameters())
    model.zero_grad()
    name = "enc.encoder.layer.0.attention.self.query.weight" if "enc.encoder.layer.0.attention.self.query.weight" in ref else next(iter(ref))
    own = torch.ones_like(ref[name])
    dict(model.named_parameters())[name].grad = own  # somebody else's tensor: autograd semantics = add to it
    backward()
    got = dict(model.named_parameters())[name].grad
    assert got.data_ptr() != eng.g[name].data_ptr() and float((got - 1.0 - ref[name]).abs().max()) <= 1e-6 + 1e-5 * float(ref[name].abs().max())
    # a torch optimizer on the views moves the parameters exactly as it does on copies
    model.zero_grad()
    backward()
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    torch.optim.SGD(model.parameters(), lr=0.5).step()
    for n, p in model.named_parameters():
        assert float((p.detach() - (before[n] - 0.5 * ref[n])).abs().max()) <= 1e-6 + 1e-5 * float(ref[n].abs().max()), n
    model.zero_grad()
    # the default (None) decides by itself: a hook on a Parameter must see its gradient arrive through autograd, so copies are handed over
    model.grad_views = None
    seen = []
    some = dict(model.named_parameters())[name]
    handle = some.register_hook(lambda g_: seen.append(tuple(g_.shape)))
    backward()
    assert seen == [tuple(some.shape)] and all(p.grad.data_ptr() != eng.g[n].data_ptr() for n, p in model.named_parameters())
    handle.remove()
    model.zero_grad()
    backward()
    assert all(p.grad.data_ptr() == eng.g[n].data_ptr() for n, p in model.named_parameters())
    model.zero_grad()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_engine_adamw_behind_the_torch_optimizer_protocol(golden_dir, dtype):
    """medvill_amd.optim.AdamW(model.parameters(), lr) -- the reference's optimizer line (train_origin.py:60) on the flat buffers: the loop
    zero_grad / backward / step moves the parameters exactly like the engine's own step, schedulers act through param_groups, the 16-bit weight
    copies are the optimizer's (no refresh in the next forward until somebody else modifies a Parameter), state round-trips, and anything but
    the whole parameter set of one CXRBERT is refused."""
    z, meta, cfg, P, b = load_case(golden_dir, "c1v1k_full")
    labels, aligned = b["txt_labels"].to(DEV), b["is_aligned"].to(DEV)
    lrs = [3e-4, 3e-4, 1.5e-4]

    def loss_of(model):
        mlm, itm = fwd(model, b)
        return mv.losses.mlm_itm_loss(mlm, itm, labels, aligned)
    ref = make_model(cfg, P, dtype)
    ref.lazy_logits = True
    ref_losses = []
    for t, lr in enumerate(lrs):
        ref.zero_grad()
        loss = loss_of(ref)
        loss.backward()
        ref.engine.adamw_step(t + 1, lr=lr, weight_decay=0.01)
        ref_losses.append(float(loss))
    model = make_model(cfg, P, dtype)
    model.lazy_logits = True
    opt = mv.optim.AdamW(model.parameters(), lr=lrs[0], weight_decay=0.01)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda e: 1.0 if e < 2 else 0.5)
    assert model._params_dirty()
    casts = []
    sync = model.engine.sync_shadow
    model.engine.sync_shadow = lambda: (casts.append(1), sync())[1]
    for t in range(len(lrs)):
        opt.zero_grad()
        loss = loss_of(model)
        assert abs(float(loss) - ref_losses[t]) < (1e-5 if dtype == torch.float32 else 2e-3), t       # (atomic sums: not bit-reproducible)
        loss.backward()
        opt.step()
        sched.step()
        assert not model._params_dirty() and not model.engine.shadow_dirty
    assert len(casts) <= 1                                  # only the first forward converted the weights (16-bit path)
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert float((p - q).abs().max()) <= (1e-7 if dtype == torch.float32 else 2e-5) + 1e-3 * lrs[0], n
    opt.zero_grad()
    before = model.engine.flat_p.clone()
    opt.step()                                              # nothing back-propagated: no update
    assert torch.equal(before, model.engine.flat_p)
    with torch.no_grad():
        next(iter(model.parameters())).add_(0.0)            # somebody else touches a Parameter in place: the copies are stale again
    assert model._params_dirty()
    sd = opt.state_dict()
    opt2 = mv.optim.AdamW(model.parameters(), lr=1.0)
    opt2.load_state_dict(sd)
    assert opt2._t == len(lrs) and opt2.param_groups[0]["weight_decay"] == 0.01 and torch.equal(opt2.state_dict()["flat_m"], sd["flat_m"])
    with pytest.raises(ValueError, match="whole flat parameter buffer"):
        mv.optim.AdamW(list(model.parameters())[:5], lr=1e-3)
    with pytest.raises(ValueError, match="not the Parameters"):
        mv.optim.AdamW(torch.nn.Linear(2, 2).parameters(), lr=1e-3)


@pytest.mark.parametrize("dtype,gop", [(torch.float32, None), (torch.bfloat16, "f16"), (torch.bfloat16, "bf16")])
def test_fused_train_step_equals_dropin_path(golden_dir, dtype, gop):
    """TrainStep (labelled rows only, fused CE) must give the same losses and gradients as
    forward() + torch CE + backward() over all positions.  16-bit path: with loss-scaled f16 gradient operands (default) and
    with round 2's bf16 gradient operands (two encodings of every stored activation)."""
    z, meta, cfg, P, b = load_case(golden_dir, "c1v1k_full")
    model = make_model(cfg, P, dtype, grad_operand=gop)
    assert model.engine.dual == (gop == "bf16")
    mlm, itm = fwd(model, b)
    ce_m, ce_i = torch.nn.CrossEntropyLoss(ignore_index=-100), torch.nn.CrossEntropyLoss()
    ml, il = ce_m(mlm.transpose(1, 2), b["txt_labels"].to(DEV)), ce_i(itm, b["is_aligned"].to(DEV))
    (ml + il).backward()
    named = dict(model.named_parameters())
    ref = torch.cat([named[n].grad.reshape(-1) for n in model._param_names]).clone()
    ts = mv.TrainStep(model, lr=0.0)
    stats = ts({k: v for k, v in b.items()}, train=True).cpu()
    got = torch.cat([model.engine.g[n].reshape(-1) for n in model._param_names])
    assert abs(float(stats[0] / stats[1]) - float(ml)) < (1e-4 if dtype == torch.float32 else 2e-3)
    assert abs(float(stats[3] / stats[4]) - float(il)) < (1e-4 if dtype == torch.float32 else 2e-3)
    denom = float(ref.abs().max())
    assert float((got - ref).abs().max()) / denom < (1e-4 if dtype == torch.float32 else 3e-2)
    lab = b["txt_labels"] != -100
    assert int(stats[1]) == int(lab.sum()) and int(stats[4]) == meta["B"]
    assert int(stats[2]) == int(((mlm.argmax(-1).cpu() == b["txt_labels"]) & lab).sum())
    assert int(stats[5]) == int((itm.argmax(-1).cpu() == b["is_aligned"]).sum())


def test_train_steps_follow_the_oracle(golden_dir):
    """Three fused steps (fp32 path, lr large enough to move the weights) against the oracle's
    loop body of train_origin.py:95-131 (autograd + HF AdamW)."""
    z, meta, cfg, P, b = load_case(golden_dir, "c1v1k_full")
    model = make_model(cfg, P, torch.float32)
    ts = mv.TrainStep(model, lr=1e-3)
    Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    Mo = {k: torch.zeros_like(v) for k, v in P.items()}
    Vo = {k: torch.zeros_like(v) for k, v in P.items()}
    for t in range(1, 4):
        stats = ts(dict(b), train=True).cpu()
        lo, mlo, ilo = O.train_step(Po, Mo, Vo, t, cfg, b, lr=1e-3, training=False)
        assert abs(float(stats[0] / stats[1]) - mlo) < 1e-3 and abs(float(stats[3] / stats[4]) - ilo) < 1e-3
    sd = model.state_dict()
    worst = max(float((sd[k].cpu() - Po[k].detach()).abs().max()) for k in P)
    print("max |param - oracle param| after 3 steps:", worst)
    assert worst < 2e-4          # 3 updates of size <= 1e-3 each; fp32 rounding in the gradients only


BASE_CASES = ["base_s2s", "base_full", "base_noncross", "base768_s2s"]     # BASELINE.json configs 3 / 2 / 4 / 5 at scale


@pytest.mark.parametrize("name", BASE_CASES)
def test_bert_base_against_reference_golden(golden_dir, name):
    """BERT-base at the benchmark's own scale against the reference's own numbers: L=512 seq2seq (B=1), L=512
    bidirectional (B=2, ragged), L=512 non-cross (n2 = 38, not tile-aligned: the block-sparse path) and L=768 seq2seq
    (max_position_embeddings 768).  Both paths are held to north_star's tolerances on EVERY stored logit:
    1e-3 (fp32) and 1e-2 (16-bit path: f16 forward operands)."""
    z, meta, cfg, P, b = load_case(golden_dir, name)
    cols = torch.from_numpy(z["cols"].astype(np.int64))
    for dtype, tol in ((torch.float32, FP32_TOL), (torch.bfloat16, BF16_TOL)):
        model = make_model(cfg, P, dtype)
        with torch.no_grad():
            mlm, itm = fwd(model, b)
        mlm, itm = mlm.float().cpu(), itm.float().cpu()
        d = np.abs(mlm[..., cols].numpy() - z["logits_cols"])
        dl = np.abs(torch.logsumexp(mlm, -1).numpy() - z["lse"])
        ml, il = O.losses(mlm, itm, b["txt_labels"], b["is_aligned"])
        print(f"{name} {dtype}: logits max-abs {d.max():.3e} mean-abs {d.mean():.3e} p99.9 {np.quantile(d, 0.999):.3e} "
              f"(logit std {z['logits_cols'].std():.3f}); lse {dl.max():.2e}; itm {np.abs(itm.numpy() - z['itm']).max():.2e}; "
              f"mlm_loss {float(ml):.5f} vs {float(z['mlm_loss']):.5f}")
        assert d.max() < tol and dl.max() < tol and np.abs(itm.numpy() - z["itm"]).max() < tol
        assert abs(float(ml) - float(z["mlm_loss"])) < tol and abs(float(il) - float(z["itm_loss"])) < tol
        # arg-max: every logit is within `tol` of the reference's, so the logit at the REFERENCE's arg-max index can trail our maximum by
        # at most 2 tol.  (Random-init logits have near-ties: on the 16-bit path 1-4 of the 512-1024 positions pick the other member of a
        # pair whose gap is under 1.2e-3 -- which ones changes with any reordering of a sum -- hence no exact agreement rate beyond a sanity bound)
        ref_idx = torch.from_numpy(z["argmax"].astype(np.int64))
        at_ref = torch.gather(mlm, -1, ref_idx.unsqueeze(-1)).squeeze(-1)
        assert float((mlm.max(-1).values - at_ref).max()) <= 2 * tol
        assert (mlm.argmax(-1).numpy() == z["argmax"]).mean() > (0.9999 if dtype == torch.float32 else 0.98)
        del model
        torch.cuda.empty_cache()


def test_pure_bf16_forward_operands_stay_within_their_measured_bound(golden_dir):
    """fwd_operand="bf16" (every operand bf16-encoded, no duplicate activations) is kept as an option.  At BERT-base it
    CANNOT meet 1e-2: rounding the weights alone to bf16 moves the logits by 1.4e-2 (profiles/r02_bf16_error.txt).
    This test pins what it does reach so that a regression is caught; the tolerance of the contract is asserted on the
    default path above."""
    z, meta, cfg, P, b = load_case(golden_dir, "base_s2s")
    cols = torch.from_numpy(z["cols"].astype(np.int64))
    model = make_model(cfg, P, torch.bfloat16, fwd_operand="bf16")
    with torch.no_grad():
        mlm, itm = fwd(model, b)
    d = np.abs(mlm.float().cpu()[..., cols].numpy() - z["logits_cols"])
    print(f"base_s2s pure-bf16 operands: logits max-abs {d.max():.3e} mean-abs {d.mean():.3e} p99.9 {np.quantile(d, 0.999):.3e}")
    assert d.mean() < 6e-3 and d.max() < 3.5e-2


@pytest.mark.parametrize("case", ["base_full", "base_full_b4"])
@pytest.mark.parametrize("dtype,rtol,pack", [(torch.float32, 1e-3, False), (torch.bfloat16, 4e-2, False), (torch.bfloat16, 4e-2, True)])
def test_bert_base_gradients_against_reference_golden(golden_dir, dtype, rtol, pack, case):
    """Config 2's family at scale (BERT-base, 12 layers, L=512, bidirectional, B=2 / B=4 ragged): the fused training step's gradient
    of EVERY parameter against the reference's loss.backward() (norm + 16 sampled entries per tensor).  pack=True is the path
    bench.py times and CXRBERT_Trainer runs -- TrainStep's defaults: mask descriptors, padding removed (packed rows), the last layer on
    the consumed rows only with those rows as its only queries (`tq`), the MLM head on the labelled rows only, f16 operands and loss-
    scaled f16 gradients -- held to the same tolerances as the padded path."""
    z, meta, cfg, P, b = load_case(golden_dir, case)
    model = make_model(cfg, P, dtype)
    batch = dict(b)
    if pack:
        batch["attn_desc"] = mv.data.MaskDesc.make("full", meta["N"], meta["S"], b["n_ids"], DEV)
        ts = mv.TrainStep(model, lr=0.0)                                      # the defaults ARE the benchmarked path
        assert ts.pack_rows and ts.tail_rows and model.engine.tail_queries
    else:
        ts = mv.TrainStep(model, lr=0.0, pack_rows=False)
    stats = ts(batch, train=True).cpu()
    eng = model.engine
    if pack:
        Lq = meta["N"] + meta["S"] + 3
        assert eng.S["cu"] is not None and eng.S["tq"] is not None and eng.S["sel"] is not None and eng.S["M"] < meta["B"] * Lq
        assert eng.dt == mv._lib.MV_F16 and eng.fdt == mv._lib.MV_F16
        assert float(eng.scaler[6]) == 0.0                                    # no non-finite gradient under the loss scale
    else:
        assert eng.S["cu"] is None
    tol = FP32_TOL if dtype == torch.float32 else BF16_TOL
    if "lab_rows" in z.files:
        # the labelled-rows-only MLM head against the reference's logits AT those rows, entry by entry (1,024 sampled columns of
        # every labelled row, the logit at the label, and each row's logsumexp / maximum over all 30,522 columns)
        R = int(z["lab_rows"].shape[0])
        assert int(stats[1]) == R
        logits = eng.S["ht_"]["logits"][:R, :cfg.vocab_size].float().cpu()
        lab_ids = torch.from_numpy(z["lab_ids"].astype(np.int64))
        lc = torch.from_numpy(z["lab_cols"].astype(np.int64))
        d_cols = float(np.abs(logits[:, lc].numpy() - z["lab_logits_cols"]).max())
        d_lab = float(np.abs(logits.gather(1, lab_ids.view(-1, 1)).reshape(-1).numpy() - z["lab_logit_at_label"]).max())
        d_lse = float(np.abs(torch.logsumexp(logits, -1).numpy() - z["lse"].reshape(-1)[z["lab_rows"]]).max())
        d_max = float(np.abs(logits.max(-1).values.numpy() - z["maxval"].reshape(-1)[z["lab_rows"]]).max())
        print(f"{case} {dtype} pack={pack}: labelled rows {R}: |dlogits| cols {d_cols:.2e} label {d_lab:.2e} lse {d_lse:.2e} max {d_max:.2e}")
        assert max(d_cols, d_lab, d_lse, d_max) < tol
    assert abs(float(stats[0] / stats[1]) - float(z["mlm_loss"])) < tol and abs(float(stats[3] / stats[4]) - float(z["itm_loss"])) < tol
    names = [str(n) for n in z["grad_names"]]
    gmax = float(z["grad_norms"].max())
    worst, table = 0.0, []
    for i, k in enumerate(names):
        g = model.engine.g[k].float().cpu()
        ref_norm = float(z["grad_norms"][i])
        got = g.reshape(-1)[torch.from_numpy(z["grad_idx"][i])].numpy()
        floor = (1e-5 if dtype == torch.float32 else 3e-2) * gmax
        efloor = floor if dtype == torch.float32 else 8 * floor / np.sqrt(g.numel())
        scale = max(ref_norm / np.sqrt(g.numel()), float(np.abs(z["grad_vals"][i]).max()), efloor)
        e1 = abs(float(g.double().norm()) - ref_norm) / max(ref_norm, floor)
        e2 = float(np.abs(got - z["grad_vals"][i]).max()) / scale
        worst = max(worst, e1, e2 * 0.25)
        table.append((e2, e1, k, ref_norm))
    table.sort(reverse=True)
    print(f"{case} {dtype} pack={pack}: worst gradient deviation {worst:.2e}; largest entry deviations: "
          + "; ".join(f"{k} e2={e2:.3f} e1={e1:.4f} |g|={n:.2e}" for e2, e1, k, n in table[:6]))
    # norm of every tensor within rtol; single entries within 4 rtol of the tensor's scale -- 6 rtol for the two head weights whose
    # gradient is a sum of B = 2 outer products (pooler, ITM): their entries are heavy-tailed (a few are tens of RMS), so one
    # operand's bf16 rounding (2^-9 of such an entry) is a visible fraction of the RMS the error is scaled by (norm error 1.2 %)
    for e2, e1, k, n in table:
        assert e1 < rtol, (k, e1, n)
        lowrank = k in ("enc.pooler.dense.weight", "itm.linear.weight")
        assert e2 < (6 if lowrank and dtype != torch.float32 else 4) * rtol + 1e-6, (k, e2)


def test_dropout_training_step_against_oracle_with_the_same_masks(golden_dir):
    """Train mode (dropout 0.1 at the embedding, attention-probability and both hidden-state sites, like the
    reference's model.train()): the kernels regenerate counter-based masks instead of storing them; feeding the very
    same masks to the oracle must reproduce loss and gradients (fp32 path), and the masks must look like Bernoulli(0.9)."""
    from medvill_amd import hip_ops as ops
    z, meta, cfg, P, b = load_case(golden_dir, "c1v1k_bar_ragged")
    model = make_model(cfg, P, torch.float32)
    model.train()
    ts = mv.TrainStep(model, lr=0.0)
    stats = ts(dict(b), train=True).cpu()
    eng = model.engine
    assert eng.S["p_drop"] == pytest.approx(0.1)
    B, Lq, H, A = meta["B"], meta["N"] + meta["S"] + 3, cfg.hidden, cfg.heads
    Lp = (Lq + 3) // 4 * 4
    masks, fracs = {}, []
    for (site, l), key in eng.S["drop_keys"].items():
        if site == eng.SITE_ATTN:
            keep, sc = ops.attn_keep_mask(eng.S["layers"][l]["dropbits"], B, Lq, A), 65536.0 / (65536.0 - 6554.0)
            masks[("attn", l)] = (keep.float() * sc).cpu()
            fracs.append(float(keep.float().mean()))
            continue
        else:
            keep, sc = ops.dropout_mask(0.1, key, B * Lq * H, DEV)
            name = {eng.SITE_EMB: "emb", eng.SITE_OUT1: ("out1", l), eng.SITE_OUT2: ("out2", l)}[site]
            full = keep.view(B * Lq, H).float() * sc
            if site != eng.SITE_EMB and l == cfg.layers - 1 and eng.S["sel"] is not None:
                # the last layer's per-row part ran on the consumed rows only: its masks are keyed by the COMPACT row index;
                # the other rows' outputs are never used, so any mask serves there
                sel = eng.S["sel"].long()
                full = torch.ones_like(full)
                full[sel] = (keep.view(-1, H)[:sel.numel()].float() * sc)
            masks[name] = full.view(B, Lq, H).cpu()
        fracs.append(float(keep.float().mean()))
        assert sc == pytest.approx(65536.0 / (65536.0 - 6554.0))        # P(drop) = 6554 / 65536 at every site
    assert all(abs(f - 0.9) < 0.012 for f in fracs), fracs
    names = list(masks)                                                       # every site / layer draws its own mask
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            a_, b_ = masks[names[i]], masks[names[j]]
            assert a_.shape != b_.shape or not torch.equal(a_, b_), (names[i], names[j])
    Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    mlm, itm = O.forward(Po, cfg, b["cls_tok"], b["input_txt"], b["attn_mask"], b["segment"], b["img_feats"], b["img_pos"],
                         b["sep_tok"], masks=masks)
    ml, il = O.losses(mlm, itm, b["txt_labels"], b["is_aligned"])
    (ml + il).backward()
    assert abs(float(stats[0] / stats[1]) - float(ml)) < 1e-4 and abs(float(stats[3] / stats[4]) - float(il)) < 1e-4
    assert abs(float(ml) - float(z["mlm_loss"])) > 1e-4                        # dropout really changed the forward
    gmax = max(float(Po[k].grad.norm()) for k in P)
    for k in P:
        got, ref = eng.g[k].cpu(), Po[k].grad
        assert float((got - ref).norm()) <= 1e-3 * float(ref.norm()) + 1e-5 * gmax, k
    # a second step draws different masks; the bf16 MFMA path runs the same schedule and stays close to the fp32 one
    ts(dict(b), train=True)
    assert eng.S["drop_keys"][(eng.SITE_EMB, 0)] != list(masks.keys()) and eng.drop_counter == 2
    m16 = make_model(cfg, P, torch.bfloat16)
    m16.train()
    m16.engine.drop_seed, m16.engine.drop_counter = eng.drop_seed, 0
    s16 = mv.TrainStep(m16, lr=0.0)(dict(b), train=True).cpu()
    assert abs(float(s16[0] / s16[1]) - float(ml)) < 1e-2
    g32 = torch.cat([Po[k].grad.reshape(-1) for k in P])
    g16 = torch.cat([m16.engine.g[k].reshape(-1) for k in P]).cpu()
    assert float((g16 - g32).norm() / g32.norm()) < 3e-2


def test_mask_descriptors_equal_materialised_masks(golden_dir):
    """masks synthesised on the device from {family, n2, vl} give bit-identical outputs to the
    reference-style int64 [B,L,L] input, for every family incl. the per-sample Mixed choice."""
    z, meta, cfg, P, b = load_case(golden_dir, "c1_s2s")
    model = make_model(cfg, P, torch.bfloat16)
    N, S = meta["N"], meta["S"]
    for fam in ("full", "s2s", "bar", "noncross", "1d", "mixed"):
        per = ["s2s", "full", "full", "s2s"] if fam == "mixed" else fam
        n_ids = torch.from_numpy(z["in_n_ids"])
        if fam == "mixed":
            mask = mv.data.mixed_mask(N, S, n_ids, [True, False, False, True])
        else:
            mask = mv.data.build_mask(fam, N, S, n_ids)
        desc = mv.data.MaskDesc.make(per, N, S, n_ids)
        with torch.no_grad():
            a = fwd(model, dict(b, attn_mask=mask))
            c = model(b["cls_tok"].to(DEV), b["input_txt"].to(DEV), desc, b["segment"].to(DEV),
                      (b["img_feats"].to(DEV), b["img_pos"].to(DEV)), b["sep_tok"].to(DEV))
        assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1]), fam


def test_retrieval_head_with_1d_masks(golden_dir):
    """CXRBertForRetrieval = enc + itm with the [B,L] mask branch (cxrbert_origin.py:76-77)."""
    z, meta, cfg, P, b = load_case(golden_dir, "c1_1d")
    for dtype, tol in ((torch.float32, FP32_TOL), (torch.bfloat16, BF16_TOL)):
        r = mv.CXRBertForRetrieval(cfg_dict(cfg), None, dtype=dtype, device=DEV)
        r.bert.load_state_dict(P)
        r.eval()
        assert b["attn_mask"].dim() == 2
        with torch.no_grad():
            itm = r(b["cls_tok"].to(DEV), b["input_txt"].to(DEV), b["attn_mask"].to(DEV), b["segment"].to(DEV),
                    (b["img_feats"].to(DEV), b["img_pos"].to(DEV)), b["sep_tok"].to(DEV))
            sc = r.score(b["cls_tok"].to(DEV), b["input_txt"].to(DEV), b["attn_mask"].to(DEV), b["segment"].to(DEV),
                         (b["img_feats"].to(DEV), b["img_pos"].to(DEV)), b["sep_tok"].to(DEV))
        assert float(np.abs(itm.float().cpu().numpy() - z["itm"]).max()) < tol
        ref = torch.softmax(torch.from_numpy(z["itm"]), -1)[:, 1]
        assert float((sc.cpu() - ref).abs().max()) < tol


@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, FP32_TOL, 1e-4), (torch.bfloat16, BF16_TOL, 3e-2)])
def test_head_submodules_are_callable_like_the_reference(golden_dir, dtype, tol, gtol):
    """The reference composes the model from callable sub-modules: CXRBERT.forward is `x_mlm, x_itm, _ = self.enc(...);
    scores, _ = self.mlm(x_mlm); itm = self.itm(x_itm)` (cxrbert_origin.py:144-149) and the retrieval model is
    `_, cls, _ = self.enc(...); self.itm(cls)` (Downstream_task/Retrieval/retrieval.py:26-31).  Both literal forms run here, on
    the HIP kernels, equal the golden logits, and back-propagate the same gradients as the fused forward()."""
    z, meta, cfg, P, b = load_case(golden_dir, "c1v1k_full")
    model = make_model(cfg, P, dtype)
    args = (b["cls_tok"].to(DEV), b["input_txt"].to(DEV), b["attn_mask"].to(DEV), b["segment"].to(DEV),
            (b["img_feats"].to(DEV), b["img_pos"].to(DEV)), b["sep_tok"].to(DEV))
    x_mlm, x_itm, _ = model.enc(*args)
    scores, none = model.mlm(x_mlm)
    itm = model.itm(x_itm)
    assert none is None and scores.shape == (meta["B"], meta["N"] + meta["S"] + 3, cfg.vocab_size) and itm.shape == (meta["B"], 2)
    assert float(np.abs(scores.detach().float().cpu().numpy() - z["mlm"]).max()) < tol
    assert float(np.abs(itm.detach().float().cpu().numpy() - z["itm"]).max()) < tol
    assert float(np.abs(model.mlm.predictions(x_mlm).detach().float().cpu().numpy() - z["mlm"]).max()) < tol     # BertLMPredictionHead
    ce_m, ce_i = torch.nn.CrossEntropyLoss(ignore_index=-100), torch.nn.CrossEntropyLoss()
    loss = ce_m(scores.float().transpose(1, 2), b["txt_labels"].to(DEV)) + ce_i(itm.float(), b["is_aligned"].to(DEV))
    loss.backward()
    got = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad()
    mlm2, itm2 = model(*args)
    (ce_m(mlm2.transpose(1, 2), b["txt_labels"].to(DEV)) + ce_i(itm2, b["is_aligned"].to(DEV))).backward()
    gmax = max(float(p.grad.abs().max()) for p in model.parameters())
    for n, p in model.named_parameters():
        scale = max(float(p.grad.abs().max()), 1e-3 * gmax)
        assert float((got[n] - p.grad).abs().max()) <= gtol * scale, (n, float((got[n] - p.grad).abs().max()), scale)
    # the retrieval composition on the 1-D mask case, under no_grad like full_dset_retrieval.py's scoring loop
    z1, meta1, cfg1, P1, b1 = load_case(golden_dir, "c1_1d")
    m1 = make_model(cfg1, P1, dtype)
    with torch.no_grad():
        _, cls, _ = m1.enc(b1["cls_tok"].to(DEV), b1["input_txt"].to(DEV), b1["attn_mask"].to(DEV), b1["segment"].to(DEV),
                           (b1["img_feats"].to(DEV), b1["img_pos"].to(DEV)), b1["sep_tok"].to(DEV))
        result = m1.itm(cls)
    assert float(np.abs(result.float().cpu().numpy() - z1["itm"]).max()) < tol
    # a head on an input that did not come from the encoder (any [.., H] tensor), gradient w.r.t. that input
    xin = torch.randn(5, cfg.hidden, device=DEV, requires_grad=True)
    out = model.itm(xin)
    W, bb = model.get_parameter("itm.linear.weight").detach(), model.get_parameter("itm.linear.bias").detach()
    assert float((out.detach() - (xin.detach() @ W.t() + bb)).abs().max()) < tol
    out[:, 1].sum().backward()
    assert float((xin.grad - W[1].expand(5, -1)).abs().max()) < tol * float(W.abs().max()) + 1e-6


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
def test_head_calls_are_reentrant_like_nn_modules(golden_dir, dtype, tol):
    """the reference's heads are plain nn.Modules -- two calls in one graph (or gradient accumulation over two forwards)
    back-propagate independently.  model.mlm(x1) and model.mlm(x2) BEFORE either backward must give the gradients of the two calls
    made one after the other, and the head's backward must leave the other heads' gradient buffers (same bucket) alone."""
    z, meta, cfg, P, b = load_case(golden_dir, "c1v1k_full")
    model = make_model(cfg, P, dtype)
    g = torch.Generator(device="cpu").manual_seed(3)
    xs = [torch.randn(n, cfg.hidden, generator=g).to(DEV) for n in (7, 5)]
    ws = [torch.randn(n, cfg.vocab_size, generator=g).to(DEV) for n in (7, 5)]
    names = [n for n, _ in model.named_parameters() if n.startswith("mlm.") or n.endswith("word_embeddings.weight")]

    def run(together):
        model.zero_grad()
        leaves = [x.clone().requires_grad_(True) for x in xs]
        if together:
            outs = [model.mlm(x)[0] for x in leaves]                       # both forwards first ...
            sum((o.float() * w).sum() for o, w in zip(outs, ws)).backward()   # ... then both backwards
        else:
            for x, w in zip(leaves, ws):
                (model.mlm(x)[0].float() * w).sum().backward()
        return [x.grad.clone() for x in leaves] + [model.get_parameter(n).grad.clone() for n in names]

    eng = model.engine
    eng.ensure_grad()
    eng.g["itm.linear.weight"].fill_(1.0)
    eng.g["enc.pooler.dense.bias"].fill_(2.0)
    one, two = run(False), run(True)
    for a_, b_, n in zip(one, two, ["x1", "x2"] + names):
        scale = max(float(a_.abs().max()), 1e-6)
        assert float((a_ - b_).abs().max()) <= tol * scale, (n, float((a_ - b_).abs().max()), scale)
    assert float(eng.g["itm.linear.weight"].min()) == 1.0 and float(eng.g["enc.pooler.dense.bias"].max()) == 2.0


def test_no_grad_forward_keeps_no_per_layer_activations():
    """under torch.no_grad() (the retrieval / evaluation loops) nothing is saved for a backward -- every layer
    reuses one scratch set -- and the results are those of the grad-mode forward."""
    cfg = mv.ModelConfig(vocab_size=512, hidden=128, layers=4, heads=2, intermediate=512, max_pos=128)
    b = mv.data.synthetic_batch(cfg.vocab_size, 4, 6, 40, "bar", seed=3, device=DEV)
    args = (b["cls_tok"], b["input_txt"], b["attn_mask"], b["segment"], (b["img_feats"], b["img_pos"]), b["sep_tok"])
    outs = []
    for grad in (False, True):
        torch.manual_seed(9)
        m = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=DEV)
        m.eval()
        with torch.set_grad_enabled(grad):
            mlm, itm = m(*args)
        torch.cuda.synchronize()
        keys = set(m.engine._ws)
        outs.append((mlm.clone(), itm.clone(), keys, sum(t.numel() * t.element_size() for t in m.engine._ws.values())))
    (m0, i0, k0, bytes0), (m1, i1, k1, bytes1) = outs
    assert torch.equal(m0, m1) and torch.equal(i0, i1)
    per_layer = [k for k in k0 if any(k == f"{stem}{l}" or k.startswith(f"{stem}{l}_") for stem in ("qkv", "ctx", "lse", "i", "dgelu", "a")
                                      for l in range(1, cfg.layers))]
    assert not per_layer, per_layer
    assert any(k.startswith("qkv1") for k in k1) and bytes0 < 0.45 * bytes1, (bytes0, bytes1)
    with torch.no_grad():
        m(*args)
    with pytest.raises(RuntimeError):
        m.engine.encoder_backward()


def test_overlapped_optimizer_equals_the_plain_step():
    """TrainStep(overlap_optimizer=True): AdamW runs range by range on the side stream and the next forward waits per range.
    Elementwise the same arithmetic on the same gradients; the gradients themselves carry float-atomic sums, so two runs of the
    SAME mode already differ in last bits (and AdamW turns a sign flip of a near-zero gradient into 2 lr): the comparison uses the
    bounds of the data-parallel test.  state_dict() taken right after a step sees the finished update."""
    cfg = mv.ModelConfig(vocab_size=2048, hidden=128, layers=3, heads=2, intermediate=512, max_pos=128)
    out = []
    for overlap in (False, True):
        torch.manual_seed(5)
        m = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=DEV)
        m.reset_parameters(seed=3)
        m.train()
        ts = mv.TrainStep(m, lr=1e-3, overlap_optimizer=overlap)
        for i in range(3):
            batch = mv.data.synthetic_batch(cfg.vocab_size, 8, 6, 40, "mixed", seed=20 + i, device=DEV)
            stats = ts(batch, train=True)
        sd = m.state_dict()                                    # waits for the side stream's kernels by itself
        ev = ts(mv.data.synthetic_batch(cfg.vocab_size, 8, 6, 40, "full", seed=99, device=DEV), train=False)
        torch.cuda.synchronize()
        out.append((m.engine.flat_p.clone(), sd, stats.cpu(), ev.cpu(), m.engine.shadow_f.clone()))
    (p0, sd0, s0, e0, sh0), (p1, sd1, s1, e1, sh1) = out
    d = (p0 - p1).abs()
    assert float(d.max()) < 3 * 2e-3 + 1e-4 and float(d.mean()) < 5e-5, (float(d.max()), float(d.mean()))
    assert torch.equal(sh1.float(), p1.to(torch.float16).float())                      # shadow refreshed from the updated weights
    assert float((s0 - s1).abs().max()) < 1e-2 * float(s0.abs().max()) and float((e0 - e1).abs().max()) < 1e-2 * float(e0.abs().max())
    off = m.engine.layout["enc.pooler.dense.weight"][0]
    assert torch.equal(sd1["enc.pooler.dense.weight"].flatten(), p1[off:off + 128 * 128])    # state_dict saw the finished step


def test_retrieval_forward_is_differentiable_through_the_itm_head(golden_dir):
    """CXRBertForRetrieval.forward (enc + itm, Retrieval/retrieval.py:26-31) runs its ITM linear on the C ABI too and stays a
    differentiable node: its parameter gradients equal those of the ITM output of the full CXRBERT.forward."""
    z, meta, cfg, P, b = load_case(golden_dir, "c1_1d")
    r = mv.CXRBertForRetrieval(cfg_dict(cfg), None, dtype=torch.float32, device=DEV)
    r.bert.load_state_dict(P)
    r.eval()
    args = (b["cls_tok"].to(DEV), b["input_txt"].to(DEV), b["attn_mask"].to(DEV), b["segment"].to(DEV),
            (b["img_feats"].to(DEV), b["img_pos"].to(DEV)), b["sep_tok"].to(DEV))
    itm = r(*args)
    assert itm.requires_grad and float(np.abs(itm.detach().cpu().numpy() - z["itm"]).max()) < FP32_TOL
    (itm[:, 1] - itm[:, 0]).sum().backward()
    g1 = {n: p.grad.clone() for n, p in r.bert.named_parameters() if p.grad is not None}
    r.bert.zero_grad()
    _, itm2 = r.bert(*args)
    (itm2[:, 1] - itm2[:, 0]).sum().backward()
    checked = 0
    for n, p in r.bert.named_parameters():
        if p.grad is None:
            continue
        ref = p.grad
        assert n in g1, n
        scale = float(ref.abs().max()) + 1e-12
        assert float((g1[n] - ref).abs().max()) <= 1e-5 * scale + 1e-9, n
        checked += 1
    assert checked > 30 and float(g1["itm.linear.weight"].abs().max()) > 0


@pytest.mark.parametrize("B", [8, 64])
def test_half_batches_sum_to_full_batch_gradient(B):
    """Size-independent linearity property at BERT-base scale, up to BASELINE.json's full size (B = 64, L = 512,
    bf16 MFMA path): the gradients of two half mini-batches, each normalised by the GLOBAL label / batch counts
    (what every DP rank computes), sum to the gradient of the full mini-batch."""
    cfg = mv.ModelConfig()
    model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=DEV)
    model.reset_parameters(seed=3)
    model.eval()                   # linearity holds for a fixed function: dropout off
    N, S = 36, 473
    full = mv.data.synthetic_batch(cfg.vocab_size, B, N, S, "mixed", seed=77, device=DEV)
    ts = mv.TrainStep(model, lr=0.0)
    ts(full, train=True)
    eng = model.engine
    g_full = eng.flat_g.clone()
    nlab = int((full["txt_labels"] != -100).sum())
    acc = torch.zeros_like(g_full)
    for h in range(2):
        sl = slice(h * B // 2, (h + 1) * B // 2)
        half = {k: (v[sl] if torch.is_tensor(v) and v.shape[:1] == (B,) else v) for k, v in full.items()
                if k not in ("label_rows", "label_ids", "attn_desc")}
        rows, ids = mv.data.label_index(half["txt_labels"])
        eng.flat_g.zero_()
        eng.encoder_forward(half["cls_tok"], half["input_txt"], half["attn_mask"], half["segment"], half["img_feats"],
                            half["img_pos"], half["sep_tok"])
        eng.heads_train(rows, ids, half["is_aligned"].to(torch.int32), mlm_scale=1.0 / nlab, itm_scale=1.0 / B)
        eng.encoder_backward()
        acc += eng.flat_g
    rel = float((acc - g_full).norm() / g_full.norm())
    print("half+half vs full gradient, relative L2:", rel)
    assert torch.isfinite(g_full).all() and rel < 2e-2


@pytest.mark.parametrize("family,B,N,S,tq", [("mixed", 7, 6, 120, True), ("full", 3, 36, 473, False), ("full", 3, 36, 473, True), ("s2s", 4, 1, 70, True)])
def test_packed_rows_reproduce_the_padded_step(family, B, N, S, tq):
    """Padding removal (TrainStep pack_rows): running the encoder on the valid rows only must give the padded run's
    loss statistics and gradients -- valid rows go through identical arithmetic (same MFMA contraction order, same key
    tiling), only the order of the row-reductions in weight gradients differs (fp32 summation order).  tq: the all-bidirectional
    batch also runs its LAST layer on reordered rows with the consumed rows as the only queries (Engine.tail_queries): the keys
    of that layer's softmax are then summed in another order, which shows at the 16-bit rounding level of its outputs."""
    cfg = mv.ModelConfig(hidden=128, heads=2, intermediate=512, layers=2, vocab_size=1024, max_pos=512, dropout=0.0)
    # ragged to the extremes: a single text token, no padding at all, and drawn lengths in between
    lens = [1, S] + [None] * (B - 2)
    drawn = mv.data.synthetic_batch(cfg.vocab_size, B, N, S, family, seed=31, device="cpu")["n_ids"] - 1
    lens = [int(drawn[i]) if v is None else v for i, v in enumerate(lens)]
    batch = mv.data.synthetic_batch(cfg.vocab_size, B, N, S, family, seed=31, device=DEV, lengths=lens)
    assert batch["n_ids"].tolist()[:2] == [2, S + 1]
    out = []
    for pack in (False, True):
        model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=DEV)
        model.reset_parameters(seed=5)
        model.train()
        model.engine.tail_queries = tq
        ts = mv.TrainStep(model, lr=0.0, pack_rows=pack)
        stats = ts(batch, train=True)
        eng = model.engine
        assert (eng.S["cu"] is not None) == pack
        assert (eng.S["tq"] is not None) == (pack and tq and family == "full")
        if pack:
            vl = batch["attn_desc"].host_desc()[:, 2]
            assert eng.S["M"] == int(vl.sum()) < B * (N + S + 3) and eng.S["layers"][0]["x"].shape[0] == eng.S["M"]
        out.append((stats.clone(), eng.flat_g.clone()))
    (s0, g0), (s1, g1) = out
    assert torch.equal(s0[[1, 2, 4, 5]], s1[[1, 2, 4, 5]])                       # counts: labels, correct predictions
    assert float((s0 - s1).abs().max() / s0.abs().max()) < 1e-5                  # nll sums
    rel = float((g0 - g1).norm() / g0.norm())
    worst = float((g0 - g1).abs().max() / g0.abs().max())
    print(f"packed vs padded gradient: rel L2 {rel:.2e}, max-abs/max {worst:.2e}")
    tol = 5e-4 if (tq and family == "full") else 1e-4
    assert rel < tol and worst < tol


@pytest.mark.parametrize("family,N,S", [("mixed", 36, 473), ("s2s", 100, 665), ("full", 36, 473)])
def test_packed_rows_reproduce_the_padded_step_at_bert_base_scale(family, N, S):
    """The same property on the production kernels' shapes (BERT-base, L = 512 / 768: 256-row GEMM tiles, persistent weight-
    gradient kernel with odd token counts, MFMA attention with per-sample row offsets), dropout off (the hidden-state
    dropout masks are keyed on row indices, which differ between the packed and the padded layout)."""
    cfg = mv.ModelConfig(max_pos=1024 if S > 511 else 512, dropout=0.0)
    batch = mv.data.synthetic_batch(cfg.vocab_size, 8, N, S, family, seed=19, device=DEV)
    out = []
    for pack in (False, True):
        model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=DEV)
        model.reset_parameters(seed=2)
        model.train()
        stats = mv.TrainStep(model, lr=0.0, pack_rows=pack)(batch, train=True)
        # "full": every sample is bidirectional, so the packed step also runs its last layer with the consumed rows as the only queries
        assert (model.engine.S["tq"] is not None) == (pack and family == "full")
        out.append((stats.clone(), model.engine.flat_g.clone()))
        del model
    (s0, g0), (s1, g1) = out
    assert torch.equal(s0[[1, 2, 4, 5]], s1[[1, 2, 4, 5]]) and float((s0 - s1).abs().max() / s0.abs().max()) < 1e-5
    rel = float((g0 - g1).norm() / g0.norm())
    print(f"BERT-base {family} L={N + S + 3}: packed vs padded gradient rel L2 {rel:.2e}")
    # "full" also reorders the last layer's rows (consumed rows first, `tq`): that layer's softmax sums its keys in another order, which
    # shows at the 16-bit rounding level of its outputs (measured 1.2e-3 here; the small-model test above allows the same 5x for it)
    assert rel < (3e-3 if family == "full" else 1e-3)


@pytest.mark.parametrize("dtype,family,layers", [(torch.float32, "bar", 2), (torch.bfloat16, "mixed", 2), (torch.bfloat16, "full", 1)])
def test_last_layer_on_consumed_rows_only_changes_nothing(dtype, family, layers):
    """TrainStep runs the last layer's per-row part (output projection, LayerNorms, FFN) only on the rows that are consumed
    downstream -- the labelled rows (MLM head) and each sample's first row (pooler -> ITM) -- and back-propagates through
    it on those rows only.  The other rows' outputs are unused and their gradients are exactly zero, so losses, counters
    and every parameter gradient must equal the all-rows run (different summation sets of the same non-zero terms)."""
    cfg = mv.ModelConfig(hidden=128, heads=2, intermediate=512, layers=layers, vocab_size=1024, max_pos=512, dropout=0.0)
    batch = mv.data.synthetic_batch(cfg.vocab_size, 5, 6, 90, family, seed=41, device=DEV)
    out = []
    for tail in (False, True):
        model = mv.CXRBERT(cfg, None, dtype=dtype, device=DEV)
        model.reset_parameters(seed=6)
        model.train()
        ts = mv.TrainStep(model, lr=0.0)
        ts.tail_rows = tail
        stats = ts(batch, train=True)
        R = int(batch["label_rows"].numel())
        assert (model.engine.S["sel"] is not None) == tail
        assert model.engine.S["hidden"].shape[0] == (R + 5 if tail else model.engine.S["M"])
        out.append((stats.clone(), model.engine.flat_g.clone()))
    (s0, g0), (s1, g1) = out
    assert torch.equal(s0[[1, 2, 4, 5]], s1[[1, 2, 4, 5]]) and float((s0 - s1).abs().max() / s0.abs().max()) < 1e-5
    rel = float((g0 - g1).norm() / g0.norm())
    print(f"consumed-rows tail vs all rows ({dtype}, {family}): gradient rel L2 {rel:.2e}")
    assert rel < (1e-5 if dtype == torch.float32 else 2e-3)


def test_packing_is_refused_where_padding_is_visible():
    cfg = mv.ModelConfig(hidden=128, heads=2, intermediate=512, layers=1, vocab_size=1024, max_pos=128)
    model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=DEV)
    b = mv.data.synthetic_batch(cfg.vocab_size, 2, 6, 40, "bar", seed=1, device=DEV)
    assert not b["attn_desc"].packable()
    with pytest.raises(ValueError):
        model.engine.encoder_forward(b["cls_tok"], b["input_txt"], b["attn_desc"], b["segment"], b["img_feats"], b["img_pos"],
                                     b["sep_tok"], pack=True)
    ts = mv.TrainStep(model, lr=0.0)           # falls back to the padded path on its own
    ts(b, train=True)
    assert model.engine.S["cu"] is None


@pytest.mark.parametrize("family", ["full", "s2s", "bar", "noncross", "1d", "mixed"])
def test_trainer_derives_descriptors_from_reference_masks_and_verifies_them(family):
    """The reference Dataset ships materialised int64 masks (dataset_origin.py:138-176).  The drop-in trainer derives the
    {family, n2, vl} descriptors from them, runs on those (packed rows where padding is invisible) and checks every mask
    entry on the host (mv_mask_verify_host, one batch ahead of the step); a single deviating entry, or a mask outside the
    families, makes the step run on the matrix itself."""
    from types import SimpleNamespace
    V, B, N, S = 2048, 4, 6, 41
    cfgd = dict(vocab_size=V, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
                max_position_embeddings=64)
    b = mv.data.synthetic_batch(V, B, N, S, family, seed=7, device="cpu")
    tup = lambda m: (b["cls_tok"], b["input_txt"], b["txt_labels"], m, (b["img_feats"], b["img_pos"]), b["segment"], b["is_aligned"],
                     b["sep_tok"], torch.zeros(B))
    args = SimpleNamespace(with_cuda=True, weight_load=False, bert_model="custom", lr=0.0, log_freq=10, mlm_task=True, itm_task=True,
                           cuda_devices=[0], dropout_prob=0.1)
    torch.manual_seed(3)
    tr = mv.CXRBERT_Trainer(args, [tup(b["attn_mask"])], None, config=cfgd, dtype=torch.bfloat16)
    tr.model.eval()
    got = tr._run_epoch(tr.train_data, 0, False)
    assert tr.n_recognised == 1
    packable = family in ("full", "s2s", "1d", "mixed")
    assert (tr.model.engine.S["cu"] is not None) == packable              # the fast layout is really the one that ran
    tr.recognise_masks = False
    ref = tr._run_epoch(tr.train_data, 0, False)
    assert tr.model.engine.S["cu"] is None
    for k in ref:
        assert abs(got[k] - ref[k]) < 2e-3, (k, got[k], ref[k])
    if b["attn_mask"].dim() == 3:
        # one entry flipped deep inside the matrix: the probes still say `family`, the device check does not
        bad = b["attn_mask"].clone()
        bad[1, N + 9, N + 5] ^= 1
        tr.recognise_masks = True
        tr.train_data = [tup(bad)]
        got_bad = tr._run_epoch(tr.train_data, 0, False)
        assert tr.n_recognised == 2 and tr.n_rejected == 1 and tr.model.engine.S["cu"] is None      # descriptors derived, then rejected
        tr.recognise_masks = False
        ref_bad = tr._run_epoch(tr.train_data, 0, False)
        for k in ref_bad:            # same kernels on the same inputs (float atomics in the loss sums: last-bit differences)
            assert abs(got_bad[k] - ref_bad[k]) < 1e-5, k


@pytest.mark.parametrize("mode,rejected", [("full", 1), ("sampled", 1), ("off", 0)])
def test_trainer_mask_verification_policies(mode, rejected):
    """args.verify_masks: "full" (default) and the first batches of "sampled" check every entry and send a batch with one flipped entry to
    the matrix; "off" trusts the probe lines (the flip is deep inside the matrix, so that batch runs on the closed form)."""
    from types import SimpleNamespace
    V, B, N, S = 2048, 4, 6, 41
    cfgd = dict(vocab_size=V, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=512, max_position_embeddings=64)

    def tup(seed, flip):
        b = mv.data.synthetic_batch(V, B, N, S, "s2s", seed=seed, device="cpu")
        m = b["attn_mask"].clone()
        if flip:
            m[3, N + 20, N + 7] ^= 1
        return (b["cls_tok"], b["input_txt"], b["txt_labels"], m, (b["img_feats"], b["img_pos"]), b["segment"], b["is_aligned"], b["sep_tok"], torch.zeros(B))
    args = SimpleNamespace(with_cuda=True, weight_load=False, bert_model="custom", lr=1e-3, log_freq=10, mlm_task=True, itm_task=True,
                           cuda_devices=[0], dropout_prob=0.1, verify_masks=mode)
    tr = mv.CXRBERT_Trainer(args, [tup(1, False), tup(2, True), tup(3, False)], None, config=cfgd, dtype=torch.bfloat16)
    res = tr.train(0)
    assert np.isfinite(res["avg_loss"]) and tr.n_recognised == 3 and tr.n_rejected == rejected


def test_trainer_mirror_runs_an_epoch_and_saves(tmp_path):
    """CXRBERT_Trainer(args, train_dl, test_dl).train(epoch) / .save(epoch, path) as main_origin.py:57-62 drives it,
    fed with the reference's 9-tuple batches (dataset_origin.py:181) on the host."""
    from types import SimpleNamespace
    V, B, N, S = 2048, 4, 6, 25
    cfgd = dict(vocab_size=V, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
                max_position_embeddings=64)

    def batches(seed0, n):
        out = []
        for i in range(n):
            b = mv.data.synthetic_batch(V, B, N, S, "bar", seed=seed0 + i, device="cpu")
            out.append((b["cls_tok"], b["input_txt"], b["txt_labels"], b["attn_mask"], (b["img_feats"], b["img_pos"]),
                        b["segment"], b["is_aligned"], b["sep_tok"], torch.zeros(B)))
        return out

    args = SimpleNamespace(with_cuda=True, weight_load=False, bert_model="custom", lr=1e-3, log_freq=10, mlm_task=True,
                           itm_task=True, cuda_devices=[0], dropout_prob=0.1)
    logged = []
    tr = mv.CXRBERT_Trainer(args, batches(1, 6) * 4, batches(100, 2), config=cfgd, dtype=torch.bfloat16,
                            logger=lambda d, step: logged.append((step, d)))
    r0 = tr.train(0)
    r1 = tr.train(1)
    assert {"avg_loss", "avg_mlm_loss", "avg_itm_loss", "itm_acc", "mlm_acc", "eval_avg_loss", "eval_mlm_loss", "eval_itm_loss",
            "eval_itm_acc", "eval_mlm_acc"} <= set(r0)
    assert np.isfinite(r0["avg_loss"]) and r1["avg_mlm_loss"] < r0["avg_mlm_loss"]        # it learns the 24 repeated batches
    assert len(logged) == 4
    tr.save(1, str(tmp_path))
    m2 = mv.CXRBERT.from_pretrained(str(tmp_path / "1"), device=DEV)
    a, c = tr.model.state_dict(), m2.state_dict()
    assert all(torch.equal(a[k].cpu(), c[k].cpu()) for k in a)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, FP32_TOL), (torch.bfloat16, BF16_TOL)])
def test_reference_default_geometry_and_ragged_text_lengths(dtype, tol):
    """main_origin.py's default shape (180 regions + 253 text -> L = 436, BAR mask; L is no multiple of any tile size)
    with the extreme text lengths in one batch: no text at all (only [SEP]), one token, and the full 253 tokens.
    Checked against the CPU oracle on the same inputs (forward logits + both losses + gradient of the fused step)."""
    cfg = O.OracleConfig(vocab_size=1024, hidden=128, layers=2, heads=2, intermediate=512, max_pos=256)
    N, S, B = 180, 253, 3
    P = O.make_params(cfg, seed=31)
    b = {k: torch.from_numpy(v) for k, v in synth.make_batch(cfg, B, N, S, "bar", seed=31).items()}
    from oracle import data_oracle as Dd
    for i, n_txt in enumerate((0, 1, S)):                      # overwrite the sampled lengths with the extremes
        toks = [10 + (300 + 7 * t) % 1000 for t in range(n_txt)]
        lab = [-100] * n_txt
        if n_txt:
            lab[0], toks[0] = toks[0], Dd.MASK
        ids, labels, seg, n_ids = Dd.assemble_sample(toks, lab, N, S)
        b["input_txt"][i], b["txt_labels"][i], b["segment"][i] = torch.from_numpy(ids), torch.from_numpy(labels), torch.from_numpy(seg)
        b["attn_mask"][i] = torch.from_numpy(Dd.build_mask("bar", N, S, n_ids))
    with torch.no_grad():
        mlm_o, itm_o = O.forward(P, cfg, b["cls_tok"], b["input_txt"], b["attn_mask"], b["segment"], b["img_feats"],
                                 b["img_pos"], b["sep_tok"])
        ml_o, il_o = O.losses(mlm_o, itm_o, b["txt_labels"], b["is_aligned"])
    model = make_model(cfg, P, dtype)
    with torch.no_grad():
        mlm, itm = fwd(model, b)
    assert mlm.shape == (B, 436, cfg.vocab_size)
    assert float((mlm.float().cpu() - mlm_o).abs().max()) < tol and float((itm.float().cpu() - itm_o).abs().max()) < tol
    stats = mv.TrainStep(model, lr=0.0)(dict(b), train=True).cpu()
    assert int(stats[1]) == 2                                   # sample 0 has no label at all
    assert abs(float(stats[0] / stats[1]) - float(ml_o)) < tol and abs(float(stats[3] / stats[4]) - float(il_o)) < tol
    assert bool(torch.isfinite(model.engine.flat_g).all())


def test_single_sample_batch_and_repeatability():
    """B = 1 (every per-batch reduction degenerates) and bit-repeatability of the forward on identical inputs."""
    cfg = O.CONFIGS["c1"]
    P = O.make_params(cfg, seed=8)
    b = {k: torch.from_numpy(v) for k, v in synth.make_batch(cfg, 1, 16, 45, "s2s", seed=8).items()}
    with torch.no_grad():
        mlm_o, itm_o = O.forward(P, cfg, b["cls_tok"], b["input_txt"], b["attn_mask"], b["segment"], b["img_feats"],
                                 b["img_pos"], b["sep_tok"])
    model = make_model(cfg, P, torch.bfloat16)
    with torch.no_grad():
        a1 = fwd(model, b)
        a2 = fwd(model, b)
    assert torch.equal(a1[0], a2[0]) and torch.equal(a1[1], a2[1])
    assert float((a1[0].float().cpu() - mlm_o).abs().max()) < BF16_TOL


def test_full_size_step_is_finite_repeatable_and_masks_pack_exactly():
    """BASELINE.json configs[1] at full size (B = 64, L = 512): mask packing bit-exact against numpy, forward
    bit-repeatable, a training step with dropout finite, loss near ln(V) for random-init weights."""
    from medvill_amd import hip_ops as ops
    cfg = mv.ModelConfig()
    model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=DEV)
    model.reset_parameters(seed=4)
    B, N, S = 64, 36, 473
    L = N + S + 3
    b = mv.data.synthetic_batch(cfg.vocab_size, B, N, S, "mixed", seed=9, device=DEV)
    bits = torch.zeros((B, L, L // 32), dtype=torch.int32, device=DEV)
    tinfo = torch.zeros((B, L // 64, L // 64), dtype=torch.uint8, device=DEV)
    ops.mask_pack(b["attn_mask"], bits, tinfo)
    m = b["attn_mask"].cpu().numpy() != 0
    want = (m.reshape(B, L, L // 32, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)
    assert np.array_equal(bits.cpu().numpy().view(np.uint32), want)
    model.eval()
    eng = model.engine
    with torch.no_grad():
        h1, p1 = eng.encoder_forward(b["cls_tok"], b["input_txt"], b["attn_desc"], b["segment"], b["img_feats"], b["img_pos"],
                                     b["sep_tok"])
        h1, p1 = h1.clone(), p1.clone()
        h2, p2 = eng.encoder_forward(b["cls_tok"], b["input_txt"], b["attn_mask"], b["segment"], b["img_feats"], b["img_pos"],
                                     b["sep_tok"])
    assert torch.equal(h1, h2) and torch.equal(p1, p2)          # descriptors == matrices, and repeatable
    model.train()
    ts = mv.TrainStep(model, lr=1e-5)
    st = ts(b, train=True).cpu()
    assert bool(torch.isfinite(eng.flat_p).all()) and bool(torch.isfinite(eng.flat_g).all())
    assert abs(float(st[0] / st[1]) - np.log(cfg.vocab_size)) < 0.5 and int(st[4]) == B


def test_fused_training_overfits_a_fixed_batch():
    """End-to-end behaviour of the fused step (bf16, dropout on, padding removal, HF AdamW): repeated steps on one
    small batch must drive both losses down monotonically-ish -- a wrong gradient sign, a stale bf16 shadow / transposed
    weight copy or a broken optimizer shows up here even when single-step parity passes."""
    cfg = mv.ModelConfig(hidden=128, heads=2, intermediate=512, layers=2, vocab_size=1024, max_pos=128)
    model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=DEV)
    model.reset_parameters(seed=11)
    model.train()
    batch = mv.data.synthetic_batch(cfg.vocab_size, 8, 6, 60, "mixed", seed=4, device=DEV)
    ts = mv.TrainStep(model, lr=1e-3)
    hist = []
    for _ in range(60):
        st = ts(batch, train=True).cpu()
        hist.append((float(st[0] / st[1]), float(st[3] / st[4])))
    assert model.engine.S["cu"] is not None                      # the packed path was the one exercised
    first = np.mean([h[0] for h in hist[:5]]), np.mean([h[1] for h in hist[:5]])
    last = np.mean([h[0] for h in hist[-5:]]), np.mean([h[1] for h in hist[-5:]])
    print("mlm / itm loss, first 5 steps:", first, "last 5:", last)
    assert last[0] < 0.5 * first[0] and last[1] < 0.5 * first[1] and np.isfinite(hist[-1]).all()


def test_retrieval_scores_on_packed_rows_equal_the_padded_ones():
    cfg = mv.ModelConfig(hidden=128, heads=2, intermediate=512, layers=2, vocab_size=1024, max_pos=128)
    r = mv.CXRBertForRetrieval(cfg, None, dtype=torch.bfloat16, device=DEV).eval()
    r.bert.reset_parameters(seed=8)
    B, N, S = 6, 6, 50
    b = mv.data.synthetic_batch(cfg.vocab_size, B, N, S, "1d", seed=12, device=DEV)
    assert b["attn_mask"].dim() == 2 and b["attn_desc"].packable()
    args = (b["cls_tok"], b["input_txt"])
    rest = (b["segment"], (b["img_feats"], b["img_pos"]), b["sep_tok"])
    padded = r.score(*args, b["attn_mask"], *rest)
    packed = r.score(*args, b["attn_desc"], *rest)
    assert r.bert.engine.S["cu"] is not None and r.bert.engine.S["M"] < B * (N + S + 3)
    assert float((padded - packed).abs().max()) < 1e-4      # the ITM linear runs in fp32 (torch) on one path, bf16 MFMA on the other


def test_graft_entry_smoke_runs():
    """The driver's round-end smoke check (forward + fused step of the tiny config against the oracle)."""
    import __graft_entry__ as g
    g.smoke()


def _kernel_dropout_masks(eng, B, Lq, H, A, n_layers):
    """keep * scale tensors of every dropout site of the engine's LAST forward, in LOGICAL [B, L, ...] coordinates -- whatever
    row layout the step ran in.  Hidden-state sites are keyed by the row index of the matrix the kernel saw: the packed row
    (mv_pack_plan's rowmap: packed row -> b*L + p) or, for the last layer's per-row part, the compact index into `sel`."""
    from medvill_amd import hip_ops as ops
    S = eng.S
    M = S["M"]
    rowmap = S["rowmap"][:M].long() if S["rowmap"] is not None else torch.arange(B * Lq, device=DEV)
    Lp = (Lq + 3) // 4 * 4
    masks = {}
    for (site, l), key in S["drop_keys"].items():
        if site == eng.SITE_ATTN:          # keep-bits tensor of the layer (blocks beyond a sample's packed length: unused garbage)
            keep = ops.attn_keep_mask(S["layers"][l]["dropbits"], B, Lq, A)
            if l == n_layers - 1 and S.get("tq") is not None:
                # the last layer ran on reordered rows (consumed rows first): entry (i, j) of the kernel's mask belongs to the logical
                # positions (perm[i], perm[j]) of the sample
                perm, cu = S["tq"][0].long(), S["cu"].long()
                out = keep.clone()
                for b in range(B):
                    lo, hi = int(cu[b]), int(cu[b + 1])
                    pb = perm[lo:hi] - lo
                    out[b][:, pb[:, None], pb[None, :]] = keep[b][:, :hi - lo, :hi - lo]
                keep = out
            masks[("attn", l)] = (keep.float() * (65536.0 / (65536.0 - 6554.0))).cpu()
            continue
        name = {eng.SITE_EMB: "emb", eng.SITE_OUT1: ("out1", l), eng.SITE_OUT2: ("out2", l)}[site]
        keep, sc = ops.dropout_mask(0.1, key, M * H, DEV)
        full = torch.ones((B * Lq, H), dtype=torch.float32, device=DEV)       # rows that do not exist: unused, any mask serves
        if site != eng.SITE_EMB and l == n_layers - 1 and S["sel"] is not None:
            sel = S["sel"].long()
            full[rowmap[sel]] = keep.view(-1, H)[:sel.numel()].float() * sc
        else:
            full[rowmap] = keep.view(M, H).float() * sc
        masks[name] = full.view(B, Lq, H).cpu()
    return masks


def test_the_benchmarked_path_against_the_oracle_at_bert_width():
    """The path bench.py times -- BERT width (H = 768, 12 heads, I = 3072), L = 512 = 36 + 476, ragged bidirectional masks so
    that the rows PACK, dropout 0.1 ON, the 16-bit path (f16 operands, loss-scaled f16 gradients), last layer on the consumed
    rows only -- against the CPU oracle (models/train_origin.py:95-131) fed with the very masks the kernels drew, mapped from
    packed / compact row indices back to logical positions.  Two layers keep the oracle's autograd pass to seconds."""
    cfg = O.OracleConfig(vocab_size=8192, hidden=768, layers=2, heads=12, intermediate=3072, max_pos=512)
    B, N, S = 4, 36, 473
    Lq = N + S + 3
    P = O.make_params(cfg, seed=77)
    bn = synth.make_batch(cfg, B, N, S, "full", seed=77)
    b = {k: torch.from_numpy(v) for k, v in bn.items()}
    model = make_model(cfg, P, torch.bfloat16)
    model.train()
    ts = mv.TrainStep(model, lr=0.0, pack_rows=True)
    assert ts.tail_rows
    batch = dict(b)
    batch["attn_desc"] = mv.data.MaskDesc.make("full", N, S, b["n_ids"], DEV)
    stats = ts(batch, train=True).cpu()
    eng = model.engine
    assert eng.S["cu"] is not None and eng.S["sel"] is not None and eng.S["M"] < B * Lq          # packed + consumed-rows layer
    assert eng.dt == mv._lib.MV_F16 and eng.fdt == mv._lib.MV_F16 and not eng.dual             # one f16 encoding
    assert float(eng.scaler[3]) == 0.0 and float(eng.scaler[4]) == 1.0                         # the step was not skipped
    masks = _kernel_dropout_masks(eng, B, Lq, cfg.hidden, cfg.heads, cfg.layers)
    Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    mlm, itm = O.forward(Po, cfg, b["cls_tok"], b["input_txt"], b["attn_mask"], b["segment"], b["img_feats"], b["img_pos"],
                         b["sep_tok"], masks=masks)
    ml, il = O.losses(mlm, itm, b["txt_labels"], b["is_aligned"])
    (ml + il).backward()
    assert abs(float(stats[0] / stats[1]) - float(ml)) < 1e-2 and abs(float(stats[3] / stats[4]) - float(il)) < 1e-2
    g_ref = torch.cat([Po[k].grad.reshape(-1) for k in P])
    g_got = torch.cat([eng.g[k].reshape(-1) for k in P]).cpu()
    assert bool(torch.isfinite(g_got).all())
    rel = float((g_got - g_ref).norm() / g_ref.norm())
    assert rel < 3e-2, rel
    worst = max(float((eng.g[k].cpu() - Po[k].grad).norm() / (Po[k].grad.norm() + 1e-3 * g_ref.norm() / len(P))) for k in P)
    print(f"packed x dropout x f16 at BERT width: loss diff {abs(float(stats[0] / stats[1]) - float(ml)):.2e}, grad rel-L2 {rel:.2e}, worst tensor {worst:.2e}")
    assert worst < 0.1


@pytest.mark.parametrize("gop", ["f16", "bf16"])
def test_f16_operands_survive_outlier_channels(gop):
    """Trained BERT checkpoints carry outlier channels (LayerNorm gains of tens, a residual channel in the hundreds); the parity
    fixtures' N(0, 0.02) weights never leave |x| < 3.  f16 saturates at 65504: an outlier fixture (a few LayerNorm gains x30, one
    residual channel at 300 via the embedding LayerNorm's bias) must stay finite, stay close to the exact fp32 path, take an
    optimizer step without tripping the loss scale, and report overflow when it is driven past f16's range on purpose.
    Gradient tolerance: profiles/r03_outlier_probe.txt (the outlier channel's LayerNorm backward cancels to 1 - xhat^2/H, so its
    relative error grows with the outlier; f16 gradient operands hold 2e-2 at 300 where bf16 ones are at 2e-1)."""
    cfg = O.OracleConfig(vocab_size=2048, hidden=768, layers=2, heads=12, intermediate=3072, max_pos=256)
    B, N, S = 2, 16, 100
    P = O.make_params(cfg, seed=13)
    P["enc.txt_embeddings.LayerNorm.bias"][37] = 300.0                        # a residual-stream outlier channel
    for l in range(cfg.layers):
        for ln in ("attention.output.LayerNorm.weight", "output.LayerNorm.weight"):
            P[f"enc.encoder.layer.{l}.{ln}"][[5, 111, 300]] *= 30.0           # outlier LayerNorm gains
    b = {k: torch.from_numpy(v) for k, v in synth.make_batch(cfg, B, N, S, "s2s", seed=13).items()}
    m32 = make_model(cfg, P, torch.float32)
    m16 = mv.CXRBERT(cfg_dict(cfg), None, dtype=torch.bfloat16, device=DEV, grad_operand=gop)
    m16.load_state_dict(P, strict=True)
    m16.eval()
    with torch.no_grad():
        l32, i32 = fwd(m32, b)
        l16, i16 = fwd(m16, b)
    assert bool(torch.isfinite(l16).all()) and bool(torch.isfinite(i16).all())
    scale = float(l32.abs().max())
    assert float((l16.float() - l32).abs().max()) < 2e-2 * max(scale, 1.0), (float((l16.float() - l32).abs().max()), scale)
    ts32, ts16 = mv.TrainStep(m32, lr=1e-4), mv.TrainStep(m16, lr=1e-4)
    s32, s16 = ts32(dict(b), train=True).cpu(), ts16(dict(b), train=True).cpu()
    assert abs(float(s16[0] / s16[1]) - float(s32[0] / s32[1])) < 2e-2 * max(1.0, float(s32[0] / s32[1]))
    g32, g16 = m32.engine.flat_g, m16.engine.flat_g
    assert bool(torch.isfinite(g16).all())
    assert float((g16 - g32).norm() / g32.norm()) < (5e-2 if gop == "f16" else 0.3)
    if gop == "f16":
        st = m16.engine.scaler.cpu()
        assert st[3] == 0 and st[4] == 1 and st[5] == 0                       # applied, nothing skipped
        # driven past f16's range on purpose: the overflow is REPORTED (skip flag, halved scale), the weights do not move
        m16.engine.reset_scaler(2.0 ** 40)
        p_before = m16.engine.flat_p.clone()
        ts16(dict(b), train=True)
        st = m16.engine.scaler.cpu()
        assert st[3] == 1 and st[5] == 1 and st[0] == 2.0 ** 39
        m16.engine.wait_optimizer()
        torch.cuda.synchronize()
        assert torch.equal(m16.engine.flat_p, p_before)
