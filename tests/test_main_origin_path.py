"""`main_origin.py` drives the build unchanged (north_star): its import line `from models.train_origin import CXRBERT_Trainer`
(main_origin.py:19) resolves to medvill_amd through the builder-written `models/` package at the repo root, and the `args`
Namespace its argparse block builds (main_origin.py:66-153; flag names and defaults RESTATED here, the reference file is not
read) is what the trainer consumes."""
import argparse
import os
import sys

import pytest
import torch

import medvill_amd as mv


def main_origin_parser():
    """Flag names, types and defaults of main_origin.py:66-153 (restated).  `type=bool` is kept as the reference has it: any
    non-empty string parses as True; `--mlm_task` / `--itm_task` are `type=str` with a bool default."""
    p = argparse.ArgumentParser()
    a = p.add_argument
    a("--train_dataset", type=str, default="/home/mimic-cxr/dataset/new_dset/Train_253.jsonl")
    a("--test_dataset", type=str, default="/home/mimic-cxr/dataset/new_dset/Valid_253.jsonl")
    a("--output_path", type=str, default="output/x")
    a("--log_freq", type=int, default=10)
    a("--with_cuda", type=bool, default=True)
    a("--cuda_devices", type=int, nargs="+", default=None)
    a("--mlm_task", type=str, default=True)
    a("--itm_task", type=str, default=True)
    a("--attn_1d", type=bool, default=False)
    a("--BAR_attn", default=True, type=bool)
    a("--Mixed", default=False, type=bool)
    a("--s2s_prob", default=1.0, type=float)
    a("--bi_prob", default=0.0, type=float)
    a("--disturbing_mask", default=False, type=bool)
    a("--epochs", type=int, default=50)
    a("--batch_size", type=int, default=36)
    a("--num_workers", type=int, default=20)
    a("--hidden_size", type=int, default=768, choices=[768, 512, 128])
    a("--embedding_size", type=int, default=768, choices=[768, 512, 128])
    a("--weight_load", type=bool, default=False)
    a("--pre_trained_model_path", type=str, default="/home/cxr-bert/clinicalbert_vlp_re35_5")
    a("--bert_model", type=str, default="bert-base-scratch")
    a("--vocab_size", type=int, default=30522, choices=[30522, 30000, 28996])
    a("--img_postion", default=True)
    a("--seq_len", type=int, default=253, choices=[128, 253])
    a("--max_seq_len", type=int, default=512)
    a("--img_hidden_sz", type=int, default=2048)
    a("--img_encoder", type=str, default="random-pixel", choices=["random-pixel", "full-fiber", "ViT"])
    a("--img_channel", type=int, default=3, choices=[1, 3])
    a("--num_image_embeds", type=int, default=180, choices=[36, 49, 180, 256])
    a("--img_size", type=int, default=512, choices=[224, 512])
    a("--img_embed_pool_type", type=str, default="max", choices=["max", "avg"])
    a("--lr", type=float, default=1e-5)
    a("--gradient_accumulation_steps", type=int, default=4)
    a("--warmup", type=float, default=0.1)
    a("--seed", type=int, default=123)
    a("--warmup_steps", type=int, default=0)
    a("--dropout_prob", type=float, default=0.1)
    a("--beta1", type=float, default=0.9)
    a("--beta2", type=float, default=0.999)
    a("--eps", type=float, default=1e-6)
    a("--weight_decay", type=float, default=0.01)
    return p


def test_models_train_origin_resolves_to_the_build():
    from models.train_origin import CXRBERT_Trainer        # the import line of main_origin.py:19, verbatim
    from models.cxrbert_origin import CXRBERT               # train_origin.py:13 / Retrieval/retrieval.py
    assert CXRBERT_Trainer is mv.CXRBERT_Trainer and CXRBERT is mv.CXRBERT
    src = os.path.dirname(sys.modules["models.train_origin"].__file__)
    assert os.path.samefile(src, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models"))


def test_args_namespace_of_main_origin_is_what_the_trainer_reads():
    from medvill_amd.trainer import BERT_CONFIGS, _str2bool
    args = main_origin_parser().parse_args([])
    assert args.bert_model in BERT_CONFIGS and args.lr == 1e-5 and args.with_cuda is True and args.weight_load is False
    assert args.BAR_attn is True and args.Mixed is False                     # the reference's default mask family is BAR
    # type=str flags with a bool default: "False" on the command line must mean False (fixed with a str2bool)
    a2 = main_origin_parser().parse_args(["--mlm_task", "False", "--itm_task", "True"])
    assert a2.mlm_task == "False" and _str2bool(a2.mlm_task) is False and _str2bool(a2.itm_task) is True
    # every --bert_model choice that names a BERT geometry has an offline stand-in for BertConfig.from_pretrained
    for name in ("bert-base-uncased", "google/bert_uncased_L-4_H-512_A-8", "google/bert_uncased_L-2_H-128_A-2",
                 "emilyalsentzer/Bio_ClinicalBERT", "bionlp/bluebert_pubmed_mimic_uncased_L-12_H-768_A-12", "bert-small-scratch",
                 "bert-base-scratch"):
        c = mv.cxrbert.model_config_from(BERT_CONFIGS[name])
        assert c.hidden % c.heads == 0 and c.hidden // c.heads == 64
    if not torch.cuda.is_available():
        from models.train_origin import CXRBERT_Trainer
        with pytest.raises(RuntimeError, match="MI355X"):                   # no CPU path: loud, not a silent fallback
            CXRBERT_Trainer(args, train_dataloader=[], test_dataloader=None)


def _hf_bert_state(name):
    """A genuine HF BertModel state dict (its own key names) at the geometry of a --bert_model choice, randomly initialised."""
    import transformers
    from medvill_amd.trainer import BERT_CONFIGS
    torch.manual_seed(7)
    return transformers.BertModel(transformers.BertConfig(**BERT_CONFIGS[name])).state_dict()


def test_bert_model_names_that_mean_pretrained_weights_never_random_init_silently(tmp_path):
    """cxrbert_origin.py:43-55: only `bert-base-scratch` / `bert-small-scratch` build a random BertModel; every other --bert_model
    value loads PRETRAINED weights.  Here (no network) such a value needs the weights handed over, else it raises."""
    import warnings
    from medvill_amd.checkpoint import from_hf_bert_keys
    from medvill_amd.trainer import BERT_CONFIGS, load_pretrained_bert, resolve_bert_model
    tiny = "google/bert_uncased_L-2_H-128_A-2"
    for name in ("bert-base-scratch", "bert-small-scratch"):
        cfg, sd = resolve_bert_model(main_origin_parser().parse_args(["--bert_model", name]), name)
        assert sd is None and cfg == BERT_CONFIGS[name]
    for name in ("bert-base-uncased", tiny, "emilyalsentzer/Bio_ClinicalBERT", "bionlp/bluebert_pubmed_mimic_uncased_L-12_H-768_A-12"):
        with pytest.raises(RuntimeError, match="PRETRAINED"):
            resolve_bert_model(main_origin_parser().parse_args(["--bert_model", name]), name)
    with pytest.raises(NotImplementedError):
        resolve_bert_model(main_origin_parser().parse_args([]), "albert-base-v2")
    args = main_origin_parser().parse_args(["--bert_model", tiny])
    args.allow_random_init = True                                            # the explicit opt-out warns
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert resolve_bert_model(args, tiny) == (BERT_CONFIGS[tiny], None)
    assert any("RANDOM" in str(x.message) for x in w)
    # weights handed over: as a state dict, and as a checkpoint directory (pytorch_model.bin + config.json)
    hf = _hf_bert_state(tiny)
    args = main_origin_parser().parse_args(["--bert_model", tiny])
    args.init_state_dict = hf
    cfg, sd = resolve_bert_model(args, tiny)
    assert sd is hf
    d = tmp_path / "tiny_bert"
    os.makedirs(d)
    torch.save({"bert." + k: v for k, v in hf.items()}, d / "pytorch_model.bin")      # BertForPreTraining-style prefix
    import json
    with open(d / "config.json", "w") as f:
        json.dump(BERT_CONFIGS[tiny], f)
    cfg2, sd2 = resolve_bert_model(main_origin_parser().parse_args([]), str(d))
    assert cfg2 == BERT_CONFIGS[tiny] and set(from_hf_bert_keys(sd2)) == set(from_hf_bert_keys(hf))
    # the weights land in the modules the reference takes from `bert` (cxrbert_origin.py:56-57,72-73), shared tensors follow
    m = mv.CXRBERT(cfg, None, device="cpu")
    before = {k: v.clone() for k, v in m.state_dict().items()}
    n = load_pretrained_bert(m, sd2)
    after = m.state_dict()
    assert n == sum(1 for k in m._param_names if k.startswith(("enc.txt_embeddings.", "enc.encoder.", "enc.pooler.")))
    assert torch.equal(after["enc.txt_embeddings.word_embeddings.weight"], hf["embeddings.word_embeddings.weight"])
    assert torch.equal(after["mlm.predictions.decoder.weight"], hf["embeddings.word_embeddings.weight"])          # tied
    assert torch.equal(after["enc.img_embeddings.LayerNorm.weight"], hf["embeddings.LayerNorm.weight"])           # shared
    assert torch.equal(after["enc.encoder.layer.1.output.dense.weight"], hf["encoder.layer.1.output.dense.weight"])
    assert torch.equal(after["enc.pooler.dense.bias"], hf["pooler.dense.bias"])
    for k in ("itm.linear.weight", "mlm.predictions.transform.dense.weight", "enc.img_embeddings.img_embeddings.weight"):
        assert torch.equal(after[k], before[k])                               # the heads / image projection keep their own init
    bad = dict(hf)
    bad.pop("encoder.layer.0.attention.self.query.weight")
    with pytest.raises(RuntimeError, match="lacks"):
        load_pretrained_bert(m, bad)


@pytest.mark.gpu
def test_main_origin_loop_through_the_models_import_path(tmp_path):
    """The body of main_origin.py:52-62 -- build the trainer from `args`, `train(epoch)`, `save(epoch, output_path)` -- through
    `models.train_origin`, fed by a host-side DataLoader of the reference's 9-tuples (dataset_origin.py:181) with the default
    BAR masks as int64 [B,L,L] matrices."""
    from torch.utils.data import DataLoader, Dataset

    from models.train_origin import CXRBERT_Trainer
    args = main_origin_parser().parse_args(["--bert_model", "google/bert_uncased_L-2_H-128_A-2", "--batch_size", "4", "--num_image_embeds", "36",
                                            "--seq_len", "128", "--lr", "1e-3", "--output_path", str(tmp_path), "--epochs", "2"])
    # this --bert_model value means "start from the pretrained BERT-Tiny" in the reference: the weights are handed over (no network)
    args.init_state_dict = _hf_bert_state(args.bert_model)
    torch.manual_seed(args.seed)                       # utils.set_seed(args.seed) of main_origin.py:25
    N, S, V = args.num_image_embeds, 40, 30522

    class Tuples(Dataset):                             # stands in for CXRDataset: one sample = the 9-tuple, per-sample tensors
        def __init__(self, n, seed):
            self.items = []
            for i in range(n):
                b = mv.data.synthetic_batch(V, 1, N, S, "bar", seed=seed + i, device="cpu")
                self.items.append((b["cls_tok"][0], b["input_txt"][0], b["txt_labels"][0], b["attn_mask"][0],
                                   (b["img_feats"][0], b["img_pos"][0]), b["segment"][0], b["is_aligned"][0], b["sep_tok"][0],
                                   torch.zeros(())))

        def __len__(self):
            return len(self.items)

        def __getitem__(self, i):
            return self.items[i]
    train_dl = DataLoader(Tuples(16, 1), batch_size=args.batch_size, num_workers=0, shuffle=True)
    test_dl = DataLoader(Tuples(4, 100), batch_size=args.batch_size, num_workers=0, shuffle=False)
    trainer = CXRBERT_Trainer(args, train_dataloader=train_dl, test_dataloader=test_dl)
    assert trainer.model.cfg.hidden == 128 and trainer.model.cfg.layers == 2
    assert torch.equal(trainer.model.state_dict()["enc.encoder.layer.0.intermediate.dense.weight"].cpu(),
                       args.init_state_dict["encoder.layer.0.intermediate.dense.weight"])
    res = []
    for epoch in range(args.epochs):
        res.append(trainer.train(epoch))
        trainer.save(epoch, args.output_path)
    assert all(torch.isfinite(torch.tensor(r["avg_loss"])) for r in res) and res[1]["avg_mlm_loss"] < res[0]["avg_mlm_loss"]
    assert os.path.exists(os.path.join(args.output_path, "1", "pytorch_model.bin"))
    assert os.path.exists(os.path.join(args.output_path, "1", "config.json"))
    assert trainer.n_recognised > 0                     # the BAR matrices were recognised (descriptors, no per-step 4 MB upload)


def test_launcher_resolves_models_to_the_build_even_from_a_checkout_with_its_own_models(tmp_path):
    """run_main_origin.py: executed from a directory that has its OWN `models` package and a main_origin.py whose first statement is
    the reference's import line, the import must land on this repo's `models/` (a script's directory otherwise shadows PYTHONPATH)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / "models").mkdir()
    (tmp_path / "models" / "__init__.py").write_text("")
    (tmp_path / "models" / "train_origin.py").write_text("class CXRBERT_Trainer:\n    ORIGIN = 'the checkout'\n")
    (tmp_path / "main_origin.py").write_text(
        "from models.train_origin import CXRBERT_Trainer  # CXR-BERT\n"
        "import sys\nprint('RESOLVED', CXRBERT_Trainer.__module__, sys.argv[1:])\n")
    r = subprocess.run([sys.executable, os.path.join(root, "run_main_origin.py"), "--epochs", "1"], cwd=str(tmp_path), capture_output=True, text=True,
                       env={**os.environ, "CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""})
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RESOLVED medvill_amd.trainer ['--epochs', '1']" in r.stdout, r.stdout
