"""Mirror of the reference trainer (models/train_origin.py): CXRBERT_Trainer(args, train_dataloader,
test_dataloader).train(epoch) / .save(epoch, path), driving the fused HIP training step.

One step (train_origin.py:95-146) = H2D of the batch -> forward -> CE(mlm, ignore -100) + CE(itm)
-> zero_grad / backward / HF-AdamW -> ITM and MLM accuracy counters.  Here that is ONE explicit
kernel schedule (TrainStep): the MLM head runs on the labelled rows only (unlabelled rows have
exactly zero loss and gradient), the losses / argmax metrics / logit gradients come from one
fused kernel, the optimizer is one fused kernel over the flat parameter buffer, and the only
host synchronisation is reading six floats when the caller asks for the numbers.
"""
from __future__ import annotations

import os

import torch

from . import data as D
from .cxrbert import CXRBERT, model_config_from
from .dist import GradAllReducer


class TrainStep:
    """Fused pretraining step over a CXRBERT model (single GPU or one rank of a DP job)."""

    def __init__(self, model: CXRBERT, lr=1e-5, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, distributed=False, group=None,
                 mlm_task=True, itm_task=True, pack_rows=True):
        # HF AdamW defaults, as effectively used by the reference: train_origin.py:60 passes only lr
        self.model, self.eng = model, model.engine
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.step_cnt = 0
        self.mlm_task, self.itm_task = mlm_task, itm_task
        # padding removal: when a batch carries mask descriptors of the full / seq2seq / 1-D families (bf16 path), the
        # encoder runs on the valid rows only; results are those of the padded run (see Engine.encoder_forward)
        self.pack_rows = pack_rows
        self.eng.ensure_opt()
        self.dp = None
        if distributed:
            self.dp = GradAllReducer(self.eng.flat_g, self.eng.layout, self.eng.n_flat, self.eng.cfg.layers, group=group)

    def _prep(self, batch):
        dev = self.eng.device
        rows, ids = batch.get("label_rows"), batch.get("label_ids")
        if rows is None:
            rows, ids = D.label_index(batch["txt_labels"])
        return rows.to(dev), ids.to(dev), batch["is_aligned"].to(dev, torch.int32)

    def __call__(self, batch, train=True):
        """batch: dict with the reference's batch fields (cls_tok, input_txt, attn_mask, segment,
        img_feats, img_pos, sep_tok, txt_labels, is_aligned) [+ label_rows/label_ids].
        Returns the device tensor stats f32[6] = [mlm_nll_sum, n_lab, mlm_correct, itm_nll_sum, B, itm_correct]
        (local to this rank); no host sync happens here."""
        eng = self.eng
        rows, ids, aligned = self._prep(batch)
        eng.training = bool(train and self.model.training)      # dropout like the reference's model.train()
        if train:
            eng.flat_g.zero_()
        desc = batch.get("attn_desc")
        pack = bool(self.pack_rows and desc is not None and eng.adt == torch.bfloat16 and desc.packable())
        eng.encoder_forward(batch["cls_tok"], batch["input_txt"], desc if pack else batch["attn_mask"], batch["segment"],
                            batch["img_feats"], batch["img_pos"], batch["sep_tok"], pack=pack)
        R, B = int(rows.numel()), int(aligned.numel())
        # loss normalisation = the reference's means over the GLOBAL mini-batch (train_origin.py:120-126)
        mlm_dev = itm_dev = None
        mlm_scale, itm_scale = 1.0 / max(R, 1), 1.0 / B
        if self.dp is not None and self.dp.world > 1:
            inv = torch.reciprocal(torch.clamp(self.dp.global_counts(R, B, eng.device), min=1.0))
            mlm_dev, itm_dev = inv[0:1], inv[1:2]
        if not self.mlm_task:
            mlm_dev, mlm_scale = None, 0.0
        if not self.itm_task:
            itm_dev, itm_scale = None, 0.0
        stats = eng.heads_train(rows, ids, aligned, mlm_scale_dev=mlm_dev, mlm_scale=mlm_scale, itm_scale=itm_scale,
                                itm_scale_dev=itm_dev, compute_grad=train)
        if train:
            eng.encoder_backward(bucket_hook=self.dp.hook if self.dp is not None else None)
            if self.dp is not None:
                self.dp.finish()
            self.step_cnt += 1
            eng.adamw_step(self.step_cnt, lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.wd)
        return stats


def _str2bool(v):
    return v if isinstance(v, bool) else str(v).lower() not in ("false", "0", "no", "")


class CXRBERT_Trainer:
    """train_origin.py:19-266.  `args` carries the reference's fields (with_cuda, weight_load,
    pre_trained_model_path, bert_model, cuda_devices, lr, log_freq, mlm_task, itm_task, ...)."""

    def __init__(self, args, train_dataloader, test_dataloader=None, config=None, dtype=torch.bfloat16, logger=None):
        self.args = args
        if not (torch.cuda.is_available() and getattr(args, "with_cuda", True)):
            raise RuntimeError("CXRBERT_Trainer needs an MI355X (ROCm) device; there is no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device())
        if getattr(args, "weight_load", False):
            self.model = CXRBERT.from_pretrained(args.pre_trained_model_path, args=args, dtype=dtype, device=self.device)
            print("training restart with mid epoch")
        else:
            if config is None:
                config = BERT_CONFIGS.get(getattr(args, "bert_model", "bert-base-scratch"), BERT_CONFIGS["bert-base-scratch"])
            # cxrbert_origin.py:59-65: anything but 'ViT' is the ResNet-50 region encoder; it is only built when the loader
            # will hand over pixels (args.pixels / an explicit request), since feature batches never touch it
            want_cnn = bool(getattr(args, "pixels", False)) and getattr(args, "img_encoder", "random-pixel") != "ViT"
            self.model = CXRBERT(config, args, dtype=dtype, device=self.device, img_encoder="resnet50" if want_cnn else None)
        self.train_data, self.test_data = train_dataloader, test_dataloader
        self.distributed = torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size() > 1
        self.mlm_task = _str2bool(getattr(args, "mlm_task", True))
        self.itm_task = _str2bool(getattr(args, "itm_task", True))
        self.step = TrainStep(self.model, lr=getattr(args, "lr", 1e-5), distributed=self.distributed, mlm_task=self.mlm_task,
                              itm_task=self.itm_task)
        self.log_freq = getattr(args, "log_freq", 10)
        self.logger = logger            # optional callable(dict, step=epoch): stands in for wandb.log
        print("Total Parameters:", sum(p.nelement() for p in self.model.parameters()))

    def _to_batch(self, data):
        cls_tok, input_ids, txt_labels, attn_masks, img, segment, is_aligned, sep_tok = data[:8]
        if torch.is_tensor(img):     # pixels [B,3,H,W] (dataset_origin.py:85-89): region features from the mirrored CNN
            if self.model.img_encoder is None:
                raise TypeError("the loader yields pixels: construct the trainer with args.pixels=True (ResNet-50 region encoder)")
            with torch.no_grad():
                img = self.model.img_encoder(img.to(self.device))
        feats, pos = img            # (region feats [B,N,2048], region positions [B,N])
        return dict(cls_tok=cls_tok, input_txt=input_ids, attn_mask=attn_masks, segment=segment, img_feats=feats, img_pos=pos,
                    sep_tok=sep_tok, txt_labels=txt_labels, is_aligned=is_aligned)

    def _run_epoch(self, loader, epoch, train):
        tot = torch.zeros(6, dtype=torch.float64)
        losses, mlm_l, itm_l = [], [], []
        for i, data in enumerate(loader):
            stats = self.step(self._to_batch(data), train=train).double().cpu()   # the one sync per step
            tot += stats
            ml = float(stats[0] / max(stats[1], 1.0))
            il = float(stats[3] / max(stats[4], 1.0))
            mlm_l.append(ml)
            itm_l.append(il)
            losses.append((ml if self.mlm_task else 0.0) + (il if self.itm_task else 0.0))
        n = max(len(losses), 1)
        pre = "" if train else "eval_"
        out = {pre + "avg_loss": sum(losses) / n, pre + "avg_mlm_loss" if train else "eval_mlm_loss": sum(mlm_l) / n,
               pre + "avg_itm_loss" if train else "eval_itm_loss": sum(itm_l) / n,
               pre + "itm_acc": float(tot[5] / max(tot[4], 1.0)) * 100, pre + "mlm_acc": float(tot[2] / max(tot[1], 1.0)) * 100}
        print(("avg loss per epoch" if train else "avg loss in testset"), out[pre + "avg_loss"])
        print(("avg itm acc per epoch" if train else "avg itm acc in testset"), round(out[pre + "itm_acc"], 3))
        if self.logger is not None:
            self.logger(out, step=epoch)
        return out

    def train(self, epoch):
        self.model.train()
        res = self._run_epoch(self.train_data, epoch, True)
        if self.test_data is not None:
            self.model.eval()
            res.update(self._run_epoch(self.test_data, epoch, False))
        return res

    def save(self, epoch, file_path):
        save_path_per_ep = os.path.join(file_path, str(epoch))
        if not os.path.exists(save_path_per_ep):
            os.makedirs(save_path_per_ep, exist_ok=True)
            os.chmod(save_path_per_ep, 0o777)
        self.model.save_pretrained(save_path_per_ep)
        print(f"EP: {epoch} Model saved on {save_path_per_ep}")
        os.chmod(save_path_per_ep + "/pytorch_model.bin", 0o777)


BERT_CONFIGS = {   # offline stand-ins for BertConfig.from_pretrained(...) at train_origin.py:36-47
    "bert-base-scratch": dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                              intermediate_size=3072, max_position_embeddings=512),
    "bert-base-uncased": dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                              intermediate_size=3072, max_position_embeddings=512),
    "bert-small-scratch": dict(vocab_size=30522, hidden_size=512, num_hidden_layers=4, num_attention_heads=8,
                               intermediate_size=2048, max_position_embeddings=512),
}
