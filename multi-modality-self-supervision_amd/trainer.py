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
                 mlm_task=True, itm_task=True, pack_rows=True, overlap_optimizer=False):
        # HF AdamW defaults, as effectively used by the reference: train_origin.py:60 passes only lr
        self.model, self.eng = model, model.engine
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.step_cnt = 0
        self.mlm_task, self.itm_task = mlm_task, itm_task
        # padding removal: when a batch carries mask descriptors of the full / seq2seq / 1-D families (bf16 path), the
        # encoder runs on the valid rows only; results are those of the padded run (see Engine.encoder_forward)
        self.pack_rows = pack_rows
        self.tail_rows = True        # last layer after its attention: consumed rows only (exact; Engine.encoder_forward)
        # AdamW range by range on the side stream, under the next step's first layers (Engine.adamw_step).  Opt-in: whoever reads
        # parameters on the current stream between steps without going through the model must call sync() first
        self.overlap_optimizer = overlap_optimizer
        self.eng.ensure_opt()
        self.dp = None
        self.time_exchange = False
        if distributed:
            self.dp = GradAllReducer(self.eng.flat_g, self.eng.layout, self.eng.n_flat, self.eng.cfg.layers, group=group,
                                     stream=self.eng.side_stream() if self.eng.device.type == "cuda" else None)
            self.dp.check_replicas(self.eng.flat_p)
            # every rank draws its own dropout masks (the same key would repeat rank 0's masks on every shard)
            self.eng.drop_seed = (self.eng.drop_seed + 0x9E3779B97F4A7C15 * (self.dp.rank + 1)) & 0xFFFFFFFFFFFFFFFF

    def exchange_exposed_ms(self):
        return self.dp.exposed_ms() if self.dp is not None else 0.0

    def exchange_timeline(self):
        """Per-bucket issue / completion times of the gradient all-reduce (dist.GradAllReducer.bucket_timeline), or None."""
        return self.dp.bucket_timeline() if self.dp is not None else None

    def _prep(self, batch):
        dev = self.eng.device
        rows, ids = batch.get("label_rows"), batch.get("label_ids")
        if rows is None:
            rows, ids = D.label_index(batch["txt_labels"])
        return rows.to(dev), ids.to(dev), batch["is_aligned"].to(dev, torch.int32)

    def __call__(self, batch, train=True, use_desc=True):
        """batch: dict with the reference's batch fields (cls_tok, input_txt, attn_mask, segment,
        img_feats, img_pos, sep_tok, txt_labels, is_aligned) [+ label_rows/label_ids] [+ attn_desc].
        Returns the device tensor stats f32[6] = [mlm_nll_sum, n_lab, mlm_correct, itm_nll_sum, B, itm_correct]
        (local to this rank); no host sync happens here.
        use_desc=False: ignore batch["attn_desc"] and run on the materialised batch["attn_mask"] (padded rows).  Which of the two a
        rank takes changes no collective (same count all-reduce, same gradient buckets), so under data parallelism every rank
        decides for itself (CXRBERT_Trainer does, from its host-side check of the shipped matrix)."""
        stats = self._run(batch, train, use_desc=use_desc)
        if train:
            self.step_cnt += 1
            # f16 gradients: overflow check of the (all-reduced) flat gradient, device-side skip / loss-scale decision
            self.eng.check_overflow()
            self.eng.adamw_step(self.step_cnt, lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.wd,
                                overlap=self.overlap_optimizer, use_scaler=True)
        return stats

    def sync(self):
        """Order the current stream behind an overlapped optimizer step (no-op otherwise)."""
        self.eng.wait_optimizer()

    def _run(self, batch, train, use_desc):
        eng = self.eng
        if self.dp is not None:
            self.dp.timing = self.time_exchange
        rows, ids, aligned = self._prep(batch)
        eng.training = bool(train and self.model.training)      # dropout like the reference's model.train()
        eng.keep_acts = bool(train)                              # eval steps keep no per-layer activations
        zero_ev = None
        if train:
            # the flat gradient is cleared on the engine's side stream, which is idle during the forward; the heads wait for it
            main = torch.cuda.current_stream() if eng.device.type == "cuda" else None
            side = eng.side_stream() if main is not None else None
            if side is not None and side is not main:
                side.wait_stream(main)                 # behind the previous step's optimizer (it read the gradients)
                with torch.cuda.stream(side):
                    eng.flat_g.zero_()
                    zero_ev = torch.cuda.Event()
                    zero_ev.record(side)
            else:
                eng.flat_g.zero_()
        desc = batch.get("attn_desc") if use_desc else None
        pack = bool(self.pack_rows and desc is not None and eng.is16 and desc.packable())
        mask = desc if desc is not None else batch["attn_mask"]      # descriptors when there are any: no [B,L,L] traffic
        # the last layer's per-row work runs only on the rows the heads consume (labelled rows + each sample's first row)
        eng.encoder_forward(batch["cls_tok"], batch["input_txt"], mask, batch["segment"],
                            batch["img_feats"], batch["img_pos"], batch["sep_tok"], pack=pack,
                            tail_rows=rows if self.tail_rows else None)
        R, B = int(rows.numel()), int(aligned.numel())
        # loss normalisation = the reference's means over the GLOBAL mini-batch (train_origin.py:120-126)
        mlm_dev = itm_dev = None
        mlm_scale, itm_scale = 1.0 / max(R, 1), 1.0 / B
        if train and self.dp is not None and (self.dp.world > 1 or self.dp.force):      # eval: local sums only, no collective (ranks may
            # hold different numbers of eval batches)
            inv = torch.reciprocal(torch.clamp(self.dp.global_counts(R, B, eng.device), min=1.0))
            mlm_dev, itm_dev = inv[0:1], inv[1:2]
        if not self.mlm_task:
            mlm_dev, mlm_scale = None, 0.0
        if not self.itm_task:
            itm_dev, itm_scale = None, 0.0
        if zero_ev is not None:
            torch.cuda.current_stream().wait_event(zero_ev)
        stats = eng.heads_train(rows, ids, aligned, mlm_scale_dev=mlm_dev, mlm_scale=mlm_scale, itm_scale=itm_scale,
                                itm_scale_dev=itm_dev, compute_grad=train)
        if train:
            eng.encoder_backward(bucket_hook=self.dp.hook if self.dp is not None else None)
            if self.dp is not None:
                from .engine import phase
                phase("exchange")          # roctx range (MV_ROCTX): the wait for the outstanding gradient buckets
                self.dp.finish()
                phase(None)
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
        # cxrbert_origin.py:59-65: anything but 'ViT' is the ResNet-50 region encoder; it is only built when the loader
        # will hand over pixels (args.pixels / an explicit request), since feature batches never touch it
        want_cnn = bool(getattr(args, "pixels", False)) and getattr(args, "img_encoder", "random-pixel") != "ViT"
        cnn = "resnet50" if want_cnn else None
        if getattr(args, "weight_load", False):
            # train_origin.py:28-34; the checkpoint's enc.img_encoder.* weights are restored into the region encoder
            self.model = CXRBERT.from_pretrained(args.pre_trained_model_path, args=args, dtype=dtype, device=self.device, img_encoder=cnn)
            print("training restart with mid epoch")
        else:
            init_sd = None
            if config is None:
                name = getattr(args, "bert_model", "bert-base-scratch")
                config, init_sd = resolve_bert_model(args, name)
            self.model = CXRBERT(config, args, dtype=dtype, device=self.device, img_encoder=cnn)
            if init_sd is not None:
                load_pretrained_bert(self.model, init_sd)
            tv = getattr(args, "resnet50_weights", None)       # path to torchvision's resnet50 state dict (no network here)
            if want_cnn and tv:
                self.model.img_encoder.load_torchvision_state_dict(torch.load(tv, map_location="cpu"))
        if want_cnn and not self.model.img_encoder.weights_loaded:
            import warnings
            warnings.warn("pixel input with a ResNet-50 region encoder that holds RANDOM weights: the reference uses torchvision's "
                          "resnet50(pretrained=True), frozen.  Pass args.resnet50_weights=<torchvision state dict> or load a "
                          "checkpoint that carries enc.img_encoder.*; the frozen random CNN only yields noise features.",
                          RuntimeWarning, stacklevel=2)
        self.train_data, self.test_data = train_dataloader, test_dataloader
        self.distributed = torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size() > 1
        # one process per GPU: the host threads of this rank (launches, the mask-check workers, pinned-memory copies) on the CPUs of its
        # GPU's NUMA node (best effort, sysfs only; a launcher that binds before the first GPU call, as bench.py does, is better still)
        self.numa = None
        if self.distributed:
            from .dist import bind_to_gpu_numa
            self.numa = bind_to_gpu_numa(int(os.environ.get("LOCAL_RANK", torch.cuda.current_device())))
        self.mlm_task = _str2bool(getattr(args, "mlm_task", True))
        self.itm_task = _str2bool(getattr(args, "itm_task", True))
        self.step = TrainStep(self.model, lr=getattr(args, "lr", 1e-5), distributed=self.distributed, mlm_task=self.mlm_task,
                              overlap_optimizer=True,
                              itm_task=self.itm_task)
        self._init_mask_policy(args)
        self._pinned = {}
        self.log_freq = getattr(args, "log_freq", 10)
        self.logger = logger            # optional callable(dict, step=epoch): stands in for wandb.log
        print("Total Parameters:", sum(p.nelement() for p in self.model.parameters()))

    def _init_mask_policy(self, args):
        """Host-side state of the mask recognition / verification (no device needed: tests/test_host_logic.py drives it on the CPU)."""
        self.recognise_masks = True     # derive {family, n2, vl} descriptors from the Dataset's materialised masks (verified)
        self.n_recognised = 0           # batches whose masks were recognised as one of the closed-form families
        self.n_rejected = 0             # ... of which the every-entry check then found a deviating entry (ran on the matrix instead)
        # How the derived descriptors are checked against the shipped matrices (see _recognise_masks / _ticket):
        #   "full" (default)  EVERY entry of EVERY batch, on the host (mv_mask_verify_host: one pass over the 134 MB at memory speed on
        #                     `verify_threads` worker threads), one batch AHEAD of the step, so the result is known before the step is
        #                     enqueued: nothing crosses PCIe, nothing is redone, no rank has to agree with another
        #   "sampled"         every entry of the first `verify_first` batches and of every `verify_every`-th one, `verify_probes` random
        #                     rows per sample otherwise (explicit opt-in: a foreign mask that matches a family on the probes only would
        #                     train with the closed form on the unchecked batches)
        #   "off"             the probe lines of the recognition only
        self.verify_masks = getattr(args, "verify_masks", "full")
        if self.verify_masks not in ("full", "sampled", "off"):
            raise ValueError("args.verify_masks must be 'full', 'sampled' or 'off'")
        self.verify_first, self.verify_every, self.verify_probes = 2, 64, 4
        self.verify_threads = int(getattr(args, "verify_threads", 4))
        self._mask_batches = {True: 0, False: 0}     # train / eval batches seen (the sampled policy counts them separately)
        self._pool = None

    def _full_check_now(self, train):
        """Counts the batch (training and evaluation batches separately) and says whether it gets the every-entry check."""
        self._mask_batches[train] += 1
        n = self._mask_batches[train]
        return self.verify_masks == "full" or (self.verify_masks == "sampled" and (n <= self.verify_first or n % self.verify_every == 0))

    def _recognise_masks(self, attn_masks, input_ids, N, txt_labels=None, probe_seed=None):
        """The reference Dataset ships a materialised int64 mask per sample (dataset_origin.py:138-176: 134 MB per batch at
        B=64, L=512).  Its five families are closed forms of {family, n2, vl}: derive the descriptors from a few probe
        entries on the host (two rows and a column per sample, all samples at once) so that the step can run on them (packed
        rows, mask bits built on the device).  probe_seed (the "sampled" policy's cheap batches): `verify_probes` random rows per
        sample are compared with the closed form as well.  A mask outside the families returns None.
        The descriptors are a HYPOTHESIS until `_ticket`'s every-entry check has confirmed them.
        Returns MaskDesc or None."""
        m = attn_masks
        if not torch.is_tensor(m) or m.dtype != torch.int64 or m.dim() not in (2, 3) or m.is_cuda:
            return None
        B, L = m.shape[0], m.shape[-1]
        S = L - N - 3
        n2 = N + 2
        if S < 1 or input_ids.shape[1] != S + 1:
            return None
        # numpy views of the host tensors: the probes touch a few KB of the 134 MB matrix, and torch's CPU operators would hand each of
        # these tiny jobs to its OpenMP pool (measured on the 16-core share of a GPU box: 3 ms .. 90 ms per batch, and the kernel
        # launches of the step that follows slowed down 3-5x by the spinning workers)
        import numpy as np
        mn = m.numpy()
        ids = input_ids.cpu().numpy()
        # valid length from the LAST non-zero id (the text [SEP] is never 0): random_word (dataset_origin.py:183-209) may put
        # id 0 at a labelled in-text position, and a tokenizer may use id 0 for a real token
        nz = ids != 0
        T = ids.shape[1]
        last = np.where(nz.any(1), T - 1 - np.argmax(nz[:, ::-1], axis=1), -1).astype(np.int64)
        n_ids = last + 1
        vl = n2 + n_ids
        j = np.arange(L).reshape(1, L)
        if txt_labels is not None and torch.is_tensor(txt_labels):
            lab = txt_labels.cpu().numpy() != -100
            if lab.shape == (B, L) and bool((lab & (j >= vl.reshape(B, 1))).any()):
                return None          # a label after the derived valid length: the packed rows would drop it
        full_row = (j < vl.reshape(B, 1)).astype(np.int64)
        if m.dim() == 2:
            if not np.array_equal(mn, full_row):
                return None
            fam_id = np.full((B,), D.FAMILY_ID["1d"], dtype=np.int32)
        else:
            r0, rl, cl = mn[:, 0, :], mn[:, L - 1, :], mn[:, :, L - 1]
            img_row = np.broadcast_to((j < n2).astype(np.int64), (B, L))
            txt_row = np.broadcast_to((j >= n2).astype(np.int64), (B, L))
            bar_col = np.broadcast_to(((j < n2) | (j == L - 1)).astype(np.int64), (B, L))
            eq = lambda a_, b_: (a_ == b_).all(1)
            one = lambda a_: (a_ == 1).all(1)
            # (the last column tells a full-length BAR sample -- whose probe ROWS are all ones too -- from a full one)
            is_full = eq(r0, full_row) & eq(rl, full_row) & eq(cl, np.broadcast_to(full_row[:, L - 1:L], (B, L)))
            is_s2s = eq(r0, img_row) & one(rl)
            is_bar = one(r0) & one(rl) & eq(cl, bar_col)
            is_non = eq(r0, img_row) & eq(rl, txt_row)
            fam_id = np.full((B,), -1, dtype=np.int32)
            # same precedence as a per-sample if / elif chain over (full, s2s, bar, noncross)
            for cond, name in ((is_non, "noncross"), (is_bar, "bar"), (is_s2s, "s2s"), (is_full, "full")):
                fam_id = np.where(cond, np.int32(D.FAMILY_ID[name]), fam_id).astype(np.int32)
            if bool((fam_id < 0).any()):
                return None
            if probe_seed is not None and self.verify_probes > 0:
                # host-side spot check of `verify_probes` random rows per sample against the closed forms
                rng = np.random.default_rng(probe_seed)
                rows = rng.integers(0, L, size=(B, self.verify_probes))
                got = mn[np.arange(B).reshape(B, 1), rows]                      # [B, P, L]
                i_ = rows.reshape(B, -1, 1)
                jj = np.arange(L).reshape(1, 1, L)
                f = fam_id.reshape(B, 1, 1)
                want = np.where(f == 1, (jj < n2) | ((i_ >= n2) & (jj >= n2) & (jj <= i_)),
                       np.where(f == 2, (i_ < n2) | (jj < n2) | (jj <= i_),
                       np.where(f == 3, (i_ < n2) == (jj < n2), jj < vl.reshape(B, 1, 1))))
                if not np.array_equal(got != 0, want):
                    return None
        dn = np.empty((B, 3), dtype=np.int32)
        dn[:, 0], dn[:, 1], dn[:, 2] = fam_id, n2, vl.astype(np.int32)
        d = torch.from_numpy(dn)
        return D.MaskDesc(d, L, host=d)          # uploaded with the batch's other integer fields (_upload_small)

    def _ticket(self, data, train):
        """Host-side look at a batch BEFORE its step: derive the mask descriptors and start the every-entry check of the shipped
        matrix against them on a worker thread (one GIL-free C call; `_prefetch` issues this one batch ahead, so it runs under
        the launches of the previous step).  -> {"data", "desc", "check"}: `check` is a future of the first mismatching entry
        (-1: none) or None when this batch is not checked entry by entry."""
        t = {"data": data, "desc": None, "check": None}
        attn_masks = data[3]
        if not self.recognise_masks or isinstance(attn_masks, D.MaskDesc):
            return t
        full = self._full_check_now(train)
        img = data[4]
        N = self.args_num_regions(img)
        probe = None if (full or self.verify_masks == "off") else self._mask_batches[train]
        desc = self._recognise_masks(attn_masks, data[1], N, data[2], probe_seed=probe) if N is not None else None
        if desc is None:
            return t
        t["desc"] = desc
        if full:
            from . import hip_ops as ops
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="medvill-maskcheck")
            t["check"] = self._pool.submit(ops.mask_verify_host, attn_masks, desc.host_desc(), self.verify_threads)
        return t

    def args_num_regions(self, img):
        """Number of image regions N of a batch without running the region encoder: region features [B,N,D] carry it; for pixel
        batches it is what ImageEncoder_cnn will sample (args.num_image_embeds, models/image.py:63-68)."""
        if isinstance(img, (tuple, list)) and len(img) == 2 and torch.is_tensor(img[0]):
            return int(img[0].shape[1])
        n = getattr(self.args, "num_image_embeds", None)
        return int(n) if n else None

    def _prefetch(self, loader, train):
        """The loader's batches with their tickets, the ticket of batch k+1 issued before batch k is handed out."""
        it = iter(loader)
        nxt = next(it, None)
        tk = self._ticket(nxt, train) if nxt is not None else None
        while tk is not None:
            cur = tk
            nxt = next(it, None)
            tk = self._ticket(nxt, train) if nxt is not None else None
            yield cur

    def _stage(self, key, n, dtype):
        """One of two reused pinned staging buffers of >= n elements (the copy that last used it has finished)."""
        slot = self._pinned.get(key)
        if slot is None or slot[0][0].numel() < n:
            cap = n + n // 4
            slot = self._pinned[key] = [[torch.empty(cap, dtype=dtype).pin_memory() for _ in range(2)], 0, [None, None]]
        bufs, i, evs = slot
        if evs[i] is not None:
            evs[i].synchronize()
        slot[1] = i ^ 1
        return bufs[i][:n], slot, i

    def _upload(self, t):
        """Host tensor -> device through a reused pinned staging buffer (asynchronous copy on the current stream)."""
        import numpy as np
        stage, slot, i = self._stage((t.dtype, "one"), t.numel(), t.dtype)
        np.copyto(stage.numpy(), t.contiguous().view(-1).numpy())          # plain memcpy, no OpenMP pool
        d = stage.view(t.shape).to(self.device, non_blocking=True)
        slot[2][i] = torch.cuda.Event()
        slot[2][i].record(torch.cuda.current_stream())
        return d

    def _to_batch(self, ticket):
        """-> (batch dict, use_desc).  Waits for the ticket's mask check: a single deviating entry sends the step to the matrix."""
        data = ticket["data"]
        cls_tok, input_ids, txt_labels, attn_masks, img, segment, is_aligned, sep_tok = data[:8]
        if torch.is_tensor(img):     # pixels [B,3,H,W] (dataset_origin.py:85-89): region features from the mirrored CNN
            if self.model.img_encoder is None:
                raise TypeError("the loader yields pixels: construct the trainer with args.pixels=True (ResNet-50 region encoder)")
            with torch.no_grad():
                img = self.model.img_encoder(img.to(self.device))
        feats, pos = img            # (region feats [B,N,2048], region positions [B,N])
        if torch.is_tensor(feats) and not feats.is_cuda:        # 19 MB per batch at B = 64: pinned staging, asynchronous copy
            feats = self._upload(feats)
        batch = dict(cls_tok=cls_tok, input_txt=input_ids, attn_mask=attn_masks, segment=segment, img_feats=feats, img_pos=pos,
                     sep_tok=sep_tok, txt_labels=txt_labels, is_aligned=is_aligned)
        if isinstance(attn_masks, D.MaskDesc):             # a loader that already ships descriptors
            batch["attn_desc"], batch["attn_mask"] = attn_masks, None
        elif ticket["desc"] is not None and ticket["desc"].L == feats.shape[1] + input_ids.shape[1] + 2:
            self.n_recognised += 1
            bad = ticket["check"].result() if ticket["check"] is not None else -1
            if bad < 0:
                batch["attn_desc"] = ticket["desc"]
            else:
                self.n_rejected += 1            # the matrix is not the closed form the probes suggested: it is the mask that counts
        self._upload_small(batch)
        return batch, ("attn_desc" in batch)

    def _upload_small(self, batch):
        """Every small integer field of a HOST batch (token ids, labels, segment, positions, the labelled-row index, mask
        descriptors) goes to the device in ONE asynchronous copy from a pinned staging buffer.  A `.to(device)` of a pageable
        tensor is ordered behind the stream's pending kernels and blocks the host until it is done -- a dozen of them per step
        serialised the host with the GPU (47-75 ms per step measured against 26 ms for resident batches)."""
        names = [k for k in ("cls_tok", "input_txt", "txt_labels", "segment", "is_aligned", "sep_tok", "img_pos")
                 if torch.is_tensor(batch.get(k)) and not batch[k].is_cuda and batch[k].dtype == torch.int64]
        # whichever of them are on the host are staged together (with pixel input the region positions come from the CNN, on the device)
        if not names:
            return
        import numpy as np
        parts = {k: batch[k].contiguous().numpy() for k in names}
        if "txt_labels" in parts:
            flat_lab = parts["txt_labels"].reshape(-1)
            rows = np.flatnonzero(flat_lab != -100)                     # the labelled-row index: R is needed on the host anyway
            parts["label_rows"], parts["label_ids"] = rows.astype(np.int64), flat_lab[rows]
        desc = batch.get("attn_desc")
        if desc is not None and desc._host is not None and not desc.desc.is_cuda:
            parts["_desc"] = desc._host.numpy().astype(np.int64)
        n = sum(v.size for v in parts.values())
        stage, slot, i = self._stage((torch.int64, "ints"), n, torch.int64)
        sn, off = stage.numpy(), 0
        for v in parts.values():
            sn[off:off + v.size] = v.reshape(-1)
            off += v.size
        dflat = stage.to(self.device, non_blocking=True)
        slot[2][i] = torch.cuda.Event()
        slot[2][i].record(torch.cuda.current_stream())
        off = 0
        for k, v in parts.items():
            t = dflat[off:off + v.size].view(v.shape)
            off += v.size
            if k in ("label_rows", "label_ids"):
                batch[k] = t.to(torch.int32)
            elif k == "_desc":
                batch["attn_desc"] = D.MaskDesc(t.to(torch.int32), desc.L, host=desc._host)
            else:
                batch[k] = t

    def _run_epoch(self, loader, epoch, train):
        # The reference reads loss.item() every step (train_origin.py:129-146): a host sync per step.  Here the six step counters
        # stay on the device and are read in blocks of `log_freq` steps, so the host keeps enqueueing (copies, launches) while the
        # GPU works; the per-step values the epoch averages need are all there afterwards.
        pending, rows = [], []

        def flush():
            if pending:
                rows.extend(torch.stack(pending).double().cpu().unbind(0))
                pending.clear()
        for ticket in self._prefetch(loader, train):
            batch, use_desc = self._to_batch(ticket)
            pending.append(self.step(batch, train=train, use_desc=use_desc))
            if len(pending) >= max(1, int(self.log_freq)):
                flush()
        flush()
        tot = torch.zeros(6, dtype=torch.float64)
        losses, mlm_l, itm_l = [], [], []
        for stats in rows:
            tot += stats[:6]
            ml = float(stats[0] / max(stats[1], 1.0))
            il = float(stats[3] / max(stats[4], 1.0))
            mlm_l.append(ml)
            itm_l.append(il)
            losses.append((ml if self.mlm_task else 0.0) + (il if self.itm_task else 0.0))
        if train:
            self.step.sync()        # parameters may be read on the current stream right after train() (overlapped AdamW)
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
        if not self.distributed or torch.distributed.get_rank() == 0:     # replicas are identical: one writer
            if not os.path.exists(save_path_per_ep):
                os.makedirs(save_path_per_ep, exist_ok=True)
                os.chmod(save_path_per_ep, 0o777)
            self.model.save_pretrained(save_path_per_ep)
            print(f"EP: {epoch} Model saved on {save_path_per_ep}")
            os.chmod(save_path_per_ep + "/pytorch_model.bin", 0o777)
        if self.distributed:
            torch.distributed.barrier()


SCRATCH_MODELS = ("bert-base-scratch", "bert-small-scratch")      # the two --bert_model values that mean "random init" (cxrbert_origin.py:49-54)


def resolve_bert_model(args, name):
    """train_origin.py:36-49 + cxrbert_origin.py:41-55.  -> (config dict, HF-BERT state dict or None).
    `bert-*-scratch` build a randomly initialised BertModel in the reference, and so here.  EVERY other name makes the reference load
    PRETRAINED text-encoder weights (`BertModel/AutoModel.from_pretrained(args.bert_model)`, a download); there is no network here, so
    the weights must be handed over -- `args.init_state_dict` (a HF BertModel / BertForPreTraining state dict) or `args.init_checkpoint`
    (a `pytorch_model.bin` or a directory holding one; `args.bert_model` itself may be such a directory, with its config.json).
    Without them this RAISES: a run the reference would start from pretrained weights must not silently start from noise.
    `args.allow_random_init = True` is the explicit opt-out (geometry only)."""
    import json
    if name == "albert-base-v2":
        raise NotImplementedError("--bert_model albert-base-v2: ALBERT (cross-layer parameter sharing, factorised embeddings) is not mirrored")
    if name == "load_pretrained_model":
        raise ValueError("--bert_model load_pretrained_model goes with --weight_load and --pre_trained_model_path (train_origin.py:28-34)")
    config = BERT_CONFIGS.get(name)
    if name in SCRATCH_MODELS:
        return config, None
    sd = getattr(args, "init_state_dict", None)
    path = getattr(args, "init_checkpoint", None) or (name if os.path.isdir(str(name)) else None)
    if sd is None and path is not None:
        f = os.path.join(path, "pytorch_model.bin") if os.path.isdir(path) else path
        sd = torch.load(f, map_location="cpu")
        cj = os.path.join(os.path.dirname(f), "config.json")
        if config is None and os.path.exists(cj):
            with open(cj) as fh:
                config = json.load(fh)
    if config is None:
        raise ValueError(f"--bert_model {name!r}: unknown geometry (no offline config for it and no config.json beside the checkpoint)")
    if sd is None:
        if getattr(args, "allow_random_init", False):
            import warnings
            warnings.warn(f"--bert_model {name!r}: args.allow_random_init is set -- training starts from RANDOM weights where the reference "
                          "would load the pretrained text encoder", RuntimeWarning, stacklevel=3)
            return config, None
        raise RuntimeError(
            f"--bert_model {name!r} means PRETRAINED text-encoder weights in the reference (cxrbert_origin.py:43-48,55: "
            "BertModel/AutoModel.from_pretrained) and they cannot be downloaded here. Pass args.init_state_dict (a HF BertModel state "
            "dict) or args.init_checkpoint (a pytorch_model.bin / a directory with one), use --bert_model bert-base-scratch for a "
            "from-scratch run, or set args.allow_random_init = True to take the geometry only.")
    return config, sd


def load_pretrained_bert(model, hf_state_dict):
    """Copies a HF BERT checkpoint into the text embeddings / encoder / pooler (the modules cxrbert_origin.py:56-57,72-73 takes from it).
    The tied MLM decoder and the image embeddings' shared position / type / LayerNorm tensors follow (same storage); the heads' own
    parameters and the image projection keep their initialisation, as in the reference.  Strict on the encoder side."""
    from .checkpoint import from_hf_bert_keys
    mapped = from_hf_bert_keys(hf_state_dict)
    want = [n for n in model._param_names if n.startswith(("enc.txt_embeddings.", "enc.encoder.", "enc.pooler."))]
    missing = [n for n in want if n not in mapped]
    if missing:
        raise RuntimeError(f"pretrained BERT checkpoint lacks {len(missing)} encoder tensors, e.g. {missing[:3]}")
    eng = model.engine
    bad = [n for n in want if tuple(mapped[n].shape) != tuple(eng.p[n].shape)]
    if bad:
        raise RuntimeError(f"pretrained BERT checkpoint does not match the model geometry, e.g. {bad[0]}: "
                           f"{tuple(mapped[bad[0]].shape)} vs {tuple(eng.p[bad[0]].shape)}")
    eng.wait_optimizer()
    with torch.no_grad():
        for n in want:
            eng.p[n].copy_(mapped[n].to(eng.device, torch.float32))
    eng.shadow_dirty = True
    return len(want)


BERT_CONFIGS = {   # offline stand-ins for BertConfig.from_pretrained(...) at train_origin.py:36-47
    "bert-base-scratch": dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                              intermediate_size=3072, max_position_embeddings=512),
    "bert-base-uncased": dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                              intermediate_size=3072, max_position_embeddings=512),
    "bert-small-scratch": dict(vocab_size=30522, hidden_size=512, num_hidden_layers=4, num_attention_heads=8,
                               intermediate_size=2048, max_position_embeddings=512),
    # the other --bert_model choices of main_origin.py:111-120 (their published config.json values)
    "google/bert_uncased_L-4_H-512_A-8": dict(vocab_size=30522, hidden_size=512, num_hidden_layers=4, num_attention_heads=8,
                                              intermediate_size=2048, max_position_embeddings=512),
    "google/bert_uncased_L-2_H-128_A-2": dict(vocab_size=30522, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                                              intermediate_size=512, max_position_embeddings=512),
    "emilyalsentzer/Bio_ClinicalBERT": dict(vocab_size=28996, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                                            intermediate_size=3072, max_position_embeddings=512),
    "bionlp/bluebert_pubmed_mimic_uncased_L-12_H-768_A-12": dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12,
                                                                 num_attention_heads=12, intermediate_size=3072,
                                                                 max_position_embeddings=512),
}
