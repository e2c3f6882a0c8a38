"""Drop-in mirror of the reference's model interface (models/cxrbert_origin.py).

    CXRBERT(config, args).forward(cls_tok, input_txt, attn_mask, segment, input_img, sep_tok)
        -> (mlm_logits [B,L,V], itm_logits [B,2])                 cxrbert_origin.py:132-149
    .enc(...) -> (last_hidden [B,L,H], pooled [B,H], None)         cxrbert_origin.py:130
    .mlm / .itm sub-modules, .state_dict() with the reference key names (incl. aliases),
    .save_pretrained(dir) / CXRBERT.from_pretrained(dir, state_dict=, config=, args=)
                                                                   train_origin.py:31-32,254-266

Same names, argument meaning and error behaviour as the reference; the arithmetic runs on the
HIP engine (engine.py) -- there is no eager / CPU path.  ``input_img`` is either the tuple
``(region_feats [B,N,2048], region_pos [B,N])`` (what the reference's ``img_encoder`` returns,
models/image.py:54-69 -- the CNN itself is out of scope) or a tensor that a
user-supplied ``img_encoder`` callable maps to that tuple.
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict
from dataclasses import replace
import weakref
from types import SimpleNamespace

import torch
import torch.nn as nn

from .engine import ALIASES, Engine, ModelConfig


def model_config_from(config, args=None) -> ModelConfig:
    """Accepts a ModelConfig, a HF-BertConfig-like object or a dict (config.json contents).  `args` (the reference's argparse
    namespace) contributes the two fields ImageBertEmbeddings reads from it (cxrbert_origin.py:19,27-31):
      args.img_postion (sic)  falsy -> the image rows get no position embedding.  Python truthiness, as in the reference
                              (`if self.args.img_postion:`; main_origin.py:129 declares the flag without a type, so the
                              command-line string "False" is truthy there as well);
      args.dropout_prob       probability of the image embeddings' own nn.Dropout (text rows use hidden_dropout_prob)."""
    if isinstance(config, ModelConfig):
        cfg = replace(config)
    else:
        get = (lambda k, d=None: config.get(k, d)) if isinstance(config, dict) else (lambda k, d=None: getattr(config, k, d))
        cfg = ModelConfig(vocab_size=get("vocab_size", 30522), hidden=get("hidden_size", 768), layers=get("num_hidden_layers", 12),
                          heads=get("num_attention_heads", 12), intermediate=get("intermediate_size", 3072),
                          max_pos=get("max_position_embeddings", 512), type_vocab=get("type_vocab_size", 2),
                          # nn.Linear(args.img_hidden_sz, args.embedding_size), cxrbert_origin.py:16 (a config entry wins)
                          img_hidden=get("img_hidden_sz", None) or int(getattr(args, "img_hidden_sz", None) or 2048), ln_eps=get("layer_norm_eps", 1e-12),
                          dropout=get("hidden_dropout_prob", 0.1))
    if args is not None:
        if hasattr(args, "img_postion"):
            cfg.img_position = bool(args.img_postion)
        if getattr(args, "dropout_prob", None) is not None:
            cfg.img_dropout = float(args.dropout_prob)
    return cfg


class _CXRBertFn(torch.autograd.Function):
    """Whole-network autograd node: forward = the engine's kernel schedule, backward = the
    engine's explicit backward schedule.  Gradients come back as (clones of) views of the flat
    gradient buffer, one per Parameter, so `loss.backward(); optimizer.step()` of the reference
    trainer (train_origin.py:129-131) works unchanged."""

    @staticmethod
    def forward(ctx, model, want_heads, cls_tok, input_txt, attn_mask, segment, feats, pos, sep_tok, *params):
        eng = model.engine
        # parameters may have been stepped by an external optimizer: the 16-bit copies are refreshed on every forward -- unless _run has just
        # done so (ahead of the mask recognition's read-back) or medvill_amd.optim.AdamW, whose kernel writes them, was the last to touch them
        if not model.__dict__.pop("_shadow_fresh", False):
            eng.shadow_dirty = eng.shadow_dirty or model._params_dirty()
        eng.training = model.training           # dropout (p = 0.1 at every site of the reference) only in train mode
        eng.keep_acts = bool(model._want_grad)  # under torch.no_grad() nothing is saved for a backward
        from .data import MaskDesc
        # lazy MLM logits (want_heads == 3): nothing downstream needs a row per position, so descriptor masks whose padding is invisible
        # run on the valid rows only, like the fused training step
        pack = want_heads == 3 and isinstance(attn_mask, MaskDesc) and eng.is16 and attn_mask.packable()
        # forward(..., txt_labels=...) under lazy logits: the labelled rows are known, so the last layer's per-row work runs on the rows the
        # heads consume only (and, in the full / 1-D families, its attention on those queries only), like the fused training step
        fwd_labels = model.__dict__.pop("_fwd_labels", None) if want_heads == 3 else None
        model._lazy_rows = fwd_labels
        hidden, pooled = eng.encoder_forward(cls_tok, input_txt, attn_mask, segment, feats, pos, sep_tok, pack=pack,
                                             tail_rows=None if fwd_labels is None else fwd_labels[0])
        ctx.model, ctx.want_heads = model, want_heads
        if want_heads == 2:                     # ITM head only (retrieval: `self.itm(cls)` on the pooled output)
            return eng._itm_forward().clone()
        if want_heads == 3:                     # the loss decides which rows of the MLM head are ever computed: see LazyLogits
            model._lazy = None
            tok = torch.zeros(1, dtype=torch.float32, device=eng.device)
            itm = eng._itm_forward().clone()
            ctx.mark_non_differentiable(itm)    # ITM logits for metrics; the ITM loss goes through losses.mlm_itm_loss with the MLM loss
            return tok, itm
        if want_heads:
            mlm, itm = eng.heads_full()
            return mlm, itm
        return hidden.clone(), pooled.clone()

    @staticmethod
    def _backward_once(ctx, g0, g1):
        model = ctx.model
        eng = model.engine
        if ctx.want_heads == 3:
            if model._lazy is None:
                raise RuntimeError("lazy logits: backward() reached the model without a loss from medvill_amd.losses.mlm_itm_loss(mlm, itm, ...)")
            rows, ids, aligned, mlm_on, itm_on, pre = model._lazy
            model._lazy = (rows, ids, aligned, mlm_on, itm_on, False)      # (a second pass -- loss-scale retry, retain_graph -- redoes the head)
            gv = float(g0) if g0 is not None else 0.0          # d(total) / d(loss), normally 1.0 (one host read on the drop-in path)
            if not (pre and gv == 1.0):
                eng.zero_grad()
                eng.heads_train(rows, ids, aligned, mlm_scale=(gv / max(int(rows.numel()), 1)) if mlm_on else 0.0,
                                itm_scale=(gv / int(aligned.numel())) if itm_on else 0.0, compute_grad=True)
            eng.encoder_backward()
            return
        eng.zero_grad()
        if ctx.want_heads == 2:
            eng.heads_full_backward(None, g0)
        elif ctx.want_heads:
            eng.heads_full_backward(g0, g1)
        else:
            S, H = eng.S, eng.cfg.hidden
            ls = eng.loss_scale_dev          # f16 gradients: the incoming f32 gradients enter the chain multiplied by S
            sc = (lambda t: t) if ls is None else (lambda t: t.float() * ls)
            dh = S["dhidden"] = eng._buf("dhidden", (S["M"], H), eng.adt)
            dh.copy_(sc(g0.reshape(S["M"], H))) if g0 is not None else dh.zero_()
            if g1 is not None:
                # pooled = tanh(hidden[:,0].Wp^T + bp)
                from . import hip_ops as ops
                B, Lq = S["B"], S["L"]
                dpre = eng._buf("dpoolpre", (B, H), eng.adt)
                ops.dact(1, sc(g1).to(eng.adt).contiguous(), S["pooled"], dpre, B * H)
                ops.colsum(dpre, H, B, H, eng.g["enc.pooler.dense.bias"], accumulate=True, unscale=eng.unscale_dev)
                eng._dW(dpre, S["hidden"], eng.g["enc.pooler.dense.weight"], H, H, B, lda=H, ldb=Lq * H)
                dh0 = eng._buf("dh0", (B, H), eng.adt)
                ops.gemm(dpre, eng.w["enc.pooler.dense.weight"], dh0, tb=True, M=B, N=H, K=H)
                rows0 = (torch.arange(B, device=eng.device, dtype=torch.int32) * Lq)
                ops.scatter_rows(dh0, H, rows0, B, H, dh, H, accumulate=True)
        eng.encoder_backward()

    @staticmethod
    def backward(ctx, g0, g1=None):
        model = ctx.model
        eng = model.engine
        # gradients a previous backward left in the flat buffer THROUGH the .grad views (no zero_grad in between, or
        # zero_grad(set_to_none=False)): keep them, this backward adds to them like autograd would
        if ctx.want_heads == 3 and model._lazy is not None and model._lazy[5]:
            held = model.__dict__.pop("_held", None)             # set aside by the loss, which has already zeroed the buffer
        else:
            held = eng.flat_g.clone() if (_use_views(model) and eng.flat_g is not None and _holds_views(model)) else None
        _CXRBertFn._backward_once(ctx, g0, g1)
        if eng.scaler is not None:
            # f16 gradient operands under a loss scale: this path hands gradients to torch (an external optimizer), so an
            # overflow cannot be turned into a skipped step -- the backward is redone with a smaller scale instead (the saved
            # activations are still there).  One host sync per try; this is the drop-in path, not the fused training step.
            for _ in range(8):
                eng.scaler[6:7].zero_()
                from . import hip_ops as ops
                ops.count_nonfinite(eng.flat_g, eng.scaler[6:7])
                if float(eng.scaler[6]) == 0.0:
                    break
                s_new = max(float(eng.scaler[0]) / 16.0, 1.0)
                eng.reset_scaler(s_new)
                _CXRBertFn._backward_once(ctx, g0, g1)
        if held is not None:
            eng.flat_g.add_(held)
        return (None,) * 9 + _hand_over_grads(model)


def _use_views(model):
    gv = model.grad_views
    if gv is not None:
        return bool(gv)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return False
    return not any(p._backward_hooks for p in model._plist)


def _holds_views(model):
    eg = model.engine.g
    for n, p in zip(model._param_names, model._plist):
        g = p.grad
        if g is not None and g.data_ptr() == eg[n].data_ptr():
            return True
    return False


def _own_grads(model):
    """Every .grad that is a view of the flat gradient buffer becomes a tensor of its own."""
    eg = model.engine.g
    for n, p in zip(model._param_names, model._plist):
        if p.grad is not None and p.grad.data_ptr() == eg[n].data_ptr():
            p.grad = p.grad.clone()


def _hand_over_grads(model):
    """The engine's gradients -> torch, for `loss.backward(); optimizer.step()` (train_origin.py:129-131).
    model.grad_views (default): every Parameter whose .grad is None -- the state `optimizer.zero_grad()` leaves -- gets the matching VIEW of
    the flat gradient buffer as its .grad and autograd receives None for it: no 440 MB copy + 200 accumulations per step (4-5 ms of host-bound
    launches at BERT-base).  Like DistributedDataParallel(gradient_as_bucket_view=True) the views are rewritten by the next backward, so a
    caller that keeps a .grad across steps must clone it.  A .grad that still IS the view (accumulation over several backward calls,
    zero_grad(set_to_none=False)) already holds old + new (see `held` in backward); any other tensor found in .grad takes the copying path
    (autograd adds a clone to it)."""
    eng = model.engine
    if not _use_views(model):
        return tuple(eng.g[n].clone() for n in model._param_names)
    out = []
    for n, p in zip(model._param_names, model._plist):
        g = eng.g[n]
        if p.grad is None or p.grad.data_ptr() == g.data_ptr():
            p.grad = g
            out.append(None)
        else:
            out.append(g.clone())
    return tuple(out)


class LazyLogits:
    """What CXRBERT.forward returns in place of the [B, L, V] MLM logits under `model.lazy_logits = True`.

    The reference's step (train_origin.py:106-131) materialises 64 x 512 x 30,522 logits and hands them to CrossEntropyLoss(ignore_index=-100),
    which reads the ~10 % labelled rows.  With lazy logits the encoder runs in forward(), and the MLM head runs inside
    `medvill_amd.losses.mlm_itm_loss(mlm, itm, txt_labels, is_aligned)` -- the one-line replacement for the two CrossEntropyLoss calls -- on
    the LABELLED rows only, fused with the loss (the fused training step's head).  `loss.backward(); optimizer.step()` work as before.
    `.stats` (after the loss): f32[6] = [mlm_nll_sum, n_labelled, mlm_correct, itm_nll_sum, B, itm_correct] (train_origin.py:133-146's
    counters); `.materialize()` computes the full logits after all (inspection)."""

    def __init__(self, model, tok, shape):
        self.model, self.tok, self.shape, self.stats = model, tok, tuple(shape), None

    def loss(self, txt_labels, is_aligned, mlm_task=True, itm_task=True):
        from .data import label_index
        eng = self.model.engine
        given = self.model._lazy_rows
        if given is not None:
            # the forward was given the labels (forward(..., txt_labels=...)) and ran its last layer on exactly their rows
            rows, ids, src = given
            if txt_labels is not src and not torch.equal(txt_labels.to(eng.device), src.to(eng.device)):
                raise ValueError("mlm_itm_loss: txt_labels differ from the labels the lazy forward was given")
            aligned = is_aligned.to(eng.device, torch.int32)
            return _LazyLossFn.apply(self.tok, self, rows, ids, aligned, bool(mlm_task), bool(itm_task))
        rows, ids = label_index(txt_labels.to(eng.device))
        if eng.S.get("inv") is not None and rows.numel() > 0:
            # the forward ran on the valid rows only (mask descriptors): a label past a sample's text [SEP] has no row.  The reference
            # Dataset never produces one (dataset_origin.py:105-135 pads the labels with -100); the device is idle here anyway
            # (label_index has just read the row count back)
            if bool((eng.S["inv"].index_select(0, rows.to(torch.int64)) < 0).any()):
                raise ValueError("txt_labels holds a label at a padded position (after the text [SEP]); the lazy forward ran on the "
                                 "valid rows only.  Set model.recognise_masks = False (and pass the mask matrix) to run every row")
        aligned = is_aligned.to(eng.device, torch.int32)
        return _LazyLossFn.apply(self.tok, self, rows, ids, aligned, bool(mlm_task), bool(itm_task))

    @torch.no_grad()
    def materialize(self):
        eng = self.model.engine
        if eng.S.get("cu") is not None or eng.S.get("sel") is not None:
            raise RuntimeError("the forward ran on packed rows (mask descriptors) or on the labelled rows only (txt_labels given): "
                               "there is no logit row per position to materialise")
        return eng.heads_full()[0]


class _LazyLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, lazy, rows, ids, aligned, mlm_on, itm_on):
        model = lazy.model
        eng = model.engine
        R, B = int(rows.numel()), int(aligned.numel())
        # A backward is coming (the handle requires grad): the head runs ONCE, with its gradient for d loss = 1 -- the flat gradient buffer
        # is zeroed here instead of in the backward (gradients held through .grad views are set aside first, see _CXRBertFn.backward).  The
        # backward reuses that when it is indeed handed 1.0 and redoes the head otherwise (an upstream factor, a loss-scale retry).
        pre = bool(ctx.needs_input_grad[0]) and model.grad_in_loss
        if pre:
            model._held = eng.flat_g.clone() if (_use_views(model) and eng.flat_g is not None and _holds_views(model)) else None
            eng.zero_grad()
        stats = eng.heads_train(rows, ids, aligned, mlm_scale=(1.0 / max(R, 1)) if (mlm_on or not pre) else 0.0,
                                itm_scale=(1.0 / B) if (itm_on or not pre) else 0.0, compute_grad=pre)
        lazy.stats = stats
        model._lazy = (rows, ids, aligned, mlm_on, itm_on, pre)
        loss = tok.new_zeros(())
        if mlm_on:
            loss = loss + stats[0] / torch.clamp(stats[1], min=1.0)      # mean over the labelled positions (ignore_index=-100)
        if itm_on:
            loss = loss + stats[3] / stats[4]
        return loss

    @staticmethod
    def backward(ctx, g):
        return g.reshape(1), None, None, None, None, None, None


_MLM_PARAMS = ("mlm.predictions.transform.dense.weight", "mlm.predictions.transform.dense.bias",
               "mlm.predictions.transform.LayerNorm.weight", "mlm.predictions.transform.LayerNorm.bias", "mlm.predictions.bias",
               "enc.txt_embeddings.word_embeddings.weight")
_ITM_PARAMS = ("itm.linear.weight", "itm.linear.bias")


def _scaled_retry(eng, run):
    """f16 gradient operands travel under the engine's loss scale S.  A head called on its own hands its gradients to torch, so an
    overflow cannot become a skipped optimizer step: `run()` (-> tensors to test) is redone with S / 16 until everything is finite."""
    outs = run()
    if eng.scaler is None:
        return outs
    for _ in range(8):
        if all(bool(torch.isfinite(t).all()) for t in outs) or float(eng.scaler[0]) <= 1.0:
            break
        eng.reset_scaler(max(float(eng.scaler[0]) / 16.0, 1.0))
        outs = run()
    return outs


class _HeadFn(torch.autograd.Function):
    """`model.itm(pooled)` / `model.mlm(hidden)` as the reference's callers use them on their own
    (cxrbert_origin.py:147-148; Downstream_task/Retrieval/retrieval.py:26-31): the head's HIP kernels over ANY input tensor,
    differentiable w.r.t. the input and the head's parameters (gradients come back as fresh tensors, like _CXRBertFn's)."""

    @staticmethod
    def forward(ctx, model, kind, x, *params):
        from . import hip_ops as ops
        from ._lib import EPI_BIAS
        eng = model.engine
        H, V = eng.cfg.hidden, eng.cfg.vocab_size
        if x.shape[-1] != H:
            raise ValueError(f"{kind}: last dimension {x.shape[-1]} != hidden size {H}")
        if eng.shadow_dirty:
            eng.sync_shadow()
        eng.wait_optimizer()
        xin = x.detach().to(eng.device).reshape(-1, H)
        R = xin.shape[0]
        xf = xin.to(eng.fadt).contiguous()
        xb = xf if not eng.dual else xin.to(eng.adt).contiguous()
        ctx.model, ctx.kind, ctx.R, ctx.xb, ctx.in_dtype = model, kind, R, xb, x.dtype
        if kind == "itm":
            out = torch.empty((R, 2), dtype=torch.float32, device=eng.device)
            ops.gemm(xf, eng.wf["itm.linear.weight"], out, M=R, N=2, K=H, bias=eng.p["itm.linear.bias"], epi=EPI_BIAS)
            return out.view(*x.shape[:-1], 2)
        if not hasattr(eng, "S"):
            eng.S = {}
        logits = eng._mlm_forward(xf, xb, R, "hm_", pad=False)
        # The engine's scratch buffers are cached by name: a second model.mlm(...) before this call's backward would overwrite what this
        # call saved.  The reference's nn.Module heads are re-entrant (two calls in one graph, gradient accumulation over two forwards),
        # so every call keeps its OWN copies of the saved activations ([R, H] each: small beside the [R, V] logits it returns).
        ctx.hs = {k: (v.clone() if (torch.is_tensor(v) and k != "logits") else v) for k, v in eng.S["hm_"].items() if k != "logits"}
        return logits.view(*x.shape[:-1], V)

    @staticmethod
    def backward(ctx, g):
        from . import hip_ops as ops
        model, kind, R = ctx.model, ctx.kind, ctx.R
        eng = model.engine
        H, V = eng.cfg.hidden, eng.cfg.vocab_size
        adt, dev = eng.adt, eng.device
        g32 = g.detach().to(dev, torch.float32).contiguous()
        eng.ensure_grad()

        def scaled(t):
            return t if eng.loss_scale_dev is None else t * eng.loss_scale_dev

        def unscaled(t):
            t = t.float()
            return t if eng.unscale_dev is None else t * eng.unscale_dev

        if kind == "itm":
            def run():
                us = eng.unscale_dev
                d8 = torch.zeros((R, 8), dtype=adt, device=dev)
                ops.cast2d(scaled(g32.view(R, 2)), 2, d8, 8, R, 2)
                gW = torch.zeros((2, H), dtype=torch.float32, device=dev)
                gb = torch.zeros((2,), dtype=torch.float32, device=dev)
                ops.colsum(d8, 8, R, 2, gb, accumulate=True, unscale=us)
                eng._dW(d8, ctx.xb, gW, 2, H, R, lda=8, ldb=H)
                dx = torch.empty((R, H), dtype=adt, device=dev)
                ops.gemm(d8, eng.w["itm.linear.weight"], dx, tb=True, M=R, N=H, K=2, lda=8, ldb=H)
                return [unscaled(dx), gW, gb]
            dx, gW, gb = _scaled_retry(eng, run)
            return None, None, dx.view(*g.shape[:-1], H).to(ctx.in_dtype), gW, gb

        Vp = (V + 7) // 8 * 8
        if _use_views(model):
            _own_grads(model)        # this backward uses ranges of the flat gradient buffer as scratch: a .grad must not be a view of them

        def run():
            eng.S["hm_"] = ctx.hs
            ctx.hs["Vp"] = Vp
            for n in _MLM_PARAMS[:-1]:        # this head's own gradients only (the tied matrix's is overwritten, not accumulated);
                eng.g[n].zero_()              # the pooler's / ITM head's gradients in the same bucket are somebody else's
            dl = torch.empty((R, Vp), dtype=adt, device=dev)
            ops.cast2d(scaled(g32.view(R, V)), V, dl, Vp, R, V)
            dxr = eng._mlm_backward(dl, "hm_")
            if eng._side is not None:
                torch.cuda.current_stream().wait_stream(eng._side)     # the decoder's / transform's parameter gradients (side stream)
            eng._dE_ev = None
            # (dxr is a view of a named scratch buffer of the engine: hand autograd a tensor of its own -- on the fp32 path `unscaled` is the
            # identity, and the next head backward would overwrite what this one returned)
            return [unscaled(dxr).clone()] + [eng.g[n].clone() for n in _MLM_PARAMS]
        outs = _scaled_retry(eng, run)
        return (None, None, outs[0].view(*g.shape[:-1], H).to(ctx.in_dtype)) + tuple(outs[1:])


class _Sub(nn.Module):
    """Namespace module so that parameters appear under the reference's dotted names."""


class CXRBERT(nn.Module):
    """Multimodal BERT: Masked Language Model + Image Text Matching (cxrbert_origin.py:132-149)."""

    def __init__(self, config, args=None, dtype=torch.bfloat16, device=None, img_encoder=None, fwd_operand=None, grad_operand=None):
        """dtype: torch.float32 (exact path) or torch.bfloat16 (16-bit MFMA path); fwd_operand / grad_operand: encodings of the
        operands of the forward / gradient products of the 16-bit path ("f16" (default) or "bf16") -- see engine.Engine."""
        super().__init__()
        self.cfg = model_config_from(config, args)
        self.config = config
        self.args = args if args is not None else SimpleNamespace()
        if getattr(self.args, "disturbing_mask", False):
            # the reference's disturbing_mask branch is shape-inconsistent and cannot run
            raise NotImplementedError("disturbing_mask model branch: use the non-cross MASK pattern with the standard branch")
        dev = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.engine = Engine(self.cfg, dtype=dtype, device=dev, fwd_operand=fwd_operand, grad_operand=grad_operand)
        if isinstance(img_encoder, str):
            if img_encoder.lower() not in ("resnet50", "cnn"):
                raise NotImplementedError(f"img_encoder {img_encoder!r}: only the ResNet-50 region encoder is mirrored")
            from .image import ImageEncoder_cnn           # models/image.py:46-69, cxrbert_origin.py:64-65
            img_encoder = ImageEncoder_cnn(self.args, dtype=dtype).to(dev)
        self.img_encoder = img_encoder
        self._param_names = list(self.engine.layout.keys())
        # drop-in switches: lazy MLM logits; recognition of the Dataset's materialised masks on the lazy path; gradients
        # handed to torch as views of the flat gradient buffer instead of copies (see _CXRBertFn.backward)
        # grad_views: None = automatic -- views unless something hooks autograd's per-parameter accumulation (a process group of more than
        # one rank is initialised: DistributedDataParallel's reducer; a Tensor.register_hook on a Parameter), True / False force it
        self.lazy_logits, self.recognise_masks, self.grad_views = False, True, None
        self.grad_in_loss = True          # lazy logits: the loss runs the head once, with its gradient (see _LazyLossFn)
        self.n_masks_seen = self.n_masks_recognised = 0
        self._lazy_rows = None
        self._opt_versions = None
        self._register()
        self.reset_parameters()

    # parameters are views of the engine's flat fp32 buffer, registered under the reference names
    def _register(self):
        for name in self._param_names:
            mod = self
            parts = name.split(".")
            for p_ in parts[:-1]:
                if p_ not in mod._modules:
                    mod.add_module(p_, _Sub())
                mod = mod._modules[p_]
            par = nn.Parameter(self.engine.p[name], requires_grad=True)
            par._medvill_model = weakref.ref(self)           # medvill_amd.optim.AdamW(model.parameters()) finds the flat buffers through it
            mod._parameters[parts[-1]] = par
        # the Parameter objects in layout order (the same objects for the model's lifetime: _rebind only re-points their .data).  Looking
        # 200 names up with get_parameter costs ~1 ms of host time per use, and the drop-in step does it with the device idle
        self._plist = [self.get_parameter(n) for n in self._param_names]
        self.enc.forward = self._enc_forward
        # the heads are callable on their own like the reference's sub-modules (cxrbert_origin.py:147-148, retrieval.py:31)
        self.itm.forward = self._itm_module_forward
        self.mlm.forward = self._mlm_module_forward
        self.mlm.predictions.forward = lambda hidden: self._mlm_module_forward(hidden)[0]       # BertLMPredictionHead, :234-238

    def _rebind(self):
        for name in self._param_names:
            mod = self
            parts = name.split(".")
            for p_ in parts[:-1]:
                mod = mod._modules[p_]
            mod._parameters[parts[-1]].data = self.engine.p[name]

    def _apply(self, fn, *a, **k):
        # .to(device) / .cuda(): move the flat buffers, then re-point the Parameter views
        probe = fn(torch.empty(0, device=self.engine.device))
        if probe.dtype != torch.float32 and probe.dtype != torch.empty(0).dtype:
            raise RuntimeError("parameters stay fp32 master weights; choose the compute dtype with CXRBERT(dtype=...)")
        self.engine.to(probe.device)
        self._rebind()
        return self

    @torch.no_grad()
    def reset_parameters(self, seed: int | None = None):
        """BertModel(config) self-init N(0, 0.02), LayerNorm (1, 0), zero biases; the heads keep
        torch's default Linear init because CXRBERT.__init__ never calls init_weights
        (cxrbert_origin.py:137-142)."""
        gen = torch.Generator(device="cpu")
        gen.manual_seed(torch.initial_seed() if seed is None else seed)
        H = self.cfg.hidden
        flat = torch.zeros(self.engine.n_flat, dtype=torch.float32)
        for name, (off, shape) in self.engine.layout.items():
            n = 1
            for s in shape:
                n *= s
            v = flat[off:off + n].view(shape)
            head_lin = name.startswith(("mlm.predictions.transform.dense", "itm.linear", "enc.img_embeddings.img_embeddings"))
            if name.endswith("LayerNorm.weight"):
                v.fill_(1.0)
            elif head_lin:
                fan_in = shape[-1] if len(shape) == 2 else {"mlm.predictions.transform.dense.bias": H, "itm.linear.bias": H,
                                                           "enc.img_embeddings.img_embeddings.bias": self.cfg.img_hidden}[name]
                bound = 1.0 / (fan_in ** 0.5)
                v.copy_((torch.rand(shape, generator=gen) * 2 - 1) * bound)
            elif name.endswith("bias"):
                v.zero_()
            else:
                v.copy_(torch.randn(shape, generator=gen) * 0.02)
        self.engine.flat_p.copy_(flat.to(self.engine.device))
        self.engine.shadow_dirty = True

    # ------------------------------------------------------------------ forward
    def _regions(self, input_img):
        if isinstance(input_img, (tuple, list)) and len(input_img) == 2:
            return input_img
        if self.img_encoder is not None:
            return self.img_encoder(input_img)
        raise TypeError("input_img must be (region_feats[B,N,2048], region_pos[B,N]); for pixels construct the model with "
                        "img_encoder='resnet50' (medvill_amd.image.ImageEncoder_cnn) or pass your own callable")

    def _params_dirty(self):
        """True unless medvill_amd.optim.AdamW (whose kernel also writes the 16-bit copies) was the last to modify the Parameters: it records
        the sum of their version counters, which every in-place operation of a torch optimizer, `load_state_dict`, `p.add_(...)` advances.
        (Writes through `p.data` leave no trace -- after such an edit set `model.engine.shadow_dirty = True`.)"""
        v = self._opt_versions
        return v is None or v != sum(p._version for p in self._plist)

    def _mask_descriptors(self, mask, input_txt, N):
        """A materialised reference mask on the device -> MaskDesc when it IS one of the closed-form families, entry by entry: the
        hypothesis of data.descriptors_from_dense is confirmed by comparing the mask words the kernels would run on (mv_mask_pack of
        the matrix against mv_mask_build of the descriptors) before it is used.  One read of the matrix, one small copy to the host
        (descriptors + verdict: the row plan needs the lengths there).  None: keep the matrix."""
        from .data import MaskDesc, descriptors_from_dense
        from . import hip_ops as ops
        hyp = descriptors_from_dense(mask, input_txt, N)
        if hyp is None:
            return None
        desc, ok = hyp
        eng = self.engine
        B, Lq = mask.shape[0], mask.shape[-1]
        W32, Tt = (Lq + 31) // 32, (Lq + 63) // 64
        words = [eng._buf(f"rec_bits{i}", (B, Lq, W32), torch.int32) for i in range(2)]
        tiles = [eng._buf(f"rec_tinfo{i}", (B, Tt, Tt), torch.uint8) for i in range(2)]
        ops.mask_pack(mask, words[0], tiles[0])
        ops.mask_build(desc, B, Lq, words[1], tiles[1])
        same = ok & (words[0] == words[1]).all()
        host = torch.cat([desc.reshape(-1), same.to(torch.int32).view(1)]).cpu()
        self.n_masks_seen += 1
        if int(host[-1]) != 1:
            return None
        self.n_masks_recognised += 1
        return MaskDesc(desc, Lq, host=host[:-1].view(B, 3).clone())

    def _run(self, want_heads, cls_tok, input_txt, attn_mask, segment, input_img, sep_tok, txt_labels=None):
        if attn_mask.dim() not in (2, 3):
            raise NotImplementedError            # cxrbert_origin.py:80-81
        feats, pos = self._regions(input_img)
        self.__dict__.pop("_fwd_labels", None)
        if want_heads == 3 and self.recognise_masks and torch.is_tensor(attn_mask) and attn_mask.is_cuda and self.engine.is16:
            # lazy logits: nothing downstream needs a row per position, so a batch whose masks are the Dataset's closed forms runs on
            # its valid rows (packed), like the fused training step does for the trainer
            if self.engine.shadow_dirty or self._params_dirty():
                self.engine.sync_shadow()        # (enqueued first: the device converts the weights while the host waits for the verdict)
            self.__dict__["_shadow_fresh"] = True
            attn_mask = self._mask_descriptors(attn_mask, input_txt, int(feats.shape[1])) or attn_mask
        if want_heads == 3 and txt_labels is not None:
            from .data import MaskDesc, label_index
            dev = self.engine.device
            rows, ids = label_index(txt_labels.to(dev))
            if isinstance(attn_mask, MaskDesc) and self.engine.is16 and attn_mask.packable() and rows.numel() > 0:
                # packed rows: a label past a sample's text [SEP] has no row (the Dataset never produces one, dataset_origin.py:105-135)
                Lq = attn_mask.L
                r64 = rows.to(torch.int64)
                vl = attn_mask.desc.to(dev)[:, 2].to(torch.int64)
                if bool((r64 % Lq >= vl.index_select(0, r64 // Lq)).any()):
                    raise ValueError("txt_labels holds a label at a padded position (after the text [SEP]); the lazy forward runs on the "
                                     "valid rows only.  Set model.recognise_masks = False (and pass the mask matrix) to run every row")
            self.__dict__["_fwd_labels"] = (rows, ids, txt_labels)
        params = self._plist
        self._want_grad = torch.is_grad_enabled()        # (grad mode is always off inside autograd.Function.forward)
        return _CXRBertFn.apply(self, want_heads, cls_tok, input_txt, attn_mask, segment, feats, pos, sep_tok, *params)

    def _enc_forward(self, cls_tok, input_txt, attn_mask, segment, input_img, sep_tok):
        hidden, pooled = self._run(False, cls_tok, input_txt, attn_mask, segment, input_img, sep_tok)
        return hidden, pooled, None

    def forward(self, cls_tok, input_txt, attn_mask, segment, input_img, sep_tok, txt_labels=None):
        """cxrbert_origin.py:144-149.  txt_labels (optional, lazy logits only; not part of the reference's signature): the MLM labels the loss
        will use -- with them the last encoder layer runs on the rows the heads consume only (what the fused training step does)."""
        if txt_labels is not None and not getattr(self, "lazy_logits", False):
            raise ValueError("forward(..., txt_labels=...) needs model.lazy_logits = True (the full logits need every row)")
        if getattr(self, "lazy_logits", False):
            # (mlm, itm) like the reference, with `mlm` a LazyLogits handle for medvill_amd.losses.mlm_itm_loss (see LazyLogits)
            tok, itm = self._run(3, cls_tok, input_txt, attn_mask, segment, input_img, sep_tok, txt_labels=txt_labels)
            return LazyLogits(self, tok, (itm.shape[0], self.engine.S["L"], self.cfg.vocab_size)), itm
        return self._run(True, cls_tok, input_txt, attn_mask, segment, input_img, sep_tok)

    def _itm_module_forward(self, x):
        """ImageTextMatching.forward (cxrbert_origin.py:172-173): x [..., H] -> [..., 2]."""
        return _HeadFn.apply(self, "itm", x, *[self.get_parameter(n) for n in _ITM_PARAMS])

    def _mlm_module_forward(self, sequence_output):
        """BertPreTrainingHeads.forward (cxrbert_origin.py:245-248): [..., H] -> (prediction_scores [..., V], None)."""
        return _HeadFn.apply(self, "mlm", sequence_output, *[self.get_parameter(n) for n in _MLM_PARAMS]), None

    def _itm_only(self, cls_tok, input_txt, attn_mask, segment, input_img, sep_tok):
        """ITM logits [B,2] of enc + itm without the MLM head (the retrieval model's forward), differentiable."""
        return self._run(2, cls_tok, input_txt, attn_mask, segment, input_img, sep_tok)

    # ------------------------------------------------------------------ state dict (reference key names)
    def state_dict(self, *a, **k):
        self.engine.wait_optimizer()
        sd = OrderedDict((n, self.engine.p[n].detach().clone()) for n in self._param_names)
        for alias, canon in ALIASES.items():
            sd[alias] = sd[canon]
        if isinstance(self.img_encoder, nn.Module):        # enc.img_encoder.model.* like the reference's checkpoints
            for k_, v in self.img_encoder.state_dict().items():
                sd["enc.img_encoder." + k_] = v.detach().clone()
        return sd

    def load_state_dict(self, sd, strict=True):
        self.engine.wait_optimizer()
        missing = []
        with torch.no_grad():
            for n in self._param_names:
                if n in sd:
                    self.engine.p[n].copy_(sd[n].to(self.engine.device, torch.float32))
                else:
                    missing.append(n)
        cnn = {k_[len("enc.img_encoder."):]: v for k_, v in sd.items() if k_.startswith("enc.img_encoder.")}
        if cnn and isinstance(self.img_encoder, nn.Module):
            self.img_encoder.load_state_dict(cnn, strict=strict)
        unexpected = [k_ for k_ in sd if k_ not in self.engine.layout and k_ not in ALIASES
                      and "position_ids" not in k_ and not k_.startswith("enc.img_encoder.")]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        self.engine.shadow_dirty = True
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    def save_pretrained(self, save_directory):
        """HF layout: config.json + pytorch_model.bin (train_origin.py:254-266)."""
        os.makedirs(save_directory, exist_ok=True)
        c = self.cfg
        cj = dict(architectures=["CXRBERT"], model_type="bert", vocab_size=c.vocab_size, hidden_size=c.hidden,
                  num_hidden_layers=c.layers, num_attention_heads=c.heads, intermediate_size=c.intermediate,
                  max_position_embeddings=c.max_pos, type_vocab_size=c.type_vocab, layer_norm_eps=c.ln_eps,
                  hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(cj, f, indent=2)
        torch.save({k_: v.cpu() for k_, v in self.state_dict().items()}, os.path.join(save_directory, "pytorch_model.bin"))

    @classmethod
    def from_pretrained(cls, path, state_dict=None, config=None, args=None, **kw):
        if config is None:
            with open(os.path.join(path, "config.json")) as f:
                config = json.load(f)
        if state_dict is None:
            state_dict = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
        m = cls(config, args, **kw)
        m.load_state_dict(state_dict, strict=False)
        return m
