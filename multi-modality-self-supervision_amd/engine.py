"""Execution engine of the CXRBERT pretraining hot path on MI355X.

One flat fp32 buffer holds every parameter (plus flat grad / Adam-moment buffers and, in bf16
mode, a flat bf16 shadow that the MFMA kernels read), so the optimizer is ONE fused kernel and
the data-parallel gradient exchange is a handful of large contiguous RCCL all-reduces.  The
forward and backward passes are explicit kernel schedules over the C ABI (include/medvill.h):
no autograd graph, no tracing compiler; activations needed by the backward live in a
workspace sized once per (B, L) -- 288 GB of HBM3E makes recomputation unnecessary.

Reference semantics implemented here (paths relative to the upstream repo):
  CXRBertEncoder.forward else-branch ......... models/cxrbert_origin.py:114-130
  ImageBertEmbeddings ........................ models/cxrbert_origin.py:22-35
  HF BertLayer x layers, BertPooler .......... call sites cxrbert_origin.py:72-73,126-130
  BertPreTrainingHeads / ImageTextMatching ... models/cxrbert_origin.py:164-173,205-248
  losses, metrics, AdamW step ................ models/train_origin.py:62-63,106-146
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from dataclasses import asdict, dataclass

import torch

from . import hip_ops as ops
from ._lib import (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GELU_D, EPI_BIAS_RES, EPI_BIAS_TANH, EPI_MUL, EPI_NONE, EPI_RES, MV_BF16, MV_F16,
                   MV_F32)


@dataclass
class ModelConfig:
    vocab_size: int = 30522
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    max_pos: int = 512
    type_vocab: int = 2
    img_hidden: int = 2048
    ln_eps: float = 1e-12
    head_ln_eps: float = 1e-5
    dropout: float = 0.1          # hidden_dropout_prob = attention_probs_dropout_prob = 0.1 in the reference's BertConfig
    img_position: bool = True     # args.img_postion (sic): image rows add P[img_pos]  (cxrbert_origin.py:27-31)
    img_dropout: float = -1.0     # args.dropout_prob: ImageBertEmbeddings' own nn.Dropout (cxrbert_origin.py:19); < 0: same as `dropout`

    def to_dict(self):
        return asdict(self)


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


def param_layout(cfg: ModelConfig):
    """name -> (offset, shape) in the flat buffer, reference state-dict names.
    q/k/v weights (and biases) of a layer are adjacent so the fused [3H,H] projection is a view."""
    H, I, V = cfg.hidden, cfg.intermediate, cfg.vocab_size
    if H % 64 or I % 64 or cfg.img_hidden % 8:
        raise ValueError("hidden and intermediate sizes must be multiples of 64")
    lay = OrderedDict()
    off = 0

    def add(name, shape):
        nonlocal off
        n = 1
        for s in shape:
            n *= s
        lay[name] = (off, tuple(shape))
        off = _align(off + n)

    e = "enc.txt_embeddings."
    add(e + "word_embeddings.weight", (V, H))
    add(e + "position_embeddings.weight", (cfg.max_pos, H))
    add(e + "token_type_embeddings.weight", (cfg.type_vocab, H))
    add(e + "LayerNorm.weight", (H,))
    add(e + "LayerNorm.bias", (H,))
    add("enc.img_embeddings.img_embeddings.weight", (H, cfg.img_hidden))
    add("enc.img_embeddings.img_embeddings.bias", (H,))
    for l in range(cfg.layers):
        p = f"enc.encoder.layer.{l}."
        for n in ("query", "key", "value"):
            add(p + f"attention.self.{n}.weight", (H, H))
        for n in ("query", "key", "value"):
            add(p + f"attention.self.{n}.bias", (H,))
        add(p + "attention.output.dense.weight", (H, H))
        add(p + "attention.output.dense.bias", (H,))
        add(p + "attention.output.LayerNorm.weight", (H,))
        add(p + "attention.output.LayerNorm.bias", (H,))
        add(p + "intermediate.dense.weight", (I, H))
        add(p + "intermediate.dense.bias", (I,))
        add(p + "output.dense.weight", (H, I))
        add(p + "output.dense.bias", (H,))
        add(p + "output.LayerNorm.weight", (H,))
        add(p + "output.LayerNorm.bias", (H,))
    add("enc.pooler.dense.weight", (H, H))
    add("enc.pooler.dense.bias", (H,))
    add("mlm.predictions.transform.dense.weight", (H, H))
    add("mlm.predictions.transform.dense.bias", (H,))
    add("mlm.predictions.transform.LayerNorm.weight", (H,))
    add("mlm.predictions.transform.LayerNorm.bias", (H,))
    add("mlm.predictions.bias", (V,))
    add("itm.linear.weight", (2, H))
    add("itm.linear.bias", (2,))
    return lay, off


ALIASES = {  # reference state-dict aliases -> canonical tensor (cxrbert_origin.py:17-20,141,231)
    "enc.img_embeddings.position_embeddings.weight": "enc.txt_embeddings.position_embeddings.weight",
    "enc.img_embeddings.token_type_embeddings.weight": "enc.txt_embeddings.token_type_embeddings.weight",
    "enc.img_embeddings.LayerNorm.weight": "enc.txt_embeddings.LayerNorm.weight",
    "enc.img_embeddings.LayerNorm.bias": "enc.txt_embeddings.LayerNorm.bias",
    "mlm.predictions.decoder.weight": "enc.txt_embeddings.word_embeddings.weight",
}


# roctx ranges around the phases of a step: MV_ROCTX=1 pushes / pops a range per phase (embed, layer<l>.attention,
# layer<l>.ffn, heads, backward.layer<l>, backward.embed, overflow_check, optimizer; the data-parallel exchange marks "exchange") so that
# `rocprofv3 --marker-trace --kernel-trace` shows which launches belong to which phase.  The ranges are HOST ranges -- the host runs
# ~17 ms ahead of the device -- so MV_ROCTX=2 also synchronises the device at every phase boundary: each range then lasts as long as its
# kernels (at the price of the two-stream overlap).  Off (the default) costs one attribute test per phase.
_ROCTX = int(os.environ.get("MV_ROCTX", "0") or 0)
_phase_open = [False]


def phase(name):
    """Close the open roctx range, open `name` (None: just close)."""
    if not _ROCTX:
        return
    if _phase_open[0]:
        if _ROCTX >= 2:
            torch.cuda.synchronize()
        torch.cuda.nvtx.range_pop()
        _phase_open[0] = False
    if name is not None:
        torch.cuda.nvtx.range_push(name)
        _phase_open[0] = True


# The engine's extra HIP streams are shared by every Engine of the process (engines never run concurrently).  HIP hands streams to a
# fixed number of hardware queues round-robin in creation order; a second model's streams could land on the queue of the first one's main
# stream and serialise its backward (measured in bench.py: the second model of a process ran L = 768 steps at 40.7 ms instead of 35.8).
_SHARED_STREAMS = {}


def _shared_stream(device, kind):
    key = (str(device), kind)
    st = _SHARED_STREAMS.get(key)
    if st is None:
        st = _SHARED_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


class Engine:
    """dtype = torch.float32: the exact path (VALU kernels).  dtype = torch.bfloat16: the 16-bit MFMA path -- fp32 master
    weights, accumulation, LayerNorm / softmax statistics, losses and optimizer; every MFMA operand is 16 bits wide.
    `fwd_operand` / `grad_operand` pick the ENCODING of the operands of the forward / gradient products of that path:
      fwd "f16", grad "f16" (default)  ONE f16 copy (11-bit significand) of every weight and stored activation.  This is
             the encoding that meets the 1e-2 logit tolerance at BERT-base (profiles/r02_bf16_error.txt: with bf16-encoded
             forward operands the weight rounding alone is 1.4e-2 max-abs).  f16 gradients need a loss scale: the loss
             gradients enter the chain multiplied by S, the kernels that write f32 parameter gradients multiply by 1/S, S
             lives on the device and backs off when a step overflows (that step is skipped) -- `scaler` below;
      fwd "f16", grad "bf16"  round 2's form: gradient products keep bf16 operands (8-bit exponent, no loss scale), so a
             forward activation that the backward needs as a gradient-product operand is stored twice (+4.7 GB of writes
             per step at the benchmark shape);
      fwd "bf16", grad "bf16"  one bf16 copy of everything (round 1)."""

    def __init__(self, cfg: ModelConfig, dtype=torch.bfloat16, device="cuda", fwd_operand=None, grad_operand=None):
        if dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("compute dtype must be float32 or bfloat16")
        if fwd_operand is None:
            fwd_operand = os.environ.get("MV_FWD_OPERAND", "f16")
        if fwd_operand not in ("f16", "bf16"):
            raise ValueError("fwd_operand must be 'f16' or 'bf16'")
        if grad_operand is None:
            grad_operand = os.environ.get("MV_GRAD_OPERAND", "f16" if fwd_operand == "f16" else "bf16")
        if grad_operand not in ("f16", "bf16") or (grad_operand == "f16" and fwd_operand != "f16"):
            raise ValueError("grad_operand must be 'f16' (with f16 forward operands) or 'bf16'")
        self.cfg = cfg
        self.is16 = dtype != torch.float32
        # encoding of gradients and of the activations the gradient products read
        self.adt = torch.float32 if not self.is16 else (torch.float16 if grad_operand == "f16" else torch.bfloat16)
        self.dt = MV_F32 if not self.is16 else (MV_F16 if grad_operand == "f16" else MV_BF16)
        # encoding of the forward operands
        self.fadt = torch.float32 if not self.is16 else (torch.float16 if fwd_operand == "f16" else torch.bfloat16)
        self.fdt = MV_F32 if not self.is16 else (MV_F16 if fwd_operand == "f16" else MV_BF16)
        self.dual = self.is16 and self.fdt != self.dt          # two encodings of every stored activation / weight
        self.device = torch.device(device)
        self.layout, self.n_flat = param_layout(cfg)
        self.flat_p = torch.zeros(self.n_flat, dtype=torch.float32, device=self.device)
        self.flat_g = None
        self.flat_m = None
        self.flat_v = None
        # 16-bit copies of the weights that the MFMA kernels read: bf16 and / or f16, whichever encodings are in use
        self.shadow = torch.zeros(self.n_flat, dtype=torch.bfloat16, device=self.device) if MV_BF16 in (self.dt, self.fdt) else None
        self.shadow_f = torch.zeros(self.n_flat, dtype=torch.float16, device=self.device) if MV_F16 in (self.dt, self.fdt) else None
        # dynamic loss scale of the f16-gradient path: device state f32[8] (include/medvill.h, mv_scaler_update)
        self.scaler = None
        self.loss_scale_init = float(os.environ.get("MV_LOSS_SCALE", 2.0 ** 15))
        self.scale_growth_interval = 2000
        if self.dt == MV_F16:
            self.scaler = torch.zeros(8, dtype=torch.float32, device=self.device)
            self.reset_scaler()
        self.shadow_dirty = True
        self._ws = {}
        self._gemm_ws = {}             # split-K workspaces, one per stream that runs split-K GEMMs (never shared across streams)
        self._side = None             # side HIP stream for the weight-gradient GEMMs of the backward
        self.training = False         # dropout is active only when True (CXRBERT.train() / TrainStep(train=True))
        self.keep_acts = True         # False: the forward keeps no per-layer activations (inference; see encoder_forward)
        # dropout stream: keyed by torch's seed (set_seed of utils/utils.py:9-16 -> torch.manual_seed), a per-rank offset
        # added by TrainStep under data parallelism, and a counter advanced once per forward (every step draws fresh masks)
        # f16 forward path: the encoder layers' LayerNorm inputs (residual sums, written by the output-projection / FFN-down GEMMs
        # and read by LayerNorm forward and backward) are stored in f16 instead of fp32: -0.47 ms/step; BERT-base logits
        # 3.2-3.4e-3 -> 3.8-4.6e-3 max-abs from the reference (tolerance 1e-2; profiles/r02_bf16_error.txt).  MV_LN_IN_16=0: fp32.
        self.ln_in_16 = os.environ.get("MV_LN_IN_16", "1") != "0"
        # dz = dproj2 . W2 (the widest input-gradient GEMM) as y = x . (W2^T)^T over a transposed bf16 copy of the FFN-down weights:
        # the 256-row LDS-DMA kernel in its row-major form is the fastest GEMM of the library and the contraction-major form on
        # 128x128 tiles is fill-bound (profiles/r02_gemm_variants.txt).  The copies (12 x 4.7 MB) are refreshed behind the
        # optimizer step on the side stream.
        self.dz_nt = os.environ.get("MV_DZ_NT", "1") != "0"
        self._w2t, self._w2t_ev, self._w2t_stale = None, None, True
        self.head_on_side = True  # the tied decoder's weight gradient on the side stream (see _mlm_backward)
        self.fused_colsum = os.environ.get("MV_FUSED_COLSUM", "1") != "0"   # bias gradients from partial sums of the producing kernels
        self.head_params_on_side = True   # MLM head: decoder-bias column sums and the transform's parameter gradients on the side stream
        self.late_opt_wait = True     # the forward's preparation kernels run under the optimizer's first kernel (encoder_forward)
        self.itm_on_side = True       # ITM head on the side stream under the MLM head's decoder GEMM (heads_train)
        # fused step, 16-bit path: MLM logits in the forward encoding (see _mlm_forward).  Off: the in-process A/B showed no gain
        # (24.60 vs 24.58 ms, profiles/r03_notes.txt), so the logits stay f32
        self.logits_16 = os.environ.get("MV_LOGITS_16", "0") == "1"
        self.tail_queries = os.environ.get("MV_TAIL_QUERIES", "1") != "0"    # last layer's attention: consumed rows only as queries (see encoder_forward)
        self._mask_stream = None      # third stream: the attention-dropout keep-bits of every layer are generated at the start of a forward
        self._dE_ev = None
        self._opt_ev = None       # overlapped AdamW: parameter range -> event (see adamw_step)
        self.head_splitk = True  # split-K for the decoder's input gradient (see _mlm_backward)
        self.dw_splitk = 0      # weight gradients: 0 = the library fills the chip with split-K slabs; n > 1 caps the slab count
        self.drop_seed = (torch.initial_seed() ^ 0x5DEECE66D) & 0xFFFFFFFFFFFFFFFF
        self.drop_counter = 0
        self._bind()

    # ------------------------------------------------------------------ loss scale (f16 gradients)
    def reset_scaler(self, scale=None):
        if self.scaler is None:
            return
        s0 = float(scale if scale is not None else self.loss_scale_init)
        self.scaler.copy_(torch.tensor([s0, 1.0 / s0, 0, 0, 0, 0, 0, 0], dtype=torch.float32))

    @property
    def loss_scale_dev(self):      # f32[1] device view: S (None when gradients are not f16)
        return None if self.scaler is None else self.scaler[0:1]

    @property
    def unscale_dev(self):         # f32[1] device view: 1 / S
        return None if self.scaler is None else self.scaler[1:2]

    def _shadow_of(self, dt):
        return self.flat_p if dt == MV_F32 else (self.shadow_f if dt == MV_F16 else self.shadow)

    # ------------------------------------------------------------------ storage
    def _bind(self):
        # p: fp32 master weights; w: what the gradient products read (16-bit shadow); wf: what the forward products read
        self.p, self.w, self.wf = {}, {}, {}
        src_w = self._shadow_of(self.dt)
        src_f = self._shadow_of(self.fdt)
        for name, (off, shape) in self.layout.items():
            n = math.prod(shape)
            self.p[name] = self.flat_p[off:off + n].view(shape)
            self.w[name] = src_w[off:off + n].view(shape)
            self.wf[name] = src_f[off:off + n].view(shape)
        self.g = {}
        if self.flat_g is not None:
            for name, (off, shape) in self.layout.items():
                self.g[name] = self.flat_g[off:off + math.prod(shape)].view(shape)

    def ensure_grad(self):
        if self.flat_g is None:
            self.flat_g = torch.zeros_like(self.flat_p)
            self._bind()

    def ensure_opt(self):
        self.ensure_grad()
        if self.flat_m is None:
            self.flat_m = torch.zeros_like(self.flat_p)
            self.flat_v = torch.zeros_like(self.flat_p)

    def to(self, device):
        device = torch.device(device)
        if device == self.device:
            return self
        if self.device.type == "cuda":
            self.wait_optimizer()          # an overlapped AdamW may still be writing the buffers that are copied below
            torch.cuda.current_stream().synchronize()
        # host storage is allowed (state-dict I/O); every kernel call requires device tensors and raises otherwise
        for k in ("flat_p", "flat_g", "flat_m", "flat_v", "shadow", "shadow_f", "scaler"):
            t = getattr(self, k)
            if t is not None:
                setattr(self, k, t.to(device))
        self.device = device
        self._ws.clear()
        self.shadow_dirty = True
        self._gemm_ws = {}
        self._bind()
        return self

    def sync_shadow(self):
        """16-bit path: refresh the copies the MFMA kernels read from the fp32 master weights."""
        self.wait_optimizer()
        if self.shadow is not None:
            ops.cast(self.flat_p, self.shadow, self.n_flat)
        if self.shadow_f is not None:
            ops.cast(self.flat_p, self.shadow_f, self.n_flat)
        self.shadow_dirty = False
        self._w2t_stale = True

    def refresh_w2t(self, on_side=True):
        """Transposed bf16 copies of the FFN-down weights ([I, H] each) for the NT form of dz; enqueued on the side stream (idle
        outside the backward) so that the main stream only waits for an event at its first dz."""
        if not (self.dz_nt and self.is16):
            return
        cfg = self.cfg
        H, I = cfg.hidden, cfg.intermediate
        if self._w2t is None:
            self._w2t = [torch.empty((I, H), dtype=self.adt, device=self.device) for _ in range(cfg.layers)]
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = _shared_stream(self.device, "side")
        side = self._side if (on_side and os.environ.get("MV_SINGLE_STREAM") != "1") else main
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for l in range(cfg.layers):
                ops.transpose(self.w[f"enc.encoder.layer.{l}.output.dense.weight"], self._w2t[l], H, I)
            self._w2t_ev = torch.cuda.Event()
            self._w2t_ev.record(side)
        self._w2t_stale = False

    def side_stream(self):
        """The engine's second HIP stream (weight gradients during the backward; idle otherwise), or the current stream under
        MV_SINGLE_STREAM=1."""
        if os.environ.get("MV_SINGLE_STREAM") == "1":
            return torch.cuda.current_stream()
        if self._side is None:
            self._side = _shared_stream(self.device, "side")
        return self._side

    def zero_grad(self):
        self.ensure_grad()
        self.flat_g.zero_()

    def qkv_views(self, l, fwd=False):
        """fused [3H,H] weight (forward encoding when `fwd`, else the gradient products' one), [3H] bias (f32) and their
        grads for layer l."""
        H = self.cfg.hidden
        name = f"enc.encoder.layer.{l}.attention.self.query.weight"
        bname = f"enc.encoder.layer.{l}.attention.self.query.bias"
        off, _ = self.layout[name]
        boff, _ = self.layout[bname]
        src_w = self._shadow_of(self.fdt if fwd else self.dt)
        W = src_w[off:off + 3 * H * H].view(3 * H, H)
        b = self.flat_p[boff:boff + 3 * H]
        gW = gb = None
        if self.flat_g is not None:
            gW = self.flat_g[off:off + 3 * H * H].view(3 * H, H)
            gb = self.flat_g[boff:boff + 3 * H]
        return W, b, gW, gb

    # ------------------------------------------------------------------ dropout keys
    SITE_EMB, SITE_ATTN, SITE_OUT1, SITE_OUT2 = 1, 2, 3, 4

    def _drop_keys(self, layers):
        """64-bit keys of the counter-based dropout masks of this forward pass: (site, layer) -> key (splitmix64)."""
        def mix(x):
            x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
            x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
            x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
            return x ^ (x >> 31)
        base = mix(self.drop_seed ^ (self.drop_counter << 24))
        keys = {(self.SITE_EMB, 0): mix(base ^ 1)}
        for l in range(layers):
            for site in (self.SITE_ATTN, self.SITE_OUT1, self.SITE_OUT2):
                keys[(site, l)] = mix(base ^ (site << 16) ^ (l + 1))
        return keys

    # ------------------------------------------------------------------ workspaces
    def _buf(self, key, shape, dtype):
        t = self._ws.get(key)
        n = math.prod(shape)
        if t is None or t.numel() < n or t.dtype != dtype:
            # packed rows: the row count changes from batch to batch -- allocate row-shaped scratch for the padded row count
            # once instead of growing it whenever a longer batch comes by
            cap = n
            rows, rows_cap = getattr(self, "_rows_now", 0), getattr(self, "_rows_cap", 0)
            if rows_cap > rows > 0 and len(shape) > 0 and shape[0] == rows:
                cap = n // rows * rows_cap
            elif t is not None and t.dtype == dtype:
                cap = n + n // 4          # a buffer that had to grow (labelled-row counts vary): leave headroom
            t = torch.empty(cap, dtype=dtype, device=self.device)
            self._ws[key] = t
        return t[:n].view(shape)

    def _pair(self, key, shape):
        """(forward-encoding buffer, gradient-product-encoding buffer) of one stored activation: two buffers on the f16
        forward path, one and the same otherwise.  The second is passed to the producing kernel as its extra output."""
        f = self._buf(key, shape, self.fadt)
        return (f, self._buf(key + "_b", shape, self.adt)) if self.dual else (f, f)

    def _gemm_workspace(self, nfloat):
        """Split-K workspace of the CURRENT stream.  The main stream and the side stream each own one: a buffer shared between
        them would be written by two split-K GEMMs at once whenever both streams take the split-K route (small-vocabulary
        models: the tied decoder's weight gradient on the side stream next to the head's input gradient on the main one)."""
        st = torch.cuda.current_stream() if self.device.type == "cuda" else None
        key = st.cuda_stream if st is not None else 0
        ws = self._gemm_ws.get(key)
        if ws is None or ws.numel() < nfloat:
            if ws is not None and st is not None:
                ws.record_stream(st)          # kernels already enqueued on this stream may still use the old buffer
            ws = self._gemm_ws[key] = torch.empty(nfloat, dtype=torch.float32, device=self.device)
        return ws

    # dW[No,Ko] = dy[Mtok,No]^T . x[Mtok,Ko]  (contraction over tokens; split-K when the tile grid is small)
    def _dW(self, dy, x, gW, No, Ko, Mtok, lda, ldb, ldc=None):
        # split-K is chosen by the library (splitk=0) from the tile grid and the workspace: up to 32 partial slabs for the small
        # [H, H] gradients (9 tiles of 256x256 need ~24 slabs to occupy the chip), 16 for the large ones
        auto = self.is16 and Mtok >= 2048 and No * Ko <= 4 * 1024 * 1024
        # the library says how much workspace its own split-K choice for this product wants (mv_gemm_workspace_bytes)
        wb = ops.gemm_workspace_bytes(self.adt, True, True, No, Ko, Mtok) if auto else 0
        auto = auto and wb > 0
        ws = self._gemm_workspace(wb // 4) if auto else None
        # f16 gradients are loss-scaled: the weight gradient is un-scaled where it is written (alpha = 1 / S from the device)
        ops.gemm(dy, x, gW, ta=True, tb=True, M=No, N=Ko, K=Mtok, lda=lda, ldb=ldb, ldc=ldc, splitk=self.dw_splitk if auto else 1, ws=ws,
                 alpha=self.unscale_dev)

    # ------------------------------------------------------------------ encoder forward
    def encoder_forward(self, cls_tok, input_txt, attn_mask, segment, img_feats, img_pos, sep_tok, pack=False, tail_rows=None):
        """pack=True (fused training path, bf16, mask descriptors of the full / seq2seq / 1-D families): the encoder runs
        on the valid rows only -- positions after a sample's text [SEP] are invisible to every valid query in those
        families and carry no label, so they contribute nothing to the loss or to any gradient (include/medvill.h,
        'packed rows').  Hidden states are then [sum(vl), H] and the first return value is that packed matrix.

        tail_rows (fused training path; int32 [R], flat logical positions b*L+i of the labelled tokens): only those rows
        and each sample's first row ([CLS] -> pooler -> ITM) of the LAST layer's output are ever consumed -- by the MLM
        head on the labelled rows and by the pooler -- so everything of the last layer that follows its attention (output
        projection, LayerNorm, FFN, LayerNorm: per-row work) runs on those R + B rows only, and so does its backward.
        Exact: the other rows' outputs are unused and their gradients are zero.  The final hidden state is then the compact
        [R + B, H] matrix (labelled rows first, in the order given, then the B first rows)."""
        cfg, dt, adt, dev = self.cfg, self.dt, self.adt, self.device
        fadt, dual = self.fadt, self.dual
        wf = self.wf
        H, A, I, D = cfg.hidden, cfg.heads, cfg.intermediate, cfg.img_hidden
        dh = H // A
        B, T = input_txt.shape
        N = img_feats.shape[1]
        Lq = N + T + 2
        M = B * Lq
        from .data import MaskDesc
        if isinstance(attn_mask, MaskDesc):
            if attn_mask.desc.shape[0] != B or attn_mask.L != Lq:
                raise ValueError(f"mask descriptors (B={attn_mask.desc.shape[0]}, L={attn_mask.L}) do not match B={B}, L={Lq}")
        elif attn_mask.shape[0] != B or attn_mask.shape[-1] != Lq:
            raise ValueError(f"attn_mask shape {tuple(attn_mask.shape)} does not match L = N+T+2 = {Lq}")
        if self.shadow_dirty:
            self.sync_shadow()
        if not self.late_opt_wait:
            self._wait_opt("embeddings_rest")
            self._wait_opt("embeddings")
        f32 = torch.float32
        S = self.S = dict(B=B, T=T, N=N, L=Lq, M=M, cu=None, rowmap=None, inv=None, sel=None, n_lab=0, tq=None)
        if pack:
            if not isinstance(attn_mask, MaskDesc) or not self.is16:
                raise ValueError("pack=True needs mask descriptors (data.MaskDesc) and the bf16 path")
            hd = attn_mask.host_desc()
            if not bool(((hd[:, 0] == 0) | (hd[:, 0] == 1) | (hd[:, 0] == 4)).all()):
                raise ValueError("pack=True: padding is visible to valid queries in the BAR / non-cross families")
            M = int(hd[:, 2].clamp(0, Lq).sum())
            S["cu"], S["rowmap"], S["inv"] = ops.pack_plan(attn_mask.desc.to(dev), B, Lq)
            S["M"] = M
        self._rows_now, self._rows_cap = M, B * Lq
        cu, rowmap = S["cu"], S["rowmap"]
        if tail_rows is not None and cfg.layers >= 1:
            lab = tail_rows.to(dev)
            if S["inv"] is not None and lab.numel() > 0:     # logical flat positions -> packed rows
                lab = S["inv"].index_select(0, lab.to(torch.int64))
            first = cu[:B] if cu is not None else torch.arange(B, device=dev, dtype=torch.int32) * Lq
            S["sel"] = torch.cat([lab.to(torch.int32), first.to(torch.int32)]).contiguous()
            S["n_lab"] = int(tail_rows.numel())
            # The consumed rows are also the only QUERIES the last layer's attention needs (its keys / values are all rows).  In the full /
            # 1-D families a packed sample's mask is "every row sees every row" whatever the order of the rows, so the last layer runs on
            # rows reordered with the consumed ones first and its attention kernels stop after them (mv_tail_perm, qlim): exact, and
            # about 70 % of that layer's attention forward and backward is not computed.
            S["tq"] = None
            if pack and self.tail_queries and cfg.layers >= 1 and ops.get_impl() == 0 and Lq <= 2048 and \
                    bool(((hd[:, 0] == 0) | (hd[:, 0] == 4)).all()):
                S["tq"] = ops.tail_perm(cu, B, Lq, S["sel"], M)          # (perm, newpos, qlim, sel_new)
        pd = S["p_drop"] = float(cfg.dropout) if self.training else 0.0
        self.drop_counter += 1
        dk = S["drop_keys"] = self._drop_keys(cfg.layers)
        i64 = torch.int64
        S["cls_tok"] = cls_tok.to(dev, i64).contiguous().view(-1)
        S["sep_tok"] = sep_tok.to(dev, i64).contiguous().view(-1)
        S["txt"] = input_txt.to(dev, i64).contiguous()
        S["segment"] = segment.to(dev, i64).contiguous()
        S["img_pos"] = img_pos.to(dev, i64).contiguous()
        # region features -> operand encodings (the gradient-product copy is kept: the image projection's weight
        # gradient needs it)
        feats_in = img_feats.to(dev).contiguous().view(B * N, D)
        if feats_in.dtype not in (f32, torch.bfloat16, torch.float16):
            feats_in = feats_in.float()

        def as_dtype(t, want, key):
            if t.dtype == want:
                return t
            o = self._buf(key, tuple(t.shape), want)
            ops.cast(t, o, t.numel())
            return o
        feats = S["feats"] = as_dtype(feats_in, adt, "feats")
        feats_f = as_dtype(feats_in, fadt, "feats_f") if dual else feats
        # packed mask
        W32, Tt = (Lq + 31) // 32, (Lq + 63) // 64
        bits = self._buf("bits", (B, Lq, W32), torch.int32)
        tinfo = self._buf("tinfo", (B, Tt, Tt), torch.uint8)
        if isinstance(attn_mask, MaskDesc):
            ops.mask_build(attn_mask.desc.to(dev), B, Lq, bits, tinfo)       # synthesised on the device, no [B,L,L] input
        else:
            ops.mask_pack(attn_mask.to(dev), bits, tinfo)
        S["bits"], S["tinfo"] = bits, tinfo
        if bits.is_cuda:
            S["bits_ev"] = torch.cuda.Event()
            S["bits_ev"].record(torch.cuda.current_stream())
        # image projection + embeddings
        e = "enc.txt_embeddings."
        # Naming below: `x` / `x_b` etc. are the two encodings of one activation -- the forward operand (f16 on the default
        # 16-bit path) and the copy the backward's gradient products read (bf16); they are the same buffer when the
        # encodings coincide.  The saved-for-backward dictionaries keep the `_b` one.
        xb2 = lambda f, b_: b_ if dual else None          # extra output of the producing kernel
        db_ev = None
        if pd > 0:
            # attention-probability dropout: the mask is a tensor of keep-bits (mv_attn_dropmask) that the forward and both backward
            # kernels select with.  It depends on the keys (and the packed lengths) only: all layers' bits are generated now, on
            # their own stream, under the embedding / first projection kernels
            if self._mask_stream is None:
                self._mask_stream = _shared_stream(dev, "mask")
            ms, main_ = self._mask_stream, torch.cuda.current_stream()
            ms.wait_stream(main_)                     # the previous step's backward has read the buffers; `cu` is ready
            db_ev = []
            with torch.cuda.stream(ms):
                for l in range(cfg.layers):
                    dbl = self._buf(f"dropbits{l}", (ops.dropbits_numel(B, Lq, A),), torch.int32)
                    ops.attn_dropmask(pd, dk[(self.SITE_ATTN, l)], B, Lq, A, dbl, cu=cu)
                    ev = torch.cuda.Event()
                    ev.record(ms)
                    db_ev.append((dbl, ev))
        phase("embed")
        imgproj = self._buf("imgproj", (B * N, H), fadt)
        # first reader of the embeddings parameter range: everything above (row plan, mask words, keep-bits of every layer) did not need
        # the optimizer's first kernel (113 us over the word table) and ran under it
        self._wait_opt("embeddings_rest")
        ops.gemm(feats_f, wf["enc.img_embeddings.img_embeddings.weight"], imgproj, M=B * N, N=H, K=D,
                 bias=self.p["enc.img_embeddings.img_embeddings.bias"], epi=EPI_BIAS)
        self._wait_opt("embeddings")                # the word table
        x, x_b = self._pair("x0", (M, H))
        pre0 = self._buf("pre0", (M, H), f32)
        mean0, rstd0 = self._buf("mean0", (M,), f32), self._buf("rstd0", (M,), f32)
        pdi = S["p_drop_img"] = (pd if cfg.img_dropout < 0 else (float(cfg.img_dropout) if self.training else 0.0))
        if not cfg.img_position:
            S["img_pos"] = None                  # args.img_postion false: no position embedding on the image rows
        ops.embed_fwd(self.fdt, S["cls_tok"], S["txt"], S["segment"], S["img_pos"], S["sep_tok"], imgproj,
                      wf[e + "word_embeddings.weight"], wf[e + "position_embeddings.weight"],
                      wf[e + "token_type_embeddings.weight"], self.p[e + "LayerNorm.weight"], self.p[e + "LayerNorm.bias"],
                      x, pre0, mean0, rstd0, B, N, T, H, cfg.vocab_size, cfg.max_pos, cfg.ln_eps, p_drop=pd,
                      drop_key=dk[(self.SITE_EMB, 0)], rowmap=rowmap, n_rows=M, x0_bf16=xb2(x, x_b), p_drop_img=pdi)
        S["layers"] = []
        # keep_acts False (a forward nothing will back-propagate through: torch.no_grad() / eval scoring): the per-layer activations are
        # not kept -- every layer writes the same scratch set (12 x 0.95 GB -> 0.95 GB at B = 64, L = 512) and S["layers"] is not a
        # valid input of encoder_backward
        keep = S["keep"] = bool(self.keep_acts)
        for l in range(cfg.layers):
            lk = l if keep else "_nk"
            phase(f"layer{l}.attention")
            self._wait_opt(f"layer{l}")
            p = f"enc.encoder.layer.{l}."
            Wqkv, bqkv, _, _ = self.qkv_views(l, fwd=True)
            a_ = {}
            tq = S["tq"] if l == cfg.layers - 1 else None
            if tq is not None:
                # last layer on the reordered rows (consumed rows first within each sample); its output only exists on the consumed rows
                xp, xp_b = self._pair("tail_xperm", (M, H))
                ops.gather_rows(x, H, tq[0], M, H, xp, H)
                if dual:
                    ops.gather_rows(x_b, H, tq[0], M, H, xp_b, H)
                x, x_b = xp, xp_b
            a_["x"] = x_b
            qkv, qkv_b = self._pair(f"qkv{lk}", (M, 3 * H))
            a_["qkv"] = qkv_b
            ops.gemm(x, Wqkv, qkv, M=M, N=3 * H, K=H, bias=bqkv, epi=EPI_BIAS, c3=xb2(qkv, qkv_b))
            ctx, ctx_b = self._pair(f"ctx{lk}", (M, H))
            a_["ctx"] = ctx_b
            lse = a_["lse"] = self._buf(f"lse{lk}", (B, A, Lq), f32)
            a_["dropbits"] = None
            if db_ev is not None:
                a_["dropbits"] = db_ev[l][0]
                torch.cuda.current_stream().wait_event(db_ev[l][1])
            ops.attn_fwd(qkv, bits, tinfo, ctx, lse, B, Lq, A, dh, p_drop=pd, cu=cu,
                         total_rows=M, ctx_bf16=xb2(ctx, ctx_b), dropbits=a_["dropbits"], qlim=tq[2] if tq is not None else None)
            Mr = M                  # rows the rest of this layer runs on
            if l == cfg.layers - 1 and S["sel"] is not None:
                # last layer: only the labelled rows and the first row of every sample are consumed downstream
                sel = S["sel"] if tq is None else tq[3]
                Mr = int(sel.numel())
                ctx_s, ctx_sb = self._pair("tail_ctx", (Mr, H))
                x_s = self._buf("tail_x", (Mr, H), fadt)
                ops.gather_rows(ctx, H, sel, Mr, H, ctx_s, H)
                ops.gather_rows(x, H, sel, Mr, H, x_s, H)
                if dual:
                    ops.gather_rows(ctx_b, H, sel, Mr, H, ctx_sb, H)
                ctx, x = ctx_s, x_s
                a_["ctx_tail"] = ctx_sb
            a_["rows"] = Mr
            M_all, M = M, Mr        # (restored after the layer; nothing follows the last layer)
            pre_dt = self.fadt if (self.fdt == MV_F16 and self.ln_in_16) else f32
            pre1 = a_["pre1"] = self._buf(f"pre1_{lk}" + ("h" if pre_dt != f32 else ""), (M, H), pre_dt)
            ops.gemm(ctx, wf[p + "attention.output.dense.weight"], pre1, M=M, N=H, K=H,
                     bias=self.p[p + "attention.output.dense.bias"], epi=EPI_BIAS_RES, r=x, p_drop=pd,
                     drop_key=dk[(self.SITE_OUT1, l)])
            a1, a1_b = self._pair(f"a{lk}", (M, H))
            a_["a"] = a1_b
            a_["mean1"], a_["rstd1"] = self._buf(f"mean1_{lk}", (M,), f32), self._buf(f"rstd1_{lk}", (M,), f32)
            ops.layernorm_fwd(pre1, self.p[p + "attention.output.LayerNorm.weight"], self.p[p + "attention.output.LayerNorm.bias"],
                              a1, a_["mean1"], a_["rstd1"], M, H, cfg.ln_eps, y_bf16=xb2(a1, a1_b))
            phase(f"layer{l}.ffn")
            act, act_b = self._pair(f"i{lk}", (M, I))
            a_["i"] = act_b
            # the second output is gelu'(z), not z: the derivative shares the forward's exp / reciprocal, and the backward
            # GEMM then only multiplies by it (an elementwise operand: either encoding serves)
            dg = a_["dgelu"] = self._buf(f"dgelu{lk}", (M, I), fadt)
            ops.gemm(a1, wf[p + "intermediate.dense.weight"], act, M=M, N=I, K=H, bias=self.p[p + "intermediate.dense.bias"],
                     epi=EPI_BIAS_GELU_D, c2=dg, c3=xb2(act, act_b))
            pre2 = a_["pre2"] = self._buf(f"pre2_{lk}" + ("h" if pre_dt != f32 else ""), (M, H), pre_dt)
            ops.gemm(act, wf[p + "output.dense.weight"], pre2, M=M, N=H, K=I, bias=self.p[p + "output.dense.bias"],
                     epi=EPI_BIAS_RES, r=a1, p_drop=pd, drop_key=dk[(self.SITE_OUT2, l)])
            x, x_b = self._pair(f"x{l + 1}" if keep else f"x_pp{l & 1}", (M, H))
            a_["mean2"], a_["rstd2"] = self._buf(f"mean2_{lk}", (M,), f32), self._buf(f"rstd2_{lk}", (M,), f32)
            ops.layernorm_fwd(pre2, self.p[p + "output.LayerNorm.weight"], self.p[p + "output.LayerNorm.bias"], x, a_["mean2"],
                              a_["rstd2"], M, H, cfg.ln_eps, y_bf16=xb2(x, x_b))
            M = M_all
            S["layers"].append(a_)
        S["hidden_f"], S["hidden"] = x, x_b
        phase("heads")
        pooled, pooled_b = self._pair("pooled", (B, H))
        S["pooled_f"], S["pooled"] = pooled, pooled_b
        if S["sel"] is not None:
            R_ = S["n_lab"]                               # compact final hidden state: the B first rows follow the labelled ones
            h0_f, S["h0"], S["h0_ld"] = x[R_:], x_b[R_:], H
        elif cu is None:
            h0_f, S["h0"], S["h0_ld"] = x, x_b, Lq * H    # first row of every sample, addressed in place
        else:
            h0_f, S["h0"] = self._pair("h0", (B, H))
            S["h0_ld"] = H
            ops.gather_rows(x, H, cu, B, H, h0_f, H)
            if dual:
                ops.gather_rows(x_b, H, cu, B, H, S["h0"], H)
        self._wait_opt("heads")
        ops.gemm(h0_f, wf["enc.pooler.dense.weight"], pooled, M=B, N=H, K=H, lda=S["h0_ld"],
                 bias=self.p["enc.pooler.dense.bias"], epi=EPI_BIAS_TANH, c3=xb2(pooled, pooled_b))
        return (x.view(B, Lq, H) if (cu is None and S["sel"] is None) else x), pooled

    # ------------------------------------------------------------------ heads (shared pieces)
    def _itm_forward(self):
        S, H = self.S, self.cfg.hidden
        itm = S["itm"] = self._buf("itm", (S["B"], 2), torch.float32)
        ops.gemm(S["pooled_f"], self.wf["itm.linear.weight"], itm, M=S["B"], N=2, K=H, bias=self.p["itm.linear.bias"], epi=EPI_BIAS)
        return itm

    def _mlm_forward(self, xr, xr_b, R, tag, pad=True, logits16=False):
        """xr [R,H] (forward encoding; xr_b = its gradient-product copy) -> logits [R, Vp] f32 (Vp = V rounded up to 8
        when `pad`; pad columns unspecified).  logits16 (fused training step, 16-bit path): the logits in the forward encoding -- they
        only feed the fused cross-entropy, which computes in f32 from them (what autocast-style mixed precision does); halves the
        decoder's output traffic and the loss kernel's input."""
        cfg, H, V = self.cfg, self.cfg.hidden, self.cfg.vocab_size
        f32 = torch.float32
        Vp = (V + 7) // 8 * 8 if pad else V
        hs = self.S[tag] = {}
        hs["xr"], hs["R"], hs["Vp"] = xr_b, R, Vp
        tact = hs["tact"] = self._buf(tag + "tact", (R, H), f32)
        tz = hs["tz"] = self._buf(tag + "tz", (R, H), f32)
        ops.gemm(xr, self.wf["mlm.predictions.transform.dense.weight"], tact, M=R, N=H, K=H,
                 bias=self.p["mlm.predictions.transform.dense.bias"], epi=EPI_BIAS_GELU, c2=tz)
        t, t_b = self._pair(tag + "t", (R, H))
        hs["t"] = t_b
        hs["mean"], hs["rstd"] = self._buf(tag + "mean", (R,), f32), self._buf(tag + "rstd", (R,), f32)
        ops.layernorm_fwd(tact, self.p["mlm.predictions.transform.LayerNorm.weight"],
                          self.p["mlm.predictions.transform.LayerNorm.bias"], t, hs["mean"], hs["rstd"], R, H, cfg.head_ln_eps,
                          y_bf16=t_b if self.dual else None)
        logits = hs["logits"] = torch.empty((R, Vp), dtype=self.fadt if (logits16 and self.is16 and not self.dual) else f32, device=self.device)
        ops.gemm(t, self.wf["enc.txt_embeddings.word_embeddings.weight"], logits, M=R, N=V, K=H, ldc=Vp,
                 bias=self.p["mlm.predictions.bias"], epi=EPI_BIAS)
        return logits

    def _mlm_backward(self, dlogits, tag):
        """dlogits [R, Vp] compute dtype (pad columns zero) -> grads of decoder/bias/transform; returns d(xr) [R,H]."""
        cfg, H, V = self.cfg, self.cfg.hidden, self.cfg.vocab_size
        hs = self.S[tag]
        R, Vp = hs["R"], hs["Vp"]
        g, us = self.g, self.unscale_dev
        # tied decoder: dE = dlogits^T . t  (the embedding scatter-add comes later, in embed_bwd).  360 tiles, no split-K, no
        # workspace: it runs on the side stream, which is idle until the encoder's backward starts; embed_bwd waits for it.
        # The decoder bias gradient (column sums over the 207 MB of dlogits) goes with it: nothing on the main chain needs it.
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = _shared_stream(self.device, "side")
        side = self._side if (self.head_on_side and os.environ.get("MV_SINGLE_STREAM") != "1") else main
        side.wait_stream(main)
        if not self.head_params_on_side:
            ops.colsum(dlogits, Vp, R, V, g["mlm.predictions.bias"], accumulate=True, unscale=us)
        with torch.cuda.stream(side):
            if self.head_params_on_side:
                ops.colsum(dlogits, Vp, R, V, g["mlm.predictions.bias"], accumulate=True, unscale=us)
            self._dW(dlogits, hs["t"], g["enc.txt_embeddings.word_embeddings.weight"], V, H, R, lda=Vp, ldb=H)
            self._dE_ev = torch.cuda.Event()
            self._dE_ev.record(side)
        if side is not main:
            dlogits.record_stream(side)     # a per-step allocation of the caller: not to be reused before the side stream has read it
        dt_ = self._buf(tag + "dt", (R, H), self.adt)
        wb = ops.gemm_workspace_bytes(self.adt, False, True, R, H, V) if (self.is16 and self.head_splitk and R * H <= 4 * 1024 * 1024) else 0
        if wb > 0:
            # dt = dlogits . E contracts over the vocabulary (K = 30,522) into a [R, 768] result: 156 tiles of 128x128 for the
            # ~3,300 labelled rows -- a fifth of the chip's tile slots, 477 K-steps each.  Split-K (partial sums in f32, one
            # reduction, one cast) spreads it over the whole chip.
            dt32 = self._buf(tag + "dt32", (R, H), torch.float32)
            ops.gemm(dlogits, self.w["enc.txt_embeddings.word_embeddings.weight"], dt32, tb=True, M=R, N=H, K=V, lda=Vp, ldb=H,
                     splitk=0, ws=self._gemm_workspace(wb // 4))
            ops.cast(dt32, dt_, R * H)
        else:
            ops.gemm(dlogits, self.w["enc.txt_embeddings.word_embeddings.weight"], dt_, tb=True, M=R, N=H, K=V, lda=Vp, ldb=H)
        dtact = self._buf(tag + "dtact", (R, H), self.adt)
        ops.layernorm_bwd(dt_, hs["tact"], hs["mean"], hs["rstd"], self.p["mlm.predictions.transform.LayerNorm.weight"], dtact,
                          g["mlm.predictions.transform.LayerNorm.weight"], g["mlm.predictions.transform.LayerNorm.bias"], None, R, H,
                          unscale=us)
        # dtz = dtact * gelu'(tz); tz is f32, dtact compute dtype -> bring tz to compute dtype first
        if self.is16:
            tzc = self._buf(tag + "tzc", (R, H), self.adt)
            ops.cast(hs["tz"], tzc, R * H)
        else:
            tzc = hs["tz"]
        dtz = self._buf(tag + "dtz", (R, H), self.adt)
        ops.dact(0, dtact, tzc, dtz, R * H)
        pside = side if self.head_params_on_side else main
        if pside is not main:          # the transform's bias / weight gradients: parameter gradients, off the main chain like the encoder's
            pside.wait_stream(main)
        with torch.cuda.stream(pside):
            ops.colsum(dtz, H, R, H, g["mlm.predictions.transform.dense.bias"], accumulate=True, unscale=us)
            self._dW(dtz, hs["xr"], g["mlm.predictions.transform.dense.weight"], H, H, R, lda=H, ldb=H)
        dxr = self._buf(tag + "dxr", (R, H), self.adt)
        ops.gemm(dtz, self.w["mlm.predictions.transform.dense.weight"], dxr, tb=True, M=R, N=H, K=H)
        return dxr

    def _itm_backward(self, ditm8):
        """ditm8 [B,8] compute dtype (cols 2..7 zero). Adds the pooler-path gradient into dhidden rows b*L."""
        S, H = self.S, self.cfg.hidden
        B, Lq = S["B"], S["L"]
        g, us = self.g, self.unscale_dev
        ops.colsum(ditm8, 8, B, 2, g["itm.linear.bias"], accumulate=True, unscale=us)
        self._dW(ditm8, S["pooled"], g["itm.linear.weight"], 2, H, B, lda=8, ldb=H)
        dpool = self._buf("dpool", (B, H), self.adt)
        ops.gemm(ditm8, self.w["itm.linear.weight"], dpool, tb=True, M=B, N=H, K=2, lda=8, ldb=H)
        dpre = self._buf("dpoolpre", (B, H), self.adt)
        ops.dact(1, dpool, S["pooled"], dpre, B * H)
        ops.colsum(dpre, H, B, H, g["enc.pooler.dense.bias"], accumulate=True, unscale=us)
        self._dW(dpre, S["h0"], g["enc.pooler.dense.weight"], H, H, B, lda=H, ldb=S["h0_ld"])
        dh0 = self._buf("dh0", (B, H), self.adt)
        ops.gemm(dpre, self.w["enc.pooler.dense.weight"], dh0, tb=True, M=B, N=H, K=H)
        if S["sel"] is not None:                         # compact final hidden state: rows n_lab .. n_lab+B-1 are the first rows
            S["dhidden"][S["n_lab"]:].copy_(dh0)
            return
        if S["cu"] is None:
            rows0 = self._buf("rows0", (B,), torch.int32)
            rows0.copy_(torch.arange(B, device=self.device, dtype=torch.int32) * Lq)
        else:
            rows0 = S["cu"]                              # packed: sample b starts at row cu[b]
        ops.scatter_rows(dh0, H, rows0, B, H, S["dhidden"], H, accumulate=True)

    # ------------------------------------------------------------------ drop-in heads: full logits
    def heads_full(self):
        """(mlm [B,L,V] f32, itm [B,2] f32) over ALL positions -- the CXRBERT.forward contract."""
        S, V = self.S, self.cfg.vocab_size
        logits = self._mlm_forward(S["hidden_f"], S["hidden"], S["M"], "hf_", pad=False)
        itm = self._itm_forward()
        return logits.view(S["B"], S["L"], V), itm.clone()

    def heads_full_backward(self, dmlm, ditm):
        S, H, V = self.S, self.cfg.hidden, self.cfg.vocab_size
        M, B = S["M"], S["B"]
        Vp = (V + 7) // 8 * 8
        self.S.setdefault("hf_", {})["Vp"] = Vp
        self.ensure_grad()
        dhid = S["dhidden"] = self._buf("dhidden", (M, H), self.adt)
        ls = self.loss_scale_dev                # f16 gradients: the incoming f32 gradients enter the chain multiplied by S
        if dmlm is not None:
            dl = torch.empty((M, Vp), dtype=self.adt, device=self.device)
            d32 = dmlm.contiguous().view(M, V).float()
            ops.cast2d(d32 if ls is None else d32 * ls, V, dl, Vp, M, V)
            dxr = self._mlm_backward(dl, "hf_")
            ops.cast(dxr, dhid, M * H)
        else:
            dhid.zero_()
        if ditm is not None:
            d8 = self._buf("ditm8", (B, 8), self.adt)
            i32_ = ditm.contiguous().float()
            ops.cast2d(i32_ if ls is None else i32_ * ls, 2, d8, 8, B, 2)
            self._itm_backward(d8)

    # ------------------------------------------------------------------ training heads: labelled rows only
    def heads_train(self, label_rows, label_ids, is_aligned, mlm_scale_dev=None, mlm_scale=None, itm_scale=None,
                    compute_grad=True, itm_scale_dev=None):
        """Fused MLM (labelled rows only) + ITM losses, metrics and their gradients into dhidden.
        label_rows int32 [R]: flat indices b*L+i of positions with a label; label_ids int32 [R].
        Returns stats f32[6] = [mlm_nll_sum, n_labelled, mlm_correct, itm_nll_sum, B, itm_correct]."""
        S, H, V = self.S, self.cfg.hidden, self.cfg.vocab_size
        M, B = S["M"], S["B"]
        R = int(label_rows.numel())
        compact = S["sel"] is not None                   # final hidden state = [labelled rows | first rows] (encoder_forward)
        if compact and R != S["n_lab"]:
            raise ValueError("heads_train: label_rows differ from the tail_rows the encoder ran with")
        if not compact and S["inv"] is not None and R > 0:               # logical flat positions -> packed rows
            label_rows = S["inv"].index_select(0, label_rows.to(torch.int64))
        stats = torch.zeros(6, dtype=torch.float32, device=self.device)
        if compute_grad:
            self.ensure_grad()
            if compact:
                dhid = S["dhidden"] = self._buf("dhidden_tail", (R + B, H), self.adt)
            else:
                dhid = S["dhidden"] = self._buf("dhidden", (M, H), self.adt)
                dhid.zero_()
        # The ITM head (pooler -> 2-way classifier, its loss and backward: a dozen latency-bound launches on [B, H] matrices, ~100 us in a
        # row) runs on the side stream, which is idle until the encoder's backward, under the MLM head's decoder GEMM on the main stream.
        # The two heads touch disjoint rows of dhidden, disjoint statistics and disjoint parameter gradients.
        main = torch.cuda.current_stream() if self.device.type == "cuda" else None
        side = main
        if main is not None and self.itm_on_side and os.environ.get("MV_SINGLE_STREAM") != "1":
            if self._side is None:
                self._side = _shared_stream(self.device, "side")
            side = self._side
        if side is not main:
            side.wait_stream(main)
            stats.record_stream(side)
            with torch.cuda.stream(side):
                self._itm_head(stats, is_aligned, B, itm_scale, itm_scale_dev, compute_grad)
                itm_ev = torch.cuda.Event()       # main joins the ITM head alone, not what _mlm_backward queues behind it on this stream
                itm_ev.record(side)
        if R > 0:
            if compact:
                xr, xr_b = S["hidden_f"][:R], S["hidden"][:R]
            else:
                xr, xr_b = self._pair("ht_xr", (R, H))
                ops.gather_rows(S["hidden_f"], H, label_rows, R, H, xr, H)
                if self.dual:
                    ops.gather_rows(S["hidden"], H, label_rows, R, H, xr_b, H)
            logits = self._mlm_forward(xr, xr_b, R, "ht_", logits16=self.logits_16)
            Vp = logits.shape[1]
            dl = torch.empty((R, Vp), dtype=self.adt, device=self.device) if compute_grad else None
            ops.ce_fwd_bwd(logits, Vp, label_ids, R, V, stats[0:3], dl, Vp, grad_scale_dev=mlm_scale_dev,
                           grad_scale=(mlm_scale if mlm_scale is not None else 1.0 / R), loss_scale_dev=self.loss_scale_dev)
            if compute_grad:
                dxr = self._mlm_backward(dl, "ht_")
                if compact:
                    dhid[:R].copy_(dxr)
                else:
                    ops.scatter_rows(dxr, H, label_rows, R, H, dhid, H, accumulate=False)
        if side is main:
            self._itm_head(stats, is_aligned, B, itm_scale, itm_scale_dev, compute_grad)
        else:
            main.wait_event(itm_ev)     # (the tied decoder's dW, the decoder-bias sums and the transform's dW stay off the main chain: _dE_ev)
        return stats

    def _itm_head(self, stats, is_aligned, B, itm_scale, itm_scale_dev, compute_grad):
        itm = self._itm_forward()
        d8 = self._buf("ditm8", (B, 8), self.adt) if compute_grad else None
        ops.ce_fwd_bwd(itm, 2, is_aligned, B, 2, stats[3:6], d8, 8, grad_scale_dev=itm_scale_dev,
                       grad_scale=(itm_scale if itm_scale is not None else 1.0 / B), loss_scale_dev=self.loss_scale_dev)
        if compute_grad:
            self._itm_backward(d8)

    # ------------------------------------------------------------------ encoder backward
    def encoder_backward(self, bucket_hook=None):
        """Consumes S['dhidden'] (grad w.r.t. the last hidden states); fills the flat gradient.

        Two HIP streams: the MAIN stream carries the dependency chain of the backward (LayerNorm backward -> dX GEMMs
        -> attention backward ...); every weight / bias gradient (dW = dy^T.x, column sums), which nothing downstream
        needs before the optimizer, is enqueued on a SIDE stream behind an event, so those MFMA-bound GEMMs fill the
        CUs that the VALU- or HBM-bound kernels of the main chain leave idle.  Per-layer gradient scratch (no buffer is
        rewritten while the side stream may still read it; ~6.6 GB at B=64, L=512).

        bucket_hook(name, event) is called when the gradients of a contiguous parameter range are final on the side
        stream ('heads', 'layer<l>', 'embeddings') so a data-parallel driver can start its all-reduce."""
        cfg, S = self.cfg, self.S
        if not S.get("keep", True):
            raise RuntimeError("encoder_backward after a forward that kept no activations (Engine.keep_acts = False / torch.no_grad())")
        H, A, I, D = cfg.hidden, cfg.heads, cfg.intermediate, cfg.img_hidden
        dh = H // A
        B, Lq, M, N, T = S["B"], S["L"], S["M"], S["N"], S["T"]
        adt, g, us = self.adt, self.g, self.unscale_dev
        dy = S["dhidden"]
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = _shared_stream(self.device, "side")
        side = main if os.environ.get("MV_SINGLE_STREAM") == "1" else self._side     # measurement switch
        side.wait_stream(main)              # the heads' gradients and the zeroed flat gradient are ordered before us

        def fork():
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)

        def side_done():
            ev = torch.cuda.Event()
            ev.record(side)
            return ev

        if bucket_hook:
            # the MLM head's parameter gradients were written on the side stream (heads_train joins the ITM head only)
            bucket_hook("heads", side_done() if side is not main else None)
        use_w2t = self.dz_nt and self.is16
        mfma_on = self.fused_colsum and self.is16 and ops.get_impl() == 0       # partial column sums exist in the MFMA kernels only
        if use_w2t:
            if self._w2t_stale or self._w2t is None:
                self.refresh_w2t(on_side=False)
            main.wait_event(self._w2t_ev)
        pd, dk = S["p_drop"], S["drop_keys"]
        dctx = self._buf("bw_dctx", (M, H), adt)
        delta = self._buf("bw_delta", (B, A, Lq), torch.float32)
        da = self._buf("bw_da", (M, H), adt)
        dxb = [self._buf("bw_dx0", (M, H), adt), self._buf("bw_dx1", (M, H), adt)]
        M_all = M
        for l in reversed(range(cfg.layers)):
            p = f"enc.encoder.layer.{l}."
            a_ = S["layers"][l]
            phase(f"backward.layer{l}")
            Wqkv, _, gWqkv, gbqkv = self.qkv_views(l)
            # the last layer's per-row part ran on the consumed rows only (encoder_forward, tail_rows): so does its backward
            tail = a_["rows"] != M_all
            M = a_["rows"]
            # with dropout the projection branch sees dpre * mask / (1-p) while the residual branch sees dpre itself
            dpre2 = self._buf(f"bw_dpre2_{l}", (M, H), adt)
            dprd2 = self._buf(f"bw_dprd2_{l}", (M, H), adt) if pd > 0 else None
            dpre1 = self._buf(f"bw_dpre1_{l}", (M, H), adt)
            dprd1 = self._buf(f"bw_dprd1_{l}", (M, H), adt) if pd > 0 else None
            dz = self._buf(f"bw_dz_{l}", (M, I), adt)
            dqkv = self._buf(f"bw_dqkv_{l}", (M_all, 3 * H), adt)
            # LN2 backward (+ bias grad of output.dense)
            ops.layernorm_bwd(dy, a_["pre2"], a_["mean2"], a_["rstd2"], self.p[p + "output.LayerNorm.weight"], dpre2,
                              g[p + "output.LayerNorm.weight"], g[p + "output.LayerNorm.bias"], g[p + "output.dense.bias"], M, H,
                              dx_drop=dprd2, p_drop=pd, drop_key=dk[(self.SITE_OUT2, l)], unscale=us)
            dproj2 = dprd2 if dprd2 is not None else dpre2
            fork()
            with torch.cuda.stream(side):
                self._dW(dproj2, a_["i"], g[p + "output.dense.weight"], H, I, M, lda=H, ldb=I)
            # FFN-up bias gradient without a second pass over dz: the dz GEMM leaves partial column sums (one row per 128-row tile
            # half) that one small kernel folds (MV_FUSED_COLSUM=0: the column-sum kernel).  The same for dqkv from the attention
            # backward was built and measured slower (profiles/r03_notes.txt): +80 us on the two kernels for 11 us saved.
            dz_part = None
            # (mv_gemm takes the 256x256 MFMA kernel for wide outputs that fill the chip: the same test as in mv_gemm.hip)
            if use_w2t and mfma_on and M >= 256 and I >= 1024 and I % 256 == 0 and ((M + 255) // 256) * ((I + 127) // 128) >= 128:
                dz_part = self._buf(f"bw_dzpart_{l}", (2 * ((M + 255) // 256), I), torch.float32)
            if use_w2t:
                ops.gemm(dproj2, self._w2t[l], dz, M=M, N=I, K=H, epi=EPI_MUL, r=a_["dgelu"], colsum_part=dz_part)
            else:
                ops.gemm(dproj2, self.w[p + "output.dense.weight"], dz, tb=True, M=M, N=I, K=H, epi=EPI_MUL, r=a_["dgelu"])
            fork()
            with torch.cuda.stream(side):
                if dz_part is not None:
                    ops.colsum_partials(dz_part, dz_part.shape[0], I, I, g[p + "intermediate.dense.bias"], unscale=us)
                else:
                    ops.colsum(dz, I, M, I, g[p + "intermediate.dense.bias"], accumulate=True, unscale=us)
                self._dW(dz, a_["a"], g[p + "intermediate.dense.weight"], I, H, M, lda=I, ldb=H)
            ops.gemm(dz, self.w[p + "intermediate.dense.weight"], da, tb=True, M=M, N=H, K=I, epi=EPI_RES, r=dpre2)
            # LN1 backward (+ bias grad of attention.output.dense)
            ops.layernorm_bwd(da, a_["pre1"], a_["mean1"], a_["rstd1"], self.p[p + "attention.output.LayerNorm.weight"], dpre1,
                              g[p + "attention.output.LayerNorm.weight"], g[p + "attention.output.LayerNorm.bias"],
                              g[p + "attention.output.dense.bias"], M, H, dx_drop=dprd1, p_drop=pd,
                              drop_key=dk[(self.SITE_OUT1, l)], unscale=us)
            dproj1 = dprd1 if dprd1 is not None else dpre1
            fork()
            with torch.cuda.stream(side):
                self._dW(dproj1, a_["ctx_tail"] if tail else a_["ctx"], g[p + "attention.output.dense.weight"], H, H, M, lda=H, ldb=H)
            if tail:
                # d(ctx) exists on the consumed rows only: everywhere else it is exactly zero
                dctx_s = self._buf("bw_dctx_tail", (M, H), adt)
                ops.gemm(dproj1, self.w[p + "attention.output.dense.weight"], dctx_s, tb=True, M=M, N=H, K=H)
                dctx.zero_()
                ops.scatter_rows(dctx_s, H, S["sel"] if S["tq"] is None else S["tq"][3], M, H, dctx, H, accumulate=False)
            else:
                ops.gemm(dproj1, self.w[p + "attention.output.dense.weight"], dctx, tb=True, M=M, N=H, K=H)
            M = M_all
            tq = S["tq"] if tail else None
            ops.attn_bwd(a_["qkv"], a_["ctx"], dctx, a_["lse"], S["bits"], S["tinfo"], dqkv, delta, B, Lq, A, dh, cu=S["cu"],
                         total_rows=M, p_drop=pd, dropbits=a_["dropbits"], qlim=tq[2] if tq is not None else None)
            fork()
            with torch.cuda.stream(side):
                ops.colsum(dqkv, 3 * H, M, 3 * H, gbqkv, accumulate=True, unscale=us)
                self._dW(dqkv, a_["x"], gWqkv, 3 * H, H, M, lda=3 * H, ldb=H)
                ev_layer = side_done()
            dx = dxb[l & 1]          # never the buffer dy currently lives in
            if tail:
                # the residual branch's gradient (dpre1) also lives on the consumed rows only
                ops.gemm(dqkv, Wqkv, dx, tb=True, M=M, N=H, K=3 * H)
                ops.scatter_rows(dpre1, H, S["sel"] if tq is None else tq[3], a_["rows"], H, dx, H, accumulate=True)
                if tq is not None:         # back to the row order of the layers below
                    dxu = self._buf("bw_dx_unperm", (M, H), adt)
                    ops.scatter_rows(dx, H, tq[0], M, H, dxu, H, accumulate=False)
                    dx = dxu
            else:
                ops.gemm(dqkv, Wqkv, dx, tb=True, M=M, N=H, K=3 * H, epi=EPI_RES, r=dpre1)
            dy = dx
            if bucket_hook:
                # LayerNorm / bias gradients of the layer were written on the main stream, the weights on the side stream
                bucket_hook(f"layer{l}", ev_layer)
        e = "enc.txt_embeddings."
        phase("backward.embed")
        dimg = self._buf("bw_dimg", (B * N, H), adt)
        if getattr(self, "_dE_ev", None) is not None:
            main.wait_event(self._dE_ev)      # the decoder's word-embedding gradient (written, not accumulated) is in place
            self._dE_ev = None
        ops.embed_bwd(self.dt, dy, self._ws["pre0"][:M * H].view(M, H), self._ws["mean0"][:M], self._ws["rstd0"][:M],
                      self.p[e + "LayerNorm.weight"], S["cls_tok"], S["txt"], S["segment"], S["img_pos"], S["sep_tok"],
                      g[e + "word_embeddings.weight"], g[e + "position_embeddings.weight"], g[e + "token_type_embeddings.weight"],
                      g[e + "LayerNorm.weight"], g[e + "LayerNorm.bias"], dimg, B, N, T, H, cfg.vocab_size, cfg.max_pos,
                      pad_token_id=0, p_drop=pd, drop_key=dk[(self.SITE_EMB, 0)], rowmap=S["rowmap"], n_rows=M, unscale=us,
                      p_drop_img=S["p_drop_img"])
        main.wait_stream(side)              # every weight gradient is final; the split-K workspace is ours again
        if N > 0:
            ops.colsum(dimg, H, B * N, H, g["enc.img_embeddings.img_embeddings.bias"], accumulate=True, unscale=us)
            self._dW(dimg, S["feats"], g["enc.img_embeddings.img_embeddings.weight"], H, D, B * N, lda=H, ldb=D)
        if bucket_hook:
            bucket_hook("embeddings", None)
        phase(None)

    # ------------------------------------------------------------------ optimizer
    def check_overflow(self):
        """f16-gradient path, after the backward (and the gradient all-reduce): count the non-finite elements of the flat
        gradient and let the device-side scaler decide -- skip flag for the optimizer, loss scale of the next step.  No host
        synchronisation; under data parallelism every rank sees the same all-reduced gradient and decides alike."""
        if self.scaler is None:
            return
        phase("overflow_check")
        # (counting everything but the embeddings bucket on the side stream under the embedding backward was measured: no change)
        ops.count_nonfinite(self.flat_g, self.scaler[6:7])
        ops.scaler_update(self.scaler, growth_interval=self.scale_growth_interval)

    def adamw_step(self, step, lr=1e-5, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True, grad_scale=1.0,
                   overlap=False, use_scaler=False):
        """HF AdamW over the flat buffer; also refreshes the 16-bit shadows.  use_scaler (f16-gradient path, after
        `check_overflow`): the kernels read the skip flag and the step count t from the device-side scaler state, so an
        overflowed step changes nothing and does not advance the bias correction.

        overlap=False: one kernel on the current stream.  overlap=True: one kernel per contiguous parameter range in FORWARD
        order (embeddings, layer 0 .. L-1, heads) on the side stream, an event after each; the next `encoder_forward` makes the
        current stream wait for a range's event right before it reads that range, so the optimizer (HBM-bound, MFMA idle) runs
        under the next step's first layers (MFMA-bound).  Anything else that reads parameters on the current stream must call
        `wait_optimizer()` first (CXRBERT.state_dict / save / load do)."""
        self.ensure_opt()
        phase("optimizer")
        scaler_state = self.scaler if (use_scaler and self.scaler is not None) else None
        if not (overlap and self.device.type == "cuda" and os.environ.get("MV_SINGLE_STREAM") != "1"):
            self.wait_optimizer()
            ops.adamw_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.shadow, self.n_flat, lr, betas[0], betas[1], eps,
                           weight_decay, step, correct_bias, grad_scale, shadow_f16=self.shadow_f, scaler_state=scaler_state)
            self.shadow_dirty = False
            self.refresh_w2t()
            phase(None)
            return
        from .dist import bucket_ranges
        cfg = self.cfg
        ranges = bucket_ranges(self.layout, self.n_flat, cfg.layers)
        main, side = torch.cuda.current_stream(), self.side_stream()
        side.wait_stream(main)                  # gradients (and their all-reduce, which the caller finished) are final
        evs = {}
        use_w2t = self.dz_nt and self.is16
        if use_w2t and self._w2t is None:
            self._w2t = [torch.empty((cfg.intermediate, cfg.hidden), dtype=self.adt, device=self.device) for _ in range(cfg.layers)]
        # the embeddings range in two kernels: the small tables / LayerNorm / image projection first (the next forward's first GEMM reads the
        # image projection), then the word table (110 us of HBM traffic that this GEMM no longer waits for)
        word_end = self.layout["enc.txt_embeddings.position_embeddings.weight"][0]
        ranges["embeddings_rest"] = (word_end, ranges["embeddings"][1])
        ranges["embeddings"] = (ranges["embeddings"][0], word_end)
        with torch.cuda.stream(side):
            for name in ["embeddings_rest", "embeddings"] + [f"layer{l}" for l in range(cfg.layers)] + ["heads"]:
                s_, e_ = ranges[name]
                sl = slice(s_, e_)
                ops.adamw_step(self.flat_p[sl], self.flat_g[sl], self.flat_m[sl], self.flat_v[sl],
                               None if self.shadow is None else self.shadow[sl], e_ - s_, lr, betas[0], betas[1], eps,
                               weight_decay, step, correct_bias, grad_scale,
                               shadow_f16=None if self.shadow_f is None else self.shadow_f[sl], scaler_state=scaler_state)
                if use_w2t and name.startswith("layer"):
                    l = int(name[5:])
                    ops.transpose(self.w[f"enc.encoder.layer.{l}.output.dense.weight"], self._w2t[l], cfg.hidden, cfg.intermediate)
                ev = torch.cuda.Event()
                ev.record(side)
                evs[name] = ev
        self._opt_ev = evs
        self._w2t_ev, self._w2t_stale = evs["heads"], False
        self.shadow_dirty = False
        phase(None)

    def _wait_opt(self, name):
        """Current stream waits for the optimizer's kernel over parameter range `name` (overlapped AdamW), once."""
        if self._opt_ev:
            ev = self._opt_ev.pop(name, None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)

    def wait_optimizer(self):
        """Current stream waits for every outstanding optimizer kernel (before parameters are read outside the engine)."""
        if self._opt_ev:
            for ev in self._opt_ev.values():
                torch.cuda.current_stream().wait_event(ev)
            self._opt_ev = None
