"""A/B of training-step variants in ONE process on ONE device (box-to-box spread of the same build is +-2 %): interleaved rounds,
median ms/step per variant.  usage: python profiles/tools/ab_step.py pack"""
import os
import statistics
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import medvill_amd as mv

dev = torch.device("cuda", 0)
cfg = mv.ModelConfig()
torch.manual_seed(1234)
model = mv.CXRBERT(cfg, None, dtype=torch.bfloat16, device=dev)
model.train()
step = mv.TrainStep(model, lr=1e-5, overlap_optimizer=True)      # as bench.py and CXRBERT_Trainer build it
B, N, S = 64, 36, 473
batches = [mv.data.synthetic_batch(cfg.vocab_size, B, N, S, "full", seed=1234 + 1000 * i, device=dev) for i in range(4)]


def timed(n=8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        step(batches[i % 4])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


VARIANTS = {
    "pack": [("padded", lambda: setattr(step, "pack_rows", False)), ("packed", lambda: setattr(step, "pack_rows", True))],
    "dwsplit": [("dW split-K auto (fills 256 CUs)", lambda: setattr(model.engine, "dw_splitk", 0)),
                ("dW split-K <= 4", lambda: setattr(model.engine, "dw_splitk", 4)),
                ("dW split-K <= 3", lambda: setattr(model.engine, "dw_splitk", 3)),
                ("dW split-K <= 2", lambda: setattr(model.engine, "dw_splitk", 2))],
    "lnin": [("LayerNorm inputs fp32", lambda: setattr(model.engine, "ln_in_16", False)),
             ("LayerNorm inputs f16", lambda: setattr(model.engine, "ln_in_16", True))],
    "dznt": [("dz on the 128x128 kernel over W2 as stored", lambda: setattr(model.engine, "dz_nt", False)),
             ("dz on the 256-row kernel over a W2^T copy", lambda: setattr(model.engine, "dz_nt", True))],
    "dwo": [("small dW: at most 16 split-K slabs", lambda: setattr(model.engine, "dw_splitk", 16)),
            ("small dW: up to 32 slabs (auto)", lambda: setattr(model.engine, "dw_splitk", 0))],
    "headk": [("decoder input gradient: one pass over K = 30,522", lambda: setattr(model.engine, "head_splitk", False)),
              ("decoder input gradient: split-K", lambda: setattr(model.engine, "head_splitk", True))],
    "dEside": [("decoder weight gradient on the main stream", lambda: setattr(model.engine, "head_on_side", False)),
               ("decoder weight gradient on the side stream", lambda: setattr(model.engine, "head_on_side", True))],
    "optov": [("AdamW on the main stream after the backward", lambda: setattr(step, "overlap_optimizer", False)),
              ("AdamW on the side stream under the next forward", lambda: setattr(step, "overlap_optimizer", True))],
    "fcs": [("FFN-up bias gradient: column-sum kernel over dz", lambda: setattr(model.engine, "fused_colsum", False)),
            ("FFN-up bias gradient: partial sums from the dz GEMM", lambda: setattr(model.engine, "fused_colsum", True))],
    "planes": [("attention-dropout mask generator: 16 bits per uniform", lambda: mv.hip_ops.set_attn_planes(16)),
               ("12 bits", lambda: mv.hip_ops.set_attn_planes(12)), ("8 bits", lambda: mv.hip_ops.set_attn_planes(8))],
    "drop": [("dropout off (eval-mode forward inside the training step)", lambda: model.eval()), ("dropout 0.1", lambda: model.train())],
    "ln": [("LayerNorm backward: one row at a time, 4 waves x 1024 blocks", lambda: mv.hip_ops.set_rowops_variant(1)),
           ("prefetch, 8 waves x 512 blocks", lambda: mv.hip_ops.set_rowops_variant(0)),
           ("prefetch, 16 waves x 256 blocks", lambda: mv.hip_ops.set_rowops_variant(3)),
           ("prefetch, 4 waves x 512 blocks", lambda: mv.hip_ops.set_rowops_variant(2 | (512 << 8)))],
    "optwait": [("forward waits for the optimizer's first kernel before its preparation kernels", lambda: setattr(model.engine, "late_opt_wait", False)),
                ("... right before the first reader of the embeddings range", lambda: setattr(model.engine, "late_opt_wait", True))],
    "hps": [("MLM head's bias / transform parameter gradients on the main chain", lambda: setattr(model.engine, "head_params_on_side", False)),
            ("... on the side stream", lambda: setattr(model.engine, "head_params_on_side", True))],
    "itm": [("ITM head after the MLM head on the main stream", lambda: setattr(model.engine, "itm_on_side", False)),
            ("ITM head on the side stream under the decoder GEMM", lambda: setattr(model.engine, "itm_on_side", True))],
    "logits": [("MLM logits f32", lambda: setattr(model.engine, "logits_16", False)),
               ("MLM logits in the forward encoding (f16)", lambda: setattr(model.engine, "logits_16", True))],
    "tq": [("last layer's attention: every row a query", lambda: setattr(model.engine, "tail_queries", False)),
           ("consumed rows only as queries (reordered rows, qlim)", lambda: setattr(model.engine, "tail_queries", True))],
    "rounds": [("ring tile height: rounds 1-4's choice (128x128 kernel for 768-column outputs)", lambda: mv.hip_ops.set_gemm_rounds(0)),
               ("ring tile height chosen for whole rounds of CUs (320 x 256 tiles: one round at ~25k rows)", lambda: mv.hip_ops.set_gemm_rounds(1))],
    "order": [("attention blocks: row block slowest (rounds 3-4)", lambda: mv.hip_ops.set_attn_order(0)),
              ("attention blocks: a pair's row blocks adjacent on one XCD", lambda: mv.hip_ops.set_attn_order(1))],
    "tail": [("last layer on all rows", lambda: setattr(step, "tail_rows", False)), ("last layer on consumed rows", lambda: setattr(step, "tail_rows", True))],
}
def _side_plain():
    from medvill_amd import engine as E
    st = E._SHARED_STREAMS.get(("plain-side",))
    if st is None:
        st = E._SHARED_STREAMS[("plain-side",)] = torch.cuda.Stream(device=dev)
    E._SHARED_STREAMS[(str(dev), "side")] = st
    model.engine._side = st
    mv.hip_ops.set_persistent_cus(0)


def _side_masked(n, grid=None):
    def f():
        from medvill_amd import engine as E
        key = ("masked-side", n)
        st = E._SHARED_STREAMS.get(key)
        if st is None:
            st = E._SHARED_STREAMS[key] = mv.hip_ops.stream_with_cus(n, dev, first=256 - n)
        E._SHARED_STREAMS[(str(dev), "side")] = st
        model.engine._side = st
        mv.hip_ops.set_persistent_cus(n if grid is None else grid)
    return f


VARIANTS["pcus"] = [(f"persistent dW kernels: {n} blocks" if n else "persistent dW kernels: one block per CU (256)",
                     (lambda n=n: mv.hip_ops.set_persistent_cus(n))) for n in (0, 224, 192, 160, 128)]
VARIANTS["cumask"] = [("side stream unmasked, 256 persistent blocks", _side_plain)] + \
    [(f"side stream on {n} CUs ({n // 8} per XCD), {n} persistent blocks", _side_masked(n)) for n in (192, 128, 96, 64)]
_orig_dW = model.engine._dW


def _dW_on(force, nj):
    def f():
        def dW(*a, **k):
            mv.hip_ops.set_gemm_variant(force, nj)
            try:
                return _orig_dW(*a, **k)
            finally:
                mv.hip_ops.set_gemm_variant(0, 0)
        model.engine._dW = dW if force else _orig_dW
    return f


VARIANTS["dwkernel"] = [("weight gradients: persistent 256x256 kernel (default)", _dW_on(0, 0)),
                        ("weight gradients: 128x128 kernel, three blocks per CU (room for the main chain's blocks)", _dW_on(1, 0)),
                        ("weight gradients: 256x256 ring kernel, one unit per block", _dW_on(2, 14)),
                        ("weight gradients: 256x128 tiles, two blocks per CU", _dW_on(2, 2))]
which = sys.argv[1] if len(sys.argv) > 1 else "pack"
arms = VARIANTS[which]
for _ in range(3):
    step(batches[0])
res = {n: [] for n, _ in arms}
for r in range(5):
    for n, f in arms:
        f()
        step(batches[0])
        res[n].append(timed())
for n, v in res.items():
    print(f"{n:36s} median {statistics.median(v):.2f} ms  (min {min(v):.2f}, max {max(v):.2f})")
