"""The step's heaviest kernels ALONE, on seeded operands -- ONE definition shared by bench.py (HIP-event timing inside the bench
line's `roofline`) and by the rocprofv3 passes (profiles/tools/r03_profile_all.sh), so that the line's numbers can be recomputed
from the committed profile summaries: both time the same kernel on the same data.

cases
  dw     the weight-gradient GEMM (persistent 256x256 kernel, TN form, split-K + reduction): dW1 = dz^T . a, [3072 x 768] over 25,483
         packed rows, f16 operands, f32 result un-scaled by 1/S -- the kernel symbol with the largest share of the step (46 launches)
  ffn1   the FFN-up projection (256x256 ring kernel, NT form): [32768, 768] x [3072, 768]^T + bias, GELU and GELU' epilogue, two f16 outputs
  attn   attention forward + backward (dQ, dK/dV) at B = 64, A = 12, L = 512, ragged packed rows, dropout 0.1 from keep-bits
  qkv wo ffn2 dz da dctx dxqkv dw2 dwqkv dwo   the other GEMM calls of an encoder layer (GEMM_FAMILY below)
usage: python3 profiles/tools/dominant.py <case> [reps] [warm-up]   |   python3 profiles/tools/dominant.py all [reps] [warm-up] [out.json]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

H, I = 768, 3072
ROWS_PACKED, ROWS_PADDED = 25483, 32768


def _rnd(shape, std, seed, dev, dtype=torch.float16):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=dev) * std).to(dtype)


def make_case(name, dev="cuda"):
    """-> (launch callable, dict(kernel, flop, bytes, launches_per_call))"""
    from medvill_amd import hip_ops as ops
    from medvill_amd._lib import EPI_BIAS_GELU_D
    if name == "dw":
        M = ROWS_PACKED
        dz, a = _rnd((M, I), 0.1, 11, dev), _rnd((M, H), 1.0, 12, dev)          # loss-scaled dz, LayerNorm output
        g = torch.zeros((I, H), device=dev)
        ws = torch.empty(16 * I * H, device=dev)
        alpha = torch.tensor([1.0 / 32768.0], device=dev)
        fn = lambda: ops.gemm(dz, a, g, ta=True, tb=True, M=I, N=H, K=M, lda=I, ldb=H, splitk=0, ws=ws, alpha=alpha)
        return fn, dict(kernel=f"gemm_pring_kernel<TN, 256x256x64, f16 operands> dW1 = dz^T.a: {I}x{H} over {M} rows, 7 split-K slabs (+ splitk_reduce_kernel)",
                        symbol="gemm_pring_kernel", flop=2.0 * M * I * H, bytes=2.0 * M * (I + H) + 4.0 * I * H)
    if name == "ffn1":
        M = int(os.environ.get("MV_FFN1_ROWS", ROWS_PADDED))          # (experiments: the packed row count of a step is ROWS_PACKED)
        x, w, b = _rnd((M, H), 1.0, 21, dev), _rnd((I, H), 0.02, 22, dev), _rnd((I,), 0.02, 23, dev, torch.float32)
        o, d = torch.empty((M, I), device=dev, dtype=torch.float16), torch.empty((M, I), device=dev, dtype=torch.float16)
        fn = lambda: ops.gemm(x, w, o, M=M, N=I, K=H, bias=b, epi=EPI_BIAS_GELU_D, c2=d)
        return fn, dict(kernel=f"gemm_ring_kernel<NT, 256x256x64, f16 operands> {M}x{I}x{H} +bias+GELU (writes gelu(z) and gelu'(z), f16)",
                        symbol="gemm_ring_kernel", flop=2.0 * M * I * H, bytes=2.0 * (M * H + I * H + 2 * M * I))
    if name == "attn":
        import medvill_amd as mv
        B, A, dh, N, S = 64, 12, 64, 36, 473
        L = N + S + 3
        g = torch.Generator().manual_seed(1)
        n_ids = torch.randint((S + 1) // 2 + 1, S + 2, (B,), generator=g)
        desc = mv.data.MaskDesc.make("full", N, S, n_ids, dev)
        bits = torch.zeros((B, L, (L + 31) // 32), dtype=torch.int32, device=dev)
        ti = torch.zeros((B, (L + 63) // 64, (L + 63) // 64), dtype=torch.uint8, device=dev)
        ops.mask_build(desc.desc, B, L, bits, ti)
        cu, _, _ = ops.pack_plan(desc.desc, B, L)
        M = int(cu[-1])
        vl = (N + 2 + n_ids).double()
        qkv, dctx = _rnd((M, 3 * H), 1.0, 31, dev), _rnd((M, H), 1.0, 32, dev)
        ctx = torch.empty((M, H), device=dev, dtype=torch.float16)
        lse, delta = torch.empty((B, A, L), device=dev), torch.empty((B, A, L), device=dev)
        dqkv = torch.empty_like(qkv)
        db = torch.empty(ops.dropbits_numel(B, L, A), dtype=torch.int32, device=dev)

        def fn():
            ops.attn_dropmask(0.1, 12345, B, L, A, db, cu=cu)
            ops.attn_fwd(qkv, bits, ti, ctx, lse, B, L, A, dh, p_drop=0.1, cu=cu, total_rows=M, dropbits=db)
            ops.attn_bwd(qkv, ctx, dctx, lse, bits, ti, dqkv, delta, B, L, A, dh, p_drop=0.1, cu=cu, total_rows=M, dropbits=db)
        fwd = 4.0 * A * dh * float((vl * vl).sum())
        return fn, dict(kernel=f"attn_dropmask + attn_fwd_mfma + attn_bwd_dq_mfma + attn_bwd_dkv_mfma, B={B} A={A} L={L} rows {M}, dropout 0.1",
                        symbol="attn_", flop=3.5 * fwd, bytes=2.0 * M * (3 * H + H) * 3)
    if name in GEMM_FAMILY:
        return _gemm_family(name, dev)
    raise ValueError(name)


# The other GEMM calls of one encoder layer, exactly as Engine.encoder_forward / encoder_backward issue them at the bench shape (25,483 packed
# rows, f16 operands, dropout 0.1): name -> (description, launches per step).  With `ffn1`, `dw` and `attn` they are the step's MFMA kernels:
# bench.py times each alone and reports them as roofline.kernels (the line covers the step, not one launch in twelve).
GEMM_FAMILY = {
    "qkv": ("y = x.Wqkv^T + b: rows x 2304 x 768 (256x256 ring kernel)", 12),
    "wo": ("attention output projection + bias + dropout + residual: rows x 768 x 768 (128x128 kernel)", 12),
    "ffn2": ("FFN-down + bias + dropout + residual: rows x 768 x 3072 (128x128 kernel)", 12),
    "dz": ("dz = (dy.W2) * gelu'(z) + per-tile column sums: rows x 3072 x 768 over the W2^T copy (256x256 ring kernel)", 12),
    "da": ("da = dz.W1 + residual gradient: rows x 768 x 3072 (128x128 kernel, NN)", 12),
    "dctx": ("dctx = dproj.Wo: rows x 768 x 768 (128x128 kernel, NN)", 12),
    "dxqkv": ("dx = dqkv.Wqkv + residual gradient: rows x 768 x 2304 (128x128 kernel, NN)", 12),
    "dw2": ("dW2 = dproj^T.gelu(z): 768 x 3072 over the rows (persistent 256x256 kernel, TN, split-K + reduction)", 12),
    "dwqkv": ("dWqkv = dqkv^T.x: 2304 x 768 over the rows (persistent kernel, TN, split-K + reduction)", 12),
    "dwo": ("dWo = dproj^T.ctx: 768 x 768 over the rows (persistent kernel, TN, split-K + reduction)", 12),
}
LAUNCHES_PER_STEP = {"ffn1": 12, "dw": 12, "attn": 12}


def _gemm_family(name, dev):
    from medvill_amd import hip_ops as ops
    from medvill_amd._lib import EPI_BIAS, EPI_BIAS_RES, EPI_MUL, EPI_RES
    M = ROWS_PACKED
    f16 = torch.float16
    e = lambda *shape: torch.empty(shape, device=dev, dtype=f16)
    alpha = torch.tensor([1.0 / 32768.0], device=dev)
    if name == "qkv":
        x, w, b, o = _rnd((M, H), 1.0, 41, dev), _rnd((3 * H, H), 0.02, 42, dev), _rnd((3 * H,), 0.02, 43, dev, torch.float32), e(M, 3 * H)
        fn, fl, by = (lambda: ops.gemm(x, w, o, M=M, N=3 * H, K=H, bias=b, epi=EPI_BIAS)), 2.0 * M * 3 * H * H, 2.0 * (M * H + 3 * H * H + M * 3 * H)
    elif name in ("wo", "ffn2"):
        K = H if name == "wo" else I
        x, w, b, r, o = _rnd((M, K), 1.0, 44, dev), _rnd((H, K), 0.02, 45, dev), _rnd((H,), 0.02, 46, dev, torch.float32), _rnd((M, H), 1.0, 47, dev), e(M, H)
        fn = lambda: ops.gemm(x, w, o, M=M, N=H, K=K, bias=b, epi=EPI_BIAS_RES, r=r, p_drop=0.1, drop_key=99)
        fl, by = 2.0 * M * H * K, 2.0 * (M * K + H * K + 2 * M * H)
    elif name == "dz":
        dy, w2t, dg, o = _rnd((M, H), 0.1, 48, dev), _rnd((I, H), 0.02, 49, dev), _rnd((M, I), 0.5, 50, dev), e(M, I)
        part = torch.empty((2 * ((M + 255) // 256), I), device=dev)
        fn = lambda: ops.gemm(dy, w2t, o, M=M, N=I, K=H, epi=EPI_MUL, r=dg, colsum_part=part)
        fl, by = 2.0 * M * I * H, 2.0 * (M * H + I * H + 2 * M * I)
    elif name in ("da", "dctx", "dxqkv"):
        K = {"da": I, "dctx": H, "dxqkv": 3 * H}[name]
        dy, w, r, o = _rnd((M, K), 0.1, 51, dev), _rnd((K, H), 0.02, 52, dev), _rnd((M, H), 0.1, 53, dev), e(M, H)
        if name == "dctx":
            fn = lambda: ops.gemm(dy, w, o, tb=True, M=M, N=H, K=K)
        else:
            fn = lambda: ops.gemm(dy, w, o, tb=True, M=M, N=H, K=K, epi=EPI_RES, r=r)
        fl, by = 2.0 * M * H * K, 2.0 * (M * K + H * K + M * H * (1 if name == "dctx" else 2))
    else:
        No, Ko = {"dw2": (H, I), "dwqkv": (3 * H, H), "dwo": (H, H)}[name]
        dy, x = _rnd((M, No), 0.1, 54, dev), _rnd((M, Ko), 1.0, 55, dev)
        g = torch.zeros((No, Ko), device=dev)
        ws = torch.empty((32 if No * Ko <= 1024 * 1024 else 16) * No * Ko, device=dev)
        fn = lambda: ops.gemm(dy, x, g, ta=True, tb=True, M=No, N=Ko, K=M, lda=No, ldb=Ko, splitk=0, ws=ws, alpha=alpha)
        fl, by = 2.0 * M * No * Ko, 2.0 * M * (No + Ko) + 4.0 * No * Ko
    return fn, dict(kernel=GEMM_FAMILY[name][0], symbol="gemm_", flop=fl, bytes=by)


def time_case(name, reps=200, warm=50, dev="cuda", batch=10):
    """HIP events on the stream the kernel is launched on.  `warm` untimed calls first (clock ramp, cold caches, first-touch page faults:
    round 4's 20-launch timing sat 25 % below its own rocprofv3 average), then `reps` timed calls in batches of `batch` (one event pair
    per batch: per-call events would put a host round trip between launches).  meta["ms"] = MEDIAN batch mean per call; ms_min / ms_max
    = fastest / slowest batch mean; ms_mean = all timed calls.  rocprofv3's per-launch average over the same run (which includes the
    warm-up launches) is what profiles/tools/r05_condense.py puts beside it."""
    import statistics
    fn, meta = make_case(name, dev)
    st = torch.cuda.current_stream()
    for _ in range(warm):
        fn()
    nb = max(1, reps // batch)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(nb + 1)]
    evs[0].record(st)
    for i in range(nb):
        for _ in range(batch):
            fn()
        evs[i + 1].record(st)
    evs[-1].synchronize()
    per = [evs[i].elapsed_time(evs[i + 1]) / batch for i in range(nb)]
    meta["ms"] = statistics.median(per)
    meta["ms_min"], meta["ms_max"], meta["ms_mean"] = min(per), max(per), sum(per) / nb
    meta["timed_calls"], meta["warmup_calls"] = nb * batch, warm
    return meta


if __name__ == "__main__":
    case = sys.argv[1] if len(sys.argv) > 1 else "dw"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    warm = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    if case == "all":
        import json
        tot, rows = 0.0, {}
        for c in ["ffn1", "dw", "attn"] + list(GEMM_FAMILY):
            m = time_case(c, reps=reps, warm=warm)
            n = LAUNCHES_PER_STEP.get(c) or GEMM_FAMILY[c][1]
            tot += n * m["ms"]
            rows[c] = dict(us_median=m["ms"] * 1e3, us_min=m["ms_min"] * 1e3, us_max=m["ms_max"] * 1e3, us_mean=m["ms_mean"] * 1e3,
                           tflops=m["flop"] / m["ms"] / 1e9, flop=m["flop"], bytes=m["bytes"], launches_per_step=n, kernel=m["kernel"])
            print(f"{c:6s} median {m['ms'] * 1e3:7.1f} us (min {m['ms_min'] * 1e3:.1f}, max {m['ms_max'] * 1e3:.1f})  {m['flop'] / m['ms'] / 1e9:5.0f} TFLOP/s  "
                  f"x{n} = {n * m['ms']:.2f} ms per step   {m['kernel']}", flush=True)
        print(f"sum over a step (12 layers, stand-alone medians): {tot:.2f} ms")
        if len(sys.argv) > 4:
            with open(sys.argv[4], "w") as f:
                json.dump(rows, f, indent=1)
        sys.exit(0)
    m = time_case(case, reps=reps, warm=warm)
    print(f"{case}: median {m['ms'] * 1e3:.1f} us per call (min {m['ms_min'] * 1e3:.1f}, max {m['ms_max'] * 1e3:.1f}, mean {m['ms_mean'] * 1e3:.1f}; HIP events, "
          f"{m['timed_calls']} calls after {m['warmup_calls']} warm-up) = {m['flop'] / m['ms'] / 1e9:.0f} TFLOP/s algorithmic; {m['kernel']}")
