"""Randomised shape sweeps of the two kernel families against fp64 restatements (GPU).  Deterministic seeds; shapes are drawn to
hit ragged tiles, odd leading dimensions, every layout / epilogue / output encoding of mv_gemm and ragged packed attention with
dropout -- the paths the fixed-shape tests pin one shape at a time."""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medvill_amd as mv                      # noqa: E402
from medvill_amd import hip_ops as ops        # noqa: E402
from medvill_amd._lib import (EPI_BIAS, EPI_BIAS_GELU_D, EPI_BIAS_RES, EPI_MUL, EPI_NONE, EPI_RES)   # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"
VARIANTS = [(0, 0), (1, 0), (2, 14), (2, 24)]      # auto / 128x128 / 256-row ring / persistent ring


def _rnd(g, shape, dtype, scale=0.5):
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(DEV)


def _relerr(got, ref):
    ref = ref.detach()
    return float((got.double() - ref).abs().max() / (ref.abs().max() + 1e-30))


def _gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _dgelu(z):
    return 0.5 * (1 + torch.erf(z / math.sqrt(2.0))) + z * torch.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)


@pytest.mark.parametrize("seed", range(120))
def test_gemm_random_configurations(seed):
    rs = np.random.RandomState(1000 + seed)
    g = torch.Generator().manual_seed(seed)
    M = int(rs.choice([1, 7, 64, 200, 256, 300, 513, 1030]))
    N = int(rs.choice([2, 8, 64, 130, 192, 256, 384, 520, 768, 1032, 3072]))
    K = int(rs.choice([8, 72, 128, 200, 768, 1032]))
    ta, tb = [(0, 0), (0, 1), (1, 1), (1, 0)][rs.randint(4)]
    f16_ops = bool(rs.randint(2)) and not (ta or tb)
    odt = torch.float16 if f16_ops else torch.bfloat16
    epi = [EPI_NONE, EPI_BIAS, EPI_BIAS_RES, EPI_RES, EPI_MUL, EPI_BIAS_GELU_D][rs.randint(6)]
    cdt = [torch.float32, torch.bfloat16, torch.float16][rs.randint(3)]
    rdt = [torch.bfloat16, torch.float16, torch.float32][rs.randint(3)]
    pad_a, pad_b, pad_c = (8 * int(rs.randint(3)) for _ in range(3))           # leading dimensions beyond the logical width
    force, nj = VARIANTS[rs.randint(len(VARIANTS))]
    up8 = lambda n: (n + 7) // 8 * 8             # 16-bit operands: leading dimensions are multiples of 8 (include/medvill.h)
    a_shape = (K, up8(M) + pad_a) if ta else (M, K + pad_a)
    b_shape = (K, up8(N) + pad_b) if tb else (N, K + pad_b)
    a = _rnd(g, a_shape, odt)
    b = _rnd(g, b_shape, odt)
    bias = _rnd(g, (N,), torch.float32, 1.0)
    r = _rnd(g, (M, N + pad_c), rdt, 1.0)
    c = torch.full((M, N + pad_c), float("nan"), dtype=cdt, device=DEV)
    c2 = torch.full((M, N + pad_c), float("nan"), dtype=cdt, device=DEV)
    want_c3 = cdt != torch.float32 and epi in (EPI_NONE, EPI_BIAS, EPI_BIAS_GELU_D) and bool(rs.randint(2))
    c3 = torch.full((M, N + pad_c), float("nan"), dtype=torch.bfloat16 if cdt == torch.float16 else torch.float16, device=DEV) if want_c3 else None
    # hidden-state dropout inside the bias + residual epilogue (mask over the index m*N + n, groups of 4 columns)
    p_drop, dkey = (0.1, 77 + seed) if (epi == EPI_BIAS_RES and N % 4 == 0 and rs.randint(2)) else (0.0, 0)
    ops.set_gemm_variant(force, nj)
    try:
        ops.gemm(a, b, c, ta=bool(ta), tb=bool(tb), M=M, N=N, K=K, lda=a_shape[1], ldb=b_shape[1], ldc=N + pad_c, bias=bias, epi=epi,
                 r=r, ldr=N + pad_c, c2=c2, ldc2=N + pad_c, c3=c3, ldc3=(N + pad_c) if want_c3 else None, p_drop=p_drop, drop_key=dkey)
    finally:
        ops.set_gemm_variant(0, 0)
    A = (a[:, :M].double().t() if ta else a[:, :K].double())
    Bm = (b[:, :N].double() if tb else b[:, :K].double().t())
    y = A @ Bm
    rr = r[:, :N].double()
    yb = y + bias.double()
    if p_drop > 0:
        keep, sc = ops.dropout_mask(p_drop, dkey, M * N, DEV)
        yb = yb * keep.view(M, N).double() * sc
    ref = {EPI_NONE: y, EPI_BIAS: y + bias.double(), EPI_BIAS_RES: yb + rr, EPI_RES: y + rr, EPI_MUL: y * rr,
           EPI_BIAS_GELU_D: _gelu(y + bias.double())}[epi]
    tol = 2e-5 * math.sqrt(K) if cdt == torch.float32 else 1.2e-2
    cfg = dict(seed=seed, M=M, N=N, K=K, ta=ta, tb=tb, f16=f16_ops, epi=epi, cdt=cdt, rdt=rdt, variant=(force, nj), p_drop=p_drop)
    got = c[:, :N]
    assert torch.isfinite(got.float()).all(), cfg
    assert _relerr(got, ref) < tol, cfg
    if epi == EPI_BIAS_GELU_D:
        assert _relerr(c2[:, :N], _dgelu(y + bias.double())) < tol, cfg
    if want_c3:
        assert _relerr(c3[:, :N], ref) < 1.2e-2, cfg
    if pad_c:                                        # columns beyond N are never written
        assert torch.isnan(c[:, N:].float()).all(), cfg


def _attn_ref(qkv_rows, cu, B, A, Lq, mask, keep_scaled):
    """fp64 attention with an explicit (already scaled) keep mask; qkv_rows are the packed rows, cu their per-sample ranges."""
    H = qkv_rows.shape[1] // 3
    dh = H // A
    outs = []
    for b in range(B):
        r0, r1 = int(cu[b]), int(cu[b + 1])
        n = r1 - r0
        q, k, v = [t.view(n, A, dh).permute(1, 0, 2) for t in qkv_rows[r0:r1].split(H, dim=-1)]
        add = (1.0 - mask[b, :n, :n].double()) * -10000.0
        p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh) + add[None], -1)
        if keep_scaled is not None:
            p = p * keep_scaled[b, :, :n, :n]
        outs.append((p @ v).permute(1, 0, 2).reshape(n, H))
    return torch.cat(outs, 0)


@pytest.mark.parametrize("seed", range(30))
def test_attention_random_configurations(seed):
    rs = np.random.RandomState(2000 + seed)
    fam = ["full", "s2s", "full", "s2s", "1d"][rs.randint(5)]
    B, A = int(rs.randint(1, 5)), int(rs.choice([1, 2, 3, 12]))
    N, S = int(rs.choice([4, 16, 36, 100])), int(rs.choice([20, 61, 150, 300, 473]))
    p_drop = float(rs.choice([0.0, 0.1]))
    key = int(rs.randint(1, 2 ** 31))
    dh, Lq = 64, N + S + 3
    H = A * dh
    n_ids = torch.from_numpy(rs.randint(2, S + 2, size=B)).to(torch.int32)
    desc = mv.data.MaskDesc.make(fam, N, S, n_ids, DEV)
    mask = mv.data.build_mask(fam, N, S, n_ids)
    if mask.dim() == 2:
        mask = mask[:, None, :].expand(B, Lq, Lq)
    mask = mask.to(DEV)
    bits = torch.zeros((B, Lq, (Lq + 31) // 32), dtype=torch.int32, device=DEV)
    tinfo = torch.zeros((B, (Lq + 63) // 64, (Lq + 63) // 64), dtype=torch.uint8, device=DEV)
    ops.mask_build(desc.desc, B, Lq, bits, tinfo)
    cu, _, _ = ops.pack_plan(desc.desc, B, Lq)
    M = int(cu[-1])
    g = torch.Generator().manual_seed(seed)
    qkv = _rnd(g, (M, 3 * H), torch.bfloat16, 1.0)
    dctx = _rnd(g, (M, H), torch.bfloat16, 1.0)
    ctx = torch.full((M, H), float("nan"), dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros((B, A, Lq), dtype=torch.float32, device=DEV)
    keep, db = None, None
    if p_drop > 0:
        db = torch.zeros(ops.dropbits_numel(B, Lq, A), dtype=torch.int32, device=DEV)
        ops.attn_dropmask(p_drop, key, B, Lq, A, db, cu=cu)
        thr = round(p_drop * 65536)
        keep = ops.attn_keep_mask(db, B, Lq, A).double() * (65536.0 / (65536.0 - thr))
    ops.attn_fwd(qkv, bits, tinfo, ctx, lse, B, Lq, A, dh, p_drop=p_drop, cu=cu, total_rows=M, dropbits=db)
    qd = qkv.double().requires_grad_(True)
    rctx = _attn_ref(qd, cu.cpu(), B, A, Lq, mask, keep)
    cfg = dict(seed=seed, fam=fam, B=B, A=A, N=N, S=S, p=p_drop, n_ids=n_ids.tolist())
    assert torch.isfinite(ctx.float()).all(), cfg
    assert _relerr(ctx, rctx) < 2e-2, cfg
    dqkv = torch.full((M, 3 * H), float("nan"), dtype=torch.bfloat16, device=DEV)
    delta = torch.zeros((B, A, Lq), dtype=torch.float32, device=DEV)
    ops.attn_bwd(qkv, ctx, dctx, lse, bits, tinfo, dqkv, delta, B, Lq, A, dh, p_drop=p_drop, cu=cu, total_rows=M, dropbits=db)
    (rctx * dctx.double()).sum().backward()
    assert torch.isfinite(dqkv.float()).all(), cfg
    assert _relerr(dqkv, qd.grad) < 2.5e-2, cfg
