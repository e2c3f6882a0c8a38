"""GPU parity tests of the individual HIP kernels, called through the C ABI (ctypes), against
plain fp32/fp64 torch restatements of the same op on the same (already rounded) inputs.
Tolerances: f32 path 1e-4 relative-to-scale (it is exact fp32 arithmetic in a different order);
bf16 path: fp32-accumulated results compared at 2e-3 before output rounding, 1e-2 after."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import medvill_amd as mv                                   # noqa: E402
from medvill_amd import hip_ops as ops                      # noqa: E402
from medvill_amd import data as D                           # noqa: E402
from medvill_amd._lib import (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GELU_D, EPI_BIAS_RES, EPI_BIAS_TANH, EPI_DGELU, EPI_MUL, EPI_NONE,  # noqa: E402
                              EPI_RES)

DEV = "cuda"


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(DEV)


def relerr(got, ref):
    ref = ref.double()
    return float((got.double() - ref).abs().max() / (ref.abs().max() + 1e-30))


def gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def dgelu(z):
    return 0.5 * (1 + torch.erf(z / math.sqrt(2.0))) + z * torch.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)


# ------------------------------------------------------------------------------------------ GEMM
GEMM_SHAPES = [(256, 256, 256), (130, 70, 200), (1024, 768, 768), (64, 2, 768), (96, 1000, 128), (517, 264, 3072)]


# set_gemm_variant(force, nj); "ring320" = the 320 x 256-tile form of the ring kernel (ta = 0 layouts; ta = 1 falls back to the persistent kernel)
VARIANT = {"mfma": (1, 0), "mfma2s": (1, 32), "mfma256k64": (2, 14), "pring256": (2, 24), "mfma_auto": (0, 0), "ring320": (2, 10)}


@pytest.mark.parametrize("impl", ["mfma", "mfma2s", "mfma256k64", "pring256", "simple_bf16", "f32", "mfma:f16", "mfma256k64:f16", "pring256:f16",
                                  "mfma_auto:f16", "ring320", "ring320:f16"])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_layouts(impl, ta, tb, M, N, K):
    """All four operand layouts on every kernel family; ':f16' = f16-encoded operands (forward products and, under the loss
    scale, the gradient products dx = dy.W and dW = dy^T.x; the fourth layout exists in bf16 only and is refused)."""
    f16 = impl.endswith(":f16")
    impl = impl.split(":")[0]
    dt = torch.float32 if impl == "f32" else (torch.float16 if f16 else torch.bfloat16)
    ops.set_impl(1 if impl == "simple_bf16" else 0)
    ops.set_gemm_variant(*VARIANT.get(impl, (0, 0)))
    try:
        if f16 and ta and not tb:
            with pytest.raises(RuntimeError, match="MV_E_DTYPE"):       # refused, never silently computed in another encoding
                z8 = torch.zeros((64, 64), dtype=dt, device=DEV)
                ops.gemm(z8, z8, torch.zeros((8, 8), dtype=torch.float32, device=DEV), ta=True, tb=False, M=8, N=8, K=8, lda=64, ldb=64)
            return
        pad = lambda n: (n + 7) // 8 * 8
        lda = pad(M if ta else K) + 8
        ldb = pad(N if tb else K) + 16
        a = rnd((K if ta else M, lda), dt, 1)
        b = rnd((K if tb else N, ldb), dt, 2)
        # zero the padding of k-contiguous operands (contract of mv_gemm for K % 8 != 0)
        if not ta:
            a[:, K:] = 0
        if not tb:
            b[:, K:] = 0
        c = torch.full((M, N + 3), 7.0, dtype=torch.float32, device=DEV)
        ops.gemm(a, b, c, ta=bool(ta), tb=bool(tb), M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=N + 3)
        A = (a[:, :M].t() if ta else a[:, :K]).float()
        Bm = (b[:, :N] if tb else b[:, :K].t()).float()
        ref = A.double() @ Bm.double()
        assert relerr(c[:, :N], ref) < (1e-5 if impl == "f32" else 2e-5) * math.sqrt(K)
        assert (c[:, N:] == 7.0).all()           # nothing written outside [M,N]
    finally:
        ops.set_impl(0)
        ops.set_gemm_variant(0, 0)


@pytest.mark.parametrize("M", [25483, 21760, 27200, 27201, 18000])
@pytest.mark.parametrize("form", ["wo", "ffn2", "da", "dctx"])
def test_one_round_ring_tiles_on_the_layer_shapes(M, form):
    """Round 5: 768-column products over ~25,500 packed rows run as ONE round of 256-column ring tiles -- 256 rows per tile when
    ceil(M/256) * 3 <= 256 CUs, else 320 (240 tiles at 25,483 rows); beyond 27,200 rows the 128 x 128 kernel as before.  The library's
    own choice (no knob) on the production shapes and epilogues against an f64 product on sampled rows -- first / last rows of the
    matrix, rows around every tile edge of both tile heights -- and nothing written past row M."""
    H, I = 768, 3072
    K = {"wo": H, "ffn2": I, "da": I, "dctx": H}[form]
    tb = form in ("da", "dctx")
    f16 = torch.float16
    a = rnd((M, K), f16, 11, 0.5)
    b = rnd((K, H) if tb else (H, K), f16, 12, 0.05)
    r = rnd((M, H), f16, 13)
    bias = rnd((H,), torch.float32, 14)
    c = torch.full((M + 8, H), 7.0, dtype=f16, device=DEV)
    if form in ("wo", "ffn2"):
        ops.gemm(a, b, c, M=M, N=H, K=K, bias=bias, epi=EPI_BIAS_RES, r=r)          # dropout off: deterministic reference
    elif form == "da":
        ops.gemm(a, b, c, tb=True, M=M, N=H, K=K, epi=EPI_RES, r=r)
    else:
        ops.gemm(a, b, c, tb=True, M=M, N=H, K=K)
    assert (c[M:] == 7.0).all()
    rows = sorted({x for e in (0, 255, 256, 319, 320, 639, 640, 12799, 12800, M // 2) for x in (e, e + 1) if x < M}
                  | {M - 1, M - 2, M - 161, (M - 1) // 320 * 320, (M - 1) // 256 * 256})
    idx = torch.tensor(rows, device=DEV)
    y = a[idx].double() @ (b.double() if tb else b.double().t())
    if form in ("wo", "ffn2"):
        y = y + bias.double() + r[idx].double()
    elif form == "da":
        y = y + r[idx].double()
    assert torch.isfinite(c[:M].float()).all()
    assert relerr(c[idx], y) < 2e-3
    # a coarse whole-matrix check against torch's own f16 matmul (different summation order, f16 output rounding on both sides)
    full = (a @ (b if tb else b.t())).float()
    if form in ("wo", "ffn2"):
        full = full + bias + r.float()
    elif form == "da":
        full = full + r.float()
    assert relerr(c[:M], full.double()) < 5e-3


@pytest.mark.parametrize("impl", ["mfma", "mfma256k64", "pring256", "ring320"])
@pytest.mark.parametrize("epi", [EPI_MUL, EPI_RES])
@pytest.mark.parametrize("rdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tb", [False, True])
@pytest.mark.parametrize("M,N,K", [(300, 512, 192), (513, 3072, 768)])
def test_gemm_elementwise_operand_16_byte_path(impl, epi, rdt, tb, M, N, K):
    """C = (A.B) * R and C = A.B + R with a 16-bit R and a 16-bit C: the 256-row kernels read R in 16-byte pieces (8 columns per
    lane, requested two 16-row groups ahead); ragged last row tile, R in either encoding, both weight layouts (dz = dy.W2 is
    the NN form, its NT form runs over a transposed weight copy)."""
    a = rnd((M, K), torch.bfloat16, 3, 0.5)
    b = rnd((K, N) if tb else (N, K), torch.bfloat16, 4, 0.5)
    r = rnd((M, N), rdt, 6)
    c = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.set_gemm_variant(*VARIANT.get(impl, (0, 0)))
    try:
        ops.gemm(a, b, c, tb=tb, M=M, N=N, K=K, epi=epi, r=r)
    finally:
        ops.set_gemm_variant(0, 0)
    y = a.double() @ (b.double() if tb else b.double().t())
    ref = y * r.double() if epi == EPI_MUL else y + r.double()
    assert torch.isfinite(c.float()).all() and relerr(c, ref) < 1e-2


@pytest.mark.parametrize("impl", ["mfma", "mfma2s", "mfma256k64", "pring256", "f32", "ring320"])
@pytest.mark.parametrize("epi", [EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_DGELU, EPI_RES, EPI_BIAS_TANH, EPI_BIAS_GELU_D, EPI_MUL])
@pytest.mark.parametrize("M,N,K,cdt", [(256, 384, 128, "bf16"), (200, 130, 72, "f32"), (128, 768, 768, "f32")])
def test_gemm_epilogues(impl, epi, M, N, K, cdt):
    dt = torch.float32 if impl == "f32" else torch.bfloat16
    cd = torch.float32 if (cdt == "f32" or impl == "f32") else torch.bfloat16
    a, b = rnd((M, K), dt, 3, 0.5), rnd((N, K), dt, 4, 0.5)
    bias = rnd((N,), torch.float32, 5)
    r = rnd((M, N), dt, 6)
    c = torch.zeros((M, N), dtype=cd, device=DEV)
    c2 = torch.zeros((M, N), dtype=cd, device=DEV)
    ops.set_gemm_variant(*VARIANT.get(impl, (0, 0)))
    try:
        ops.gemm(a, b, c, M=M, N=N, K=K, bias=bias, epi=epi, r=r, c2=c2)
    finally:
        ops.set_gemm_variant(0, 0)
    y = a.double() @ b.double().t()
    rr = r.double()
    if epi == EPI_BIAS:
        ref = y + bias
    elif epi == EPI_BIAS_GELU:
        z = y + bias
        ref = gelu(z)
        assert relerr(c2, z) < (1e-5 if cd == torch.float32 else 1e-2)
    elif epi == EPI_BIAS_GELU_D:
        z = y + bias
        ref = gelu(z)
        assert relerr(c2, dgelu(z)) < (1e-5 if cd == torch.float32 else 1e-2)
    elif epi == EPI_BIAS_RES:
        ref = y + bias + rr
    elif epi == EPI_DGELU:
        ref = y * dgelu(rr)
    elif epi == EPI_MUL:
        ref = y * rr
    elif epi == EPI_RES:
        ref = y + rr
    else:
        ref = torch.tanh(y + bias)
    assert relerr(c, ref) < (2e-5 if cd == torch.float32 else 1e-2)


@pytest.mark.parametrize("impl", ["mfma", "mfma2s", "mfma256k64", "pring256", "f32"])
def test_gemm_splitk_and_accumulate(impl):
    dt = torch.float32 if impl == "f32" else torch.bfloat16
    Mt, No, Ko = 4096, 200, 136
    dy, x = rnd((Mt, No), dt, 7, 0.3), rnd((Mt, Ko), dt, 8, 0.3)
    ref = dy.double().t() @ x.double()
    ops.set_gemm_variant(*VARIANT.get(impl, (0, 0)))
    for sk in (1, 4, 7, 0):
        c = torch.zeros((No, Ko), dtype=torch.float32, device=DEV)
        ws = torch.empty(max(sk, 16) * No * Ko, dtype=torch.float32, device=DEV)
        ops.gemm(dy, x, c, ta=True, tb=True, M=No, N=Ko, K=Mt, lda=No, ldb=Ko, splitk=sk, ws=ws)
        assert relerr(c, ref) < 2e-4
    c = torch.ones((No, Ko), dtype=torch.float32, device=DEV)
    ops.gemm(dy, x, c, ta=True, tb=True, M=No, N=Ko, K=Mt, lda=No, ldb=Ko, accumulate=True)
    ops.set_gemm_variant(0, 0)
    assert relerr(c, ref + 1.0) < 2e-4


@pytest.mark.parametrize("ta,tb,M,N,K", [(0, 0, 1024, 768, 768), (0, 1, 2048, 768, 2304), (1, 1, 768, 3072, 8192), (1, 1, 768, 3072, 25483),
                                         (0, 0, 515, 2304, 264), (0, 1, 1000, 304, 1032)])
def test_gemm_auto_dispatch_large(ta, tb, M, N, K):
    """Shapes that take the 256-row LDS-DMA kernel under the automatic tile choice, incl. auto split-K."""
    a = rnd((K, M) if ta else (M, K), torch.bfloat16, 12, 0.5)
    b = rnd((K, N) if tb else (N, K), torch.bfloat16, 13, 0.5)
    c = torch.zeros((M, N), dtype=torch.float32, device=DEV)
    ws = torch.empty(16 * M * N, dtype=torch.float32, device=DEV)
    ops.gemm(a, b, c, ta=bool(ta), tb=bool(tb), M=M, N=N, K=K, splitk=0, ws=ws)
    A = (a.t() if ta else a).double()
    Bm = (b if tb else b.t()).double()
    assert relerr(c, A @ Bm) < 2e-5 * math.sqrt(K)


@pytest.mark.parametrize("impl", ["pring256", "mfma256k64"])
@pytest.mark.parametrize("ta,tb,M,N,K,epi", [(0, 0, 8200, 3000, 768, EPI_BIAS_GELU), (0, 0, 8192, 3072, 264, EPI_BIAS_RES),
                                             (0, 1, 9000, 2304, 520, EPI_DGELU), (0, 0, 16384, 768, 3072, EPI_BIAS),
                                             (0, 0, 8192, 3072, 768, EPI_BIAS_GELU_D), (0, 1, 8200, 3072, 768, EPI_MUL),
                                             (1, 1, 768, 3072, 8192, EPI_NONE), (1, 1, 2304, 776, 4104, EPI_NONE), (1, 1, 768, 768, 25483, EPI_NONE),
                                             (1, 0, 1000, 2048, 1032, EPI_RES)])
def test_gemm_persistent_many_units_per_block(impl, ta, tb, M, N, K, epi):
    """More (tile, K-slice) units than CUs: every block of the persistent kernel walks several units, with the ring
    carried across them, ragged tiles at both edges and the fused epilogues / split-K partial stores in between."""
    a = rnd((K, M) if ta else (M, K), torch.bfloat16, 21, 0.5)
    b = rnd((K, N) if tb else (N, K), torch.bfloat16, 22, 0.5)
    bias, r = rnd((N,), torch.float32, 23), rnd((M, N), torch.bfloat16, 24)
    split = epi == EPI_NONE
    cd = torch.float32 if split else torch.bfloat16
    c, c2 = torch.zeros((M, N), dtype=cd, device=DEV), torch.zeros((M, N), dtype=cd, device=DEV)
    ws = torch.empty(16 * M * N, dtype=torch.float32, device=DEV) if split else None
    ops.set_gemm_variant(*VARIANT[impl])
    try:
        for _ in range(2):                       # twice: the second launch must not depend on state left in LDS / ws
            ops.gemm(a, b, c, ta=bool(ta), tb=bool(tb), M=M, N=N, K=K, bias=bias, epi=epi, r=r, c2=c2, splitk=0 if split else 1, ws=ws)
    finally:
        ops.set_gemm_variant(0, 0)
    y = (a.t() if ta else a).double() @ (b if tb else b.t()).double()
    rr = r.double()
    if epi == EPI_BIAS_GELU:
        assert relerr(c2, y + bias) < 1e-2
        ref = gelu(y + bias)
    elif epi == EPI_BIAS_GELU_D:
        assert relerr(c2, dgelu(y + bias)) < 1e-2
        ref = gelu(y + bias)
    else:
        ref = {EPI_BIAS: y + bias, EPI_BIAS_RES: y + bias + rr, EPI_DGELU: y * dgelu(rr), EPI_RES: y + rr, EPI_NONE: y,
               EPI_MUL: y * rr}[epi]
    assert relerr(c, ref) < (2e-5 * math.sqrt(K) if split else 1e-2)
    # element-wise: a misplaced or stale tile would hide in a max-norm test of a smooth matrix; compare tile by tile
    err = (c.double() - ref).abs()
    tol = (2e-5 * math.sqrt(K) if split else 1e-2) * float(ref.abs().max())
    assert float(err.max()) < tol


def test_gemm_vocab_sized_tails():
    """Decoder shapes: N = V = 30522 (not a multiple of 8) forward; K = V with zero-padded rows backward."""
    V, H, R = 30522, 128, 96
    Vp = (V + 7) // 8 * 8
    t, E = rnd((R, H), torch.bfloat16, 9), rnd((V, H), torch.bfloat16, 10, 0.05)
    logits = torch.zeros((R, Vp), dtype=torch.float32, device=DEV)
    ops.gemm(t, E, logits, M=R, N=V, K=H, ldc=Vp)
    assert relerr(logits[:, :V], t.double() @ E.double().t()) < 1e-4
    dl = torch.zeros((R, Vp), dtype=torch.bfloat16, device=DEV)
    dl[:, :V] = rnd((R, V), torch.bfloat16, 11, 0.01)
    dt_ = torch.zeros((R, H), dtype=torch.float32, device=DEV)
    ops.gemm(dl, E, dt_, tb=True, M=R, N=H, K=V, lda=Vp, ldb=H)
    assert relerr(dt_, dl[:, :V].double() @ E.double()) < 1e-3
    dE = torch.zeros((V, H), dtype=torch.float32, device=DEV)
    ops.gemm(dl, t, dE, ta=True, tb=True, M=V, N=H, K=R, lda=Vp, ldb=H)
    assert relerr(dE, dl[:, :V].double().t() @ t.double()) < 1e-4


def test_gemm_rejects_bad_arguments():
    a = torch.zeros((8, 12), dtype=torch.bfloat16, device=DEV)
    c = torch.zeros((8, 8), dtype=torch.float32, device=DEV)
    with pytest.raises(RuntimeError, match="MV_E_SHAPE"):
        ops.gemm(a, a, c, M=8, N=8, K=12)                  # lda = 12 is not a multiple of 8
    with pytest.raises(RuntimeError, match="MV_E_ARG"):
        ops.gemm(a, a, c, M=0, N=8, K=8, lda=16, ldb=16)


# ------------------------------------------------------------------------------------------ attention
def attn_ref(qkv, mask, A):
    B, Lq, H3 = qkv.shape
    H = H3 // 3
    dh = H // A
    q, k, v = [t.view(B, Lq, A, dh).permute(0, 2, 1, 3) for t in qkv.double().split(H, dim=-1)]
    add = (1.0 - (mask if mask.dim() == 3 else mask[:, None, :].expand(B, Lq, Lq)).double()) * -10000.0
    s = q @ k.transpose(-1, -2) / math.sqrt(dh) + add[:, None]
    p = torch.softmax(s, -1)
    ctx = (p @ v).permute(0, 2, 1, 3).reshape(B, Lq, H)
    return ctx, torch.logsumexp(s, -1)


def _masks(name, B, N, S):
    Lq = N + S + 3
    g = torch.Generator().manual_seed(5)
    n_ids = torch.randint(2, S + 2, (B,), generator=g)
    if name == "random":
        m = (torch.rand((B, Lq, Lq), generator=g) < 0.6).long()
        m[:, :, 0] = 1
        return m
    if name == "deadrow":                 # one query row fully masked: additive -10000 on every key
        m = D.build_mask("bar", N, S, n_ids)
        m[0, Lq // 2, :] = 0
        return m
    if name == "mixed":
        return D.mixed_mask(N, S, n_ids, torch.arange(B) % 2 == 0)
    return D.build_mask(name, N, S, n_ids)


ATT_CASES = [("full", 2, 2, 16, 45), ("s2s", 2, 2, 16, 45), ("bar", 3, 2, 5, 29), ("noncross", 2, 2, 16, 45), ("1d", 2, 2, 16, 45),
             ("random", 2, 2, 7, 90), ("deadrow", 2, 2, 16, 45), ("s2s", 2, 4, 36, 473), ("mixed", 4, 3, 36, 150),
             ("full", 1, 12, 100, 665)]


@pytest.mark.parametrize("impl", ["mfma", "mfma_f16", "simple_bf16", "simple_f16", "f32"])
@pytest.mark.parametrize("fam,B,A,N,S", ATT_CASES)
def test_attention_fwd_bwd(impl, fam, B, A, N, S):
    """'_f16': qkv, ctx, dctx and dqkv all f16-encoded (the single-encoding 16-bit path; gradients at unit scale here)."""
    dt = torch.float32 if impl == "f32" else (torch.float16 if impl.endswith("f16") else torch.bfloat16)
    ops.set_impl(1 if impl.startswith("simple") else 0)
    try:
        dh, Lq = 64, N + S + 3
        H = A * dh
        mask = _masks(fam, B, N, S).to(DEV)
        qkv = rnd((B, Lq, 3 * H), dt, 21, 1.0)
        dctx = rnd((B, Lq, H), dt, 22, 1.0)
        bits = torch.zeros((B, Lq, (Lq + 31) // 32), dtype=torch.int32, device=DEV)
        tinfo = torch.zeros((B, (Lq + 63) // 64, (Lq + 63) // 64), dtype=torch.uint8, device=DEV)
        ops.mask_pack(mask, bits, tinfo)
        ctx = torch.zeros((B, Lq, H), dtype=dt, device=DEV)
        lse = torch.zeros((B, A, Lq), dtype=torch.float32, device=DEV)
        ops.attn_fwd(qkv.view(B * Lq, 3 * H), bits, tinfo, ctx, lse, B, Lq, A, dh)
        qd = qkv.double().requires_grad_(True)
        rctx, rlse = attn_ref(qd, mask, A)
        # a fully masked row keeps only the fp32-rounded differences of (score - 10000): the fp64 restatement
        # here does not round them, the fp32 reference does (ulp(1e4) = 9.8e-4)
        tol = (1e-3 if fam == "deadrow" else 1e-5) if impl == "f32" else (3e-3 if dt == torch.float16 else 1.5e-2)
        assert relerr(ctx, rctx) < tol
        assert float((lse.double() - rlse).abs().max()) < ((2e-3 if fam == "deadrow" else 1e-4) if impl == "f32" else 2e-2)
        # backward, with the kernel's own (rounded) ctx as the saved output
        dqkv = torch.zeros((B, Lq, 3 * H), dtype=dt, device=DEV)
        delta = torch.zeros((B, A, Lq), dtype=torch.float32, device=DEV)
        ops.attn_bwd(qkv.view(B * Lq, 3 * H), ctx, dctx, lse, bits, tinfo, dqkv, delta, B, Lq, A, dh)
        (rctx * dctx.double()).sum().backward()
        assert relerr(dqkv, qd.grad) < ((1e-3 if fam == "deadrow" else 1e-5) if impl == "f32" else (4e-3 if dt == torch.float16 else 2e-2))
    finally:
        ops.set_impl(0)


@pytest.mark.parametrize("impl", ["mfma", "mfma_f16", "simple_bf16", "f32"])
@pytest.mark.parametrize("fam,B,A,N,S", [("full", 2, 2, 16, 45), ("s2s", 2, 3, 36, 150), ("bar", 2, 2, 36, 221), ("full", 1, 12, 100, 409)])
def test_attention_dropout_follows_the_mask_function(impl, fam, B, A, N, S):
    """Attention-probability dropout (HF BertSelfAttention.dropout): the mask is the keep-bit tensor of mv_attn_dropmask; the forward
    and both backward kernels (MFMA: select masks / per-key dwords; plain VALU: bit look-ups) must all apply exactly it -- checked by
    decoding the bits, applying them in an fp64 restatement and differentiating that.  P(drop) = 6554 / 65536."""
    dt = torch.float32 if impl == "f32" else (torch.float16 if impl.endswith("f16") else torch.bfloat16)
    ops.set_impl(1 if impl.startswith("simple") else 0)
    try:
        dh, Lq, key, p = 64, N + S + 3, 0x1234567, 0.1
        H = A * dh
        mask = _masks(fam, B, N, S).to(DEV)
        qkv = rnd((B, Lq, 3 * H), dt, 41, 1.0)
        dctx = rnd((B, Lq, H), dt, 42, 1.0)
        bits = torch.zeros((B, Lq, (Lq + 31) // 32), dtype=torch.int32, device=DEV)
        tinfo = torch.zeros((B, (Lq + 63) // 64, (Lq + 63) // 64), dtype=torch.uint8, device=DEV)
        ops.mask_pack(mask, bits, tinfo)
        db = torch.zeros(ops.dropbits_numel(B, Lq, A), dtype=torch.int32, device=DEV)
        ops.attn_dropmask(p, key, B, Lq, A, db)
        keep = ops.attn_keep_mask(db, B, Lq, A).double()
        sc = 65536.0 / (65536.0 - 6554.0)
        frac = float(keep.mean())
        assert abs(frac - (1 - 6554 / 65536)) < 4 * math.sqrt(0.1 * 0.9 / keep.numel()) + 1e-4
        db2 = torch.zeros_like(db)
        ops.attn_dropmask(p, key + 1, B, Lq, A, db2)
        assert not torch.equal(db, db2)                                            # another key, another mask
        k2 = ops.attn_keep_mask(db2, B, Lq, A).double()
        assert abs(float((keep * k2).mean()) - frac * float(k2.mean())) < 5 * math.sqrt(0.09 / keep.numel()) + 1e-4   # and an independent one
        ctx = torch.zeros((B, Lq, H), dtype=dt, device=DEV)
        lse = torch.zeros((B, A, Lq), dtype=torch.float32, device=DEV)
        ops.attn_fwd(qkv.view(B * Lq, 3 * H), bits, tinfo, ctx, lse, B, Lq, A, dh, p_drop=p, dropbits=db)
        with pytest.raises(RuntimeError, match="MV_E_ARG"):                        # no mask tensor, no dropout: never a silent p = 0
            ops.attn_fwd(qkv.view(B * Lq, 3 * H), bits, tinfo, ctx.clone(), lse.clone(), B, Lq, A, dh, p_drop=p)
        qd = qkv.double().requires_grad_(True)
        q, k, v = [t.view(B, Lq, A, dh).permute(0, 2, 1, 3) for t in qd.split(H, dim=-1)]
        add = (1.0 - (mask if mask.dim() == 3 else mask[:, None, :].expand(B, Lq, Lq)).double()) * -10000.0
        pr = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh) + add[:, None], -1) * keep * sc
        rctx = (pr @ v).permute(0, 2, 1, 3).reshape(B, Lq, H)
        tol = 1e-5 if impl == "f32" else 2e-2
        assert relerr(ctx, rctx) < tol
        dqkv = torch.zeros((B, Lq, 3 * H), dtype=dt, device=DEV)
        delta = torch.zeros((B, A, Lq), dtype=torch.float32, device=DEV)
        ops.attn_bwd(qkv.view(B * Lq, 3 * H), ctx, dctx, lse, bits, tinfo, dqkv, delta, B, Lq, A, dh, p_drop=p, dropbits=db)
        (rctx * dctx.double()).sum().backward()
        assert relerr(dqkv, qd.grad) < (1e-5 if impl == "f32" else 2.5e-2)
    finally:
        ops.set_impl(0)


def test_mask_pack_bits_and_tile_classes():
    B, N, S = 3, 36, 120
    Lq = N + S + 3
    for fam in ("full", "s2s", "bar", "noncross", "1d", "random", "deadrow"):
        mask = _masks(fam, B, N, S).to(DEV)
        W, T = (Lq + 31) // 32, (Lq + 63) // 64
        bits = torch.zeros((B, Lq, W), dtype=torch.int32, device=DEV)
        tinfo = torch.full((B, T, T), 9, dtype=torch.uint8, device=DEV)
        ops.mask_pack(mask, bits, tinfo)
        m3 = (mask if mask.dim() == 3 else mask[:, None, :].expand(B, Lq, Lq)).cpu().numpy() != 0
        pad = np.zeros((B, Lq, W * 32), bool)
        pad[:, :, :Lq] = m3
        want = (pad.reshape(B, Lq, W, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)
        assert np.array_equal(bits.cpu().numpy().view(np.uint32), want), fam     # bit-exact
        ti = tinfo.cpu().numpy()
        rows_ok = m3.any(-1)
        for b in range(B):
            for tq in range(T):
                for tk in range(T):
                    blk = m3[b, tq * 64:(tq + 1) * 64, tk * 64:(tk + 1) * 64]
                    c = 1 if blk.all() else (0 if (not blk.any() and rows_ok[b, tq * 64:(tq + 1) * 64].all()) else 2)
                    assert ti[b, tq, tk] == c, (fam, b, tq, tk)


def test_mask_build_from_descriptors_equals_packed_reference_matrices():
    """On-device synthesis {family, n2, vl} -> bits must be bit-identical to packing the Dataset-style matrices."""
    B, N, S = 5, 36, 90
    Lq = N + S + 3
    W, T = (Lq + 31) // 32, (Lq + 63) // 64
    n_ids = torch.tensor([1, 2, 46, 90, 91])
    fams = ["full", "s2s", "bar", "noncross", "1d"]
    for fam in fams + ["per-sample"]:
        per = fams if fam == "per-sample" else [fam] * B
        mats = torch.stack([(D.build_mask(f, N, S, n_ids[b:b + 1])[0] if f != "1d" else
                             D.build_mask("1d", N, S, n_ids[b:b + 1])[0][None, :].expand(Lq, Lq)) for b, f in enumerate(per)])
        b1 = torch.zeros((B, Lq, W), dtype=torch.int32, device=DEV)
        t1 = torch.zeros((B, T, T), dtype=torch.uint8, device=DEV)
        ops.mask_pack(mats.contiguous().to(DEV), b1, t1)
        b2, t2 = torch.zeros_like(b1), torch.zeros_like(t1)
        desc = D.MaskDesc.make(per, N, S, n_ids, DEV)
        ops.mask_build(desc.desc, B, Lq, b2, t2)
        assert torch.equal(b1, b2) and torch.equal(t1, t2), fam


# ------------------------------------------------------------------------------------------ batch assembly
class _Replay:
    """python `random` stand-in replaying per-token draws: random() -> u[i] for token i, randrange() -> rnd[i]."""

    def __init__(self, u, r):
        self.u, self.r, self.i = u, r, -1

    def random(self):
        self.i += 1
        return float(self.u[self.i])

    def randrange(self, n):
        return int(self.r[self.i])


@pytest.mark.parametrize("B,N,S,V", [(7, 6, 40, 1024), (64, 36, 473, 30522), (3, 0, 5, 30522), (5, 100, 665, 30522)])
def test_mlm_corrupt_equals_random_word_bit_exact(B, N, S, V):
    """mv_mlm_corrupt against the numpy restatement of dataset_origin.py:102-135,183-209 on shared draws."""
    from oracle import data_oracle as DO
    g = torch.Generator().manual_seed(B * 1000 + S)
    lengths = torch.randint(1, S + 1, (B,), generator=g).to(torch.int32)
    lengths[0], lengths[-1] = 1, S
    ids = torch.randint(1000 if V > 2000 else 200, V, (B, S), generator=g)
    u, r = ops.mlm_draws(0xABCDEF12345 + B, B, S, V, DEV)
    u = u.cpu()
    # decision boundaries in the float32 neighbourhood of 0.15, 0.15*0.8 and 0.15*0.9, and one sample with no selection
    edge = torch.tensor([0.15, 0.12, 0.135], dtype=torch.float64)
    edge32 = torch.cat([edge.to(torch.float32), torch.nextafter(edge.to(torch.float32), torch.tensor(0.0)),
                        torch.nextafter(edge.to(torch.float32), torch.tensor(1.0)), torch.tensor([0.0, 0.99999994])])
    k = min(S, edge32.numel())
    u[1 % B, :k] = edge32[:k]
    u[2 % B] = 0.5                                          # nothing selected -> token 0 forced
    fam = torch.tensor([b % 5 for b in range(B)], dtype=torch.int32)
    out = ops.mlm_corrupt(ids.to(DEV), lengths.to(DEV), u.to(DEV), r, N, family=fam.to(DEV))
    torch.cuda.synchronize()
    r = r.cpu()
    n_tot, rows_ref, ids_ref = 0, [], []
    Lq = S + N + 3
    for b in range(B):
        n = int(lengths[b])
        toks, labs = DO.random_word(ids[b, :n].tolist(), _Replay(u[b].numpy(), r[b].numpy()), V)
        t_ref, l_ref, s_ref, n_ids = DO.assemble_sample(toks, labs, N, S)
        assert np.array_equal(out["input_txt"][b].cpu().numpy(), t_ref), b
        assert np.array_equal(out["txt_labels"][b].cpu().numpy(), l_ref), b
        assert np.array_equal(out["segment"][b].cpu().numpy(), s_ref), b
        assert int(out["n_ids"][b]) == n_ids
        assert out["desc"][b].tolist() == [int(fam[b]), N + 2, N + 2 + n_ids]
        nz = np.nonzero(l_ref != -100)[0]
        assert int(out["counts"][b]) == len(nz) >= 1
        rows_ref += [b * Lq + int(i) for i in nz]
        ids_ref += [int(l_ref[i]) for i in nz]
        n_tot += len(nz)
    assert int(out["n_labels"]) == n_tot
    assert out["label_rows"][:n_tot].tolist() == rows_ref and out["label_ids"][:n_tot].tolist() == ids_ref


def test_mlm_draws_are_counter_based_and_well_distributed():
    B, S, V = 64, 473, 30522
    u1, r1 = ops.mlm_draws(11, B, S, V, DEV)
    u2, r2 = ops.mlm_draws(11, B, S, V, DEV)
    u3, r3 = ops.mlm_draws(12, B, S, V, DEV)
    assert torch.equal(u1, u2) and torch.equal(r1, r2) and not torch.equal(u1, u3) and not torch.equal(r1, r3)
    assert float(u1.min()) >= 0.0 and float(u1.max()) < 1.0 and int(r1.min()) >= 0 and int(r1.max()) < V
    n = B * S
    assert abs(float((u1 < 0.15).float().mean()) - 0.15) < 4 * math.sqrt(0.15 * 0.85 / n)
    assert abs(float(u1.mean()) - 0.5) < 4 * math.sqrt(1 / 12 / n) and abs(float(r1.float().mean()) / V - 0.5) < 0.02
    # a prefix of a larger batch is the same stream (index = b*S + t only through the linear counter)
    u4, _ = ops.mlm_draws(11, 2 * B, S, V, DEV)
    assert torch.equal(u4[:B], u1)


def test_assemble_batch_feeds_the_fused_step_contract():
    """data.assemble_batch output == label_index / MaskDesc.make on the same tensors (host restatements)."""
    B, N, S, V = 9, 6, 40, 1024
    g = torch.Generator().manual_seed(3)
    lengths = torch.randint(1, S + 1, (B,), generator=g)
    ids = torch.randint(200, V, (B, S), generator=g)
    fams = ["full", "s2s", "bar", "noncross", "full", "s2s", "s2s", "full", "bar"]
    a = D.assemble_batch(ids.to(DEV), lengths.to(DEV), N, V, key=99, family=fams)
    rows, lids = D.label_index(a["txt_labels"])
    assert torch.equal(rows, a["label_rows"]) and torch.equal(lids, a["label_ids"])
    want = D.MaskDesc.make(fams, N, S, lengths + 1, DEV)
    assert torch.equal(want.desc, a["attn_desc"].desc) and a["attn_desc"].L == S + N + 3
    assert torch.equal(a["n_ids"].cpu(), lengths + 1)


# ------------------------------------------------------------------------------------------ row kernels
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,H", [(37, 128), (513, 768), (64, 1024), (10, 2048)])
def test_layernorm_fwd_bwd(dt, M, H):
    x = rnd((M, H), torch.float32, 31, 2.0) + 0.5
    g, b = rnd((H,), torch.float32, 32) * 0.1 + 1.0, rnd((H,), torch.float32, 33) * 0.1
    dy = rnd((M, H), dt, 34)
    y = torch.zeros((M, H), dtype=dt, device=DEV)
    mean, rstd = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
    eps = 1e-12
    ops.layernorm_fwd(x, g, b, y, mean, rstd, M, H, eps)
    xd = x.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (H,), gd, bd, eps)
    assert relerr(y, ref) < (1e-5 if dt == torch.float32 else 1e-2)
    dx = torch.zeros((M, H), dtype=dt, device=DEV)
    dg, db, cs = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    ops.layernorm_bwd(dy, x, mean, rstd, g, dx, dg, db, cs, M, H)
    (ref * dy.double()).sum().backward()
    tol = 1e-4 if dt == torch.float32 else 1e-2
    assert relerr(dx, xd.grad) < tol and relerr(dg, gd.grad) < tol and relerr(db, bd.grad) < tol
    assert relerr(cs, dx.double().sum(0)) < (1e-4 if dt == torch.float32 else 2e-2)


@pytest.mark.parametrize("variant", [0, 2, 3])
@pytest.mark.parametrize("M,H", [(5000, 768), (37, 256), (700, 512), (3000, 1024)])
def test_layernorm_bwd_kernel_variants(variant, M, H):
    """The prefetching LayerNorm backward (16 / 4 / 8 waves per block; 16-bit rows of up to 768 elements take the wide blocks, others
    fall back) against the one-row-at-a-time kernel: dx and the dropped copy bit for bit, the column sums to f32 summation order."""
    f16 = torch.float16
    x = (rnd((M, H), torch.float32, 31, 2.0) + 0.5).to(f16)
    g = rnd((H,), torch.float32, 32) * 0.1 + 1.0
    dy = rnd((M, H), f16, 34)
    mean, rstd = x.float().mean(1), 1.0 / torch.sqrt(x.float().var(1, unbiased=False) + 1e-12)
    us = torch.tensor([0.25], device=DEV)
    out = {}
    try:
        for v in (1, variant):
            ops.set_rowops_variant(v)
            dx, dxd = torch.zeros((M, H), dtype=f16, device=DEV), torch.zeros((M, H), dtype=f16, device=DEV)
            dg, db, cs = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
            ops.layernorm_bwd(dy, x, mean, rstd, g, dx, dg, db, cs, M, H, dx_drop=dxd, p_drop=0.1, drop_key=77, unscale=us)
            out[v] = (dx, dxd, dg, db, cs)
    finally:
        ops.set_rowops_variant(0)
    a, b = out[1], out[variant]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for i in (2, 3, 4):
        assert relerr(b[i], a[i]) < 1e-5
    xd = x.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (H,), g.double(), None, 1e-12)
    (ref * dy.double()).sum().backward()
    assert relerr(b[0], xd.grad) < 2e-3
    assert relerr(b[4], 0.25 * b[1].double().sum(0)) < 1e-3


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_embed_fwd_bwd(dt):
    B, N, T, H, V, maxpos = 3, 5, 30, 128, 1024, 512
    Lq = N + T + 2
    g_ = torch.Generator().manual_seed(41)
    cls_tok = torch.full((B,), 101, dtype=torch.int64, device=DEV)
    sep_tok = torch.full((B,), 102, dtype=torch.int64, device=DEV)
    txt = torch.randint(0, V, (B, T), generator=g_).to(DEV)
    txt[:, -5:] = 0                                     # pads collide on row 0 (atomic contention path)
    txt[:, 2:9] = 103                                   # [MASK]: the hot row the backward sums per block instead of row by row
    seg = torch.ones((B, T), dtype=torch.int64, device=DEV)
    pos = torch.sort(torch.randperm(256, generator=g_)[:N])[0].view(1, N).expand(B, N).contiguous().to(DEV)
    E, P, Ty = rnd((V, H), dt, 42, 0.05), rnd((maxpos, H), dt, 43, 0.05), rnd((2, H), dt, 44, 0.05)
    imgproj = rnd((B, N, H), dt, 45, 0.5)
    g, b = rnd((H,), torch.float32, 46) * 0.1 + 1.0, rnd((H,), torch.float32, 47) * 0.1
    x0 = torch.zeros((B, Lq, H), dtype=dt, device=DEV)
    pre = torch.zeros((B, Lq, H), device=DEV)
    mean, rstd = torch.zeros(B * Lq, device=DEV), torch.zeros(B * Lq, device=DEV)
    dtn = 0 if dt == torch.float32 else 1
    ops.embed_fwd(dtn, cls_tok, txt, seg, pos, sep_tok, imgproj, E, P, Ty, g, b, x0, pre, mean, rstd, B, N, T, H, V, maxpos, 1e-12)
    Ed, Pd, Td, Id = [t.double().requires_grad_(True) for t in (E, P, Ty, imgproj)]
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    z = torch.zeros((B, 1), dtype=torch.int64, device=DEV)
    emb = lambda ids: torch.nn.functional.embedding(ids, Ed, padding_idx=0)      # HF BertEmbeddings: padding_idx = 0
    rows = torch.cat([emb(cls_tok)[:, None] + Pd[z] + Td[z], Id + Pd[pos] + Td[0][None, None],
                      emb(sep_tok)[:, None] + Pd[z] + Td[z], emb(txt) + Pd[torch.arange(T, device=DEV)][None] + Td[seg]], 1)
    ref = torch.nn.functional.layer_norm(rows, (H,), gd, bd, 1e-12)
    assert relerr(x0, ref) < (1e-5 if dt == torch.float32 else 1e-2)
    dx0 = rnd((B, Lq, H), dt, 48)
    dE, dP, dTy = torch.zeros((V, H), device=DEV), torch.zeros((maxpos, H), device=DEV), torch.zeros((2, H), device=DEV)
    dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    dimg = torch.zeros((B, N, H), dtype=dt, device=DEV)
    ops.embed_bwd(dtn, dx0, pre, mean, rstd, g, cls_tok, txt, seg, pos, sep_tok, dE, dP, dTy, dg, db, dimg, B, N, T, H, V, maxpos)
    (ref * dx0.double()).sum().backward()
    tol = 1e-4 if dt == torch.float32 else 1e-2
    for got, want in ((dE, Ed.grad), (dP, Pd.grad), (dTy, Td.grad), (dg, gd.grad), (db, bd.grad), (dimg, Id.grad)):
        assert relerr(got, want) < tol
    assert float(dE[0].abs().max()) == 0.0          # no look-up gradient for the [PAD] row
    assert relerr(dE[103], Ed.grad[103]) < tol and float(dE[103].abs().max()) > 0


def test_embed_without_image_positions_and_with_its_own_image_dropout():
    """args.img_postion false (cxrbert_origin.py:27-31): img_pos = NULL -> image rows = LN(imgproj + Ty[0]), no dP contribution from them.
    args.dropout_prob (cxrbert_origin.py:19): the image rows' dropout probability is separate from the text rows'."""
    B, N, T, H, V, maxpos = 3, 5, 30, 128, 1024, 512
    Lq = N + T + 2
    g_ = torch.Generator().manual_seed(41)
    cls_tok = torch.full((B,), 101, dtype=torch.int64, device=DEV)
    sep_tok = torch.full((B,), 102, dtype=torch.int64, device=DEV)
    txt = torch.randint(1, V, (B, T), generator=g_).to(DEV)
    seg = torch.ones((B, T), dtype=torch.int64, device=DEV)
    dt = torch.float32
    E, P, Ty = rnd((V, H), dt, 42, 0.05), rnd((maxpos, H), dt, 43, 0.05), rnd((2, H), dt, 44, 0.05)
    imgproj = rnd((B, N, H), dt, 45, 0.5)
    g, b = rnd((H,), torch.float32, 46) * 0.1 + 1.0, rnd((H,), torch.float32, 47) * 0.1
    x0 = torch.zeros((B, Lq, H), dtype=dt, device=DEV)
    pre = torch.zeros((B, Lq, H), device=DEV)
    mean, rstd = torch.zeros(B * Lq, device=DEV), torch.zeros(B * Lq, device=DEV)
    ops.embed_fwd(0, cls_tok, txt, seg, None, sep_tok, imgproj, E, P, Ty, g, b, x0, pre, mean, rstd, B, N, T, H, V, maxpos, 1e-12)
    Ed, Pd, Td, Id = [t.double().requires_grad_(True) for t in (E, P, Ty, imgproj)]
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    z = torch.zeros((B, 1), dtype=torch.int64, device=DEV)
    rows = torch.cat([Ed[cls_tok][:, None] + Pd[z] + Td[z], Id + Td[0][None, None],
                      Ed[sep_tok][:, None] + Pd[z] + Td[z], Ed[txt] + Pd[torch.arange(T, device=DEV)][None] + Td[seg]], 1)
    ref = torch.nn.functional.layer_norm(rows, (H,), gd, bd, 1e-12)
    assert relerr(x0, ref) < 1e-5
    dx0 = rnd((B, Lq, H), dt, 48)
    dE, dP, dTy = torch.zeros((V, H), device=DEV), torch.zeros((maxpos, H), device=DEV), torch.zeros((2, H), device=DEV)
    dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    dimg = torch.zeros((B, N, H), dtype=dt, device=DEV)
    ops.embed_bwd(0, dx0, pre, mean, rstd, g, cls_tok, txt, seg, None, sep_tok, dE, dP, dTy, dg, db, dimg, B, N, T, H, V, maxpos,
                  pad_token_id=-1)
    (ref * dx0.double()).sum().backward()
    for got, want in ((dE, Ed.grad), (dP, Pd.grad), (dTy, Td.grad), (dg, gd.grad), (db, bd.grad), (dimg, Id.grad)):
        assert relerr(got, want) < 1e-4
    assert float(dP[T:].abs().max()) == 0.0          # only the text / [CLS] / [SEP] positions 0..T-1 received anything
    # separate dropout probabilities: text rows p = 0.1, image rows p = 0.5 (same key, same element index)
    B2, N2, T2 = 8, 64, 64
    L2 = N2 + T2 + 2
    pos = torch.arange(N2, device=DEV).view(1, N2).expand(B2, N2).contiguous()
    txt2 = torch.randint(1, V, (B2, T2), generator=g_).to(DEV)
    seg2 = torch.ones((B2, T2), dtype=torch.int64, device=DEV)
    cls2, sep2 = torch.full((B2,), 101, dtype=torch.int64, device=DEV), torch.full((B2,), 102, dtype=torch.int64, device=DEV)
    img2 = rnd((B2, N2, H), dt, 49, 0.5)
    xs = []
    for pt, pi in ((0.0, 0.0), (0.1, 0.5)):
        x = torch.zeros((B2, L2, H), device=DEV)
        pre2 = torch.zeros((B2, L2, H), device=DEV)
        m2, r2 = torch.zeros(B2 * L2, device=DEV), torch.zeros(B2 * L2, device=DEV)
        ops.embed_fwd(0, cls2, txt2, seg2, pos, sep2, img2, E, P, Ty, g, b, x, pre2, m2, r2, B2, N2, T2, H, V, maxpos, 1e-12,
                      p_drop=pt, drop_key=1234, p_drop_img=pi)
        xs.append(x)
    clean, dropped = xs
    zi = float((dropped[:, 1:N2 + 1] == 0).float().mean())
    zt = float((dropped[:, N2 + 2:] == 0).float().mean())
    assert abs(zi - 0.5) < 0.01 and abs(zt - 0.1) < 0.01, (zi, zt)
    kept_i = dropped[:, 1:N2 + 1] != 0
    assert relerr(dropped[:, 1:N2 + 1][kept_i], clean[:, 1:N2 + 1][kept_i] * 2.0) < 1e-5            # survivors scaled by 1 / (1 - 0.5)
    # the backward regenerates the same two masks
    dxo = torch.ones((B2, L2, H), device=DEV)
    dEa, dPa, dTa = torch.zeros((V, H), device=DEV), torch.zeros((maxpos, H), device=DEV), torch.zeros((2, H), device=DEV)
    dga, dba = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    dimg2 = torch.zeros((B2, N2, H), device=DEV)
    ops.embed_bwd(0, dxo, pre2, m2, r2, g, cls2, txt2, seg2, pos, sep2, dEa, dPa, dTa, dga, dba, dimg2, B2, N2, T2, H, V, maxpos,
                  p_drop=0.1, drop_key=1234, p_drop_img=0.5)
    keep = (dropped != 0).double()
    keep[:, 1:N2 + 1] *= 2.0
    keep[:, :1] *= 1.0 / 0.9
    keep[:, N2 + 1:] *= 1.0 / 0.9
    assert relerr(dba, keep.sum((0, 1))) < 2e-3          # dbeta = column sums of the masked, rescaled incoming gradient


@pytest.mark.parametrize("V,ld", [(2, 2), (1000, 1000), (30522, 30528)])
@pytest.mark.parametrize("ddt", [torch.float32, torch.bfloat16])
def test_cross_entropy_fused(V, ld, ddt):
    R = 37
    logits = torch.zeros((R, ld), device=DEV)
    logits[:, :V] = rnd((R, V), torch.float32, 51, 2.0)
    g_ = torch.Generator().manual_seed(52)
    labels = torch.randint(0, V, (R,), generator=g_)
    labels[::5] = -100
    labels = labels.to(DEV)
    logits[3, int(labels[3])] = 50.0                     # a guaranteed argmax hit
    out = torch.zeros(3, device=DEV)
    ldd = (V + 7) // 8 * 8
    dl = torch.full((R, ldd), 3.0, dtype=ddt, device=DEV)
    scale = torch.tensor([0.125], device=DEV)
    ops.ce_fwd_bwd(logits, ld, labels.to(torch.int32), R, V, out, dl, ldd, grad_scale_dev=scale)
    ld_ = logits[:, :V].double().requires_grad_(True)
    nll = torch.nn.functional.cross_entropy(ld_, labels, ignore_index=-100, reduction="sum")
    nlab = int((labels != -100).sum())
    corr = int(((ld_.argmax(-1) == labels) & (labels != -100)).sum())
    assert abs(float(out[0]) - float(nll)) < 1e-3 * max(1.0, abs(float(nll)))
    assert int(out[1]) == nlab and int(out[2]) == corr and corr >= 1
    (nll * 0.125).backward()
    assert relerr(dl[:, :V], ld_.grad) < (1e-5 if ddt == torch.float32 else 1e-2)
    assert (dl[:, V:] == 0).all()


def test_gather_scatter_colsum_cast_add():
    M, H, R = 300, 768, 41
    for dt in (torch.float32, torch.bfloat16):
        src = rnd((M, H), dt, 61)
        rows = torch.randperm(M)[:R].to(torch.int32).to(DEV)
        dst = torch.zeros((R, H), dtype=dt, device=DEV)
        ops.gather_rows(src, H, rows, R, H, dst, H)
        assert torch.equal(dst, src[rows.long()])
        back = torch.zeros((M, H), dtype=dt, device=DEV)
        ops.scatter_rows(dst, H, rows, R, H, back, H)
        assert torch.equal(back[rows.long()], dst)
        untouched = torch.ones(M, dtype=torch.bool, device=DEV)
        untouched[rows.long()] = False
        assert (back[untouched] == 0).all()
        cs = torch.ones(H, device=DEV)
        ops.colsum(src, H, M, H, cs, accumulate=True)
        assert relerr(cs, src.double().sum(0) + 1.0) < 1e-4
        c = torch.zeros((M, H), dtype=dt, device=DEV)
        ops.add(src, src, c, M * H)
        assert torch.equal(c, (src.float() * 2).to(dt))
    a = rnd((1001,), torch.float32, 62)
    b = torch.zeros(1001, dtype=torch.bfloat16, device=DEV)
    ops.cast(a, b, 1001)
    assert torch.equal(b, a.to(torch.bfloat16))
    z = rnd((64, 30), torch.float32, 63)
    zp = torch.full((64, 32), 5.0, dtype=torch.bfloat16, device=DEV)
    ops.cast2d(z, 30, zp, 32, 64, 30)
    assert torch.equal(zp[:, :30], z.to(torch.bfloat16)) and (zp[:, 30:] == 0).all()
    dy, zz = rnd((16, 64), torch.float32, 64), rnd((16, 64), torch.float32, 65)
    o = torch.zeros_like(dy)
    ops.dact(0, dy, zz, o, 16 * 64)
    assert relerr(o, dy.double() * dgelu(zz.double())) < 1e-5
    ops.dact(1, dy, torch.tanh(zz), o, 16 * 64)
    assert relerr(o, dy.double() * (1 - torch.tanh(zz.double()) ** 2)) < 1e-5


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,cols", [(768, 3072), (2304, 768), (65, 130), (1, 7)])
def test_transpose(dt, rows, cols):
    x = rnd((rows, cols + 3), dt, 77)
    y = torch.full((cols, rows + 5), 9.0, dtype=dt, device=DEV)
    ops.transpose(x, y, rows, cols, lds=cols + 3, ldd=rows + 5)
    assert torch.equal(y[:, :rows], x[:, :cols].t()) and bool((y[:, rows:] == 9.0).all())


def test_gather_scatter_skip_negative_rows():
    src = rnd((6, 8), torch.bfloat16, 3)
    rows = torch.tensor([4, -1, 0], dtype=torch.int32, device=DEV)
    dst = torch.full((3, 8), 9.0, dtype=torch.bfloat16, device=DEV)
    ops.gather_rows(src, 8, rows, 3, 8, dst, 8)
    assert torch.equal(dst[0], src[4]) and torch.equal(dst[2], src[0]) and bool((dst[1] == 0).all())
    out = torch.full((6, 8), 5.0, dtype=torch.bfloat16, device=DEV)
    ops.scatter_rows(dst, 8, rows, 3, 8, out, 8, accumulate=False)
    assert torch.equal(out[4], dst[0]) and torch.equal(out[0], dst[2]) and bool((out[[1, 2, 3, 5]] == 5.0).all())


def test_adamw_known_answer(golden_dir):
    """HF-AdamW KAT (tests/golden/adamw.npz, computed with python floats) on the fused kernel."""
    import os
    z = np.load(os.path.join(golden_dir, "adamw.npz"))
    lr, b1, b2, eps, wd = [float(x) for x in z["hyper"]]
    p = torch.tensor(z["p0"], dtype=torch.float32, device=DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    sh = torch.zeros(16, dtype=torch.bfloat16, device=DEV)
    for t in range(1, 4):
        g = torch.tensor(z["grads"][t - 1], dtype=torch.float32, device=DEV)
        ops.adamw_step(p, g, m, v, sh, 16, lr, b1, b2, eps, wd, t)
        assert np.abs(p.cpu().numpy() - z["p"][t - 1]).max() < 2e-6
    assert torch.equal(sh, p.to(torch.bfloat16))


# ------------------------------------------------------------------------------------------ f16 forward operands
# The forward products of the 16-bit path read f16-encoded operands (MV_F16) and write a forward activation twice: the
# f16 copy for the next forward product and the bf16 copy for the backward's gradient products (include/medvill.h).
@pytest.mark.parametrize("impl", ["mfma", "mfma256k64", "simple"])
@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (130, 70, 200), (1024, 768, 768), (64, 2, 768), (515, 2304, 264), (1030, 3072, 768),
                                   (2100, 3072, 64), (700, 520, 136)])
def test_gemm_f16_operands(impl, M, N, K):
    ops.set_impl(1 if impl == "simple" else 0)
    ops.set_gemm_variant(*VARIANT.get(impl, (0, 0)))
    try:
        pad = lambda n: (n + 7) // 8 * 8
        lda, ldb = pad(K) + 8, pad(K) + 16
        a, b = rnd((M, lda), torch.float16, 1), rnd((N, ldb), torch.float16, 2)
        a[:, K:] = 0
        b[:, K:] = 0
        c = torch.full((M, N + 3), 7.0, dtype=torch.float32, device=DEV)
        ops.gemm(a, b, c, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=N + 3)
        ref = a[:, :K].double() @ b[:, :K].double().t()
        assert relerr(c[:, :N], ref) < 2e-5 * math.sqrt(K)
        assert (c[:, N:] == 7.0).all()
    finally:
        ops.set_impl(0)
        ops.set_gemm_variant(0, 0)


@pytest.mark.parametrize("impl", ["mfma", "mfma256k64"])
@pytest.mark.parametrize("epi", [EPI_BIAS, EPI_BIAS_RES, EPI_BIAS_GELU_D, EPI_BIAS_TANH])
@pytest.mark.parametrize("M,N,K", [(256, 384, 128), (200, 132, 72), (300, 1024, 768)])
def test_gemm_f16_dual_outputs(impl, epi, M, N, K):
    """C (f16) and C3 (bf16) are the same fp32 result rounded to each encoding; the residual operand may be f16."""
    a, b = rnd((M, K), torch.float16, 3, 0.5), rnd((N, K), torch.float16, 4, 0.5)
    bias, r = rnd((N,), torch.float32, 5), rnd((M, N), torch.float16, 6)
    c = torch.zeros((M, N), dtype=torch.float16, device=DEV)
    c2 = torch.zeros((M, N), dtype=torch.float16, device=DEV)
    c3 = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
    ops.set_gemm_variant(*VARIANT.get(impl, (0, 0)))
    try:
        ops.gemm(a, b, c, M=M, N=N, K=K, bias=bias, epi=epi, r=r, c2=c2, c3=c3)
    finally:
        ops.set_gemm_variant(0, 0)
    y = a.double() @ b.double().t() + bias.double()
    ref = {EPI_BIAS: y, EPI_BIAS_RES: y + r.double(), EPI_BIAS_GELU_D: gelu(y), EPI_BIAS_TANH: torch.tanh(y)}[epi]
    assert relerr(c, ref) < 1.5e-3 and relerr(c3, ref) < 1e-2
    if epi == EPI_BIAS_GELU_D:
        assert relerr(c2, dgelu(y)) < 1.5e-3
    # both copies come from the same accumulator: the bf16 copy equals the f16 copy's value up to one bf16 rounding
    assert float((c3.double() - c.double()).abs().max()) <= float(ref.abs().max()) * 2.0 ** -8


@pytest.mark.parametrize("impl", ["mfma", "simple"])
@pytest.mark.parametrize("fam,B,A,N,S", [("full", 2, 2, 16, 45), ("s2s", 2, 4, 36, 473), ("noncross", 2, 2, 16, 45), ("mixed", 4, 3, 36, 150),
                                         ("bar", 3, 2, 5, 29)])
def test_attention_fwd_f16(impl, fam, B, A, N, S):
    """f16 q / k / v / P operands: 8x tighter than the bf16 kernel on the same inputs; the bf16 context copy and the
    backward on the bf16 copy of qkv stay consistent with it."""
    ops.set_impl(1 if impl == "simple" else 0)
    try:
        dh, Lq = 64, N + S + 3
        H = A * dh
        mask = _masks(fam, B, N, S).to(DEV)
        qkv = rnd((B, Lq, 3 * H), torch.float16, 21, 1.0)
        bits = torch.zeros((B, Lq, (Lq + 31) // 32), dtype=torch.int32, device=DEV)
        tinfo = torch.zeros((B, (Lq + 63) // 64, (Lq + 63) // 64), dtype=torch.uint8, device=DEV)
        ops.mask_pack(mask, bits, tinfo)
        ctx = torch.zeros((B, Lq, H), dtype=torch.float16, device=DEV)
        ctx_b = torch.zeros((B, Lq, H), dtype=torch.bfloat16, device=DEV)
        lse = torch.zeros((B, A, Lq), dtype=torch.float32, device=DEV)
        ops.attn_fwd(qkv.view(B * Lq, 3 * H), bits, tinfo, ctx, lse, B, Lq, A, dh, ctx_bf16=ctx_b)
        rctx, rlse = attn_ref(qkv.double(), mask, A)
        assert relerr(ctx, rctx) < 2e-3
        assert float((lse.double() - rlse).abs().max()) < 1e-3
        assert relerr(ctx_b, rctx) < 1e-2 and float((ctx_b.double() - ctx.double()).abs().max()) <= float(rctx.abs().max()) * 2.0 ** -8
        if impl == "mfma":
            # the gradient products read the bf16 copies (as the engine does): same tolerance as the all-bf16 kernels
            qkv_b, dctx = qkv.to(torch.bfloat16), rnd((B, Lq, H), torch.bfloat16, 22, 1.0)
            dqkv = torch.zeros((B, Lq, 3 * H), dtype=torch.bfloat16, device=DEV)
            delta = torch.zeros((B, A, Lq), dtype=torch.float32, device=DEV)
            ops.attn_bwd(qkv_b.view(B * Lq, 3 * H), ctx_b, dctx, lse, bits, tinfo, dqkv, delta, B, Lq, A, dh)
            qd = qkv.double().requires_grad_(True)
            (attn_ref(qd, mask, A)[0] * dctx.double()).sum().backward()
            assert relerr(dqkv, qd.grad) < 2e-2
    finally:
        ops.set_impl(0)


def test_layernorm_embed_adamw_cast_f16():
    M, H = 513, 768
    x = rnd((M, H), torch.float32, 31, 2.0) + 0.5
    g, b = rnd((H,), torch.float32, 32) * 0.1 + 1.0, rnd((H,), torch.float32, 33) * 0.1
    y, yb = torch.zeros((M, H), dtype=torch.float16, device=DEV), torch.zeros((M, H), dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
    ops.layernorm_fwd(x, g, b, y, mean, rstd, M, H, 1e-12, y_bf16=yb)
    ref = torch.nn.functional.layer_norm(x.double(), (H,), g.double(), b.double(), 1e-12)
    assert relerr(y, ref) < 1e-3 and relerr(yb, ref) < 1e-2
    with pytest.raises(RuntimeError, match="MV_E_DTYPE"):      # the bf16 copy exists for f16 outputs only
        ops.layernorm_fwd(x, g, b, yb, mean, rstd, M, H, 1e-12, y_bf16=torch.zeros_like(yb))
    # casts between the three encodings
    src = rnd((1000, 7), torch.float32, 40)
    h = torch.zeros((1000, 7), dtype=torch.float16, device=DEV)
    ops.cast(src, h, src.numel())
    assert torch.equal(h, src.to(torch.float16))
    bb = torch.zeros((1000, 7), dtype=torch.bfloat16, device=DEV)
    ops.cast(h, bb, h.numel())
    assert torch.equal(bb, h.to(torch.bfloat16))
    f = torch.zeros((1000, 7), dtype=torch.float32, device=DEV)
    ops.cast(h, f, h.numel())
    assert torch.equal(f, h.float())
    # gather of 16-bit rows is encoding-agnostic
    rows = torch.tensor([5, 0, -1, 999], dtype=torch.int32, device=DEV)
    h8 = rnd((1000, 8), torch.float16, 41)
    out = torch.ones((4, 8), dtype=torch.float16, device=DEV)
    ops.gather_rows(h8, 8, rows, 4, 8, out, 8)
    assert torch.equal(out[0], h8[5]) and torch.equal(out[3], h8[999]) and bool((out[2] == 0).all())
    # AdamW refreshes both shadows
    n = 1003
    p, gr = rnd((n,), torch.float32, 50), rnd((n,), torch.float32, 51)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    sb, sh = torch.zeros(n, dtype=torch.bfloat16, device=DEV), torch.zeros(n, dtype=torch.float16, device=DEV)
    ops.adamw_step(p, gr, m, v, sb, n, 1e-3, 0.9, 0.999, 1e-6, 0.0, 1, shadow_f16=sh)
    assert torch.equal(sb, p.to(torch.bfloat16)) and torch.equal(sh, p.to(torch.float16))


# ------------------------------------------------------------------------------------------ f16 gradients under a loss scale
# No counterpart in the reference (fp32 gradients, train_origin.py:129-131).  Contract (include/medvill.h): the loss gradients
# enter the 16-bit chain multiplied by S (mv_ce_fwd_bwd), every writer of an f32 parameter gradient multiplies by 1/S, both
# read from the device; mv_count_nonfinite + mv_scaler_update turn an overflow into a skipped optimizer step and a smaller S.
@pytest.mark.parametrize("splitk", [1, 0, 4])
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_weight_gradient_gemm_unscales_with_alpha(dt, splitk):
    M, N, K = 768, 512, 4100                                   # dW [M,N] over K tokens (ragged last K tile)
    S = 1024.0
    dy, x = rnd((K, M), torch.float32, 1, 0.01), rnd((K, N), torch.float32, 2)
    dy_s = (dy * S).to(dt)
    x16 = x.to(dt)
    alpha = torch.tensor([1.0 / S], dtype=torch.float32, device=DEV)
    c = torch.full((M, N), 3.0, dtype=torch.float32, device=DEV)
    ws = torch.empty(8 * M * N, dtype=torch.float32, device=DEV) if splitk != 1 else None
    ops.gemm(dy_s, x16, c, ta=True, tb=True, M=M, N=N, K=K, lda=M, ldb=N, splitk=splitk, ws=ws, alpha=alpha, accumulate=True)
    ref = 3.0 + (dy_s.double().t() @ x16.double()) / S
    assert relerr(c, ref) < 1e-5 * math.sqrt(K)
    with pytest.raises(RuntimeError, match="MV_E_ARG"):        # alpha belongs to plain f32 products only
        ops.gemm(dy_s, x16, torch.zeros((M, N), dtype=dt, device=DEV), ta=True, tb=True, M=M, N=N, K=K, lda=M, ldb=N, alpha=alpha)


def test_row_kernels_apply_loss_scale_and_unscale():
    M, H, V = 300, 256, 1000
    S = 4096.0
    ls = torch.tensor([S], dtype=torch.float32, device=DEV)
    us = torch.tensor([1.0 / S], dtype=torch.float32, device=DEV)
    # cross-entropy: f16 gradient = (softmax - onehot) * grad_scale * S
    logits = rnd((M, V), torch.float32, 1)
    labels = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(2)).to(torch.int32).to(DEV)
    labels[::7] = -100
    out = torch.zeros(3, dtype=torch.float32, device=DEV)
    dl = torch.zeros((M, V), dtype=torch.float16, device=DEV)
    ops.ce_fwd_bwd(logits, V, labels, M, V, out, dl, V, grad_scale=1.0 / M, loss_scale_dev=ls)
    p = torch.softmax(logits.double(), -1)
    oh = torch.zeros_like(p)
    keep = labels >= 0
    oh[keep, labels[keep].long()] = 1.0
    ref = (p - oh) / M * S * keep.view(-1, 1)
    assert float((dl.double() - ref).abs().max()) < 2e-3 * float(ref.abs().max())
    # column sums and LayerNorm backward: what reaches the f32 gradients is divided by S again
    x = (rnd((M, H), torch.float32, 3, 0.01) * S).to(torch.float16)
    cs = torch.zeros(H, dtype=torch.float32, device=DEV)
    ops.colsum(x, H, M, H, cs, accumulate=True, unscale=us)
    assert relerr(cs, x.double().sum(0) / S) < 1e-5
    pre = rnd((M, H), torch.float32, 4).to(torch.float16)
    mean, rstd = pre.float().mean(1), 1.0 / torch.sqrt(pre.float().var(1, unbiased=False) + 1e-12)
    g = rnd((H,), torch.float32, 5) + 1.0
    dx = torch.zeros((M, H), dtype=torch.float16, device=DEV)
    dg, db, cs2 = (torch.zeros(H, dtype=torch.float32, device=DEV) for _ in range(3))
    ops.layernorm_bwd(x, pre, mean, rstd, g, dx, dg, db, cs2, M, H, unscale=us)
    xh = (pre.double() - mean.double().view(-1, 1)) * rstd.double().view(-1, 1)
    assert relerr(dg, (x.double() * xh).sum(0) / S) < 1e-4 and relerr(db, x.double().sum(0) / S) < 1e-4
    gd = x.double() * g.double()
    dx_ref = rstd.double().view(-1, 1) * (gd - gd.mean(1, keepdim=True) - xh * (gd * xh).mean(1, keepdim=True))
    assert relerr(dx, dx_ref) < 2e-3                                        # the chain itself stays scaled
    assert relerr(cs2, dx_ref.sum(0) / S) < 2e-3


def test_dynamic_loss_scale_state_machine_and_skipped_adamw_step():
    st = torch.tensor([1024.0, 1.0 / 1024.0, 0, 0, 0, 0, 0, 0], dtype=torch.float32, device=DEV)
    n = 1000
    p0 = rnd((n,), torch.float32, 1)
    p, g = p0.clone(), rnd((n,), torch.float32, 2)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    sh = torch.zeros(n, dtype=torch.float16, device=DEV)
    # clean step: t = 1, applied
    ops.count_nonfinite(g, st[6:7])
    ops.scaler_update(st, growth_interval=2)
    assert st.tolist()[:7] == [1024.0, 1.0 / 1024.0, 1.0, 0.0, 1.0, 0.0, 0.0]
    ops.adamw_step(p, g, m, v, None, n, 1e-2, 0.9, 0.999, 1e-6, 0.0, 77, shadow_f16=sh, scaler_state=st)      # host step ignored
    ref = p0 - 1e-2 * math.sqrt(1 - 0.999) / (1 - 0.9) * (0.1 * g) / (torch.sqrt(0.001 * g * g) + 1e-6)
    assert float((p - ref).abs().max()) < 1e-6 and torch.equal(sh, p.to(torch.float16))
    # overflowed step: nothing moves, the scale halves, t stays
    p1, m1, v1 = p.clone(), m.clone(), v.clone()
    gb = g.clone()
    gb[17], gb[500], gb[999] = float("inf"), float("nan"), float("-inf")
    ops.count_nonfinite(gb, st[6:7])
    assert float(st[6]) == 3.0
    ops.scaler_update(st, growth_interval=2)
    assert st.tolist()[:7] == [512.0, 1.0 / 512.0, 0.0, 1.0, 1.0, 1.0, 0.0]
    ops.adamw_step(p, gb, m, v, None, n, 1e-2, 0.9, 0.999, 1e-6, 0.0, 78, shadow_f16=sh, scaler_state=st)
    assert torch.equal(p, p1) and torch.equal(m, m1) and torch.equal(v, v1)
    # two clean steps: the scale grows back, t = 3
    for _ in range(2):
        ops.count_nonfinite(g, st[6:7])
        ops.scaler_update(st, growth_interval=2)
    assert st.tolist()[:7] == [1024.0, 1.0 / 1024.0, 0.0, 0.0, 3.0, 1.0, 0.0]
    ops.scaler_update(st, growth_interval=1, max_scale=1024.0)            # capped
    assert float(st[0]) == 1024.0


# ------------------------------------------------------------------------------------------ bias gradients from partial column sums
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("epi", [EPI_MUL, EPI_BIAS, EPI_NONE])
@pytest.mark.parametrize("M,N,K", [(700, 512, 192), (2048, 3072, 768), (130, 256, 64)])
def test_gemm_writes_partial_column_sums(dt, epi, M, N, K):
    """colsum_part: the 256x256 kernel leaves sum over each 128-row tile half of every output column (the dz GEMM's bias gradient
    without re-reading dz); mv_colsum_partials folds the rows, un-scaled."""
    a, b = rnd((M, K), dt, 3, 0.5), rnd((N, K), dt, 4, 0.5)
    r, bias = rnd((M, N), torch.float16, 6), rnd((N,), torch.float32, 7)
    c = torch.zeros((M, N), dtype=dt, device=DEV)
    P = 2 * ((M + 255) // 256)
    part = torch.full((P, N), float("nan"), dtype=torch.float32, device=DEV)
    ops.set_gemm_variant(2, 14)
    try:
        ops.gemm(a, b, c, M=M, N=N, K=K, epi=epi, r=r if epi == EPI_MUL else None, bias=bias if epi == EPI_BIAS else None, colsum_part=part)
        with pytest.raises(RuntimeError, match="MV_E_SHAPE"):           # a ragged last column tile: refused, never silently wrong
            ops.gemm(a, b[:N - 8], c[:, :N - 8], M=M, N=N - 8, K=K, ldc=N, colsum_part=part)
    finally:
        ops.set_gemm_variant(0, 0)
    y = a.double() @ b.double().t()
    ref = y * r.double() if epi == EPI_MUL else (y + bias.double() if epi == EPI_BIAS else y)
    assert relerr(c, ref) < (2e-3 if dt == torch.float16 else 1e-2)
    out = torch.full((N,), 2.0, dtype=torch.float32, device=DEV)
    us = torch.tensor([0.25], dtype=torch.float32, device=DEV)
    ops.colsum_partials(part, P, N, N, out, unscale=us)
    assert torch.isfinite(part).all()
    assert relerr(out, 2.0 + 0.25 * ref.sum(0)) < 1e-4
    with pytest.raises(RuntimeError, match="MV_E_SHAPE"):               # the 128x128 kernel has no such epilogue
        ops.set_gemm_variant(1, 0)
        try:
            ops.gemm(a, b, c, M=M, N=N, K=K, colsum_part=part)
        finally:
            ops.set_gemm_variant(0, 0)


# ------------------------------------------------------------------------------------------ attention dropout: keep-bits
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("fam,B,A,N,S,packed", [("full", 2, 2, 16, 45, False), ("s2s", 3, 4, 36, 473, False), ("bar", 2, 2, 36, 221, False),
                                                 ("full", 4, 3, 36, 300, True), ("mixed", 4, 2, 36, 150, True), ("full", 1, 12, 100, 665, True)])
def test_attention_dropout_keep_bits_every_tile_class_and_packed_rows(dt, fam, B, A, N, S, packed):
    """The MFMA kernels' dropout (select masks from scalar loads in the forward / dQ kernels, per-key dwords through the LDS-DMA ring
    in the dK/dV kernel) on every tile class -- fully visible, mixed, ragged tail, skipped -- over padded and packed rows, against an
    fp64 restatement that applies the decoded keep-bits; blocks beyond a sample's packed length are never written."""
    dh, Lq, p, key = 64, N + S + 3, 0.1, 0xABCDEF12345
    H = A * dh
    g = torch.Generator().manual_seed(9)
    n_ids = torch.randint(2, S + 2, (B,), generator=g)
    fams = [("s2s" if i % 2 else "full") for i in range(B)] if fam == "mixed" else fam
    desc = D.MaskDesc.make(fams, N, S, n_ids, DEV)
    mask = (D.mixed_mask(N, S, n_ids, torch.arange(B) % 2 == 1) if fam == "mixed" else D.build_mask(fam, N, S, n_ids)).to(DEV)
    bits = torch.zeros((B, Lq, (Lq + 31) // 32), dtype=torch.int32, device=DEV)
    tinfo = torch.zeros((B, (Lq + 63) // 64, (Lq + 63) // 64), dtype=torch.uint8, device=DEV)
    ops.mask_build(desc.desc, B, Lq, bits, tinfo)
    cu, M, lens = None, B * Lq, [Lq] * B
    if packed:
        cu, rowmap, inv = ops.pack_plan(desc.desc, B, Lq)
        M = int(cu[-1])
        lens = (cu[1:] - cu[:-1]).tolist()
    db = torch.full((ops.dropbits_numel(B, Lq, A),), 0x5A5A5A5A, dtype=torch.int32, device=DEV)
    ops.attn_dropmask(p, key, B, Lq, A, db, cu=cu)
    w = db.view(B, A, (Lq + 31) // 32, (Lq + 63) // 64, 64)
    for b in range(B):                                       # unwritten exactly where no query or no key exists
        for qb in range(w.shape[2]):
            for kt in range(w.shape[3]):
                untouched = bool((w[b, :, qb, kt] == 0x5A5A5A5A).all())
                assert untouched == (qb * 32 >= lens[b] or kt * 64 >= lens[b]), (b, qb, kt)
    keep = ops.attn_keep_mask(db, B, Lq, A).double() * (65536.0 / (65536.0 - 6554.0))
    qkv, dctx = rnd((M, 3 * H), dt, 51), rnd((M, H), dt, 52)
    ctx = torch.zeros((M, H), dtype=dt, device=DEV)
    lse = torch.zeros((B, A, Lq), dtype=torch.float32, device=DEV)
    ops.attn_fwd(qkv, bits, tinfo, ctx, lse, B, Lq, A, dh, p_drop=p, cu=cu, total_rows=M, dropbits=db)
    dq = torch.zeros((M, 3 * H), dtype=dt, device=DEV)
    delta = torch.zeros((B, A, Lq), dtype=torch.float32, device=DEV)
    ops.attn_bwd(qkv, ctx, dctx, lse, bits, tinfo, dq, delta, B, Lq, A, dh, p_drop=p, cu=cu, total_rows=M, dropbits=db)
    # fp64 restatement sample by sample on the rows that exist
    qd = qkv.double().requires_grad_(True)
    outs, row0 = [], 0
    for b in range(B):
        n = lens[b]
        x = qd[row0:row0 + n]
        q, k, v = [t.view(n, A, dh).permute(1, 0, 2) for t in x.split(H, dim=-1)]
        add = (1.0 - mask[b, :n, :n].double()) * -10000.0
        pr = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh) + add[None], -1) * keep[b, :, :n, :n]
        outs.append((pr @ v).permute(1, 0, 2).reshape(n, H))
        row0 += n
    rctx = torch.cat(outs)
    assert relerr(ctx, rctx) < (4e-3 if dt == torch.float16 else 2e-2)
    (rctx * dctx.double()).sum().backward()
    assert relerr(dq, qd.grad) < (6e-3 if dt == torch.float16 else 2.5e-2)


# ------------------------------------------------------------------------------------------ last layer: consumed rows first, query limits
def test_tail_perm_orders_the_consumed_rows_first():
    g = torch.Generator().manual_seed(3)
    B, Lq = 7, 515
    lens = torch.randint(40, Lq + 1, (B,), generator=g)
    lens[3] = Lq
    cu = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)]).to(torch.int32)
    M = int(cu[-1])
    sel = []
    for b in range(B):
        n = int(lens[b])
        k = int(torch.randint(1, min(n, 90), (1,), generator=g))
        sel.append(int(cu[b]) + torch.randperm(n, generator=g)[:k])
    sel = torch.cat(sel)
    sel = sel[torch.randperm(sel.numel(), generator=g)].to(torch.int32)          # any order: the heads' order is not the row order
    perm, newpos, qlim, sel_new = ops.tail_perm(cu.to(DEV), B, Lq, sel.to(DEV), M)
    perm, newpos, qlim, sel_new = perm.cpu().long(), newpos.cpu().long(), qlim.cpu(), sel_new.cpu().long()
    assert torch.equal(newpos[perm], torch.arange(M)) and torch.equal(perm[newpos], torch.arange(M))
    flag = torch.zeros(M, dtype=torch.bool)
    flag[sel.long()] = True
    for b in range(B):
        lo, hi = int(cu[b]), int(cu[b + 1])
        nq = int(flag[lo:hi].sum())
        assert int(qlim[b]) == nq
        old = perm[lo:hi]
        assert bool((old >= lo).all()) and bool((old < hi).all())
        assert bool(flag[old[:nq]].all()) and not bool(flag[old[nq:]].any())
        assert torch.equal(old[:nq], old[:nq].sort().values) and torch.equal(old[nq:], old[nq:].sort().values)
    assert torch.equal(sel_new, newpos[sel.long()])


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_attention_query_limits_equal_the_full_run_on_the_limited_rows(dt, p):
    """qlim: only the first qlim[b] rows of a sample are queries.  The context of those rows, and dqkv when the other rows' dctx is
    zero, must be what the unrestricted kernels give (the skipped work contributes exact zeros)."""
    B, A, N, S, dh = 5, 3, 36, 473, 64
    Lq, H = N + S + 3, A * dh
    g = torch.Generator().manual_seed(21)
    n_ids = torch.randint(2, S + 2, (B,), generator=g)
    n_ids[0] = S + 1
    desc = D.MaskDesc.make("full", N, S, n_ids, DEV)
    bits = torch.zeros((B, Lq, (Lq + 31) // 32), dtype=torch.int32, device=DEV)
    tinfo = torch.zeros((B, (Lq + 63) // 64, (Lq + 63) // 64), dtype=torch.uint8, device=DEV)
    ops.mask_build(desc.desc, B, Lq, bits, tinfo)
    cu, rowmap, inv = ops.pack_plan(desc.desc, B, Lq)
    M = int(cu[-1])
    lens = (cu[1:] - cu[:-1]).cpu()
    qlim = torch.tensor([1, 54, 64, 131, 0], dtype=torch.int32)
    qlim = torch.minimum(qlim, lens.to(torch.int32))
    qlim[0] = lens[0]                                      # one sample unrestricted, one with no query at all
    db = None
    if p > 0:
        db = torch.zeros((ops.dropbits_numel(B, Lq, A),), dtype=torch.int32, device=DEV)
        ops.attn_dropmask(p, 99, B, Lq, A, db, cu=cu)
    qkv, dctx = rnd((M, 3 * H), dt, 51), rnd((M, H), dt, 52)
    isq = torch.zeros(M, dtype=torch.bool)
    for b in range(B):
        isq[int(cu[b]):int(cu[b]) + int(qlim[b])] = True
    dctx[~isq.to(DEV)] = 0
    res = []
    for ql in (None, qlim.to(DEV)):
        ctx = torch.full((M, H), 7.0, dtype=dt, device=DEV)
        lse = torch.zeros((B, A, Lq), dtype=torch.float32, device=DEV)
        ops.attn_fwd(qkv, bits, tinfo, ctx, lse, B, Lq, A, dh, p_drop=p, cu=cu, total_rows=M, dropbits=db, qlim=ql)
        dq = torch.full((M, 3 * H), 3.0, dtype=dt, device=DEV)
        delta = torch.zeros((B, A, Lq), dtype=torch.float32, device=DEV)
        ops.attn_bwd(qkv, ctx, dctx, lse, bits, tinfo, dq, delta, B, Lq, A, dh, p_drop=p, cu=cu, total_rows=M, dropbits=db, qlim=ql)
        res.append((ctx, dq))
    (c0, d0), (c1, d1) = res
    m = isq.to(DEV)
    assert torch.equal(c0[m], c1[m])
    assert bool((c1[~m] == 7.0).all())                     # rows that are keys only: context untouched
    assert bool(torch.isfinite(d1.float()).all())
    assert float((d1.float() - d0.float()).abs().max()) <= 1e-6 * float(d0.float().abs().max()) + 0.0
    assert bool((d1[~m][:, :H] == 0).all())                # their dQ rows are zeros
