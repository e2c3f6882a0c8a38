"""CPU-reachable half of the C ABI: every entry point validates its arguments BEFORE it touches the HIP runtime, so the rejections
(null pointers, shapes and alignments the kernels do not support, dtype codes, workspaces that are too small) and the one HOST
function (mv_mask_verify_host) can be exercised without a GPU -- also under AddressSanitizer / UBSan (tools/asan_host_check.sh
builds the library's host side with -fsanitize=address,undefined and runs this file and tests/test_abi.py against it)."""
import ctypes as C

import numpy as np
import pytest

import medvill_amd  # noqa: F401
from medvill_amd import _lib

E_ARG, E_SHAPE, E_DTYPE, E_WS = -1, -2, -3, -4
F32, BF16, F16 = 0, 1, 2
P = 0x1000          # a non-null, 16-byte aligned address that is never dereferenced: every call below must return before a launch


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def gemm(lib, dtype=BF16, ta=0, tb=0, M=64, N=64, K=64, A=P, lda=64, B=P, ldb=64, Cp=P, ldc=64, c_dtype=BF16, bias=None, epi=0, R=None,
         ldr=64, r_dtype=BF16, C2=None, ldc2=64, C3=None, ldc3=64, c3_dtype=BF16, splitk=1, ws=None, ws_bytes=0, accumulate=0,
         alpha=None, cpart=None):
    return lib.mv_gemm(dtype, ta, tb, M, N, K, A, lda, B, ldb, Cp, ldc, c_dtype, bias, epi, R, ldr, r_dtype, C2, ldc2, C3, ldc3, c3_dtype,
                       splitk, ws, ws_bytes, accumulate, 0.0, 0, alpha, cpart, None)


def test_gemm_rejects(lib):
    assert gemm(lib, A=None) == E_ARG and gemm(lib, B=None) == E_ARG and gemm(lib, Cp=None) == E_ARG
    assert gemm(lib, M=0) == E_ARG and gemm(lib, K=-3) == E_ARG
    assert gemm(lib, dtype=7) == E_DTYPE and gemm(lib, c_dtype=-1) == E_DTYPE
    assert gemm(lib, C3=P, c3_dtype=F32) == E_DTYPE and gemm(lib, C3=P, ldc3=8) == E_DTYPE
    assert gemm(lib, epi=11) == E_ARG and gemm(lib, epi=-1) == E_ARG
    for epi in (1, 2, 3, 6, 7, 9, 10):
        assert gemm(lib, epi=epi) == E_ARG                     # bias missing
    for epi in (3, 4, 5, 8, 10):
        assert gemm(lib, epi=epi, bias=P) == E_ARG             # elementwise operand missing
    assert gemm(lib, epi=5, R=P, r_dtype=9) == E_ARG
    assert gemm(lib, epi=2, bias=P) == E_ARG and gemm(lib, epi=7, bias=P) == E_ARG          # second output missing
    assert gemm(lib, lda=8) == E_SHAPE and gemm(lib, ldb=8) == E_SHAPE and gemm(lib, ldc=8) == E_SHAPE
    assert gemm(lib, ta=1, M=128, lda=64) == E_SHAPE and gemm(lib, tb=1, N=128, ldb=64, ldc=128) == E_SHAPE
    assert gemm(lib, epi=5, R=P, ldr=8) == E_SHAPE
    assert gemm(lib, splitk=4, c_dtype=F32, epi=1, bias=P) == E_SHAPE and gemm(lib, splitk=4) == E_SHAPE       # split-K: plain f32 output only
    assert gemm(lib, accumulate=1) == E_SHAPE
    assert gemm(lib, splitk=4, c_dtype=F32) == E_WS
    assert gemm(lib, splitk=4, c_dtype=F32, ws=P, ws_bytes=4 * 64 * 64 * 4 - 1) == E_WS
    assert gemm(lib, alpha=P) == E_ARG                         # alpha: f32 output without an epilogue only


def test_mask_and_plan_rejects(lib):
    assert lib.mv_mask_pack(None, 3, 2, 64, P, P, None) == E_ARG and lib.mv_mask_pack(P, 3, 0, 64, P, P, None) == E_ARG
    assert lib.mv_mask_pack(P, 4, 2, 64, P, P, None) == E_SHAPE and lib.mv_mask_pack(P, 1, 2, 64, P, P, None) == E_SHAPE      # NotImplementedError
    assert lib.mv_mask_pack(P, 3, 2, 64 * 64 + 1, P, P, None) == E_SHAPE
    assert lib.mv_mask_build(None, 2, 64, P, P, None) == E_ARG and lib.mv_mask_build(P, 2, 0, P, P, None) == E_ARG
    assert lib.mv_mask_build(P, 2, 5000, P, P, None) == E_SHAPE
    assert lib.mv_pack_plan(None, 2, 64, P, P, P, None) != 0
    assert lib.mv_tail_perm(None, 2, 64, P, 4, P, P, P, P, None) != 0


def test_attention_rejects(lib):
    fwd = lambda **k: lib.mv_attn_fwd(k.get("dt", BF16), k.get("qkv", P), P, P, k.get("ctx", P), k.get("ctx2", None), P, k.get("B", 2),
                                      k.get("L", 64), 2, k.get("dh", 64), k.get("p", 0.0), k.get("db", None), k.get("cu", None),
                                      k.get("rows", 0), None, None)
    assert fwd(qkv=None) == E_ARG and fwd(B=0) == E_ARG and fwd(dh=0) == E_ARG
    assert fwd(dt=5) == E_DTYPE and fwd(dt=BF16, ctx2=P) == E_DTYPE
    assert fwd(dh=32) == E_SHAPE                               # the MFMA kernels are built for dh = 64
    assert fwd(qkv=P + 4) == E_SHAPE                           # 16-byte alignment of the fused projection
    assert fwd(cu=P, rows=0) == E_ARG and fwd(cu=P, rows=2 * 64 + 1) == E_ARG
    assert fwd(p=0.1) != 0                                     # dropout needs the keep-bit tensor
    assert lib.mv_attn_bwd(BF16, None, P, P, P, P, P, P, P, 2, 64, 2, 64, 0.0, None, None, 0, None, None) == E_ARG
    assert lib.mv_attn_dropmask(0.1, 1, 2, 64, 2, None, None, None) == E_ARG


def test_rowop_rejects(lib):
    assert lib.mv_layernorm_fwd(BF16, None, F32, P, P, P, None, P, P, 4, 128, 1e-12, None) == E_ARG
    assert lib.mv_layernorm_fwd(BF16, P, F32, P, P, P, None, P, P, 4, 130, 1e-12, None) == E_SHAPE          # H % 4
    emb = lambda **k: lib.mv_embed_fwd(k.get("dt", BF16), P, P, P, k.get("pos", P), P, k.get("img", P), P, P, P, P, P, P, k.get("x2", None), P, P,
                                       P, k.get("B", 2), k.get("N", 4), k.get("T", 8), k.get("H", 128), 1000, k.get("maxpos", 64), 1e-12,
                                       0.0, 0.0, 0, k.get("rowmap", None), k.get("n_rows", 0), None)
    assert emb(B=0) == E_ARG and emb(img=None) == E_ARG and emb(H=130) == E_SHAPE
    assert emb(T=65) == E_SHAPE                                 # text positions 0..T-1 must exist in the position table
    assert emb(rowmap=P, n_rows=0) == E_ARG and emb(rowmap=P, n_rows=2 * 14 + 1) == E_ARG
    assert emb(dt=BF16, x2=P) == E_DTYPE and emb(dt=4) == E_DTYPE
    assert lib.mv_ce_fwd_bwd(None, F32, 8, P, 2, 8, P, None, 0, 0, None, 1.0, None, None) == E_ARG
    assert lib.mv_adamw_step(None, P, P, P, None, None, 16, 1e-3, 0.9, 0.999, 1e-6, 0.0, 1, 1, 1.0, None, None) == E_ARG
    assert lib.mv_gather_rows(BF16, None, 8, P, 2, 8, P, 8, None) == E_ARG
    assert lib.mv_colsum(BF16, None, 8, 2, 8, P, 1, None, None) == E_ARG
    assert lib.mv_cast(None, F32, P, BF16, 16, None) == E_ARG
    assert lib.mv_count_nonfinite(None, 16, P, None) == E_ARG


def test_host_mask_check_through_the_raw_abi(lib):
    """Real host work (the only entry point that computes on the CPU): ragged geometry, every family, 1-D masks, threads > samples."""
    out = C.c_longlong(0)
    rng = np.random.default_rng(3)
    for L, n2 in ((37, 7), (64, 18), (95, 4), (512, 38)):
        B = 5
        i, j = np.arange(L).reshape(L, 1), np.arange(L).reshape(1, L)
        vls = rng.integers(n2 + 1, L + 1, size=B)
        for fam in range(5):
            desc = np.stack([np.full(B, fam), np.full(B, n2), vls], 1).astype(np.int32)
            forms = {0: lambda vl: np.broadcast_to(j < vl, (L, L)), 1: lambda vl: (j < n2) | ((i >= n2) & (j >= n2) & (j <= i)),
                     2: lambda vl: (i < n2) | (j < n2) | (j <= i), 3: lambda vl: (i < n2) == (j < n2)}
            if fam == 4:
                m = np.stack([(np.arange(L) < vl) for vl in vls]).astype(np.int64)
            else:
                m = np.stack([forms[fam](vl) for vl in vls]).astype(np.int64)
            m = np.ascontiguousarray(m)
            assert lib.mv_mask_verify_host(m.ctypes.data, m.ndim, desc.ctypes.data, B, L, 8, C.byref(out)) == 0 and out.value == -1, (L, fam)
            idx = tuple(int(rng.integers(0, s)) for s in m.shape)
            m[idx] = 5 if m[idx] == 0 else 0                  # any non-zero value is "visible", like the device packer's `!= 0`
            assert lib.mv_mask_verify_host(m.ctypes.data, m.ndim, desc.ctypes.data, B, L, 3, C.byref(out)) == 0
            assert out.value == int(np.ravel_multi_index(idx, m.shape)), (L, fam, idx)
    m = np.zeros((2, 8, 8), dtype=np.int64)
    d = np.zeros((2, 3), dtype=np.int32)
    assert lib.mv_mask_verify_host(None, 3, d.ctypes.data, 2, 8, 1, C.byref(out)) == E_ARG
    assert lib.mv_mask_verify_host(m.ctypes.data, 4, d.ctypes.data, 2, 8, 1, C.byref(out)) == E_SHAPE
    d[1, 0] = 5
    assert lib.mv_mask_verify_host(m.ctypes.data, 3, d.ctypes.data, 2, 8, 1, C.byref(out)) == E_ARG


def test_comm_rejects(lib):
    """mv_comm_*: argument checks come before RCCL is even loaded (it is bound at run time: the library has no link-time dependency on it)."""
    h = C.c_void_p()
    assert lib.mv_comm_unique_id(None) == E_ARG
    assert lib.mv_comm_init(None, 0, 1, P) == E_ARG and lib.mv_comm_init(C.byref(h), 0, 1, None) == E_ARG
    assert lib.mv_comm_init(C.byref(h), 2, 2, P) == E_ARG and lib.mv_comm_init(C.byref(h), 0, 0, P) == E_ARG
    assert lib.mv_comm_allreduce_async(None, P, 16, F32, None) == E_ARG and lib.mv_comm_wait(None, None) == E_ARG
    assert lib.mv_comm_destroy(None) == E_ARG
    import subprocess
    out = subprocess.run(["bash", "-c", f"readelf -d {_lib.LIB_PATH} | grep NEEDED"], capture_output=True, text=True).stdout
    assert "rccl" not in out and "nccl" not in out, out
