"""The C-ABI shared library: it loads, exports every symbol include/medvill.h declares, and the
ctypes prototype table matches the header's parameter counts.  No compute calls (CPU only)."""
import ctypes
import os
import re

import pytest

import medvill_amd  # noqa: F401
from medvill_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(name="medvill.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|void|size_t|const char\*)\s+(mv_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[m.group(1)] = n
    return out


def test_library_is_built_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = _lib.load()
    assert lib.mv_abi_version() == _lib.ABI_VERSION == 6
    assert b"gfx950" in lib.mv_build_info()


def test_every_declared_symbol_is_exported_with_matching_arity():
    decl = header_functions()
    assert len(decl) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name, nargs in decl.items():
        assert hasattr(raw, name), f"{name} declared in medvill.h but not exported"
        assert name in _lib.PROTOTYPES, f"{name} has no ctypes prototype"
        assert len(_lib.PROTOTYPES[name]) == nargs, (name, len(_lib.PROTOTYPES[name]), nargs)
    assert set(_lib.PROTOTYPES) == set(decl)


def _exports(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("mv_")}


def test_product_library_exports_the_header_and_nothing_else_and_has_no_knobs():
    """no global mutable state.  The kernel-forcing knobs live in libmedvill_hip_dbg.so only (include/medvill_debug.h);
    the product library exports exactly what medvill.h declares."""
    assert _exports(_lib.LIB_PATH) == set(header_functions())
    dbg = header_functions("medvill_debug.h")
    assert set(dbg) == set(_lib.DEBUG_PROTOTYPES) == {"mv_debug_set_knob", "mv_debug_get_knob"}
    for name, nargs in dbg.items():
        assert len(_lib.DEBUG_PROTOTYPES[name]) == nargs
    assert os.path.exists(_lib.DBG_LIB_PATH), "run __graft_entry__.build() first"
    assert _exports(_lib.DBG_LIB_PATH) == set(header_functions()) | set(dbg)
    raw = ctypes.CDLL(_lib.DBG_LIB_PATH)
    for name, (kid, default) in _lib.KNOBS.items():           # the debug library starts from the product's constants
        assert raw.mv_debug_get_knob(kid) == default, name
    assert raw.mv_debug_set_knob(99, 0) == -1 and raw.mv_debug_set_knob(4, 10) == -1
    raw.mv_build_info.restype = ctypes.c_char_p
    assert b"debug-knobs" in raw.mv_build_info()
    assert b"debug-knobs" not in _lib.load().mv_build_info()


def test_workspace_sizing_functions_are_pure_and_match_the_split_k_choice():
    """mv_gemm_workspace_bytes / mv_workspace_bytes: what a host that is not hip_ops.py needs to size split-K workspaces.
    Pure host arithmetic (no kernel, no device memory): callable here."""
    lib = _lib.load()
    H, I, V, M = 768, 3072, 30522, 25483
    dw1 = lib.mv_gemm_workspace_bytes(_lib.MV_F16, 1, 1, I, H, M)          # dW1 = dz^T . a: 36 tiles of 256 x 256 -> 7 slabs fill 256 CUs
    assert dw1 == 7 * I * H * 4
    assert lib.mv_gemm_workspace_bytes(_lib.MV_F16, 1, 1, H, H, M) == 24 * H * H * 4          # 9 tiles -> 28 slabs wanted, 24 of >= 1024 rows
    assert lib.mv_gemm_workspace_bytes(_lib.MV_F16, 0, 0, M, I, H) == 0                       # y = x.W^T over 1,200 tiles never splits
    assert lib.mv_gemm_workspace_bytes(_lib.MV_F32, 1, 1, I, H, M) == 0                       # f32 data: plain kernels, no split
    assert lib.mv_gemm_workspace_bytes(_lib.MV_F16, 1, 1, 0, H, M) == 0
    step = lib.mv_workspace_bytes(H, I, V, 2048, M, 3300, 64 * 36)
    assert step == max(dw1, lib.mv_gemm_workspace_bytes(_lib.MV_F16, 1, 1, H, H, M), lib.mv_gemm_workspace_bytes(_lib.MV_F16, 0, 1, 3300, H, V))
    assert lib.mv_workspace_bytes(0, I, V, 2048, M, 3300, 2304) == 0


def test_signatures_have_no_torch_types():
    src = open(os.path.join(ROOT, "include", "medvill.h")).read()
    assert 'extern "C"' in src
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)            # declarations only, comments stripped
    assert "torch" not in code.lower() and "at::" not in code and "std::" not in code and "Tensor" not in code


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "multi-modality-self-supervision_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "/root/reference" not in txt, f


def test_kernels_refuse_cpu_tensors():
    import torch
    from medvill_amd import hip_ops as ops
    a = torch.zeros((8, 8), dtype=torch.bfloat16)
    c = torch.zeros((8, 8), dtype=torch.float32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(a, a, c, M=8, N=8, K=8)
