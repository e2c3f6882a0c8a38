"""The C-ABI shared library: it loads, exports every symbol include/medvill.h declares, and the
ctypes prototype table matches the header's parameter counts.  No compute calls (CPU only)."""
import ctypes
import os
import re

import pytest

import medvill_amd  # noqa: F401
from medvill_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "medvill.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|void|const char\*)\s+(mv_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[m.group(1)] = n
    return out


def test_library_is_built_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = _lib.load()
    assert lib.mv_abi_version() == 5
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
