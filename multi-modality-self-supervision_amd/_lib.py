"""ctypes binding of libmedvill_hip.so (the C ABI declared in include/medvill.h).

The product path FAILS LOUDLY when the HIP library is missing or a call returns a
non-zero status: there is no CPU / eager fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MV_LIB_PATH") or os.path.join(HERE, "libmedvill_hip.so")   # MV_LIB_PATH: kernel-variant experiments
# the test / experiment build (include/medvill_debug.h): same kernels + a knob table; loaded only while a knob is off its default
DBG_LIB_PATH = os.environ.get("MV_DBG_LIB_PATH") or (LIB_PATH[:-3] + "_dbg.so")

MV_F32, MV_BF16, MV_F16 = 0, 1, 2
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES, EPI_DGELU, EPI_RES, EPI_BIAS_TANH, EPI_BIAS_GELU_D, EPI_MUL, EPI_BIAS_RELU, EPI_BIAS_RES_RELU = range(11)
_ERR = {-1: "MV_E_ARG (null pointer / bad size)", -2: "MV_E_SHAPE (unsupported shape or alignment)",
        -3: "MV_E_DTYPE", -4: "MV_E_WORKSPACE (workspace too small)", -5: "MV_E_NO_RCCL (librccl could not be loaded)"}

vp, i32, i64, f32, sz, u64 = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t, C.c_ulonglong

# name -> argtypes; every function returns int unless listed in _RESTYPE.  This table is also
# what tests/test_abi.py checks against include/medvill.h.
PROTOTYPES = {
    "mv_abi_version": [],
    "mv_build_info": [],
    "mv_stream_create_cumask": [vp, i32, C.POINTER(C.c_void_p)],
    "mv_stream_destroy": [vp],
    "mv_comm_unique_id": [vp],
    "mv_comm_init": [C.POINTER(C.c_void_p), i32, i32, vp],
    "mv_comm_allreduce_async": [vp, vp, sz, i32, vp],
    "mv_comm_wait": [vp, vp],
    "mv_comm_destroy": [vp],
    "mv_gemm": [i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, i32, vp, i32, vp, i32, i32, vp, i32, vp, i32, i32,
                i32, vp, sz, i32, f32, u64, vp, vp, vp],
    "mv_gemm_workspace_bytes": [i32, i32, i32, i32, i32, i32],
    "mv_workspace_bytes": [i32, i32, i32, i32, i32, i32, i32],
    "mv_mask_pack": [vp, i32, i32, i32, vp, vp, vp],
    "mv_mask_build": [vp, i32, i32, vp, vp, vp],
    "mv_mask_verify_host": [vp, i32, vp, i32, i32, i32, C.POINTER(C.c_longlong)],
    "mv_mlm_draws": [u64, i32, i32, i32, vp, vp, vp],
    "mv_mlm_corrupt": [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "mv_pack_plan": [vp, i32, i32, vp, vp, vp, vp],
    "mv_tail_perm": [vp, i32, i32, vp, i32, vp, vp, vp, vp, vp],
    "mv_attn_fwd": [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, i32, vp, vp],
    "mv_attn_dropmask": [f32, u64, i32, i32, i32, vp, vp, vp],
    "mv_attn_bwd": [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, i32, vp, vp],
    "mv_layernorm_fwd": [i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp],
    "mv_layernorm_bwd": [i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, f32, u64, vp, vp],
    "mv_embed_fwd": [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, f32, f32,
                     u64, vp, i32, vp],
    "mv_embed_bwd": [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, f32,
                     u64, vp, i32, vp, vp],
    "mv_dropout_mask": [f32, u64, sz, vp, C.POINTER(C.c_float), vp],
    "mv_ce_fwd_bwd": [vp, i32, i32, vp, i32, i32, vp, vp, i32, i32, vp, f32, vp, vp],
    "mv_gather_rows": [i32, vp, i32, vp, i32, i32, vp, i32, vp],
    "mv_scatter_rows": [i32, vp, i32, vp, i32, i32, vp, i32, i32, vp],
    "mv_colsum": [i32, vp, i32, i32, i32, vp, i32, vp, vp],
    "mv_colsum_partials": [vp, i32, i32, i32, vp, vp, vp],
    "mv_add": [i32, vp, vp, vp, sz, vp],
    "mv_dact": [i32, i32, vp, vp, vp, sz, vp],
    "mv_cast2d": [vp, i32, i64, vp, i32, i64, i32, i32, vp],
    "mv_cast": [vp, i32, vp, i32, sz, vp],
    "mv_transpose": [i32, vp, i64, vp, i64, i32, i32, vp],
    "mv_nchw_to_nhwc": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "mv_im2col": [i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp],
    "mv_conv2d": [i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp],
    "mv_col_stats": [i32, vp, i32, i32, i32, vp, vp],
    "mv_bn_finalize": [vp, i32, i64, f32, f32, vp, vp, vp, vp, vp],
    "mv_bn_act": [i32, vp, i32, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp],
    "mv_maxpool3x3s2": [i32, vp, vp, i32, i32, i32, i32, vp],
    "mv_adamw_step": [vp, vp, vp, vp, vp, vp, sz, f32, f32, f32, f32, f32, i32, i32, f32, vp, vp],
    "mv_count_nonfinite": [vp, sz, vp, vp],
    "mv_scaler_update": [vp, i32, f32, f32, f32, f32, vp],
}
# include/medvill_debug.h: exported by libmedvill_hip_dbg.so only
DEBUG_PROTOTYPES = {"mv_debug_set_knob": [i32, i32], "mv_debug_get_knob": [i32]}
_RESTYPE = {"mv_build_info": C.c_char_p, "mv_gemm_workspace_bytes": C.c_size_t, "mv_workspace_bytes": C.c_size_t}
ABI_VERSION = 6

# knob name -> (id, default): csrc/mv_common.h / include/medvill_debug.h
KNOBS = {"impl": (0, 0), "gemm_force": (1, 0), "gemm_nj": (2, 0), "gemm_dbg": (3, 0), "attn_planes": (4, 16), "persistent_cus": (5, 0),
         "rowops_variant": (6, 0), "attn_order": (7, 0), "attn_fwd": (8, 0), "gemm_rounds": (9, 1)}

_lib = None          # the product library
_dbg = None          # the debug library (lazily)
_knobs = {k: d for k, (_, d) in KNOBS.items()}


def _open(path, protos):
    lib = C.CDLL(path)
    for name, args in protos.items():
        fn = getattr(lib, name)          # AttributeError if an exported symbol is missing
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, C.c_int)
    if lib.mv_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version mismatch")
    return lib


def load(build_if_missing: bool = False):
    """The ctypes handle every wrapper calls through: the product library -- or the debug library while a knob is off its default
    (set_knob).  Loaded once; raises if the library is absent."""
    global _lib
    if _dbg is not None and any(_knobs[k] != d for k, (_, d) in KNOBS.items()):
        return _dbg
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if build_if_missing:
            from ._build import build
            build()
        else:
            raise RuntimeError(
                f"{LIB_PATH} is missing: the gfx950 HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no fallback path.")
    _lib = _open(LIB_PATH, PROTOTYPES)
    return _lib


def set_knob(name: str, value: int):
    """Force a kernel choice (tests, timing experiments).  The product library has no such state: a knob off its default routes every
    call of this process through libmedvill_hip_dbg.so until it is set back."""
    global _dbg
    kid, default = KNOBS[name]
    value = int(value)
    if _dbg is None:
        if value == default:
            return
        if not os.path.exists(DBG_LIB_PATH):
            raise RuntimeError(f"{DBG_LIB_PATH} is missing (the debug build of the library: __graft_entry__.build() makes both)")
        _dbg = _open(DBG_LIB_PATH, {**PROTOTYPES, **DEBUG_PROTOTYPES})
    check(_dbg.mv_debug_set_knob(kid, value), f"mv_debug_set_knob({name}, {value})")
    _knobs[name] = value


def get_knob(name: str) -> int:
    return _knobs[name]


def _knobs_from_env():
    """MV_KNOBS="attn_fwd=1,gemm_force=2": run a whole process (a test session, a bench) with kernels forced -- experiments only."""
    spec = os.environ.get("MV_KNOBS", "")
    for item in filter(None, (x.strip() for x in spec.split(","))):
        k, _, v = item.partition("=")
        set_knob(k.strip(), int(v))


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc <= -1000:                 # MV_E_COMM_BASE - ncclResult_t (mv_comm_* only)
        raise RuntimeError(f"{what}: ncclResult_t {-1000 - rc}")
    if rc < 0:
        raise RuntimeError(f"{what}: {_ERR.get(rc, rc)}")
    raise RuntimeError(f"{what}: hipError_t {rc}")


def dt_of(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return MV_F32
    if t.dtype == torch.bfloat16:
        return MV_BF16
    if t.dtype == torch.float16:
        return MV_F16
    raise TypeError(f"unsupported dtype {t.dtype}")


def torch_dtype(dt: int):
    return {MV_F32: torch.float32, MV_BF16: torch.bfloat16, MV_F16: torch.float16}[dt]


def ptr(t):
    return None if t is None else t.data_ptr()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("medvill HIP kernels need CUDA(ROCm) tensors; there is no CPU fallback")


_knobs_from_env()
