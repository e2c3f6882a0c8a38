"""Build libmedvill_hip.so (hand-written gfx950 HIP kernels + the C ABI of include/medvill.h) and libmedvill_hip_dbg.so (the same
objects with the debug build of csrc/mv_api.hip: include/medvill_debug.h -- kernel-forcing knobs for tests and experiments).

In-tree build with plain hipcc (no cmake/ninja needed): one object per .hip file, compiled in
parallel, linked into ``multi-modality-self-supervision_amd/libmedvill_hip.so``.  hipcc
cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build container too.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libmedvill_hip.so")
LIB_DBG = os.path.join(HERE, "libmedvill_hip_dbg.so")
SOURCES = ["mv_gemm_ring_tn.hip", "mv_gemm_ring_tn4.hip", "mv_gemm_ring_nt.hip", "mv_gemm_ring_nn.hip", "mv_gemm_ring_tnn.hip", "mv_gemm.hip", "mv_attn.hip",
           "mv_rowops.hip", "mv_batch.hip", "mv_conv.hip", "mv_hostpack.hip", "mv_comm.hip", "mv_api.hip"]      # slowest translation units first
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "mv_common.h"), os.path.join(CSRC, "mv_gemm_common.h"), os.path.join(CSRC, "mv_gemm_ring.h"),
               os.path.join(os.path.dirname(HERE), "include", "medvill.h")]
    dbg_header = os.path.join(os.path.dirname(HERE), "include", "medvill_debug.h")      # only the debug build of mv_api.hip includes it
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj, []))
    api_dbg = os.path.join(OBJ, "mv_api_dbg.o")
    if force or _stale(api_dbg, [os.path.join(CSRC, "mv_api.hip"), dbg_header] + headers):
        jobs.append((os.path.join(CSRC, "mv_api.hip"), api_dbg, ["-DMV_DEBUG_KNOBS"]))

    def cc(job):
        src, obj, extra = job
        cmd = [hipcc] + FLAGS + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    objs_dbg = [api_dbg if o.endswith("mv_api.o") else o for o in objs]
    for lib, oo in ((LIB, objs), (LIB_DBG, objs_dbg)):
        if force or jobs or _stale(lib, oo):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + oo
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
