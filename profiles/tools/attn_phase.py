"""Where the waves of the attention kernels spend their cycles: runs profiles/tools/dominant.py's attention case on a variant library built
with -DATT_ABL=4096 (profiles/tools/attn_ablate.sh build 4096) and prints the per-phase shader-clock totals of thread 0 of every block.
usage: MV_LIB_PATH=.../libmedvill_abl4096.so python profiles/tools/attn_phase.py"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from medvill_amd import _lib
from dominant import make_case

fn, meta = make_case("attn")
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 48)()
for _ in range(2):
    fn()
torch.cuda.synchronize()
assert lib.mv_debug_attn_prof(buf) == 0
reps = 5
for _ in range(reps):
    fn()
torch.cuda.synchronize()
assert lib.mv_debug_attn_prof(buf) == 0
names = ["prologue", "wait for the tile's LDS-DMA", "barrier", "tile body", "issue next tile + bookkeeping", "epilogue"]
for k, kn in enumerate(("forward", "backward dQ", "backward dK/dV")):
    v = [buf[8 * k + i] for i in range(8)]
    nb = max(v[7], 1)
    tot = sum(v[:6])
    print(f"{kn}: {nb // reps} blocks with work per launch, {tot / nb:.0f} shader cycles per block (thread 0)")
    for i in range(6):
        print(f"    {names[i]:32s} {v[i] / nb:9.0f} cycles per block  {100.0 * v[i] / tot:5.1f} %")
tn = ["score MFMAs (K fragment reads + 8 MFMAs)", "row maximum + cross-half shuffle", "exponentials + row sums", "dropout selects (scalar mask loads)", "P.V (16 transposed reads + 8 MFMAs)"]
v = [buf[24 + i] for i in range(6)]
nb = max(buf[7], 1)
print("forward, inside the tile body (cycles per block, thread 0):")
for i in range(5):
    print(f"    {tn[i]:44s} {v[i] / nb:9.0f}  {100.0 * v[i] / max(sum(v[:5]), 1):5.1f} %")
