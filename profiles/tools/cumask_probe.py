"""Does a CU-masked HIP stream (hipExtStreamCreateWithCUMask, mv_stream_create_cumask) behave as a partition of the chip?  Times the
weight-gradient GEMM (persistent kernel), an HBM-bound kernel (AdamW over 28 M parameters) and the FFN-up GEMM (one block per tile)
alone on streams masked to n CUs, for masks taken from the low / high end and strided.  usage: python profiles/tools/cumask_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes
import torch
import medvill_amd as mv
from medvill_amd import hip_ops as ops
from medvill_amd import _lib as L
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dominant import make_case

dev = torch.device("cuda", 0)
dw, _ = make_case("dw", dev)
ffn1, _ = make_case("ffn1", dev)
n = 28 * 1024 * 1024
p, g, m, v = (torch.zeros(n, device=dev) for _ in range(4))
sh = torch.zeros(n, device=dev, dtype=torch.float16)
adam = lambda: ops.adamw_step(p, g, m, v, None, n, 1e-5, 0.9, 0.999, 1e-6, 0.0, 1, True, 1.0, shadow_f16=sh)


def masked(bits):
    words = (ctypes.c_uint32 * 8)()
    for i in bits:
        words[i // 32] |= 1 << (i % 32)
    out = ctypes.c_void_p()
    L.check(L.load().mv_stream_create_cumask(words, 8, ctypes.byref(out)), "cumask")
    return torch.cuda.ExternalStream(out.value, device=dev)


def t(fn, st, reps=10):
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            fn()
        e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


plain = torch.cuda.Stream(device=dev)
print(f"unmasked stream: dW {t(dw, plain):.0f} us | AdamW(28M) {t(adam, plain):.0f} us | FFN-up {t(ffn1, plain):.0f} us")
for name, bits in (("all 256 bits", range(256)), ("low 128", range(128)), ("high 128", range(128, 256)), ("even bits (128)", range(0, 256, 2)),
                   ("low 64", range(64)), ("high 64", range(192, 256)), ("every 4th (64)", range(0, 256, 4)), ("low 192", range(192)),
                   ("low 32", range(32)), ("low 8", range(8)), ("bits 0,8,16,..,248 (32)", range(0, 256, 8))):
    st = masked(list(bits))
    nb = len(list(bits))
    ops.set_persistent_cus(0)
    a = t(dw, st)
    ops.set_persistent_cus(nb)
    b = t(dw, st)
    ops.set_persistent_cus(0)
    print(f"{name:28s}: dW 256 blocks {a:.0f} us, {nb} blocks {b:.0f} us | AdamW {t(adam, st):.0f} us | FFN-up {t(ffn1, st):.0f} us")
