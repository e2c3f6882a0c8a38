import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import medvill_amd as mv
DEV = "cuda"
enc = mv.ImageEncoder_cnn(num_image_embeds=36, dtype=torch.bfloat16).to(DEV)
for B in (16, 64):
    x = torch.randn(B, 3, 512, 512, device=DEV)
    for training in (False, True):
        enc.train(training)
        for _ in range(2): enc.trunk(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): y, h, w = enc.trunk(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        fl = 2 * 21.5e9 * B        # ~4.1 GMAC at 224^2 x (512/224)^2
        print(f"B={B} 512x512 training={training}: {dt*1e3:7.1f} ms  ({fl/dt/1e12:5.0f} TFLOP/s conv-only convention), out {tuple(y.shape)} {h}x{w}, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
