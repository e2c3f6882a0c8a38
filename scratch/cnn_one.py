import sys, os
sys.path.insert(0, os.getcwd())
import torch
import medvill_amd as mv
enc = mv.ImageEncoder_cnn(num_image_embeds=36, dtype=torch.bfloat16).to("cuda").train()
x = torch.randn(64, 3, 512, 512, device="cuda")
for _ in range(3): enc.trunk(x)
torch.cuda.synchronize()
