import sys, os
sys.path.insert(0, os.getcwd())
import torch
import medvill_amd as mv
from oracle import resnet_oracle as R
import torch.nn.functional as F
DEV="cuda"
def enc_of(dtype):
    torch.manual_seed(0)
    enc = mv.ImageEncoder_cnn(num_image_embeds=6, dtype=dtype)
    return enc.to(DEV)
x = torch.randn(8, 3, 160, 128, generator=torch.Generator().manual_seed(9))
for dtype in (torch.float32, torch.bfloat16):
    for training in (False, True):
        enc = enc_of(dtype); enc.train(training)
        sd = {k: v.detach().cpu().float().clone() for k, v in enc.state_dict().items()}
        ref = R.trunk(sd, x, training)
        y, h, w = enc.trunk(x.to(DEV))
        got = y.view(8, h, w, 2048).permute(0, 3, 1, 2).float().cpu()
        print(dtype, training, "L2", float((got-ref).norm()/ref.norm()), "max", float((got-ref).abs().max()/ref.abs().max()))
# stem only, bf16 train
enc = enc_of(torch.bfloat16); enc.train(True)
sd = {k: v.detach().cpu().float().clone() for k, v in enc.state_dict().items()}
m = enc.model
B,H,W = 8,160,128
xin = torch.empty((B*H*W, 8), dtype=torch.bfloat16, device=DEV)
mv.hip_ops.nchw_to_nhwc(x.to(DEV), xin, B, 3, H, W, 8)
y, H1, W1 = enc._conv(xin, B, H, W, 8, m[0])
ref = F.conv2d(x, sd["model.0.weight"], stride=2, padding=3)
got = y.view(B,H1,W1,64).permute(0,3,1,2).float().cpu()
print("stem conv L2", float((got-ref).norm()/ref.norm()))
yb = enc._bn(y, m[1])
refb = F.relu(F.batch_norm(ref, sd["model.1.running_mean"].clone(), sd["model.1.running_var"].clone(), sd["model.1.weight"], sd["model.1.bias"], training=True))
gotb = yb.view(B,H1,W1,64).permute(0,3,1,2).float().cpu()
print("stem bn L2", float((gotb-refb).norm()/refb.norm()))
