"""Region-feature extractor: mirror of the reference's `ImageEncoder_cnn` (models/image.py:46-69).

ResNet-50 trunk (torchvision `resnet50`, children()[:-2]: conv1, bn1, relu, maxpool, layer1..4) -> [B, 2048, h, w] ->
[B, h*w, 2048] -> a sorted random sample of `num_image_embeds` positions shared by the batch, returned with the
positions themselves (the position ids of the image embeddings, cxrbert_origin.py:26-29).

The torch modules below are parameter CONTAINERS only (torchvision's names and shapes, so `enc.img_encoder.model.*`
keys of the released checkpoints load); the arithmetic runs through the C ABI: every convolution is `mv_gemm` over
the NHWC activation matrix (after `mv_im2col` unless it is 1x1 / stride 1), BatchNorm(+residual)(+ReLU) is `mv_bn_act`
with batch statistics (`mv_col_stats`) under train() -- the reference keeps the CNN frozen but in train() mode, so
BatchNorm normalises with batch statistics and updates its running ones -- or the running statistics under eval().
Forward only: nothing in the reference's CNN requires a gradient (cxrbert_origin.py:66-70 unfreezes
`list(self.img_encoder.children())[5:]`, which is empty: the module has a single child).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import hip_ops as ops

LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)


class _Bottleneck(nn.Module):          # torchvision.models.resnet.Bottleneck (v1.5: the stride sits on conv2)
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
        self.stride = stride


def _trunk():
    mods = [nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.MaxPool2d(3, stride=2, padding=1)]
    inplanes = 64
    for li, (n, planes) in enumerate(zip(LAYERS, PLANES)):
        blocks = []
        for bi in range(n):
            stride = 2 if (bi == 0 and li > 0) else 1
            blocks.append(_Bottleneck(inplanes, planes, stride, downsample=(bi == 0)))
            inplanes = planes * 4
        mods.append(nn.Sequential(*blocks))
    model = nn.Sequential(*mods)
    for m in model.modules():              # torchvision's initialisation
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    return model


class ImageEncoder_cnn(nn.Module):
    def __init__(self, args=None, num_image_embeds=None, dtype=torch.bfloat16):
        super().__init__()
        self.args = args
        self.num_image_embeds = int(num_image_embeds if num_image_embeds is not None else getattr(args, "num_image_embeds", 36))
        self.model = _trunk()
        for p in self.parameters():        # cxrbert_origin.py:66-70: everything stays frozen
            p.requires_grad = False
        self.adt = dtype
        self._wcache = {}
        self._bncache = {}
        self._foldcache = {}
        self.weights_loaded = False       # True once trained weights arrived: the reference builds resnet50(pretrained=True)
        self.fold_bn = True               # eval(), bf16: BatchNorm folded into the convolutions' weights and epilogues (test switch)
        self.implicit_conv = True         # bf16 path: mv_conv2d instead of mv_im2col + mv_gemm (test switch)

    # ---- weights: the reference's trunk is torchvision's pretrained resnet50 (models/image.py:50-52)
    _TV_PREFIX = {"conv1.": "model.0.", "bn1.": "model.1.", "layer1.": "model.4.", "layer2.": "model.5.", "layer3.": "model.6.",
                  "layer4.": "model.7."}

    def load_torchvision_state_dict(self, sd, strict=True):
        """Load a torchvision `resnet50` state dict (keys conv1.*, bn1.*, layer1-4.*, fc.* -- e.g. the file behind
        `resnet50(pretrained=True)`); `fc.*` is dropped like the reference's `children()[:-2]` does."""
        mapped = {}
        for k, v in sd.items():
            for src, dst in self._TV_PREFIX.items():
                if k.startswith(src):
                    mapped[dst + k[len(src):]] = v
                    break
        return self.load_state_dict(mapped, strict=strict)

    def load_state_dict(self, sd, strict=True, **kw):
        r = super().load_state_dict(sd, strict=strict, **kw)
        self.weights_loaded = True
        return r

    # ---- weights in GEMM layout: [Cout, kh*kw*Cin_padded], (ky, kx, c) order, compute dtype
    def _w2d(self, conv: nn.Conv2d, cin_pad: int):
        key = id(conv)
        ver = conv.weight._version
        hit = self._wcache.get(key)
        if hit is not None and hit[0] == ver and hit[1].device == conv.weight.device:
            return hit[1]
        O, I, kh, kw = conv.weight.shape
        w = conv.weight.detach().permute(0, 2, 3, 1)
        if cin_pad != I:
            w = torch.nn.functional.pad(w, (0, cin_pad - I))
        w = w.reshape(O, kh * kw * cin_pad).to(self.adt).contiguous()
        self._wcache[key] = (ver, w)
        return w

    # ---- eval(): BatchNorm folded into the convolution (w' = w * gamma * rstd per output channel, b' = beta - mean * gamma * rstd)
    def _folded(self, conv: nn.Conv2d, bn: nn.BatchNorm2d, cin_pad: int):
        key = (id(conv), id(bn))
        ver = (conv.weight._version, bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version)
        hit = self._foldcache.get(key)
        if hit is not None and hit[0] == ver and hit[1].device == conv.weight.device:
            return hit[1], hit[2]
        O, I, kh, kw = conv.weight.shape
        sc = bn.weight.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps)
        w = (conv.weight.detach().float() * sc.view(O, 1, 1, 1)).permute(0, 2, 3, 1)
        if cin_pad != I:
            w = torch.nn.functional.pad(w, (0, cin_pad - I))
        w = w.reshape(O, kh * kw * cin_pad).to(self.adt).contiguous()
        b = (bn.bias.detach().float() - bn.running_mean.float() * sc).contiguous()
        self._foldcache[key] = (ver, w, b)
        return w, b

    def _conv_bn_eval(self, x, B, H, W, C, conv: nn.Conv2d, bn: nn.BatchNorm2d, residual=None, relu=True):
        """eval(), bf16: convolution + BatchNorm (+ residual) (+ ReLU) as ONE GEMM with a fused epilogue."""
        kh, kw = conv.kernel_size
        s, pad = conv.stride[0], conv.padding[0]
        Ho, Wo = (H + 2 * pad - kh) // s + 1, (W + 2 * pad - kw) // s + 1
        w, b = self._folded(conv, bn, C)
        rows, K, O = B * Ho * Wo, w.shape[1], w.shape[0]
        epi = ops.EPI_BIAS_RES_RELU if residual is not None else (ops.EPI_BIAS_RELU if relu else ops.EPI_BIAS)
        y = torch.empty((rows, O), dtype=self.adt, device=x.device)
        if kh == 1 and kw == 1 and s == 1:
            step = max(256, ((1 << 31) - (1 << 20)) // (K * x.element_size()) // 256 * 256)
            for r0 in range(0, rows, step):
                n = min(step, rows - r0)
                ops.gemm(x[r0:r0 + n], w, y[r0:r0 + n], M=n, N=O, K=K, bias=b, epi=epi,
                         r=None if residual is None else residual[r0:r0 + n])
        else:
            ops.conv2d(x, w, y, B, H, W, C, O, kh, kw, s, pad, bias=b, epi=epi, r=residual)
        return y, Ho, Wo

    def _conv(self, x, B, H, W, C, conv: nn.Conv2d):
        """x: [B*H*W, C] activation matrix (NHWC) -> ([B*Ho*Wo, Cout], Ho, Wo)."""
        kh, kw = conv.kernel_size
        s, pad = conv.stride[0], conv.padding[0]
        Ho, Wo = (H + 2 * pad - kh) // s + 1, (W + 2 * pad - kw) // s + 1
        w = self._w2d(conv, C)
        rows, K, O = B * Ho * Wo, w.shape[1], w.shape[0]
        y = torch.empty((rows, O), dtype=self.adt, device=x.device)
        if kh == 1 and kw == 1 and s == 1:
            a = x                                         # a 1x1 / stride-1 convolution is the GEMM over the activation matrix itself
        elif self.adt == torch.bfloat16 and self.implicit_conv:
            ops.conv2d(x, w, y, B, H, W, C, O, kh, kw, s, pad)        # taps gathered inside the GEMM's operand staging
            return y, Ho, Wo
        else:
            a = torch.empty((rows, K), dtype=self.adt, device=x.device)
            ops.im2col(x, a, B, H, W, C, kh, kw, s, pad, K)
        # mv_gemm addresses an operand through a 2-GiB buffer descriptor: large activation matrices go in row slabs
        step = max(256, ((1 << 31) - (1 << 20)) // (K * a.element_size()) // 256 * 256)
        for r0 in range(0, rows, step):
            n = min(step, rows - r0)
            ops.gemm(a[r0:r0 + n], w, y[r0:r0 + n], M=n, N=O, K=K)
        return y, Ho, Wo

    def _bn(self, x, bn: nn.BatchNorm2d, residual=None, relu=True):
        rows, C = x.shape
        dev = x.device
        if self.training:
            st = torch.empty((2, C), dtype=torch.float32, device=dev)
            mean, rstd = torch.empty((C,), dtype=torch.float32, device=dev), torch.empty((C,), dtype=torch.float32, device=dev)
            ops.col_stats(x, C, rows, C, st)
            m = bn.momentum if bn.momentum is not None else 0.1
            ops.bn_finalize(st, C, rows, bn.eps, m, mean, rstd, bn.running_mean, bn.running_var)   # updates the running buffers
            bn.num_batches_tracked += 1
        else:
            key = id(bn)
            hit = self._bncache.get(key)
            ver = (bn.running_mean._version, bn.running_var._version)
            if hit is None or hit[0] != ver or hit[1].device != dev:
                hit = (ver, bn.running_mean.float().contiguous(), torch.rsqrt(bn.running_var.float() + bn.eps).contiguous())
                self._bncache[key] = hit
            mean, rstd = hit[1], hit[2]
        y = torch.empty(x.shape, dtype=self.adt, device=dev)
        ops.bn_act(x, mean, rstd, bn.weight, bn.bias, y, rows, C, residual=residual, relu=relu)
        return y

    def _block(self, y, B, H, W, C, blk: _Bottleneck):
        """one bottleneck on the activation matrix y [B*H*W, C] -> (matrix, H', W', C')"""
        if not self.training and self.fold_bn and self.adt == torch.bfloat16:
            o, _, _ = self._conv_bn_eval(y, B, H, W, C, blk.conv1, blk.bn1)
            o, H2, W2 = self._conv_bn_eval(o, B, H, W, blk.conv1.out_channels, blk.conv2, blk.bn2)
            idt = y
            if blk.downsample is not None:
                idt, _, _ = self._conv_bn_eval(y, B, H, W, C, blk.downsample[0], blk.downsample[1], relu=False)
            o, _, _ = self._conv_bn_eval(o, B, H2, W2, blk.conv2.out_channels, blk.conv3, blk.bn3, residual=idt)
            return o, H2, W2, blk.conv3.out_channels
        idt = y
        o, _, _ = self._conv(y, B, H, W, C, blk.conv1)
        o = self._bn(o, blk.bn1)
        o, H2, W2 = self._conv(o, B, H, W, blk.conv1.out_channels, blk.conv2)
        o = self._bn(o, blk.bn2)
        o, _, _ = self._conv(o, B, H2, W2, blk.conv2.out_channels, blk.conv3)
        if blk.downsample is not None:
            idt, _, _ = self._conv(y, B, H, W, C, blk.downsample[0])
            idt = self._bn(idt, blk.downsample[1], relu=False)
        y = self._bn(o, blk.bn3, residual=idt, relu=True)
        return y, H2, W2, blk.conv3.out_channels

    def trunk(self, x: torch.Tensor):
        """pixels f32 [B,3,H,W] -> ([B*h*w, 2048] NHWC activation matrix, h, w)."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("ImageEncoder_cnn expects [B,3,H,W] pixels")
        B, _, H, W = x.shape
        m = self.model
        xin = torch.empty((B * H * W, 8), dtype=self.adt, device=x.device)        # 3 channels padded to 8: 16-byte gathers
        ops.nchw_to_nhwc(x.float().contiguous(), xin, B, 3, H, W, 8)
        if not self.training and self.fold_bn and self.adt == torch.bfloat16:
            y, H, W = self._conv_bn_eval(xin, B, H, W, 8, m[0], m[1])
        else:
            y, H, W = self._conv(xin, B, H, W, 8, m[0])
            y = self._bn(y, m[1])
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        p = torch.empty((B * Ho * Wo, 64), dtype=self.adt, device=x.device)
        ops.maxpool3x3s2(y, p, B, H, W, 64)
        y, H, W, C = p, Ho, Wo, 64
        for li in range(4, 8):
            for blk in m[li]:
                y, H, W, C = self._block(y, B, H, W, C, blk)
        return y, H, W

    def forward(self, x):
        y, h, w = self.trunk(x)
        B = x.shape[0]
        out = y.view(B, h * w, y.shape[1])                                       # B x M x 2048 (image.py:57-58)
        n = out.shape[1]
        sel, _ = torch.sort(torch.randperm(n)[:self.num_image_embeds])            # image.py:63-65 (CPU generator, per forward)
        sel = sel.to(out.device)
        return out[:, sel].contiguous(), sel.view(1, -1).expand(B, -1).contiguous()
