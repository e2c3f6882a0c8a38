"""CPU oracle for the region-feature extractor (models/image.py:46-69): torchvision's ResNet-50 trunk restated with
torch.nn.functional on NCHW fp32 tensors, driven by a torchvision-layout state dict (`model.<idx>...` keys).

TEST INFRASTRUCTURE ONLY (see oracle/cxrbert_oracle.py for who may import it).

PARITY UNPINNED: torchvision is neither in /root/reference nor in this image, so this restatement follows the published
torchvision definition (Bottleneck v1.5: stride on the 3x3 convolution; conv1 7x7/2 pad 3; maxpool 3x3/2 pad 1;
BatchNorm eps 1e-5, momentum 0.1, biased variance for normalisation and unbiased for the running estimate) without a
fixture produced by torchvision itself.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

LAYERS = (3, 4, 6, 3)


def _bn(x, sd, pre, training, momentum=0.1, eps=1e-5):
    return F.batch_norm(x, sd[pre + ".running_mean"], sd[pre + ".running_var"], sd[pre + ".weight"], sd[pre + ".bias"],
                        training=training, momentum=momentum, eps=eps)


def block(sd, p, y, stride, training, rnd=None):
    """torchvision Bottleneck `p` (v1.5) on y [B,C,H,W]."""
    r = rnd if rnd is not None else (lambda t: t)
    w = lambda k: r(sd[k])
    idt = y
    o = r(F.relu(_bn(r(F.conv2d(y, w(p + ".conv1.weight"))), sd, p + ".bn1", training)))
    o = r(F.relu(_bn(r(F.conv2d(o, w(p + ".conv2.weight"), stride=stride, padding=1)), sd, p + ".bn2", training)))
    o = _bn(r(F.conv2d(o, w(p + ".conv3.weight"))), sd, p + ".bn3", training)
    if (p + ".downsample.0.weight") in sd:
        idt = r(_bn(r(F.conv2d(y, w(p + ".downsample.0.weight"), stride=stride)), sd, p + ".downsample.1", training))
    return r(F.relu(o + idt))


def trunk(sd: dict, x: torch.Tensor, training: bool, rnd=None) -> torch.Tensor:
    """sd: state dict with keys model.0.weight, model.1.*, model.4.0.conv1.weight ... (float32; running statistics are
    updated IN PLACE when training, like nn.BatchNorm2d).  x [B,3,H,W] -> [B,2048,h,w].
    rnd: optional rounding applied where the bf16 product stores bf16 (pixels, convolution weights, every convolution
    output and every BatchNorm output): a randomly initialised ResNet under batch-statistics BatchNorm amplifies bf16 rounding ~100x over its 53
    convolutions, so the bf16 path is checked against this restatement WITH its rounding points."""
    r = rnd if rnd is not None else (lambda t: t)
    w = lambda k: r(sd[k])
    y = r(F.conv2d(r(x), w("model.0.weight"), stride=2, padding=3))
    y = r(F.relu(_bn(y, sd, "model.1", training)))
    y = F.max_pool2d(y, 3, stride=2, padding=1)
    for li, n in enumerate(LAYERS):
        for bi in range(n):
            y = block(sd, f"model.{4 + li}.{bi}", y, 2 if (bi == 0 and li > 0) else 1, training, rnd)
    return y


def region_features(sd, x, positions, training):
    """image.py:54-69 with the sampled positions injected: ([B,N,2048], [B,N])."""
    out = torch.flatten(trunk(sd, x, training), start_dim=2).transpose(1, 2).contiguous()
    return out[:, positions], positions.view(1, -1).expand(out.shape[0], -1)
