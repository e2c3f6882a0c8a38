"""GPU parity of the region-feature extractor (ResNet-50 trunk through mv_gemm / mv_im2col / mv_bn_act) against the
torch.nn.functional restatement in oracle/resnet_oracle.py on the same torchvision-layout weights.
Tolerances (relative L2; max-abs within 10x): fp32 path 2e-4 (exact fp32 arithmetic in another order, 53 convolutions
deep); bf16 path 2e-2 against the restatement with the same bf16 rounding points."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import medvill_amd as mv                                    # noqa: E402
from medvill_amd import hip_ops as ops                       # noqa: E402
from oracle import resnet_oracle as R                        # noqa: E402

DEV = "cuda"


def _encoder(dtype, seed=0, n=6):
    torch.manual_seed(seed)
    enc = mv.ImageEncoder_cnn(num_image_embeds=n, dtype=dtype)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():                                   # non-trivial BatchNorm parameters and running statistics
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
    return enc.to(DEV)


def _sd_cpu(enc):
    return {k: v.detach().cpu().float().clone() for k, v in enc.state_dict().items()}


@pytest.mark.parametrize("training", [False, True])
def test_trunk_against_the_torchvision_fixture_when_one_exists(golden_dir, training):
    """SURVEY 8(f) rank 4's pin: tests/golden/resnet50.npz holds feature maps of torchvision's own resnet50 trunk on splitmix weights
    (oracle/gen_resnet_golden.py, to be run where torchvision is installed -- it is in neither the reference tree nor this image).  Until
    that file is committed the HIP trunk is compared with the restatement only and the row stays PARITY UNPINNED (README)."""
    import os
    path = os.path.join(golden_dir, "resnet50.npz")
    if not os.path.exists(path):
        pytest.skip("parity unpinned: tests/golden/resnet50.npz absent (python oracle/gen_resnet_golden.py where torchvision exists)")
    from oracle import gen_resnet_golden as G
    z = np.load(path)
    enc = mv.ImageEncoder_cnn(num_image_embeds=4, dtype=torch.float32)
    enc.load_state_dict({"model." + k: v for k, v in G.fill_state({k[len("model."):]: v for k, v in enc.state_dict().items()},
                                                                   seed=int(z["w_seed"])).items()})
    enc = enc.to(DEV)
    enc.train(training)
    x = G.make_input(int(z["in_seed"]))
    y, h, w = enc.trunk(x.to(DEV))
    got = y.view(x.shape[0], h, w, 2048).permute(0, 3, 1, 2).float().cpu().numpy()
    ref = z["out_train" if training else "out_eval"]
    assert got.shape == ref.shape and np.abs(got - ref).max() < (1e-3 if training else 2e-4) * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("training", [False, True])
def test_trunk_matches_the_functional_restatement(dtype, tol, training):
    enc = _encoder(dtype)
    enc.train(training)
    sd = _sd_cpu(enc)
    # the bf16 path is compared with the restatement WITH its rounding points (see oracle/resnet_oracle.trunk: under batch
    # statistics a randomly initialised ResNet amplifies bf16 rounding ~100x; against the unrounded fp32 restatement the
    # eval() output differs by ~1 % and the train() output by ~45 % -- and so does the rounded restatement itself)
    bf = dtype == torch.bfloat16
    if bf and training:
        pytest.skip("end-to-end bf16 under batch statistics is chaotic at random init: checked block by block below")
    rnd = (lambda t: t.to(torch.bfloat16).float()) if bf else None
    Bx, Hx, Wx = 3, 96, 64
    x = torch.randn(Bx, 3, Hx, Wx, generator=torch.Generator().manual_seed(9))
    ref = R.trunk(sd, x, training, rnd=rnd)                                  # [B,2048,h,w]; updates sd's running stats
    y, h, w = enc.trunk(x.to(DEV))
    got = y.view(Bx, h, w, 2048).permute(0, 3, 1, 2).float().cpu()
    assert got.shape == ref.shape == (Bx, 2048, Hx // 32, Wx // 32)
    err = float((got - ref).abs().max() / ref.abs().max())
    l2 = float((got - ref).norm() / ref.norm())
    print(f"trunk {dtype} training={training}: max-abs error / scale = {err:.2e}, relative L2 = {l2:.2e}")
    assert l2 < tol and err < 10 * tol
    if training:                                                            # running statistics moved like nn.BatchNorm2d's
        new = _sd_cpu(enc)
        for k in ("model.1.running_mean", "model.1.running_var", "model.7.2.bn3.running_mean", "model.7.2.bn3.running_var"):
            d = float((new[k] - sd[k]).abs().max() / (sd[k].abs().max() + 1e-6))
            assert d < (1e-4 if dtype == torch.float32 else 3e-2), (k, d)
        assert int(new["model.1.num_batches_tracked"]) == 1


@pytest.mark.parametrize("li,bi", [(4, 0), (5, 0), (6, 3), (7, 2)])
def test_bf16_blocks_under_batch_statistics(li, bi):
    """bf16 path, train() BatchNorm, one bottleneck at a time (first blocks have the strided 3x3 + downsample branch):
    same bf16 input, same rounding points as the product -> agreement at bf16 rounding level, no deep amplification."""
    enc = _encoder(torch.bfloat16).train()
    sd = _sd_cpu(enc)
    blk = enc.model[li][bi]
    C = blk.conv1.in_channels
    B, H, W = 4, 12, 10
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(li * 10 + bi)).abs())      # post-ReLU-like input
    ref = R.block(sd, f"model.{li}.{bi}", x, blk.stride, True, rnd=bf)
    y = x.permute(0, 2, 3, 1).reshape(B * H * W, C).to(torch.bfloat16).to(DEV).contiguous()
    out, H2, W2, C2 = enc._block(y, B, H, W, C, blk)
    got = out.view(B, H2, W2, C2).permute(0, 3, 1, 2).float().cpu()
    l2 = float((got - ref).norm() / ref.norm())
    print(f"block model.{li}.{bi}: relative L2 {l2:.2e}")
    assert got.shape == ref.shape and l2 < 1e-2


def test_forward_contract_and_pixels_into_cxrbert():
    """image.py:54-69: sorted, unique sampled positions shared by the batch, features gathered at them; and the whole
    pixels -> region features -> CXRBERT.forward path runs with the encoder attached."""
    enc = _encoder(torch.bfloat16, n=5).eval()
    x = torch.randn(2, 3, 128, 128, generator=torch.Generator().manual_seed(3)).to(DEV)
    feats, pos = enc(x)
    assert feats.shape == (2, 5, 2048) and pos.shape == (2, 5) and pos.dtype == torch.int64
    p = pos[0].tolist()
    assert p == sorted(set(p)) and max(p) < 16 and torch.equal(pos[0], pos[1])
    full, h, w = enc.trunk(x)
    assert torch.equal(feats, full.view(2, h * w, 2048)[:, pos[0]])
    cfg = mv.ModelConfig(hidden=128, heads=2, intermediate=512, layers=1, vocab_size=1024, max_pos=64)
    from types import SimpleNamespace
    model = mv.CXRBERT(cfg, SimpleNamespace(num_image_embeds=5), dtype=torch.bfloat16, device=DEV, img_encoder="resnet50").eval()
    b = mv.data.synthetic_batch(cfg.vocab_size, 2, 5, 20, "full", seed=1, device=DEV)
    mlm, itm = model(b["cls_tok"], b["input_txt"], b["attn_mask"], b["segment"], x, b["sep_tok"])
    assert mlm.shape == (2, 5 + 20 + 3, 1024) and itm.shape == (2, 2) and torch.isfinite(mlm).all()
    sd = model.state_dict()
    assert "enc.img_encoder.model.0.weight" in sd and "enc.img_encoder.model.7.2.bn3.running_var" in sd
    model.load_state_dict(sd)


def test_conv_support_kernels_against_torch():
    g = torch.Generator().manual_seed(5)
    B, C, H, W = 2, 16, 9, 7
    x = torch.randn(B, C, H, W, generator=g)
    nhwc = torch.empty((B * H * W, C), dtype=torch.bfloat16, device=DEV)
    ops.nchw_to_nhwc(x.to(DEV), nhwc, B, C, H, W, C)
    assert torch.equal(nhwc.view(B, H, W, C).float().cpu(), x.permute(0, 2, 3, 1).to(torch.bfloat16).float())
    for (k, s, p) in ((3, 1, 1), (3, 2, 1), (1, 2, 0), (7, 2, 3)):
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        ldk = (k * k * C + 7) // 8 * 8 + 8
        cols = torch.full((B * Ho * Wo, ldk), 7.0, dtype=torch.bfloat16, device=DEV)
        ops.im2col(nhwc, cols, B, H, W, C, k, k, s, p, ldk)
        ref = torch.nn.functional.unfold(x.to(torch.bfloat16).float(), k, padding=p, stride=s)          # [B, C*k*k, L], (c, ky, kx) order
        ref = ref.view(B, C, k * k, Ho * Wo).permute(0, 3, 2, 1).reshape(B * Ho * Wo, k * k * C)
        assert torch.equal(cols[:, :k * k * C].float().cpu(), ref) and bool((cols[:, k * k * C:] == 0).all()), (k, s, p)
    mp = torch.empty((B * 5 * 4, C), dtype=torch.bfloat16, device=DEV)
    ops.maxpool3x3s2(nhwc, mp, B, H, W, C)
    ref = torch.nn.functional.max_pool2d(x.to(torch.bfloat16).float(), 3, 2, 1).permute(0, 2, 3, 1).reshape(-1, C)
    assert torch.equal(mp.float().cpu(), ref)
    st = torch.empty((2, C), dtype=torch.float32, device=DEV)
    ops.col_stats(nhwc, C, B * H * W, C, st)
    xf = nhwc.float()
    assert torch.allclose(st[0], xf.sum(0), rtol=1e-5, atol=1e-4) and torch.allclose(st[1], (xf * xf).sum(0), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("C,O,k,s,p,H,W", [(64, 64, 3, 1, 1, 20, 17), (128, 128, 3, 2, 1, 19, 22), (256, 512, 1, 2, 0, 14, 9),
                                           (8, 64, 7, 2, 3, 37, 40), (512, 512, 3, 1, 1, 7, 5)])
def test_implicit_gemm_convolution_against_torch(C, O, k, s, p, H, W):
    """mv_conv2d (taps gathered while the MFMA kernel stages its operand) == conv2d on the same bf16 inputs, and ==
    the materialised-patch path (mv_im2col + mv_gemm) it replaces."""
    g = torch.Generator().manual_seed(C + O + k)
    B = 3
    x = torch.randn(B, C, H, W, generator=g).to(torch.bfloat16)
    w = (torch.randn(O, C, k, k, generator=g) / math.sqrt(C * k * k)).to(torch.bfloat16)
    ref = torch.nn.functional.conv2d(x.float(), w.float(), stride=s, padding=p)
    Ho, Wo = ref.shape[2:]
    xm = x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(DEV)
    wm = w.permute(0, 2, 3, 1).reshape(O, k * k * C).contiguous().to(DEV)
    y = torch.full((B * Ho * Wo + 1, O), 7.0, dtype=torch.float32, device=DEV)
    ops.conv2d(xm, wm, y[:-1], B, H, W, C, O, k, k, s, p)
    got = y[:-1].view(B, Ho, Wo, O).permute(0, 3, 1, 2).cpu()
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-5 * math.sqrt(C * k * k) and bool((y[-1] == 7.0).all())
    cols = torch.empty((B * Ho * Wo, k * k * C), dtype=torch.bfloat16, device=DEV)
    ops.im2col(xm, cols, B, H, W, C, k, k, s, p, k * k * C)
    y2 = torch.empty((B * Ho * Wo, O), dtype=torch.float32, device=DEV)
    ops.gemm(cols, wm, y2, M=B * Ho * Wo, N=O, K=k * k * C)
    assert float((y2 - y[:-1]).abs().max() / ref.abs().max()) < 1e-5


def test_trunk_is_the_same_with_materialised_patches():
    enc = _encoder(torch.bfloat16).eval()
    x = torch.randn(2, 3, 96, 64, generator=torch.Generator().manual_seed(2)).to(DEV)
    a, h, w = enc.trunk(x)
    enc.implicit_conv = False
    b, _, _ = enc.trunk(x)
    assert float((a.float() - b.float()).norm() / b.float().norm()) < 2e-3


def test_trainer_consumes_pixel_batches():
    """main_origin.py's loop with the Dataset's real 9-tuple (pixels in slot 4): the trainer runs the region encoder."""
    from types import SimpleNamespace
    V, B, N, S = 1024, 2, 4, 20
    args = SimpleNamespace(num_image_embeds=N, lr=1e-4, pixels=True, img_encoder="random-pixel", bert_model="x")
    cfg = dict(vocab_size=V, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=512, max_position_embeddings=64)
    b = mv.data.synthetic_batch(V, B, N, S, "full", seed=2, device="cpu")
    px = torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    loader = [(b["cls_tok"], b["input_txt"], b["txt_labels"], b["attn_mask"], px, b["segment"], b["is_aligned"], b["sep_tok"], None)]
    tr = mv.CXRBERT_Trainer(args, train_dataloader=loader, test_dataloader=None, config=cfg)
    assert isinstance(tr.model.img_encoder, mv.ImageEncoder_cnn)
    out = tr.train(0)
    assert np.isfinite(out["avg_loss"])


def test_folded_batchnorm_equals_the_separate_pass_in_eval():
    """eval(), bf16: convolution + folded BatchNorm (+ residual)(+ ReLU) in the GEMM epilogue vs convolution, then
    mv_bn_act with the running statistics (different rounding points: folded weights are rounded after scaling)."""
    enc = _encoder(torch.bfloat16).eval()
    x = torch.randn(2, 3, 96, 64, generator=torch.Generator().manual_seed(6)).to(DEV)
    a, h, w = enc.trunk(x)
    enc.fold_bn = False
    b, _, _ = enc.trunk(x)
    l2 = float((a.float() - b.float()).norm() / b.float().norm())
    print("folded vs separate BatchNorm (eval, bf16): relative L2", l2)
    assert l2 < 2e-2
    ref = R.trunk(_sd_cpu(enc), x.cpu(), False)
    got = a.view(2, h, w, 2048).permute(0, 3, 1, 2).float().cpu()
    assert float((got - ref).norm() / ref.norm()) < 3e-2


@pytest.mark.parametrize("epi", ["relu", "res_relu"])
def test_relu_epilogues(epi):
    from medvill_amd._lib import EPI_BIAS_RELU, EPI_BIAS_RES_RELU
    g = torch.Generator().manual_seed(4)
    M, N, K = 300, 136, 72
    a, b = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV), (torch.randn(N, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    bias, r = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(torch.bfloat16).to(DEV)
    y = torch.empty((M, N), dtype=torch.float32, device=DEV)
    ref = a.double() @ b.double().t() + bias.double()
    if epi == "relu":
        ops.gemm(a, b, y, M=M, N=N, K=K, bias=bias, epi=EPI_BIAS_RELU)
    else:
        ops.gemm(a, b, y, M=M, N=N, K=K, bias=bias, epi=EPI_BIAS_RES_RELU, r=r)
        ref = ref + r.double()
    ref = ref.clamp(min=0)
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 1e-4
