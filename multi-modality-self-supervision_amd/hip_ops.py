"""Tensor-level wrappers over the C ABI (include/medvill.h).  torch is used for device memory
and streams only; every wrapper enqueues on torch's current stream and returns immediately."""
from __future__ import annotations

import torch

from . import _lib as L
from ._lib import (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GELU_D, EPI_BIAS_RELU, EPI_BIAS_RES, EPI_BIAS_RES_RELU, EPI_BIAS_TANH, EPI_DGELU,  # noqa: F401
                   EPI_MUL, EPI_NONE, EPI_RES, MV_BF16, MV_F16,
                   MV_F32)


def _lib():
    return L.load()


def gemm(a, b, c, *, ta=False, tb=False, M, N, K, lda=None, ldb=None, ldc=None, bias=None, epi=EPI_NONE, r=None, ldr=None,
         c2=None, ldc2=None, c3=None, ldc3=None, splitk=1, ws=None, accumulate=False, p_drop=0.0, drop_key=0, alpha=None, colsum_part=None):
    """c[M,N] = epi(op(a)[M,K] . op(b)[K,N]); c3 (optional): the same result in a second 16-bit encoding; alpha (optional,
    f32 device scalar): factor on the product (weight gradients of the f16-gradient path); see mv_gemm in include/medvill.h."""
    L.require_cuda(a, b, c, bias, r, c2, c3, ws, alpha, colsum_part)
    if colsum_part is not None and (colsum_part.dtype != torch.float32 or colsum_part.numel() < 2 * ((M + 255) // 256) * N):
        raise ValueError("colsum_part: f32 [2*ceil(M/256), N]")
    lda = lda if lda is not None else (M if ta else K)
    ldb = ldb if ldb is not None else (N if tb else K)
    ldc = ldc if ldc is not None else N
    ldr = ldr if ldr is not None else N
    ldc2 = ldc2 if ldc2 is not None else N
    ldc3 = ldc3 if ldc3 is not None else N
    if a.dtype != b.dtype:
        raise TypeError("gemm operands must share a dtype")
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError("bias must be f32")
    rc = _lib().mv_gemm(L.dt_of(a), int(ta), int(tb), M, N, K, L.ptr(a), lda, L.ptr(b), ldb, L.ptr(c), ldc, L.dt_of(c),
                        L.ptr(bias), epi, L.ptr(r), ldr, L.dt_of(r) if r is not None else 0, L.ptr(c2), ldc2,
                        L.ptr(c3), ldc3, L.dt_of(c3) if c3 is not None else 0, splitk, L.ptr(ws), (ws.numel() * 4) if ws is not None else 0, int(accumulate), float(p_drop), int(drop_key),
                        L.ptr(alpha), L.ptr(colsum_part), L.stream_ptr())
    L.check(rc, f"mv_gemm(M={M},N={N},K={K},ta={ta},tb={tb},epi={epi})")
    return c


def mask_pack(mask, bits, tileinfo):
    L.require_cuda(mask, bits, tileinfo)
    if mask.dtype != torch.int64:
        raise TypeError("attn_mask must be int64 (as built by the reference Dataset)")
    if mask.dim() not in (2, 3):
        raise NotImplementedError          # cxrbert_origin.py:80-81
    B, Lq = mask.shape[0], mask.shape[-1]
    rc = _lib().mv_mask_pack(L.ptr(mask.contiguous()), mask.dim(), B, Lq, L.ptr(bits), L.ptr(tileinfo), L.stream_ptr())
    L.check(rc, "mv_mask_pack")


def mask_build(desc, B, Lq, bits, tileinfo):
    """bits / tile classes from int32 [B,3] descriptors {family, n2, vl} (see mv_mask_build)."""
    L.require_cuda(desc, bits, tileinfo)
    if desc.dtype != torch.int32 or tuple(desc.shape) != (B, 3):
        raise TypeError("desc must be int32 [B,3]")
    rc = _lib().mv_mask_build(L.ptr(desc.contiguous()), B, Lq, L.ptr(bits), L.ptr(tileinfo), L.stream_ptr())
    L.check(rc, "mv_mask_build")


def mask_verify_host(mask, desc, threads=4):
    """HOST tensors: int64 [B,L,L] / [B,L] reference mask against int32 [B,3] descriptors (see mv_mask_verify_host).
    -> -1 when every entry agrees, else the linear index of the first mismatch.  Pure host work; releases the GIL."""
    import ctypes
    if mask.is_cuda or desc.is_cuda or mask.dtype != torch.int64 or desc.dtype != torch.int32:
        raise TypeError("mask_verify_host: host int64 mask and host int32 [B,3] descriptors")
    if mask.dim() not in (2, 3):
        raise NotImplementedError          # cxrbert_origin.py:80-81
    B, Lq = mask.shape[0], mask.shape[-1]
    if tuple(desc.shape) != (B, 3):
        raise TypeError("desc must be int32 [B,3]")
    mask, desc = mask.contiguous(), desc.contiguous()
    out = ctypes.c_longlong(0)
    rc = _lib().mv_mask_verify_host(mask.data_ptr(), mask.dim(), desc.data_ptr(), B, Lq, int(threads), ctypes.byref(out))
    L.check(rc, "mv_mask_verify_host")
    return int(out.value)


def mlm_draws(key, B, S, vocab, device):
    """(u f32 [B,S], rnd int32 [B,S]): the counter-based stand-ins for random_word's two random sources."""
    u = torch.empty((B, S), dtype=torch.float32, device=device)
    rnd = torch.empty((B, S), dtype=torch.int32, device=device)
    L.require_cuda(u)
    rc = _lib().mv_mlm_draws(int(key) & 0xFFFFFFFFFFFFFFFF, B, S, vocab, L.ptr(u), L.ptr(rnd), L.stream_ptr())
    L.check(rc, "mv_mlm_draws")
    return u, rnd


def mlm_corrupt(ids, lengths, u, rnd, N, family=None, want_index=True):
    """On-device sample assembly (see mv_mlm_corrupt).  Returns a dict of device tensors; `n_labels` stays on the
    device (one int32) so the caller decides when to synchronise."""
    L.require_cuda(ids, lengths, u, rnd, family)
    B, S = ids.shape
    T, Lq = S + 1, S + N + 3
    if ids.dtype != torch.int64 or lengths.dtype != torch.int32 or u.dtype != torch.float32 or rnd.dtype != torch.int32:
        raise TypeError("mlm_corrupt: ids int64, lengths int32, u f32, rnd int32")
    if tuple(lengths.shape) != (B,) or tuple(u.shape) != (B, S) or tuple(rnd.shape) != (B, S):
        raise ValueError("mlm_corrupt: shape mismatch")
    if family is not None and (family.dtype != torch.int32 or tuple(family.shape) != (B,)):
        raise TypeError("mlm_corrupt: family must be int32 [B]")
    dev = ids.device
    out = dict(input_txt=torch.empty((B, T), dtype=torch.int64, device=dev),
               segment=torch.empty((B, T), dtype=torch.int64, device=dev),
               txt_labels=torch.empty((B, Lq), dtype=torch.int64, device=dev),
               n_ids=torch.empty((B,), dtype=torch.int32, device=dev),
               desc=torch.empty((B, 3), dtype=torch.int32, device=dev),
               counts=torch.empty((B,), dtype=torch.int32, device=dev))
    if want_index:
        out["label_rows"] = torch.empty((B * S,), dtype=torch.int32, device=dev)
        out["label_ids"] = torch.empty((B * S,), dtype=torch.int32, device=dev)
        out["n_labels"] = torch.empty((1,), dtype=torch.int32, device=dev)
    rc = _lib().mv_mlm_corrupt(L.ptr(ids.contiguous()), L.ptr(lengths.contiguous()), L.ptr(u.contiguous()), L.ptr(rnd.contiguous()),
                               L.ptr(family), B, N, S, L.ptr(out["input_txt"]), L.ptr(out["segment"]), L.ptr(out["txt_labels"]),
                               L.ptr(out["n_ids"]), L.ptr(out["desc"]), L.ptr(out["counts"]), L.ptr(out.get("label_rows")),
                               L.ptr(out.get("label_ids")), L.ptr(out.get("n_labels")), L.stream_ptr())
    L.check(rc, "mv_mlm_corrupt")
    return out


def pack_plan(desc, B, Lq):
    """(cu int32 [B+1], rowmap int32 [B*L], inv int32 [B*L]) from mask descriptors (see mv_pack_plan); device tensors."""
    L.require_cuda(desc)
    if desc.dtype != torch.int32 or tuple(desc.shape) != (B, 3):
        raise TypeError("desc must be int32 [B,3]")
    dev = desc.device
    cu = torch.empty((B + 1,), dtype=torch.int32, device=dev)
    rowmap = torch.empty((B * Lq,), dtype=torch.int32, device=dev)
    inv = torch.empty((B * Lq,), dtype=torch.int32, device=dev)
    rc = _lib().mv_pack_plan(L.ptr(desc.contiguous()), B, Lq, L.ptr(cu), L.ptr(rowmap), L.ptr(inv), L.stream_ptr())
    L.check(rc, "mv_pack_plan")
    return cu, rowmap, inv


def tail_perm(cu, B, Lq, sel, total_rows):
    """(perm, newpos, qlim, sel_new) of mv_tail_perm: the last layer's row order with the consumed rows `sel` (packed row indices) first."""
    L.require_cuda(cu, sel)
    if cu.dtype != torch.int32 or sel.dtype != torch.int32:
        raise TypeError("tail_perm: int32 tensors")
    dev = cu.device
    perm = torch.empty((total_rows,), dtype=torch.int32, device=dev)
    newpos = torch.empty((total_rows,), dtype=torch.int32, device=dev)
    qlim = torch.empty((B,), dtype=torch.int32, device=dev)
    sel_new = torch.empty((sel.numel(),), dtype=torch.int32, device=dev)
    rc = _lib().mv_tail_perm(L.ptr(cu), B, Lq, L.ptr(sel.contiguous()), int(sel.numel()), L.ptr(perm), L.ptr(newpos), L.ptr(qlim), L.ptr(sel_new),
                             L.stream_ptr())
    L.check(rc, "mv_tail_perm")
    return perm, newpos, qlim, sel_new


def dropbits_numel(B, Lq, A):
    """uint32 elements of one layer's attention-dropout keep-bits (mv_attn_dropmask)."""
    return B * A * ((Lq + 31) // 32) * ((Lq + 63) // 64) * 64


def attn_dropmask(p_drop, drop_key, B, Lq, A, out, cu=None):
    """Keep-bits of the attention-probability dropout mask of (p_drop, drop_key), in the layout the MFMA kernels select with."""
    L.require_cuda(out, cu)
    if out.dtype != torch.int32 or out.numel() < dropbits_numel(B, Lq, A):
        raise TypeError("attn_dropmask: out must be int32 with dropbits_numel(B, L, A) elements")
    rc = _lib().mv_attn_dropmask(float(p_drop), int(drop_key), B, Lq, A, L.ptr(cu), L.ptr(out), L.stream_ptr())
    L.check(rc, "mv_attn_dropmask")
    return out


def attn_keep_mask(dropbits, B, Lq, A):
    """The keep-bits of mv_attn_dropmask decoded to a bool tensor [B, A, L, L] (keep[b, h, q, k]) -- inspection / tests.  Blocks the
    generator skipped (beyond a sample's packed length) decode to whatever the buffer held."""
    NQB, NKT = (Lq + 31) // 32, (Lq + 63) // 64
    w = dropbits[:dropbits_numel(B, Lq, A)].view(B, A, NQB, NKT, 2, 16, 2).to(torch.int64) & 0xFFFFFFFF     # [b, h, qb, kt, kk, r, half]
    i = torch.arange(32, device=dropbits.device, dtype=torch.int64)
    bit = ((w.unsqueeze(-1) >> i) & 1).bool()                      # [..., r, half, query-in-block]
    r = torch.arange(16, device=dropbits.device)
    key_of = ((r & 3) + 8 * (r >> 2)).view(16, 1) + 4 * torch.arange(2, device=dropbits.device).view(1, 2)      # [r, half] -> key in the 32-half
    out = torch.zeros((B, A, NQB, NKT, 2, 32, 32), dtype=torch.bool, device=dropbits.device)             # [.., kk, key32, q32]
    out[:, :, :, :, :, key_of.reshape(-1), :] = bit.reshape(B, A, NQB, NKT, 2, 32, 32)
    # -> [b, h, qb, q32, kt, kk, key32]
    out = out.permute(0, 1, 2, 6, 3, 4, 5).reshape(B, A, NQB * 32, NKT * 64)
    return out[:, :, :Lq, :Lq]


def attn_fwd(qkv, bits, tileinfo, ctx, lse, B, Lq, A, dh, p_drop=0.0, cu=None, total_rows=0, ctx_bf16=None, dropbits=None, qlim=None):
    """p_drop > 0 needs `dropbits` (attn_dropmask).  qlim (int32 [B], optional): only the first qlim[b] rows of a sample are queries."""
    if ctx.dtype != qkv.dtype or (ctx_bf16 is not None and ctx_bf16.dtype != torch.bfloat16):
        raise TypeError("attn_fwd: ctx shares qkv's encoding; the second output is bf16")
    L.require_cuda(dropbits)
    rc = _lib().mv_attn_fwd(L.dt_of(qkv), L.ptr(qkv), L.ptr(bits), L.ptr(tileinfo), L.ptr(ctx), L.ptr(ctx_bf16), L.ptr(lse), B, Lq, A, dh,
                            float(p_drop), L.ptr(dropbits), L.ptr(cu), int(total_rows), L.ptr(qlim), L.stream_ptr())
    L.check(rc, "mv_attn_fwd")


def attn_bwd(qkv, ctx, dctx, lse, bits, tileinfo, dqkv, delta, B, Lq, A, dh, p_drop=0.0, cu=None, total_rows=0, dropbits=None, qlim=None):
    L.require_cuda(dropbits, qlim)
    rc = _lib().mv_attn_bwd(L.dt_of(qkv), L.ptr(qkv), L.ptr(ctx), L.ptr(dctx), L.ptr(lse), L.ptr(bits), L.ptr(tileinfo),
                            L.ptr(dqkv), L.ptr(delta), B, Lq, A, dh, float(p_drop), L.ptr(dropbits), L.ptr(cu), int(total_rows),
                            L.ptr(qlim), L.stream_ptr())
    L.check(rc, "mv_attn_bwd")


def layernorm_fwd(x, gamma, beta, y, mean, rstd, M, H, eps, y_bf16=None):
    if y_bf16 is not None and y_bf16.dtype != torch.bfloat16:
        raise TypeError("layernorm_fwd: the second output is bf16")
    rc = _lib().mv_layernorm_fwd(L.dt_of(y), L.ptr(x), L.dt_of(x), L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(y_bf16), L.ptr(mean),
                                 L.ptr(rstd), M, H, float(eps), L.stream_ptr())
    L.check(rc, "mv_layernorm_fwd")


def layernorm_bwd(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, colsum, M, H, dx_drop=None, p_drop=0.0, drop_key=0, unscale=None):
    rc = _lib().mv_layernorm_bwd(L.dt_of(dy), L.ptr(dy), L.ptr(x), L.dt_of(x), L.ptr(mean), L.ptr(rstd), L.ptr(gamma),
                                 L.ptr(dx), L.ptr(dgamma), L.ptr(dbeta), L.ptr(colsum), M, H, L.ptr(dx_drop), float(p_drop),
                                 int(drop_key), L.ptr(unscale), L.stream_ptr())
    L.check(rc, "mv_layernorm_bwd")


def embed_fwd(dt, cls_tok, txt, segment, img_pos, sep_tok, imgproj, E, P, Ty, gamma, beta, x0, pre, mean, rstd, B, N, T, H, V,
              maxpos, eps, p_drop=0.0, drop_key=0, rowmap=None, n_rows=0, x0_bf16=None, p_drop_img=None):
    """img_pos None: the image rows get no position embedding (args.img_postion false); p_drop_img: dropout probability of the image
    rows (args.dropout_prob, cxrbert_origin.py:19; default = p_drop)."""
    if any(L.dt_of(t) != dt for t in (E, P, Ty, x0)) or (imgproj is not None and L.dt_of(imgproj) != dt):
        raise TypeError("embed_fwd: tables, imgproj and x0 must be in the encoding `dt`")
    rc = _lib().mv_embed_fwd(dt, L.ptr(cls_tok), L.ptr(txt), L.ptr(segment), L.ptr(img_pos), L.ptr(sep_tok), L.ptr(imgproj),
                             L.ptr(E), L.ptr(P), L.ptr(Ty), L.ptr(gamma), L.ptr(beta), L.ptr(x0), L.ptr(x0_bf16), L.ptr(pre), L.ptr(mean),
                             L.ptr(rstd), B, N, T, H, V, maxpos, float(eps), float(p_drop),
                             float(p_drop if p_drop_img is None else p_drop_img), int(drop_key), L.ptr(rowmap), int(n_rows),
                             L.stream_ptr())
    L.check(rc, "mv_embed_fwd")


def embed_bwd(dt, dx0, pre, mean, rstd, gamma, cls_tok, txt, segment, img_pos, sep_tok, dE, dP, dTy, dgamma, dbeta, dimgproj, B,
              N, T, H, V, maxpos, pad_token_id=0, p_drop=0.0, drop_key=0, rowmap=None, n_rows=0, unscale=None, p_drop_img=None):
    rc = _lib().mv_embed_bwd(dt, L.ptr(dx0), L.ptr(pre), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(cls_tok), L.ptr(txt),
                             L.ptr(segment), L.ptr(img_pos), L.ptr(sep_tok), L.ptr(dE), L.ptr(dP), L.ptr(dTy), L.ptr(dgamma),
                             L.ptr(dbeta), L.ptr(dimgproj), B, N, T, H, V, maxpos, int(pad_token_id), float(p_drop),
                             float(p_drop if p_drop_img is None else p_drop_img), int(drop_key),
                             L.ptr(rowmap), int(n_rows), L.ptr(unscale), L.stream_ptr())
    L.check(rc, "mv_embed_bwd")


def ce_fwd_bwd(logits, ld, labels, R, V, out, dlogits=None, ldd=0, grad_scale_dev=None, grad_scale=1.0, loss_scale_dev=None):
    if labels.dtype != torch.int32:
        raise TypeError("labels must be int32")
    rc = _lib().mv_ce_fwd_bwd(L.ptr(logits), L.dt_of(logits), ld, L.ptr(labels), R, V, L.ptr(out), L.ptr(dlogits),
                              L.dt_of(dlogits) if dlogits is not None else 0, ldd, L.ptr(grad_scale_dev), float(grad_scale),
                              L.ptr(loss_scale_dev), L.stream_ptr())
    L.check(rc, "mv_ce_fwd_bwd")


def gather_rows(src, lds, rows, R, H, dst, ldd):
    rc = _lib().mv_gather_rows(L.dt_of(src), L.ptr(src), lds, L.ptr(rows), R, H, L.ptr(dst), ldd, L.stream_ptr())
    L.check(rc, "mv_gather_rows")


def scatter_rows(src, lds, rows, R, H, dst, ldd, accumulate=False):
    rc = _lib().mv_scatter_rows(L.dt_of(src), L.ptr(src), lds, L.ptr(rows), R, H, L.ptr(dst), ldd, int(accumulate),
                                L.stream_ptr())
    L.check(rc, "mv_scatter_rows")


def colsum(x, ldx, M, N, out, accumulate=True, unscale=None):
    rc = _lib().mv_colsum(L.dt_of(x), L.ptr(x), ldx, M, N, L.ptr(out), int(accumulate), L.ptr(unscale), L.stream_ptr())
    L.check(rc, "mv_colsum")


def colsum_partials(part, P, ld, N, out, unscale=None):
    """out[n] += [unscale] * sum_p part[p, n] (the partial column sums of gemm(colsum_part=))."""
    L.require_cuda(part, out, unscale)
    rc = _lib().mv_colsum_partials(L.ptr(part), P, ld, N, L.ptr(out), L.ptr(unscale), L.stream_ptr())
    L.check(rc, "mv_colsum_partials")


def add(a, b, c, n):
    rc = _lib().mv_add(L.dt_of(a), L.ptr(a), L.ptr(b), L.ptr(c), n, L.stream_ptr())
    L.check(rc, "mv_add")


def dact(mode, dy, z, out, n):
    rc = _lib().mv_dact(L.dt_of(dy), mode, L.ptr(dy), L.ptr(z), L.ptr(out), n, L.stream_ptr())
    L.check(rc, "mv_dact")


def cast(src, dst, n):
    rc = _lib().mv_cast(L.ptr(src), L.dt_of(src), L.ptr(dst), L.dt_of(dst), n, L.stream_ptr())
    L.check(rc, "mv_cast")


def transpose(src, dst, rows, cols, lds=None, ldd=None):
    """dst[c, r] = src[r, c] (same dtype)."""
    L.require_cuda(src, dst)
    if src.dtype != dst.dtype:
        raise TypeError("transpose: dtypes differ")
    rc = _lib().mv_transpose(L.dt_of(src), L.ptr(src), lds if lds is not None else cols, L.ptr(dst), ldd if ldd is not None else rows,
                             rows, cols, L.stream_ptr())
    L.check(rc, "mv_transpose")
    return dst


def nchw_to_nhwc(src, dst, B, C, H, W, Cp):
    L.require_cuda(src, dst)
    if src.dtype != torch.float32:
        raise TypeError("nchw_to_nhwc: source pixels must be float32")
    rc = _lib().mv_nchw_to_nhwc(L.ptr(src), L.ptr(dst), L.dt_of(dst), B, C, H, W, Cp, L.stream_ptr())
    L.check(rc, "mv_nchw_to_nhwc")


def im2col(src, dst, B, H, W, C, kh, kw, stride, pad, ldk):
    L.require_cuda(src, dst)
    rc = _lib().mv_im2col(L.dt_of(src), L.ptr(src), B, H, W, C, kh, kw, stride, pad, L.ptr(dst), ldk, L.stream_ptr())
    L.check(rc, "mv_im2col")


def conv2d(x, w, y, B, H, W, C, O, kh, kw, stride, pad, bias=None, epi=EPI_NONE, r=None):
    """implicit-GEMM convolution over NHWC x [B*H*W, C] with w [O, kh*kw*C] -> y [B*Ho*Wo, O] (see mv_conv2d)."""
    L.require_cuda(x, w, y, bias, r)
    rc = _lib().mv_conv2d(L.dt_of(x), L.ptr(x), L.ptr(w), L.ptr(y), L.dt_of(y), B, H, W, C, O, kh, kw, stride, pad, L.ptr(bias), epi,
                          L.ptr(r), L.dt_of(r) if r is not None else 0, L.stream_ptr())
    L.check(rc, "mv_conv2d")


def col_stats(x, ldx, rows, C, stats):
    L.require_cuda(x, stats)
    rc = _lib().mv_col_stats(L.dt_of(x), L.ptr(x), ldx, rows, C, L.ptr(stats), L.stream_ptr())
    L.check(rc, "mv_col_stats")


def bn_finalize(stats, C, rows, eps, momentum, mean, rstd, running_mean=None, running_var=None):
    L.require_cuda(stats, mean, rstd, running_mean, running_var)
    rc = _lib().mv_bn_finalize(L.ptr(stats), C, int(rows), float(eps), float(momentum), L.ptr(mean), L.ptr(rstd),
                               L.ptr(running_mean), L.ptr(running_var), L.stream_ptr())
    L.check(rc, "mv_bn_finalize")


def bn_act(x, mean, rstd, gamma, beta, y, rows, C, residual=None, relu=True):
    L.require_cuda(x, mean, rstd, gamma, beta, y, residual)
    rc = _lib().mv_bn_act(L.dt_of(y), L.ptr(x), L.dt_of(x), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(beta), L.ptr(residual), L.ptr(y),
                          int(rows), C, int(relu), L.stream_ptr())
    L.check(rc, "mv_bn_act")


def maxpool3x3s2(x, y, B, H, W, C):
    L.require_cuda(x, y)
    rc = _lib().mv_maxpool3x3s2(L.dt_of(x), L.ptr(x), L.ptr(y), B, H, W, C, L.stream_ptr())
    L.check(rc, "mv_maxpool3x3s2")


def cast2d(src, lds, dst, ldd, rows, cols):
    rc = _lib().mv_cast2d(L.ptr(src), L.dt_of(src), lds, L.ptr(dst), L.dt_of(dst), ldd, rows, cols, L.stream_ptr())
    L.check(rc, "mv_cast2d")


def adamw_step(p, g, m, v, shadow, n, lr, b1, b2, eps, wd, step, correct_bias=True, grad_scale=1.0, shadow_f16=None, scaler_state=None):
    rc = _lib().mv_adamw_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.ptr(shadow), L.ptr(shadow_f16), n, float(lr), float(b1), float(b2),
                              float(eps), float(wd), int(step), int(correct_bias), float(grad_scale), L.ptr(scaler_state), L.stream_ptr())
    L.check(rc, "mv_adamw_step")


def count_nonfinite(x, counter):
    """counter[0] += number of inf / nan elements of the f32 tensor x (device)."""
    L.require_cuda(x, counter)
    if x.dtype != torch.float32 or counter.dtype != torch.float32:
        raise TypeError("count_nonfinite: f32 tensors")
    rc = _lib().mv_count_nonfinite(L.ptr(x), x.numel(), L.ptr(counter), L.stream_ptr())
    L.check(rc, "mv_count_nonfinite")


def scaler_update(state, growth_interval=2000, growth=2.0, backoff=0.5, max_scale=2.0 ** 24, min_scale=1.0):
    """Dynamic loss scale: consume the non-finite count in state[6], set the skip flag / step count, adapt the scale."""
    L.require_cuda(state)
    rc = _lib().mv_scaler_update(L.ptr(state), int(growth_interval), float(growth), float(backoff), float(max_scale), float(min_scale),
                                 L.stream_ptr())
    L.check(rc, "mv_scaler_update")


def dropout_mask(p_drop, drop_key, n, device):
    """(keep uint8 [n], scale) of the kernels' counter-based dropout for linear indices 0..n-1."""
    import ctypes
    keep = torch.empty(n, dtype=torch.uint8, device=device)
    sc = ctypes.c_float(1.0)
    rc = _lib().mv_dropout_mask(float(p_drop), int(drop_key), n, L.ptr(keep), ctypes.byref(sc), L.stream_ptr())
    L.check(rc, "mv_dropout_mask")
    return keep, float(sc.value)


# ---- kernel-forcing knobs: tests and timing experiments only (include/medvill_debug.h).  The product library has none of this state;
# a knob off its default routes the process's calls through libmedvill_hip_dbg.so (medvill_amd._lib.set_knob). ----
def set_impl(impl: int):
    L.set_knob("impl", 1 if impl else 0)


def get_impl() -> int:
    return L.get_knob("impl")


def set_attn_planes(planes: int):
    """Bits per uniform of the attention-dropout mask generator (16, 12 or 8)."""
    L.set_knob("attn_planes", planes if planes in (8, 12) else 16)


def attn_drop_prob(p_drop: float) -> float:
    """The drop probability attn_dropmask realises for p_drop at the current plane count."""
    n = L.get_knob("attn_planes")
    return round(p_drop * (1 << n)) / float(1 << n)


def set_rowops_variant(v: int = 0):
    """Experiment hook of mv_layernorm_bwd (see include/medvill_debug.h)."""
    L.set_knob("rowops_variant", v)


def set_gemm_rounds(on: int = 1):
    """1 (default): mv_gemm picks the ring tile height that minimises whole rounds of CUs (csrc/mv_gemm.hip: gemm_route); 0: off."""
    L.set_knob("gemm_rounds", 1 if on else 0)


def set_attn_fwd(variant: int = 0):
    """Reserved: selects the two-sub-tile forward kernel when profiles/r05_two_subtile_attention_experiment.patch is applied."""
    L.set_knob("attn_fwd", variant)


def set_attn_order(order: int = 0):
    """Attention block order (see att_block in csrc/mv_attn.hip): 0 row block slowest, 1 a pair's row blocks adjacent on one XCD."""
    L.set_knob("attn_order", order)


class RcclComm:
    """The C ABI's own RCCL communicator (mv_comm_*: include/medvill.h), for hosts that do not go through torch.distributed.
    `medvill_amd.dist.GradAllReducer` (the product's exchange) uses torch's nccl backend instead -- the same library, one communicator
    per process.  unique_id(): rank 0 makes the 128-byte id and hands it to the other ranks by any transport."""

    def __init__(self, rank: int, world: int, unique_id: bytes):
        import ctypes
        if len(unique_id) != 128:
            raise ValueError("unique_id: 128 bytes (RcclComm.unique_id() on rank 0)")
        self._h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        L.check(_lib().mv_comm_init(ctypes.byref(self._h), int(rank), int(world), buf), "mv_comm_init")
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id() -> bytes:
        import ctypes
        buf = ctypes.create_string_buffer(128)
        L.check(_lib().mv_comm_unique_id(buf), "mv_comm_unique_id")
        return buf.raw

    def allreduce_async(self, t, stream=None):
        """In-place sum all-reduce of a contiguous f32 / f16 / bf16 device tensor, enqueued on `stream` (default: the current one)."""
        L.require_cuda(t)
        if not t.is_contiguous():
            raise ValueError("allreduce_async: contiguous tensor")
        st = (stream or torch.cuda.current_stream()).cuda_stream
        L.check(_lib().mv_comm_allreduce_async(self._h, L.ptr(t), t.numel(), L.dt_of(t), st), "mv_comm_allreduce_async")

    def wait(self, stream=None):
        """`stream` (default: the current one) waits on the device for every collective issued so far."""
        st = (stream or torch.cuda.current_stream()).cuda_stream
        L.check(_lib().mv_comm_wait(self._h, st), "mv_comm_wait")

    def destroy(self):
        if self._h:
            L.check(_lib().mv_comm_destroy(self._h), "mv_comm_destroy")
            self._h = None


def set_persistent_cus(n: int = 0):
    """The persistent (weight-gradient) GEMM kernels launch at most n blocks; 0 = one per CU."""
    L.set_knob("persistent_cus", max(int(n), 0))


def stream_with_cus(n_cus: int, device, first: int = 0, total: int = 256, n_xcd: int = 8):
    """torch stream whose kernels only run on `n_cus` compute units (a multiple of n_xcd: n_cus / n_xcd CUs of every XCD), starting at
    CU `first` (also a multiple of n_xcd).  Bit i of the mask is CU i // n_xcd of XCD i % n_xcd (see mv_stream_create_cumask)."""
    import ctypes
    if n_cus % n_xcd or first % n_xcd or n_cus <= 0 or first + n_cus > total:
        raise ValueError("n_cus and first must be multiples of the XCD count and fit the device")
    words = (ctypes.c_uint32 * ((total + 31) // 32))()
    for i in range(first, first + n_cus):
        words[i // 32] |= 1 << (i % 32)
    out = ctypes.c_void_p()
    with torch.cuda.device(device):
        L.check(_lib().mv_stream_create_cumask(words, len(words), ctypes.byref(out)), "mv_stream_create_cumask")
    return torch.cuda.ExternalStream(out.value, device=device)


def gemm_workspace_bytes(dtype, ta, tb, M, N, K) -> int:
    """Bytes of split-K workspace mv_gemm(splitk=0) would like for this product (0: it never splits); see include/medvill.h."""
    dt = dtype if isinstance(dtype, int) else {torch.float32: MV_F32, torch.bfloat16: MV_BF16, torch.float16: MV_F16}[dtype]
    return int(_lib().mv_gemm_workspace_bytes(dt, int(ta), int(tb), int(M), int(N), int(K)))


def workspace_bytes(hidden, intermediate, vocab, img_hidden, max_rows, max_label_rows, max_regions) -> int:
    """The largest split-K workspace any GEMM of a pretraining step asks for at this geometry (mv_workspace_bytes)."""
    return int(_lib().mv_workspace_bytes(int(hidden), int(intermediate), int(vocab), int(img_hidden), int(max_rows), int(max_label_rows), int(max_regions)))


def set_gemm_variant(force: int = 0, nj: int = 0):
    L.set_knob("gemm_force", int(force) & 0xff)
    L.set_knob("gemm_nj", int(nj))
    L.set_knob("gemm_dbg", int(force) >> 8)
