"""TensorFlow-semantics primitives used by the oracle (test infrastructure only).

All tensors are NHWC ``torch`` CPU tensors (like the reference's TF tensors); dtype is
whatever the caller passes (fp32 by default, fp64 to bound the oracle's own rounding).
Each function cites the TF op whose behaviour it restates and the reference call site.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def same_padding(in_size: int, k: int, stride: int):
    """TF 'SAME' padding: out = ceil(in/stride); total = max((out-1)*stride + k - in, 0);
    before = total // 2, after = total - before (the extra pixel goes AFTER, which makes
    stride-2 convs asymmetric -- SURVEY F8)."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    return total // 2, total - total // 2


def conv2d(x, kernel, bias=None, stride=1, padding='same'):
    """Keras ``Conv2D(filters, k, strides, padding)`` on NHWC input with an HWIO kernel.

    reference call sites: update.py:10-11, 22-24, 43-49, 73-76, 91-95, 138-140;
    extractor.py:26-27, 37, 95, 102.  ``padding`` is 'same' or 'valid' (Keras default).
    """
    kh, kw, cin, cout = kernel.shape
    assert x.shape[-1] == cin, (x.shape, kernel.shape)
    xc = x.permute(0, 3, 1, 2)
    if padding == 'same':
        pt, pb = same_padding(x.shape[1], kh, stride)
        pl, pr = same_padding(x.shape[2], kw, stride)
        if pt or pb or pl or pr:
            xc = F.pad(xc, (pl, pr, pt, pb))
    elif padding != 'valid':
        raise ValueError(padding)
    w = kernel.permute(3, 2, 0, 1)
    y = F.conv2d(xc, w, bias, stride=stride)
    return y.permute(0, 2, 3, 1)


def instance_norm(x, gamma, beta, eps=1e-3):
    """``tfa.layers.InstanceNormalization`` (GroupNormalization with groups == channels,
    tfa 0.11.1): per-sample per-channel moments over H x W, biased variance, epsilon 1e-3,
    ``(x - mean) * rsqrt(var + eps) * gamma + beta``.  reference extractor.py:11-12."""
    mean = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


def batch_norm(x, gamma, beta, moving_mean, moving_var, training=False, eps=1e-3):
    """Keras ``BatchNormalization`` (epsilon 1e-3).  Inference uses the moving statistics;
    ``training=True`` normalises with the batch moments (biased variance) as Keras does
    (moving-stat updates are training plumbing and out of scope).  reference extractor.py:9-10."""
    if training:
        mean = x.mean(dim=(0, 1, 2), keepdim=True)
        var = ((x - mean) ** 2).mean(dim=(0, 1, 2), keepdim=True)
    else:
        mean, var = moving_mean, moving_var
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


def avg_pool_2x2_valid(x):
    """``tf.nn.avg_pool2d(x, 2, 2, 'VALID')`` on NHWC: floor on odd sizes.  reference corr.py:113."""
    n, h, w, c = x.shape
    h2, w2 = h // 2, w // 2
    x = x[:, :h2 * 2, :w2 * 2, :].reshape(n, h2, 2, w2, 2, c)
    return x.mean(dim=(2, 4))


def extract_patches_same(x, k: int):
    """``tf.image.extract_patches(x, sizes=(1,k,k,1), strides 1, rates 1, padding='SAME')``:
    zero padding, patch depth ordered (ky, kx, channel).  reference model.py:55-59."""
    n, h, w, c = x.shape
    pt, pb = same_padding(h, k, 1)
    pl, pr = same_padding(w, k, 1)
    xp = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb)).permute(0, 2, 3, 1)
    cols = []
    for ky in range(k):
        for kx in range(k):
            cols.append(xp[:, ky:ky + h, kx:kx + w, :])
    return torch.cat(cols, dim=-1)


def extract_patches_valid(x, k: int):
    """Same op with padding='VALID' (used only to pin the ordering against the reference's
    known-answer arrays, tests/test_model.py:16-35)."""
    n, h, w, c = x.shape
    ho, wo = h - k + 1, w - k + 1
    cols = []
    for ky in range(k):
        for kx in range(k):
            cols.append(x[:, ky:ky + ho, kx:kx + wo, :])
    return torch.cat(cols, dim=-1)


def depth_to_space(x, block: int):
    """``tf.nn.depth_to_space`` (NHWC, DCR order): out[b, y*bs+i, x*bs+j, c] =
    in[b, y, x, (i*bs + j)*C + c].  reference model.py:66."""
    n, h, w, d = x.shape
    c = d // (block * block)
    x = x.reshape(n, h, w, block, block, c).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(n, h * block, w * block, c)


def resize_bilinear(x, new_h: int, new_w: int):
    """``tf.image.resize(x, size, 'bilinear')`` in TF2: half-pixel centres, no antialias.
    Per axis: src = (dst + 0.5) * (in/out) - 0.5; lo = max(floor(src), 0);
    hi = min(ceil(src), in-1); lerp = src - floor(src).  reference corr.py:93-96."""
    n, h, w, c = x.shape

    def axis_weights(in_size, out_size):
        scale = in_size / out_size
        dst = torch.arange(out_size, dtype=torch.float64)
        src = (dst + 0.5) * scale - 0.5
        fl = torch.floor(src)
        lo = torch.clamp(fl, min=0).long()
        hi = torch.clamp(torch.ceil(src), max=in_size - 1).long()
        lerp = (src - fl).to(x.dtype)
        return lo, hi, lerp

    ylo, yhi, yl = axis_weights(h, new_h)
    xlo, xhi, xl = axis_weights(w, new_w)
    top = x[:, ylo]
    bot = x[:, yhi]

    def lerp_x(rows):
        left = rows[:, :, xlo]
        right = rows[:, :, xhi]
        return left + (right - left) * xl.view(1, 1, -1, 1)

    t = lerp_x(top)
    b = lerp_x(bot)
    return t + (b - t) * yl.view(1, -1, 1, 1)


def glorot_limit(kh, kw, cin, cout):
    return math.sqrt(6.0 / (kh * kw * cin + kh * kw * cout))
