"""Oracle restatement of reference tf_raft/layers/corr.py (test infrastructure only)."""
from __future__ import annotations

import math

import torch

from . import tf_ops


def bilinear_sampler(image, coords):
    """reference corr.py:28-69, statement for statement.

    image: (N, h, w, 1); coords: (N, kh, kw, 2) xy order.  Returns (N, kh, kw, 1).
    Weights are ``ceil(g) - g`` and ``g - floor(g)`` after clamping, so a sample whose
    (clamped) x or y is an exact integer evaluates to 0 -- including every out-of-range
    sample.
    """
    n, h, w, _ = image.shape
    gx, gy = coords[..., 0], coords[..., 1]                       # corr.py:40
    gx = torch.clamp(gx, 0, w - 1)                                # corr.py:41
    gy = torch.clamp(gy, 0, h - 1)                                # corr.py:42
    gx0, gx1 = torch.floor(gx), torch.ceil(gx)                    # corr.py:45-46
    gy0, gy1 = torch.floor(gy), torch.ceil(gy)                    # corr.py:47-48
    c00 = (gy1 - gy) * (gx1 - gx)                                 # corr.py:57-60
    c01 = (gy1 - gy) * (gx - gx0)
    c10 = (gy - gy0) * (gx1 - gx)
    c11 = (gy - gy0) * (gx - gx0)
    img = image[..., 0]
    bidx = torch.arange(n).view(n, 1, 1).expand_as(gx)

    def gather(yy, xx):                                           # corr.py:63-66 (gather_nd, batch_dims=1)
        return img[bidx, yy.long(), xx.long()]

    out = (c00 * gather(gy0, gx0) + c01 * gather(gy0, gx1)
           + c10 * gather(gy1, gx0) + c11 * gather(gy1, gx1))     # corr.py:68
    return out.unsqueeze(-1)


def coords_grid(batch_size, height, width, dtype=torch.float32):
    """reference corr.py:72-90: (bs, h, w, 2) with [..., 0] = x, [..., 1] = y."""
    gy, gx = torch.meshgrid(torch.arange(height, dtype=dtype),
                            torch.arange(width, dtype=dtype), indexing='ij')
    coords = torch.stack([gx, gy], dim=-1)
    return coords.unsqueeze(0).repeat(batch_size, 1, 1, 1)


def upflow8(flow):
    """reference corr.py:93-96: ``8 * tf.image.resize(flow, (8h, 8w), 'bilinear')``."""
    _, h, w, _ = flow.shape
    return 8 * tf_ops.resize_bilinear(flow, 8 * h, 8 * w)


class CorrBlock:
    """reference corr.py:99-162."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.fmap1 = fmap1
        self.fmap2 = fmap2
        self.num_levels = num_levels
        self.radius = radius

        corr = self.correlation(fmap1, fmap2)                     # corr.py:106
        bs, h1, w1, _, h2, w2 = corr.shape
        corr = corr.reshape(bs * h1 * w1, h2, w2, 1)              # corr.py:108
        self.corr_pyramid = [corr]
        for _ in range(num_levels - 1):                           # corr.py:112-114
            corr = tf_ops.avg_pool_2x2_valid(corr)
            self.corr_pyramid.append(corr)

    def retrieve(self, coords):
        """reference corr.py:116-152.  NB the window axis quirk: delta is
        stack([dy, dx]) but is added to (x, y) coords, so window axis 0 offsets x."""
        r = self.radius
        bs, h, w, _ = coords.shape
        out_pyramid = []
        for i in range(self.num_levels):
            corr = self.corr_pyramid[i]
            d = torch.arange(-r, r + 1, dtype=coords.dtype)
            dy, dx = torch.meshgrid(d, d, indexing='ij')          # corr.py:134
            delta = torch.stack([dy, dx], dim=-1)                 # corr.py:136
            delta_lvl = delta.reshape(1, 2 * r + 1, 2 * r + 1, 2)
            centroid_lvl = coords.reshape(bs * h * w, 1, 1, 2) / 2 ** i   # corr.py:141
            coords_lvl = centroid_lvl + delta_lvl                 # corr.py:143
            s = bilinear_sampler(corr, coords_lvl)                # corr.py:146
            out_pyramid.append(s.reshape(bs, h, w, -1))           # corr.py:148
        return torch.cat(out_pyramid, dim=-1)                     # corr.py:151

    @staticmethod
    def correlation(fmap1, fmap2):
        """reference corr.py:154-162: matmul then divide by sqrt(C)."""
        bs, h, w, nch = fmap1.shape
        f1 = fmap1.reshape(bs, h * w, nch)
        f2 = fmap2.reshape(bs, h * w, nch)
        corr = torch.matmul(f1, f2.transpose(1, 2))
        corr = corr.reshape(bs, h, w, 1, h, w)
        return corr / math.sqrt(float(nch))
