"""``tensorflow_addons`` 0.11.1 stand-in: InstanceNormalization / GroupNormalization layers and ``image.resampler`` (test
infrastructure only; see oracle/tfstub/README.md)."""
from __future__ import annotations

import types as _types

import torch as _torch

from oracle import tf_ops as _ops
from tensorflow.keras.layers import Layer as _Layer


class InstanceNormalization(_Layer):
    """tfa InstanceNormalization = GroupNormalization(groups = channels): epsilon 1e-3, per-sample per-channel moments over
    H x W; ``training`` does not change it (reference extractor.py:11-12)."""

    def call(self, inputs):
        if 'gamma' not in self._vars:
            c = inputs.shape[-1]
            self._vars.update(gamma=_torch.ones(c, dtype=inputs.dtype), beta=_torch.zeros(c, dtype=inputs.dtype))
        return _ops.instance_norm(inputs, self._vars['gamma'], self._vars['beta'])


class GroupNormalization(_Layer):
    """Constructible (ResBlock's default ``norm_type='group'``, extractor.py:7-8) but out of scope: no model of the reference
    instantiates it (RAFT uses instance / batch, SmallRAFT instance / none)."""

    def __init__(self, groups=2, **kwargs):
        super().__init__(**kwargs)
        self.groups = groups

    def call(self, inputs):
        raise NotImplementedError('GroupNormalization is outside the forward-prediction path (SURVEY section 2)')


layers = _types.SimpleNamespace(InstanceNormalization=InstanceNormalization, GroupNormalization=GroupNormalization)


def _resampler(data, warp):
    """``tfa.image.resampler``: bilinear interpolation of ``data`` (N,H,W,C) at ``warp[..., (x, y)]`` with floor / floor+1
    corners and zero for corners outside the image.  Written independently of the reference's ``bilinear_sampler`` (it is the
    comparator of reference tests/layers/test_corr.py:15-27)."""
    n, h, w, c = data.shape
    x, y = warp[..., 0], warp[..., 1]
    x0, y0 = _torch.floor(x), _torch.floor(y)
    out = _torch.zeros(warp.shape[:-1] + (c,), dtype=data.dtype)
    b = _torch.arange(n).view((n,) + (1,) * (x.dim() - 1)).expand(x.shape)
    for dy in (0, 1):
        for dx in (0, 1):
            xi, yi = x0 + dx, y0 + dy
            wgt = (1 - (x - xi).abs()) * (1 - (y - yi).abs())
            ok = (xi >= 0) & (xi <= w - 1) & (yi >= 0) & (yi <= h - 1)
            v = data[b, yi.clamp(0, h - 1).long(), xi.clamp(0, w - 1).long()]
            out = out + (wgt * ok.to(data.dtype)).unsqueeze(-1) * v
    return out


image = _types.SimpleNamespace(resampler=_resampler)
