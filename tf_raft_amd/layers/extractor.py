"""Device mirror of reference ``tf_raft/layers/extractor.py`` (BasicEncoder / SmallEncoder).

Inference (``training=False``, the forward-prediction path) runs on the hand-written halo-tiled
fp32-MFMA convolutions of ``csrc/conv_halo.h`` through ``raft_encoder_f32`` (``csrc/encoder.hip``):
batch norm folded into the weights, instance norm fused into the consumer convolution.
``training=True`` (batch statistics, dropout -- training plumbing, not on the prediction path) keeps
the PyTorch-ROCm implementation below, which BASELINE.json's north star allows for the encoders.
TensorFlow semantics that differ from PyTorch defaults are written out explicitly:
  * Keras 'same' padding with stride 2 is asymmetric (extra pixel after)   -- SURVEY F8
  * the 1x1 downsample conv is 'valid' with stride s (extractor.py:37)
  * InstanceNormalization / BatchNormalization use epsilon 1e-3, biased variance -- SURVEY F9
"""
from __future__ import annotations

from typing import Dict, Optional

import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F

from .. import _dev, _ffi, packing
from .. import weights as weights_mod
from .._ffi import check


def _same_pad(in_size: int, k: int, stride: int):
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    return total // 2, total - total // 2


class _Encoder:
    variant = 'basic'

    def __init__(self, output_dim=128, norm_type='batch', drop_rate=0.0,
                 weights: Optional[Dict[str, np.ndarray]] = None, prefix='fnet', seed=0):
        if norm_type not in ('batch', 'instance', None):
            if norm_type == 'group':
                raise NotImplementedError("norm_type 'group' is unused by RAFT/SmallRAFT and not implemented")
            raise ValueError(f'Invalid norm_type specified: {norm_type}')       # extractor.py:16
        self.output_dim = output_dim
        self.norm_type = norm_type
        self.drop_rate = drop_rate
        self.prefix = prefix
        if weights is None:
            entries = weights_mod.encoder_entries(prefix, self.variant, norm_type, output_dim)
            full = {}
            rng_w = weights_mod.init_weights  # noqa: F841  (documented default: Keras initialisers)
            rng = np.random.default_rng(seed)
            for name, kind, info in entries:
                if kind == 'conv':
                    full[f'{name}/kernel'] = weights_mod._glorot_uniform(rng, info)
                    full[f'{name}/bias'] = np.zeros((info[3],), np.float32)
                else:
                    full[f'{name}/gamma'] = np.ones((info,), np.float32)
                    full[f'{name}/beta'] = np.zeros((info,), np.float32)
                    if kind == 'bn':
                        full[f'{name}/moving_mean'] = np.zeros((info,), np.float32)
                        full[f'{name}/moving_variance'] = np.ones((info,), np.float32)
            weights = full
        self.set_weights(weights)

    def set_weights(self, weights: Dict[str, np.ndarray]) -> None:
        dev = _dev.require_gpu()
        p = self.prefix + '/'
        self.t = {}
        for k, v in weights.items():
            if not k.startswith(p):
                continue
            a = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
            if k.endswith('/kernel'):       # HWIO -> OIHW, channels_last
                a = a.permute(3, 2, 0, 1).contiguous().to(dev).contiguous(memory_format=torch.channels_last)
            else:
                a = a.to(dev)
            self.t[k[len(p):]] = a
        self._pack_device(weights, dev)

    def _pack_device(self, weights, dev):
        """Pack the weights for ``raft_encoder_f32`` (one device blob + the C struct)."""
        convs, norms, dims = packing.pack_encoder(weights, self.prefix, self.norm_type)
        chunks, offs, pos = [], {}, 0

        def add(key, arr):
            nonlocal pos
            flat = np.ascontiguousarray(arr, dtype=np.float32).ravel()
            offs[key] = pos
            chunks.append(flat)
            pad = (-flat.size) % 4
            if pad:
                chunks.append(np.zeros(pad, dtype=np.float32))
            pos += flat.size + pad

        for field, wp, b, _ in convs:
            add((field, 'w'), wp)
            add((field, 'b'), b)
        for idx, g, b in norms:
            add((idx, 'g'), g)
            add((idx, 'be'), b)
        self._blob = torch.from_numpy(np.concatenate(chunks)).to(dev)
        base = self._blob.data_ptr()
        c = _ffi.EncoderWeights()
        c.c0, c.c1, c.c2, c.c3, c.cout = dims
        c.norm = {'instance': _ffi.NORM_INSTANCE, 'batch': _ffi.NORM_FOLDED, None: _ffi.NORM_NONE}[self.norm_type]
        for field, _, _, npad in convs:
            cw = _ffi.ConvWeights(wp=base + 4 * offs[(field, 'w')], bias=base + 4 * offs[(field, 'b')], npad=npad)
            if field == 'conv1':
                c.conv1 = cw
            elif field == 'conv2':
                c.conv2 = cw
            elif field[0] == 'block_w':
                c.block_w[field[1]][field[2]] = cw
            elif field[0] == 'block_w44':
                c.block_w44[field[1]][field[2]] = cw
            else:
                c.block[field[1]][field[2]] = cw
        for idx, _, _ in norms:
            c.in_gamma[idx] = base + 4 * offs[(idx, 'g')]
            c.in_beta[idx] = base + 4 * offs[(idx, 'be')]
        self.c = c
        self._ws = None

    def forward_device(self, images: torch.Tensor, input_affine: bool = False, images_b: torch.Tensor = None) -> torch.Tensor:
        """(n, H, W, 3) device tensor -> (n, ceil(H/8), ceil(W/8), output_dim) through ``raft_encoder_f32``.
        ``input_affine`` applies the model's ``2 * (image / 255) - 1`` while staging (model.py:70-71).  With ``images_b`` (same
        shape) the batch is [images | images_b] (extractor.py:114-116), staged from the two tensors without a concatenation."""
        n, H, W, ch = images.shape
        if ch != 3:
            raise ValueError(f'images must be (n, H, W, 3), got {tuple(images.shape)}')
        if images_b is not None:
            if images_b.shape != images.shape:
                raise ValueError(f'image batches differ: {tuple(images.shape)} / {tuple(images_b.shape)}')
            n = 2 * n
        lib = _dev.lib()
        need = lib.raft_encoder_workspace_floats(C.byref(self.c), n, H, W)
        if need <= 0:
            raise ValueError('invalid encoder geometry')
        if self._ws is None or self._ws.numel() < need or self._ws.device != images.device:
            self._ws = torch.empty((need,), device=images.device, dtype=torch.float32)
        ho, wo = H, W
        for _ in range(3):
            ho, wo = (ho + 1) // 2, (wo + 1) // 2
        out = torch.empty((n, ho, wo, self.output_dim), device=images.device, dtype=torch.float32)
        if images_b is not None:
            check(lib.raft_encoder_pair_f32(C.byref(self.c), _dev.ptr(images), _dev.ptr(images_b), n // 2, H, W, 1 if input_affine else 0,
                                            _dev.ptr(out), _dev.ptr(self._ws), _dev.stream_ptr()), 'encoder_pair')
        else:
            check(lib.raft_encoder_f32(C.byref(self.c), _dev.ptr(images), n, H, W, 1 if input_affine else 0,
                                       _dev.ptr(out), _dev.ptr(self._ws), _dev.stream_ptr()), 'encoder')
        return out

    # ---- building blocks ----------------------------------------------------------------------
    def _conv(self, name, x, stride=1, padding='same'):
        w, b = self.t[f'{name}/kernel'], self.t[f'{name}/bias']
        kh, kw = w.shape[2], w.shape[3]
        if padding == 'same':
            pt, pb = _same_pad(x.shape[2], kh, stride)
            pl, pr = _same_pad(x.shape[3], kw, stride)
            if pt == pb and pl == pr:
                return F.conv2d(x, w, b, stride=stride, padding=(pt, pl))
            x = F.pad(x, (pl, pr, pt, pb))
        return F.conv2d(x, w, b, stride=stride)

    def _norm(self, name, x, training):
        if f'{name}/moving_mean' in self.t:
            g, b = self.t[f'{name}/gamma'], self.t[f'{name}/beta']
            if training:
                return F.batch_norm(x, None, None, g, b, True, 0.0, 1e-3)
            return F.batch_norm(x, self.t[f'{name}/moving_mean'], self.t[f'{name}/moving_variance'], g, b,
                                False, 0.0, 1e-3)
        if f'{name}/gamma' in self.t:
            return F.instance_norm(x, weight=self.t[f'{name}/gamma'], bias=self.t[f'{name}/beta'], eps=1e-3)
        return x

    def _res_block(self, name, x, strides, training):
        """reference extractor.py:41-49."""
        fx = F.relu(self._norm(f'{name}/norm1', self._conv(f'{name}/conv1', x, strides), training))
        fx = F.relu(self._norm(f'{name}/norm2', self._conv(f'{name}/conv2', fx), training))
        if strides != 1:
            x = self._conv(f'{name}/downsample/0', x, strides, padding='valid')
            x = self._norm(f'{name}/downsample/1', x, training)
        return F.relu(x + fx)

    def __call__(self, inputs, training=False, _raw_images=False):
        """reference extractor.py:113-130 / 158-175.  NHWC in, NHWC out; a list input is
        concatenated along the batch and split again."""
        is_list = isinstance(inputs, (tuple, list))
        if is_list and not training and len(inputs) == 2:
            a, b = (_dev.to_device(i) for i in inputs)
            y = self.forward_device(a, input_affine=bool(_raw_images), images_b=b)      # no concatenation: staged from both tensors
            half = y.shape[0] // 2
            return [_dev.wrap(y[:half]), _dev.wrap(y[half:])]
        if is_list:
            x = torch.cat([_dev.to_device(i) for i in inputs], dim=0)
        else:
            x = _dev.to_device(inputs)
        if not training:
            y = self.forward_device(x, input_affine=bool(_raw_images))
            if is_list:
                half = y.shape[0] // 2
                return [_dev.wrap(y[:half]), _dev.wrap(y[half:])]
            return _dev.wrap(y)
        if _raw_images:
            x = 2 * (x / 255.0) - 1.0
        x = x.permute(0, 3, 1, 2)            # NHWC storage viewed as NCHW == channels_last
        x = F.relu(self._norm('norm1', self._conv('conv1', x, 2), training))
        for li, s in ((1, 1), (2, 2), (3, 2)):
            x = self._res_block(f'layer{li}/0', x, s, training)
            x = self._res_block(f'layer{li}/1', x, 1, training)
        x = self._conv('conv2', x, padding='valid')
        if self.drop_rate > 0 and training:
            x = F.dropout(x, self.drop_rate, True)
        x = x.permute(0, 2, 3, 1).contiguous()
        if is_list:
            half = x.shape[0] // 2
            return [_dev.wrap(x[:half]), _dev.wrap(x[half:])]
        return _dev.wrap(x)


class BasicEncoder(_Encoder):
    """reference extractor.py:88-130 (channels 64/64/96/128)."""
    variant = 'basic'


class SmallEncoder(_Encoder):
    """reference extractor.py:133-175 (channels 32/32/64/96, ResBlock-based)."""
    variant = 'small'
