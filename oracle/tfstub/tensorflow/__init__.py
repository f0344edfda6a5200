"""Minimal eager stand-in for the ``tensorflow`` 2.3 symbols the reference's tf_raft package touches.

TEST INFRASTRUCTURE ONLY (see oracle/tfstub/README.md).  Tensors are torch CPU tensors (subclass ``Tensor`` so that
``ndarray - tensor`` behaves as in TF); every op with non-trivial TensorFlow semantics delegates to ``oracle.tf_ops``, which
states those semantics and cites the reference call sites.  Symbols are listed in the order the reference uses them:
model.py, layers/corr.py, layers/update.py, layers/extractor.py, losses/losses.py.
"""
from __future__ import annotations

import builtins as _bi
import types as _types

import numpy as _np
import torch as _torch

from oracle import tf_ops as _ops

__version__ = '2.3.0-stub'


class Tensor(_torch.Tensor):
    """torch tensor with TF's reflected-operator behaviour for NumPy operands (``ndarray - Tensor`` converts the array to the
    tensor's dtype, reference tests/losses/test_losses.py:41) and ``.numpy()``."""

    def _conv(self, other):
        return convert_to_tensor(other, dtype=self.dtype if self.dtype.is_floating_point else None)

    def __rsub__(self, other):
        return self._conv(other) - self

    def __radd__(self, other):
        return self._conv(other) + self

    def __rmul__(self, other):
        return self._conv(other) * self

    def __rtruediv__(self, other):
        return self._conv(other) / self

    def __rand__(self, other):
        return self._conv(other) & self


# dtypes (module attributes so that oracle.reference_runner.floatx() can rebind float32 for an fp64 run of the same source)
float32 = _torch.float32
float64 = _torch.float64
int32 = _torch.int32
int64 = _torch.int64
bool = _torch.bool      # noqa: A001  (tf.bool)


def _t(x, dtype=None):
    if isinstance(x, _torch.Tensor):
        y = x if dtype is None else x.to(dtype)
    else:
        a = _np.asarray(x)
        if dtype is None and a.dtype == _np.float64 and not isinstance(x, _np.ndarray):
            dtype = float32                     # python floats / lists of floats are float32 in TF
        y = _torch.as_tensor(a) if dtype is None else _torch.as_tensor(a).to(dtype)
    return y if isinstance(y, Tensor) else y.as_subclass(Tensor)


def convert_to_tensor(value, dtype=None):
    return _t(value, dtype)


def cast(x, dtype):
    return _t(x, dtype)


# ---------------------------------------------------------------- shape ops

def reshape(tensor, shape):
    return _t(tensor).reshape(tuple(int(s) for s in shape))


def concat(values, axis):
    return _torch.cat([_t(v) for v in values], dim=axis)


def stack(values, axis=0):
    return _torch.stack([_t(v) for v in values], dim=axis)


def unstack(value, axis=0):
    return list(_torch.unbind(_t(value), dim=axis))


def split(value, num_or_size_splits, axis=0):
    """tf.split: an int is the NUMBER of equal pieces (extractor.py:128), a list the piece sizes (model.py:84)."""
    value = _t(value)
    if isinstance(num_or_size_splits, int):
        n = value.shape[axis]
        assert n % num_or_size_splits == 0, (n, num_or_size_splits)
        return list(_torch.split(value, n // num_or_size_splits, dim=axis))
    return list(_torch.split(value, [int(s) for s in num_or_size_splits], dim=axis))


def tile(input, multiples):   # noqa: A002
    return _t(input).repeat(*[int(m) for m in multiples])


def expand_dims(input, axis):   # noqa: A002
    return _t(input).unsqueeze(axis)


def range(start, limit=None, delta=1, dtype=None):   # noqa: A001
    if limit is None:
        start, limit = 0, start
    return _torch.arange(start, limit, delta, dtype=dtype).as_subclass(Tensor)


def meshgrid(*args, indexing='xy'):
    return list(_torch.meshgrid(*[_t(a) for a in args], indexing=indexing))


def gather_nd(params, indices, batch_dims=0):
    """tf.gather_nd: ``indices[..., :K]`` index the K axes of ``params`` after the ``batch_dims`` leading ones; the index
    tensor's leading ``batch_dims`` axes are matched element for element with those of ``params`` (reference corr.py:63-66:
    params (N,h,w,1), indices (N,k,k,2) int32, batch_dims=1 -> (N,k,k,1))."""
    params, indices = _t(params), _t(indices).long()
    k = indices.shape[-1]
    idx = [indices[..., i] for i in _bi.range(k)]
    if batch_dims == 0:
        return params[tuple(idx)]
    assert batch_dims == 1, batch_dims
    n = params.shape[0]
    b = _torch.arange(n).view((n,) + (1,) * (indices.dim() - 2)).expand(indices.shape[:-1])
    return params[(b,) + tuple(idx)]


# ---------------------------------------------------------------- elementwise / reductions

def floor(x):
    return _torch.floor(_t(x))


def sqrt(x):
    return _torch.sqrt(_t(x))


def abs(x):   # noqa: A001
    return _torch.abs(_t(x))


def tanh(x):
    return _torch.tanh(_t(x))


def clip_by_value(t, clip_value_min, clip_value_max):
    return _torch.clamp(_t(t), clip_value_min, clip_value_max)


def reduce_sum(input_tensor, axis=None, keepdims=False):
    x = _t(input_tensor)
    return x.sum() if axis is None else x.sum(dim=axis, keepdim=keepdims)


def reduce_mean(input_tensor, axis=None, keepdims=False):
    x = _t(input_tensor)
    return x.mean() if axis is None else x.mean(dim=axis, keepdim=keepdims)


def matmul(a, b, transpose_a=False, transpose_b=False):
    a, b = _t(a), _t(b)
    if transpose_a:
        a = a.transpose(-1, -2)
    if transpose_b:
        b = b.transpose(-1, -2)
    return _torch.matmul(a, b)


def cond(pred, true_fn, false_fn):
    return true_fn() if _bi.bool(pred) else false_fn()


def clip_by_global_norm(t_list, clip_norm):
    norm = _torch.sqrt(sum((g.double() ** 2).sum() for g in t_list)).to(float32)
    scale = clip_norm / _torch.maximum(norm, _torch.as_tensor(float(clip_norm)))
    return [g * scale for g in t_list], norm


class GradientTape:
    def __enter__(self):
        raise NotImplementedError('oracle/tfstub covers the forward path only (no autodiff)')

    def __exit__(self, *a):
        return False


# ---------------------------------------------------------------- tf.math / tf.nn / tf.image / tf.random

math = _types.SimpleNamespace(
    ceil=lambda x: _torch.ceil(_t(x)),
    floor=floor, sqrt=sqrt, abs=abs, tanh=tanh)


def _softmax(logits, axis=-1):
    return _torch.softmax(_t(logits), dim=axis)


def _avg_pool2d(input, ksize, strides, padding):   # noqa: A002
    """tf.nn.avg_pool2d: only the reference's use (corr.py:113: 2, 2, 'VALID')."""
    assert ksize == 2 and strides == 2 and padding == 'VALID', (ksize, strides, padding)
    return _ops.avg_pool_2x2_valid(_t(input))


def _depth_to_space(input, block_size):   # noqa: A002
    return _ops.depth_to_space(_t(input), int(block_size))


nn = _types.SimpleNamespace(
    relu=lambda x: _torch.relu(_t(x)),
    sigmoid=lambda x: _torch.sigmoid(_t(x)),
    tanh=tanh,
    softmax=_softmax,
    avg_pool2d=_avg_pool2d,
    depth_to_space=_depth_to_space)


def _extract_patches(images, sizes, strides, rates, padding):
    """tf.image.extract_patches for square patches, stride 1, rate 1 (model.py:55-59 'SAME'; tests/test_model.py:29-33 'VALID')."""
    assert tuple(strides) == (1, 1, 1, 1) and tuple(rates) == (1, 1, 1, 1), (strides, rates)
    _, kh, kw, _ = sizes
    assert kh == kw
    fn = _ops.extract_patches_same if padding == 'SAME' else _ops.extract_patches_valid
    return fn(_t(images), int(kh))


def _resize(images, size, method='bilinear'):
    assert method == 'bilinear', method
    return _ops.resize_bilinear(_t(images), int(size[0]), int(size[1]))


image = _types.SimpleNamespace(extract_patches=_extract_patches, resize=_resize)

_gen = _torch.Generator().manual_seed(0)


def _set_seed(seed):
    _gen.manual_seed(int(seed))


def _normal(shape, mean=0.0, stddev=1.0, dtype=None):
    return (_torch.randn(tuple(shape), generator=_gen, dtype=dtype or float32) * stddev + mean).as_subclass(Tensor)


def _uniform(shape, minval=0, maxval=1, dtype=None):
    u = _torch.rand(tuple(shape), generator=_gen, dtype=dtype or float32)
    return (u * (maxval - minval) + minval).as_subclass(Tensor)


random = _types.SimpleNamespace(set_seed=_set_seed, normal=_normal, uniform=_uniform)

from . import keras  # noqa: E402,F401
