"""``tensorflow.keras.layers`` stand-in (test infrastructure only).

Layers create their variables lazily on the first call with the Keras defaults (glorot_uniform kernels, zero biases, unit
gamma, zero beta; BatchNormalization momentum 0.99, epsilon 1e-3) unless ``oracle.reference_runner.assign_weights`` put
Keras-layout arrays in place first.  Sub-layers are tracked by ATTRIBUTE NAME (and by position inside a Sequential), which
gives every variable the path ``tf_raft_amd.weights`` uses, e.g. ``fnet/layer2/0/downsample/0/kernel``.
"""
from __future__ import annotations

import inspect

import torch

from oracle import tf_ops

_init_gen = torch.Generator().manual_seed(0)


class Layer:
    def __init__(self, name=None, **kwargs):
        object.__setattr__(self, '_children', {})
        object.__setattr__(self, '_vars', {})
        self.name = name

    def __setattr__(self, key, value):
        if '_children' not in self.__dict__:          # subclass assigned before super().__init__()
            object.__setattr__(self, '_children', {})
            object.__setattr__(self, '_vars', {})
        if isinstance(value, Layer):
            self._children[key] = value
        elif key in self._children:
            del self._children[key]
        object.__setattr__(self, key, value)

    # -- Keras call protocol: ``training`` is forwarded only to a ``call`` that takes it
    def __call__(self, inputs, *args, **kwargs):
        if 'training' in kwargs and not _accepts(self.call, 'training', len(args)):
            kwargs.pop('training')
        return self.call(inputs, *args, **kwargs)

    def call(self, inputs):
        raise NotImplementedError

    # -- variable bookkeeping
    def named_layers(self, prefix=''):
        yield prefix, self
        for k, c in self._children.items():
            yield from c.named_layers(f'{prefix}/{k}' if prefix else k)

    def named_weights(self):
        for p, layer in self.named_layers():
            for k, v in layer._vars.items():
                yield (f'{p}/{k}' if p else k), v


def _accepts(fn, name, n_positional):
    try:
        sig = inspect.signature(fn)
    except (TypeError, ValueError):
        return False
    params = list(sig.parameters.values())
    if any(p.kind is p.VAR_KEYWORD for p in params):
        return True
    return any(p.name == name for p in params)


class Sequential(Layer):
    def __init__(self, layers=None, **kwargs):
        super().__init__(**kwargs)
        self.layers = list(layers or [])
        for i, l in enumerate(self.layers):
            self._children[str(i)] = l

    def call(self, inputs, training=None):
        x = inputs
        for l in self.layers:
            x = l(x, training=training)
        return x


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


class Conv2D(Layer):
    """``Conv2D(filters, kernel_size, strides=1, padding='valid')``: NHWC input, HWIO kernel, bias added."""

    def __init__(self, filters, kernel_size, strides=1, padding='valid', **kwargs):
        super().__init__(**kwargs)
        self.filters = int(filters)
        self.kernel_size = _pair(kernel_size)
        self.strides = _pair(strides)
        assert self.strides[0] == self.strides[1]
        self.padding = padding.lower()

    def call(self, inputs):
        if 'kernel' not in self._vars:
            kh, kw = self.kernel_size
            cin = inputs.shape[-1]
            lim = tf_ops.glorot_limit(kh, kw, cin, self.filters)
            k = (torch.rand((kh, kw, cin, self.filters), generator=_init_gen) * 2 - 1) * lim
            self._vars['kernel'] = k.to(inputs.dtype)
            self._vars['bias'] = torch.zeros(self.filters, dtype=inputs.dtype)
        return tf_ops.conv2d(inputs, self._vars['kernel'], self._vars['bias'], self.strides[0], self.padding)


class ReLU(Layer):
    def call(self, inputs):
        return torch.relu(inputs)


class Lambda(Layer):
    def __init__(self, function, **kwargs):
        super().__init__(**kwargs)
        self.function = function

    def call(self, inputs):
        return self.function(inputs)


class Dropout(Layer):
    """Inference: identity.  Training: inverted dropout from a seeded generator (training plumbing, not on the forward-prediction path)."""

    def __init__(self, rate, **kwargs):
        super().__init__(**kwargs)
        self.rate = float(rate)

    def call(self, inputs, training=None):
        if not training or self.rate == 0:
            return inputs
        keep = (torch.rand(inputs.shape, generator=_init_gen) >= self.rate).to(inputs.dtype)
        return inputs * keep / (1.0 - self.rate)


class BatchNormalization(Layer):
    """Keras defaults: axis -1, momentum 0.99, epsilon 1e-3.  training=True normalises with the batch moments and moves the
    moving statistics (training plumbing); inference uses the moving statistics (reference extractor.py:9-10)."""

    momentum = 0.99

    def _build(self, x):
        c = x.shape[-1]
        self._vars.update(gamma=torch.ones(c, dtype=x.dtype), beta=torch.zeros(c, dtype=x.dtype),
                          moving_mean=torch.zeros(c, dtype=x.dtype), moving_variance=torch.ones(c, dtype=x.dtype))
        for k in ('moving_mean', 'moving_variance'):
            self._vars[k]._trainable = False

    def call(self, inputs, training=None):
        if 'gamma' not in self._vars:
            self._build(inputs)
        v = self._vars
        y = tf_ops.batch_norm(inputs, v['gamma'], v['beta'], v['moving_mean'], v['moving_variance'], bool(training))
        if training:
            m = inputs.mean(dim=(0, 1, 2))
            var = ((inputs - m) ** 2).mean(dim=(0, 1, 2))
            v['moving_mean'] = v['moving_mean'] * self.momentum + m * (1 - self.momentum)
            v['moving_variance'] = v['moving_variance'] * self.momentum + var * (1 - self.momentum)
        return y
