"""``tensorflow.keras`` stand-in: Model / Sequential / layers / metrics / callbacks (test infrastructure only)."""
from __future__ import annotations

import torch as _torch

from . import layers  # noqa: F401
from .layers import Layer, Sequential  # noqa: F401


class Model(Layer):
    """``tf.keras.Model``: a Layer whose ``compile`` accepts and ignores Keras' keyword arguments (reference model.py:111-116)."""

    def compile(self, **kwargs):   # noqa: A003
        self._compiled = True

    @property
    def trainable_weights(self):
        return [v for _, v in self.named_weights() if getattr(v, '_trainable', True)]


class _Mean:
    def __init__(self, name=None):
        self.name = name
        self.reset_states()

    def update_state(self, value):
        self.total += float(value)
        self.count += 1

    def result(self):
        return _torch.as_tensor(self.total / max(self.count, 1), dtype=_torch.float32)

    def reset_states(self):
        self.total, self.count = 0.0, 0


class _Variable:
    def __init__(self):
        self.value = _torch.zeros((), dtype=_torch.float32)
        self.dtype = self.value.dtype

    def assign_add(self, v):
        self.value = self.value + _torch.as_tensor(v, dtype=_torch.float32)

    def __truediv__(self, other):
        return self.value / other


class _Metric:
    """``tf.keras.metrics.Metric`` as far as reference losses.py:46-86 uses it."""

    def __init__(self, name=None, **kwargs):
        self.name = name

    def add_weight(self, name, initializer='zeros'):
        assert initializer == 'zeros'
        return _Variable()


import types as _types  # noqa: E402

metrics = _types.SimpleNamespace(Mean=_Mean, Metric=_Metric)


class _Callback:
    def __init__(self, **kwargs):
        self.model = None


callbacks = _types.SimpleNamespace(Callback=_Callback)
