"""Mirror of reference ``tf_raft/training.py`` (learning-rate scale functions) and of the optimizer objects the reference's
training scripts build (train_sintel.py:83-93): ``tfa.optimizers.CyclicalLearningRate`` and ``tfa.optimizers.AdamW``
(tensorflow-addons 0.11.1 / Keras Adam of TF 2.3 semantics), with the update itself a HIP kernel (``raft_adamw_step_f32``)
and ``tf.clip_by_global_norm`` reduced on the device (``raft_sumsq_f32``).  ``VisFlowCallback`` (Keras callback plumbing) is
out of scope.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _dev
from ._ffi import check


def first_cycle_scaler(cycle):
    """reference training.py:10-15: 1 in the first cycle, 0 afterwards."""
    return 1.0 if cycle == 1 else 0.0


def inverse_scaler(cycle):
    """reference training.py:18-23."""
    return 1.0 / cycle


class CyclicalLearningRate:
    """``tfa.optimizers.CyclicalLearningRate`` (triangular): train_sintel.py:83-89."""

    def __init__(self, initial_learning_rate, maximal_learning_rate, step_size, scale_fn=lambda cycle: 1.0, scale_mode='cycle'):
        self.initial_learning_rate = float(initial_learning_rate)
        self.maximal_learning_rate = float(maximal_learning_rate)
        self.step_size = float(step_size)
        self.scale_fn = scale_fn
        if scale_mode not in ('cycle', 'iterations'):
            raise ValueError(f'scale_mode must be "cycle" or "iterations", got {scale_mode!r}')
        self.scale_mode = scale_mode

    def __call__(self, step):
        cycle = math.floor(1 + step / (2 * self.step_size))
        x = abs(step / self.step_size - 2 * cycle + 1)
        mode_step = cycle if self.scale_mode == 'cycle' else step
        return self.initial_learning_rate + (self.maximal_learning_rate - self.initial_learning_rate) * max(0.0, 1 - x) \
            * self.scale_fn(mode_step)


class AdamW:
    """``tfa.optimizers.AdamW(weight_decay, learning_rate)`` on device tensors: decoupled weight decay ``var -= wd * var``
    (tfa 0.11.1: not scaled by the learning rate), then Keras Adam with ``beta_1 = 0.9, beta_2 = 0.999, epsilon = 1e-7``.
    ``learning_rate`` is a float or a schedule called with ``iterations`` (0 for the first step)."""

    def __init__(self, weight_decay, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.weight_decay = weight_decay
        self.learning_rate = learning_rate
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        self.iterations = 0
        self._slots = {}
        self._norm = None

    def _lr(self):
        lr = self.learning_rate
        return float(lr(self.iterations)) if callable(lr) else float(lr)

    def global_norm_sq(self, grads):
        """Squared global norm of ``grads`` (dict or list of device tensors) as a float64 device scalar: one multi-tensor
        launch per 64 tensors + one ordered final sum (``raft_sumsq_multi_f32``)."""
        lib = _dev.lib()
        tensors = [g.as_subclass(torch.Tensor).contiguous() for g in (grads.values() if isinstance(grads, dict) else grads)]
        dev = tensors[0].device
        n = len(tensors)
        need = int(lib.raft_sumsq_multi_workspace_doubles(n))
        if self._norm is None or self._norm[0].device != dev or self._norm[1].numel() < need:
            self._norm = (torch.zeros((1,), dtype=torch.float64, device=dev), torch.empty((need,), dtype=torch.float64, device=dev))
        out, ws = self._norm
        ptrs = (C.c_void_p * n)(*[_dev.ptr(g) for g in tensors])
        sizes = (C.c_int64 * n)(*[g.numel() for g in tensors])
        check(lib.raft_sumsq_multi_f32(ptrs, sizes, n, _dev.ptr(out), _dev.ptr(ws), _dev.stream_ptr()), 'sumsq_multi')
        return out

    def apply_gradients(self, grads, variables, clip_norm=None):
        """``variables`` / ``grads``: dicts name -> fp32 device tensor (variables are updated in place).  ``clip_norm``:
        ``tf.clip_by_global_norm`` (reference model.py:134) folded into the update kernel.  All tensors go through
        ``raft_adamw_step_multi_f32``: one launch per 64 tensors."""
        lib = _dev.lib()
        t = self.iterations + 1
        lr = self._lr()
        wd = float(self.weight_decay(self.iterations)) if callable(self.weight_decay) else float(self.weight_decay)
        lr_t = lr * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)
        gn = self.global_norm_sq({k: grads[k] for k in variables}) if clip_norm is not None else None
        names = list(variables)
        if not names:
            self.iterations += 1
            return gn
        gs = [grads[k].as_subclass(torch.Tensor).contiguous() for k in names]
        for name in names:
            if name not in self._slots:
                self._slots[name] = (torch.zeros_like(variables[name]), torch.zeros_like(variables[name]))
        n = len(names)
        arr = lambda ts: (C.c_void_p * n)(*[_dev.ptr(x) for x in ts])      # noqa: E731
        sizes = (C.c_int64 * n)(*[variables[k].numel() for k in names])
        check(lib.raft_adamw_step_multi_f32(arr([variables[k] for k in names]), arr(gs), arr([self._slots[k][0] for k in names]),
                                            arr([self._slots[k][1] for k in names]), sizes, n, lr_t, self.beta_1, self.beta_2,
                                            self.epsilon, wd, _dev.ptr(gn) if gn is not None else None,
                                            float(clip_norm) if clip_norm is not None else 0.0, _dev.stream_ptr()), 'adamw_step_multi')
        from . import grad as _grad
        _grad.bump_pack_version()      # the variables changed in place through raw pointers: packed copies of them are stale
        self.iterations += 1
        return gn
