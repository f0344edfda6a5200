"""Device mirror of reference ``tf_raft/losses/losses.py`` -- the evaluation side of the forward pass.

    sequence_loss(y_true, y_pred, gamma=0.8, max_flow=400)        losses.py:4-21
    end_point_error(y_true, y_pred, max_flow=400)                 losses.py:24-43
    EndPointError(max_flow=400)                                   losses.py:46-90

``y_true = (flow_gt, valid)`` with ``flow_gt`` (bs, H, W, 2) float and ``valid`` (bs, H, W) bool; ``y_pred`` is the list
of flow predictions the model returns (``sequence_loss``) or one prediction (``end_point_error``).  Arrays may be
NumPy or torch, host or device.  The reductions run as single-pass HIP kernels (csrc/metrics.hip) on the current
stream; results are 0-d fp32 device tensors that answer ``.numpy()`` / ``float()`` like TF eager scalars.  There is no
CPU fallback.
"""
from __future__ import annotations

import torch

from . import _dev
from ._ffi import check

_WS = {}


def _workspace(device):
    key = (device.type, device.index)
    ws = _WS.get(key)
    if ws is None:
        ws = torch.empty((int(_dev.lib().raft_metrics_workspace_doubles()),), dtype=torch.float64, device=device)
        _WS[key] = ws
    return ws


def _truth(y_true):
    try:
        flow_gt, valid = y_true
    except (TypeError, ValueError) as exc:
        raise ValueError('y_true must be the pair (flow_gt, valid)') from exc
    flow_gt = _dev.to_device(flow_gt).as_subclass(torch.Tensor).to(torch.float32).contiguous()
    if flow_gt.dim() < 2 or flow_gt.shape[-1] != 2:
        raise ValueError(f'flow_gt must be (..., 2), got {tuple(flow_gt.shape)}')
    valid = torch.as_tensor(valid) if not isinstance(valid, torch.Tensor) else valid.as_subclass(torch.Tensor)
    if tuple(valid.shape) != tuple(flow_gt.shape[:-1]):
        raise ValueError(f'valid must be {tuple(flow_gt.shape[:-1])}, got {tuple(valid.shape)}')
    valid = (valid != 0).to(device=flow_gt.device, dtype=torch.uint8).contiguous()
    return flow_gt, valid


def _prediction(p, flow_gt):
    p = _dev.to_device(p).as_subclass(torch.Tensor).to(torch.float32)
    if tuple(p.shape) != tuple(flow_gt.shape):
        raise ValueError(f'prediction must be {tuple(flow_gt.shape)}, got {tuple(p.shape)}')
    return p


def sequence_loss(y_true, y_pred, gamma=0.8, max_flow=400):
    """reference losses.py:4-21: sum_i gamma**(n-i-1) * mean(valid * |y_pred[i] - flow_gt|)."""
    flow_gt, valid = _truth(y_true)
    preds = [_prediction(p, flow_gt) for p in y_pred]
    n = len(preds)
    if n == 0:
        return _dev.wrap(torch.zeros((), dtype=torch.float32, device=flow_gt.device))   # flow_loss = 0.0
    if n > 64:
        raise ValueError(f'sequence_loss: at most 64 predictions, got {n}')
    npix = flow_gt.numel() // 2
    # the model returns views of one (iters, bs, H, W, 2) buffer: use it in place, otherwise gather the list once
    stride = npix * 2
    base = preds[0]
    in_place = all(p.is_contiguous() and p.data_ptr() == base.data_ptr() + 4 * stride * i for i, p in enumerate(preds))
    stacked = base if in_place else torch.stack([p.contiguous() for p in preds])
    out = torch.empty((), dtype=torch.float32, device=flow_gt.device)
    check(_dev.lib().raft_sequence_loss_f32(_dev.ptr(flow_gt), _dev.ptr(valid), _dev.ptr(stacked), stride, n, npix,
                                            float(gamma), float(max_flow), _dev.ptr(out), _dev.ptr(_workspace(out.device)),
                                            _dev.stream_ptr()), 'sequence_loss')
    return _dev.wrap(out)


def _metrics(y_true, y_pred, max_flow):
    flow_gt, valid = _truth(y_true)
    pred = _prediction(y_pred, flow_gt).contiguous()
    out = torch.empty((5,), dtype=torch.float32, device=flow_gt.device)
    check(_dev.lib().raft_flow_metrics_f32(_dev.ptr(flow_gt), _dev.ptr(valid), _dev.ptr(pred), flow_gt.numel() // 2,
                                           float(max_flow), _dev.ptr(out), _dev.ptr(_workspace(out.device)),
                                           _dev.stream_ptr()), 'flow_metrics')
    return out


def end_point_error(y_true, y_pred, max_flow=400):
    """reference losses.py:24-43: {'epe', 'u1', 'u3', 'u5'} over the valid pixels with |flow_gt| < max_flow."""
    out = _metrics(y_true, y_pred, max_flow)
    return {k: _dev.wrap(out[i]) for i, k in enumerate(('epe', 'u1', 'u3', 'u5'))}


class EndPointError:
    """reference losses.py:46-90 (a keras Metric): running means of the per-batch EPE / u1 / u3 / u5 of ``y_pred[-1]``."""

    def __init__(self, max_flow=400, **kwargs):
        self.name = kwargs.pop('name', 'end_point_error')
        if kwargs:
            raise TypeError(f'unexpected keyword arguments {sorted(kwargs)}')
        self.max_flow = max_flow
        self.reset_states()

    def reset_states(self):
        self._sum = None      # device tensor [epe, u1, u3, u5]
        self.count = 0

    def update_state(self, y_true, y_pred):
        out = _metrics(y_true, y_pred[-1], self.max_flow)[:4]       # losses.py:68: the last prediction
        self._sum = out.clone() if self._sum is None else self._sum + out
        self.count += 1

    def result(self):
        if self._sum is None:
            nan = float('nan')                                       # 0 / 0 in the reference
            return {'epe': nan, 'u1': nan, 'u3': nan, 'u5': nan}
        mean = self._sum / float(self.count)
        return {k: _dev.wrap(mean[i]) for i, k in enumerate(('epe', 'u1', 'u3', 'u5'))}


class Mean:
    """The part of ``tf.keras.metrics.Mean`` the reference uses (model.py:118-124): a running mean of scalars."""

    def __init__(self, name='mean'):
        self.name = name
        self.reset_states()

    def reset_states(self):
        self.total = 0.0
        self.count = 0

    def update_state(self, value):
        # a device scalar stays on the device (no host synchronisation inside a training step: the caller converts the result
        # when it wants the number, as with the Keras metric's tensor)
        if isinstance(value, torch.Tensor) and value.is_cuda:
            v = value.detach().as_subclass(torch.Tensor).to(torch.float32).reshape(())
            self.total = v.clone() if self.count == 0 else self.total + v
        else:
            self.total = self.total + float(value)
        self.count += 1

    def result(self):
        """Running mean.  Fed device scalars (train_step / test_step) it is a 0-dim DEVICE tensor -- like the Keras metric's
        tensor, no host synchronisation inside a step -- otherwise a Python float (0.0 when nothing was fed: keras
        divide_no_nan).  ``float(m.result())`` works on both (and synchronises); logging code that wants plain numbers
        (``json.dumps``, formatting) uses ``result_float()`` or ``losses.to_floats(logs)``."""
        if not self.count:
            return 0.0                                              # keras: divide_no_nan
        mean = self.total / self.count
        return _dev.wrap(mean) if isinstance(mean, torch.Tensor) else mean

    def result_float(self) -> float:
        """``result()`` as a Python float (one device synchronisation when the mean lives on the device)."""
        return float(self.result())


def to_floats(logs):
    """The dict a ``train_step`` / ``test_step`` returns (0-dim device tensors) as plain Python floats: ONE device-to-host copy
    for all entries.  For logging / ``json.dumps``; the step functions themselves never synchronise."""
    keys = list(logs)
    dev = [k for k in keys if isinstance(logs[k], torch.Tensor)]
    out = {k: float(logs[k]) for k in keys if k not in dev}
    if dev:
        vals = torch.stack([logs[k].detach().as_subclass(torch.Tensor).to(torch.float32).reshape(()) for k in dev]).cpu().tolist()
        out.update(dict(zip(dev, vals)))
    return {k: out[k] for k in keys}
