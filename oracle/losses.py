"""Oracle restatement of reference tf_raft/losses/losses.py (NumPy; test infrastructure only).
The parity harness uses ``end_point_error``'s EPE formula for its max-EPE metric."""
from __future__ import annotations

import numpy as np


def sequence_loss(y_true, y_pred, gamma=0.8, max_flow=400):
    """reference losses.py:4-21."""
    flow_gt, valid = y_true
    flow_gt = np.asarray(flow_gt, dtype=np.float32)
    n_predictions = len(y_pred)
    mag = np.sqrt(np.sum(flow_gt ** 2, axis=-1))
    valid = np.asarray(valid, dtype=bool) & (mag < max_flow)
    valid = valid.astype(np.float32)[..., None]
    flow_loss = 0.0
    for i in range(n_predictions):
        i_weight = gamma ** (n_predictions - i - 1)
        i_loss = np.abs(np.asarray(y_pred[i], dtype=np.float32) - flow_gt)
        flow_loss += i_weight * np.mean(valid * i_loss)
    return flow_loss


def end_point_error(y_true, y_pred, max_flow=400):
    """reference losses.py:24-43."""
    flow_gt, valid = y_true
    flow_gt = np.asarray(flow_gt, dtype=np.float32)
    mag = np.sqrt(np.sum(flow_gt ** 2, axis=-1))
    valid = np.asarray(valid, dtype=bool) & (mag < max_flow)
    epe = np.sqrt(np.sum((np.asarray(y_pred, dtype=np.float32) - flow_gt) ** 2, axis=-1))
    epe = epe[valid]
    return {
        'epe': float(np.mean(epe)),
        'u1': float(np.mean((epe < 1).astype(np.float32))),
        'u3': float(np.mean((epe < 3).astype(np.float32))),
        'u5': float(np.mean((epe < 5).astype(np.float32))),
    }


def max_epe(flow_a, flow_b):
    """Parity metric (BASELINE.json): max over pixels of ||flow_a - flow_b||_2."""
    d = np.asarray(flow_a, dtype=np.float64) - np.asarray(flow_b, dtype=np.float64)
    return float(np.sqrt((d ** 2).sum(axis=-1)).max())
