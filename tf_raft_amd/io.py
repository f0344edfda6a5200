"""Output-side helpers of the prediction path: Middlebury ``.flo`` files and the flow colour coding.

Mirrors what a user of the reference calls right after ``model([image1, image2])``:
``tf_raft/datasets/frame_utils.py:12-31`` (readFlow), ``70-99`` (writeFlow) and
``tf_raft/datasets/flow_viz.py:20-132`` (make_colorwheel / flow_uv_to_colors / flow_to_image: the Middlebury
colour wheel of Baker et al., ICCV 2007).  NumPy on the host: these touch one (H, W, 2) array per image pair and
are not on the device path.  Accepts NumPy arrays, torch tensors (any device) and the model's output tensors.
"""
from __future__ import annotations

import numpy as np

FLO_MAGIC = np.float32(202021.25)        # the bytes b'PIEH' read as a little-endian float


def _to_numpy(x) -> np.ndarray:
    if hasattr(x, 'detach'):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


def write_flo(filename, uv, v=None) -> None:
    """reference frame_utils.py:70-99.  ``uv``: (H, W, 2) flow, or ``uv`` = u and ``v`` = v as (H, W) planes.
    Layout: float32 magic 202021.25, int32 width, int32 height, then H*W interleaved (u, v) float32 pairs."""
    u = _to_numpy(uv)
    if v is None:
        if u.ndim != 3 or u.shape[2] != 2:
            raise ValueError(f'flow must have shape (H, W, 2), got {u.shape}')
        flow = u
    else:
        v = _to_numpy(v)
        if u.ndim != 2 or u.shape != v.shape:
            raise ValueError(f'u and v must be (H, W) planes of one shape, got {u.shape} and {v.shape}')
        flow = np.stack([u, v], axis=-1)
    h, w = flow.shape[:2]
    with open(filename, 'wb') as f:
        f.write(FLO_MAGIC.astype('<f4').tobytes())
        f.write(np.array([w, h], dtype='<i4').tobytes())
        f.write(np.ascontiguousarray(flow, dtype='<f4').tobytes())


def read_flo(filename) -> np.ndarray:
    """reference frame_utils.py:12-31: (H, W, 2) float32.  Raises ``ValueError`` on a bad magic number or a
    truncated file (the reference prints a message and returns ``None`` for the former)."""
    with open(filename, 'rb') as f:
        head = f.read(12)
        if len(head) < 12 or np.frombuffer(head[:4], '<f4')[0] != FLO_MAGIC:
            raise ValueError(f'{filename}: magic number incorrect, invalid .flo file')
        w, h = (int(t) for t in np.frombuffer(head[4:], '<i4'))
        data = np.frombuffer(f.read(), '<f4')
    if w <= 0 or h <= 0 or data.size < 2 * w * h:
        raise ValueError(f'{filename}: header says {w}x{h} but the file holds {data.size} floats')
    return data[:2 * w * h].reshape(h, w, 2).astype(np.float32)


# colour wheel: six hue segments, each ramping one RGB channel up or down (reference flow_viz.py:20-67)
_SEGMENTS = ((15, 0, 1, +1),   # red -> yellow   : R = 255, G rises
             (6, 1, 0, -1),    # yellow -> green : G = 255, R falls
             (4, 1, 2, +1),    # green -> cyan   : G = 255, B rises
             (11, 2, 1, -1),   # cyan -> blue    : B = 255, G falls
             (13, 2, 0, +1),   # blue -> magenta : B = 255, R rises
             (6, 0, 2, -1))    # magenta -> red  : R = 255, B falls


def make_colorwheel() -> np.ndarray:
    """(55, 3) float64 colour wheel, values 0..255."""
    rows = []
    for n, full, ramp, sign in _SEGMENTS:
        seg = np.zeros((n, 3))
        seg[:, full] = 255
        step = np.floor(255 * np.arange(n) / n)
        seg[:, ramp] = step if sign > 0 else 255 - step
        rows.append(seg)
    return np.concatenate(rows, axis=0)


def flow_uv_to_colors(u, v, convert_to_bgr=False) -> np.ndarray:
    """reference flow_viz.py:70-106: colour of unit-normalised flow components u, v (H, W) -> uint8 (H, W, 3)."""
    u, v = _to_numpy(u), _to_numpy(v)
    wheel = make_colorwheel()
    ncols = wheel.shape[0]
    rad = np.sqrt(np.square(u) + np.square(v))
    fk = (np.arctan2(-v, -u) / np.pi + 1) / 2 * (ncols - 1)
    k0 = np.floor(fk).astype(np.int32)
    k1 = np.where(k0 + 1 == ncols, 0, k0 + 1)
    f = (fk - k0)[..., None]
    col = (1 - f) * (wheel[k0] / 255.0) + f * (wheel[k1] / 255.0)          # (H, W, 3)
    inside = (rad <= 1)[..., None]
    col = np.where(inside, 1 - rad[..., None] * (1 - col), col * 0.75)       # saturate with radius; dim out of range
    img = np.floor(255 * col).astype(np.uint8)
    return img[..., ::-1] if convert_to_bgr else img


def flow_to_image(flow_uv, clip_flow=None, convert_to_bgr=False) -> np.ndarray:
    """reference flow_viz.py:109-132: (H, W, 2) flow -> uint8 (H, W, 3), normalised by the largest magnitude."""
    flow_uv = _to_numpy(flow_uv)
    if flow_uv.ndim != 3 or flow_uv.shape[2] != 2:
        raise ValueError(f'input flow must have shape (H, W, 2), got {flow_uv.shape}')
    if clip_flow is not None:
        flow_uv = np.clip(flow_uv, 0, clip_flow)                              # reference flow_viz.py:124 clips to [0, clip]
    u, v = flow_uv[:, :, 0], flow_uv[:, :, 1]
    rad_max = np.max(np.sqrt(np.square(u) + np.square(v)))
    eps = 1e-5
    return flow_uv_to_colors(u / (rad_max + eps), v / (rad_max + eps), convert_to_bgr)
