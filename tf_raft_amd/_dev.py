"""Device-memory plumbing: PyTorch-ROCm tensors own HBM buffers and provide the HIP stream;
all compute on the hot path goes through ``libraft_hip.so`` (``_ffi``)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _ffi


_GPU_SEEN = False


def require_gpu() -> torch.device:
    global _GPU_SEEN
    if not _GPU_SEEN:                          # asked once: torch.cuda.is_available() costs ~8 us per call on ROCm
        if not torch.cuda.is_available():
            raise RuntimeError('tf_raft_amd needs a ROCm GPU (MI355X / gfx950): no device is visible and '
                               'there is no CPU fallback')
        _GPU_SEEN = True
    return torch.device('cuda', torch.cuda.current_device())


class DeviceTensor(torch.Tensor):
    """A device tensor that also answers ``.numpy()`` / ``np.asarray`` like the TF eager tensors
    the reference returns (reference model.py:109), by copying to the host first."""

    def numpy(self):   # type: ignore[override]
        return self.detach().as_subclass(torch.Tensor).cpu().numpy()

    def __array__(self, dtype=None, copy=None):   # noqa: D105
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)


def wrap(t: torch.Tensor) -> torch.Tensor:
    return t.as_subclass(DeviceTensor)


def to_device(x, device=None, dtype=torch.float32) -> torch.Tensor:
    """numpy / torch (any device) -> contiguous device tensor of ``dtype``."""
    if isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == dtype and device is None and x.is_contiguous():
        return x.detach().as_subclass(torch.Tensor)      # already where and what it should be (the training path: thousands of calls per step)
    device = device or require_gpu()
    if isinstance(x, torch.Tensor):
        t = x.detach().as_subclass(torch.Tensor)
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
    if t.device != device and t.element_size() < torch.empty((), dtype=dtype).element_size():
        t = t.to(device=device)                 # narrow host data (uint8 images) crosses PCIe as it is; cast in HBM
    return t.to(device=device, dtype=dtype).contiguous()


def ptr(t: torch.Tensor) -> int:
    return t.data_ptr()


def stream_ptr() -> int:
    """Raw hipStream_t of torch's CURRENT stream on the current device (honours ``torch.cuda.stream(...)`` contexts).  Through the
    C binding directly: ``torch.cuda.current_stream().cuda_stream`` costs ~9 us per call (device-index resolution in Python), and
    a training step asks 5000 times."""
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:                      # another torch build: the public route
        return torch.cuda.current_stream().cuda_stream


# Side streams of the forward pass, ONE set per device and process, shared by every model object.  HIP maps streams onto a few
# hardware queues in creation order (4 by default); a second model with streams of its own put its flow branch on the main
# chain's queue and the two serialised (measured in round 4: the fifth RAFT object of a process ran at 211 instead of 324
# pairs/s).  Work enqueued by different models is still ordered per model by events / wait_stream; two host threads driving two
# models on one device share the side streams and interleave there.
_SIDE_STREAMS = {}


def side_stream(device, role: str) -> 'torch.cuda.Stream':
    """The process-wide side stream of ``role`` ('flow', 'mask', 'encoder', 'loop') on ``device``."""
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), role)
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return s


def i64_array(values):
    return (C.c_int64 * len(values))(*values)


def lib():
    return _ffi.load_library()
