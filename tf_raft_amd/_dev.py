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


class Pending:
    """Results still being computed on another stream than the caller's (the pipelined forward of ``tf_raft_amd.model``: the
    recurrent loop runs on the process-wide 'loop' stream so that the NEXT call's encoders can run under it).  ``join()`` makes
    the CURRENT stream wait for them, once per stream; every route from a ``DeviceTensor`` to its bytes calls it first."""

    __slots__ = ('event', 'device', '_joined')

    def __init__(self, event, device):
        self.event, self.device, self._joined = event, device, set()

    def join(self):
        try:
            sid = torch._C._cuda_getCurrentRawStream(self.device.index)
        except AttributeError:
            sid = torch.cuda.current_stream(self.device).cuda_stream
        if sid not in self._joined:
            torch.cuda.current_stream(self.device).wait_event(self.event)
            self._joined.add(sid)

    def __reduce__(self):
        # pickling / deepcopy of a result tensor copies its __dict__: the copy's bytes are read on the current stream (ordered
        # behind the loop here), and an event cannot travel -- the copy carries no Pending
        self.join()
        return (_no_pending, ())


def _no_pending():
    return None


# attribute reads that say nothing about the tensor's bytes: no ordering needed
_METADATA = frozenset(('shape', 'dtype', 'device', 'is_cuda', 'ndim', 'layout', 'requires_grad', 'grad', 'grad_fn', 'is_leaf',
                       'names', 'is_sparse', 'is_quantized', 'is_meta', 'output_nr', '_version', 'is_cpu', 'itemsize', 'nbytes'))


def _join_nested(items):
    """Join every pending DeviceTensor among ``items``, looking into lists / tuples / dicts at any depth (``torch.cat(tensors=[..])``,
    ``torch.stack(tensors=(..))``, nested containers)."""
    for a in items:
        t = type(a)
        if t is DeviceTensor:
            p = a.__dict__.get('_pending')
            if p is not None:
                p.join()
        elif t in (list, tuple):
            _join_nested(a)
        elif t is dict:
            _join_nested(a.values())


class DeviceTensor(torch.Tensor):
    """A device tensor that also answers ``.numpy()`` / ``np.asarray`` like the TF eager tensors
    the reference returns (reference model.py:109), by copying to the host first.

    A tensor returned by a pipelined forward call carries a ``Pending``: the first operation that touches its data (any
    torch function or method, ``data_ptr()``, ``as_subclass``, ``numpy()``) makes the current stream wait for the loop that
    produces it.  Reading ``shape`` / ``dtype`` / ``device`` does not."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if not (getattr(func, '__name__', '') == '__get__' and getattr(getattr(func, '__self__', None), '__name__', '') in _METADATA):
            _join_nested(args)
            if kwargs:
                _join_nested(kwargs.values())
        return super().__torch_function__(func, types, args, kwargs or {})

    def join(self):
        """Make the current stream wait for this tensor's producer (no-op for a tensor that is not pending)."""
        p = self.__dict__.get('_pending')
        if p is not None:
            p.join()
        return self

    def as_subclass(self, cls):   # type: ignore[override]   (not routed through __torch_function__)
        self.join()
        return super().as_subclass(cls)

    def __deepcopy__(self, memo):
        self.join()
        c = wrap(torch.Tensor.as_subclass(self, torch.Tensor).detach().clone())     # a copy is complete: no Pending
        memo[id(self)] = c
        return c

    def __dlpack__(self, *args, **kwargs):   # type: ignore[override]   (torch.from_dlpack / np.from_dlpack / other frameworks)
        self.join()
        return super().__dlpack__(*args, **kwargs)

    def numpy(self):   # type: ignore[override]
        return self.detach().as_subclass(torch.Tensor).cpu().numpy()

    def __array__(self, dtype=None, copy=None):   # noqa: D105
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)


def wrap(t: torch.Tensor, pending: 'Pending' = None) -> torch.Tensor:
    d = t.as_subclass(DeviceTensor)
    if pending is not None:
        d.__dict__['_pending'] = pending
    return d


def join(t):
    """Order the current stream behind whatever still produces ``t`` (a tensor or a list of them); returns ``t``."""
    for x in (t if isinstance(t, (list, tuple)) else (t,)):
        if isinstance(x, DeviceTensor):
            x.join()
    return t


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


# Priority of EVERY side stream of the package (one pool: streams of two priorities come from two pools whose members collide on the
# hardware queues, see model.py _loop_priority).  RAFT_STREAM_PRIORITY=-1: all of them high -- i.e. above the caller's stream.
import os as _os
STREAM_PRIORITY = int(_os.environ.get('RAFT_STREAM_PRIORITY', '0'))


def side_stream(device, role: str, priority: int = None) -> 'torch.cuda.Stream':
    """The process-wide side stream of ``role`` ('flow', 'mask', 'encoder', 'loop'; lane k > 0 of the pipelined forward:
    'loop1', 'flow1', ...) and ``priority`` on ``device``.  Streams of different priorities are different streams (a
    stream's priority is fixed when it is created)."""
    device = torch.device(device)
    priority = STREAM_PRIORITY if priority is None else priority
    key = (device.index if device.index is not None else torch.cuda.current_device(), role, int(priority))
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device, priority=priority)
    return s


def reserve_streams(device, lanes: int = 3) -> None:
    """Create the package's side streams on ``device`` in ONE canonical order the first time any model is built: the loop lanes
    first, then the flow / mask / encoder streams of the serial schedule.  HIP assigns hardware queues in creation order, so this
    makes the mapping of the lanes independent of which kind of model a process happens to build first (a multi-lane model built
    after a serial one found its lanes on queues the serial streams had taken: 367 against 377 pairs/s, profiles/r12w_config_bench.txt)."""
    for role in [('loop' if k == 0 else f'loop{k}') for k in range(lanes)] + ['flow', 'mask', 'encoder']:
        side_stream(device, role)


def i64_array(values):
    return (C.c_int64 * len(values))(*values)


def lib():
    return _ffi.load_library()
