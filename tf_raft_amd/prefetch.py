"""Host batches -> HBM ahead of the compute stream.

The reference feeds the model from ``tf.data`` pipelines that end in ``.prefetch(buffer_size=1)`` with uint8 images
(reference train_sintel.py:50-56, 73-76; train_chairs.py:57, 72): the next batch is made ready while the current one is
being computed.  ``prefetch_to_device`` is that stage for host-resident batches: every array of batch ``i + 1`` is
copied into a pinned staging buffer and uploaded on a separate copy stream while batch ``i`` runs, in the dtype it
arrived in (uint8 images cross PCIe as bytes; the cast to fp32 happens on the device).

Plumbing only: no arithmetic on the prediction path is done here.
"""
from __future__ import annotations

from typing import Iterable, Iterator, Sequence, Tuple

import numpy as np
import torch

from . import _dev


class _Slot:
    """One set of pinned staging buffers plus the event that says the last upload out of it has finished."""

    def __init__(self):
        self.pins = []
        self.done = None

    def stage(self, arrays: Sequence[np.ndarray]):
        if self.done is not None:
            self.done.synchronize()                       # the copy engine may still be reading these buffers
        if len(self.pins) != len(arrays) or any(
                tuple(p.shape) != a.shape or p.numpy().dtype != a.dtype for p, a in zip(self.pins, arrays)):
            self.pins = [torch.from_numpy(np.empty(a.shape, a.dtype)).pin_memory() for a in arrays]
        for p, a in zip(self.pins, arrays):
            np.copyto(p.numpy(), a)
        return self.pins


def _as_host_arrays(batch, device) -> tuple:
    """Host arrays of a batch as contiguous NumPy arrays; tensors that already live on ``device`` pass through untouched."""
    out = []
    for a in batch:
        if isinstance(a, torch.Tensor):
            if a.device == device:
                out.append(a.detach())
                continue
            a = a.detach().cpu().numpy()
        a = np.ascontiguousarray(np.asarray(a))
        if a.dtype == np.float64:
            a = a.astype(np.float32)
        out.append(a)
    return tuple(out)


def prefetch_to_device(batches: Iterable, buffer_size: int = 1, device=None) -> Iterator[Tuple[torch.Tensor, ...]]:
    """Iterate ``batches`` (each a tuple / list of host arrays) as tuples of device tensors, ``buffer_size`` batches
    ahead of the consumer.  The tensors of a yielded batch are safe to use on the consumer's current stream."""
    if buffer_size < 1:
        raise ValueError(f'buffer_size must be >= 1, got {buffer_size}')
    device = device or _dev.require_gpu()
    copy_stream = torch.cuda.Stream(device=device)
    slots = [_Slot() for _ in range(buffer_size + 1)]
    it = iter(batches)
    pending = []                                          # (device tensors, upload-finished event, resident flags), oldest first
    n = 0

    def issue() -> bool:
        nonlocal n
        try:
            batch = next(it)
        except StopIteration:
            return False
        slot = slots[n % len(slots)]
        n += 1
        arrays = _as_host_arrays(batch, device)
        host = [a for a in arrays if not isinstance(a, torch.Tensor)]
        pins = iter(slot.stage(host))
        with torch.cuda.stream(copy_stream):
            dev = tuple(a if isinstance(a, torch.Tensor) else next(pins).to(device, non_blocking=True) for a in arrays)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        slot.done = ev
        pending.append((dev, ev, tuple(isinstance(a, torch.Tensor) for a in arrays)))
        return True

    for _ in range(buffer_size):
        if not issue():
            break
    while pending:
        dev, ev, resident = pending.pop(0)
        issue()                                           # the upload of the next batch overlaps this batch's compute
        cur = torch.cuda.current_stream(device)
        cur.wait_event(ev)
        for t, was_resident in zip(dev, resident):
            if not was_resident:
                t.record_stream(cur)                      # allocated on the copy stream, consumed on this one
        yield tuple(_dev.wrap(t) for t in dev)
