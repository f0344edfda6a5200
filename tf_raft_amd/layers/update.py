"""Device mirror of reference ``tf_raft/layers/update.py``: ``BasicUpdateBlock`` and
``SmallUpdateBlock`` with the reference's call signature
``block([net, inp, corr, flow]) -> (net, mask, delta_flow)``, executed by the fp32-MFMA
convolution kernels of ``csrc/conv.hip`` through the C ABI.

The recurrent loop of the model does not go through ``__call__`` (which has to marshal the four
inputs into the fused state buffers on every call); it drives ``raft_iterate_*`` on a persistent
:class:`UpdateState` instead.  ``__call__`` exists for API parity and for the per-block tests.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from .. import _dev, _ffi, packing
from .. import weights as weights_mod
from .._ffi import check

_GEOM = {
    # variant: hdim, cdim, x_ld, flow_slot, corr_ld, corr_used, radius, mask
    'raft': dict(hdim=128, cdim=128, x_ld=256, flow_slot=254, corr_ld=352, corr_used=324, radius=4, mask=576, ctx=768),
    'small': dict(hdim=96, cdim=64, x_ld=160, flow_slot=144, corr_ld=224, corr_used=196, radius=3, mask=0, ctx=0),
}


class UpdateState:
    """Caller-owned device buffers of the recurrent loop (``raft_state`` in include/raft_hip.h)."""

    def __init__(self, variant: str, B: int, h: int, w: int, device):
        g = _GEOM[variant]
        self.variant, self.B, self.h, self.w, self.g = variant, B, h, w, g
        M = B * h * w
        lib = _dev.lib()
        ws = (lib.raft_update_workspace_floats if variant == 'raft' else lib.raft_small_update_workspace_floats)(B, h, w)
        f = dict(device=device, dtype=torch.float32)
        self.net = torch.empty((B, h, w, g['hdim']), **f)
        self.x = torch.empty((B, h, w, g['x_ld']), **f)
        self.corr = torch.empty((B, h, w, g['corr_ld']), **f)
        self.coords1 = torch.empty((B, h, w, 2), **f)
        self.flow = torch.empty((B, h, w, 2), **f)
        self.delta = torch.empty((B, h, w, 2), **f)
        self.mask = torch.empty((B, h, w, max(g['mask'], 1)), **f)
        self.ws = torch.empty((ws,), **f)
        # loop-invariant GRU context terms (raft_gru_context_f32); SmallRAFT does not use them
        self.ctx = torch.empty((B, h, w, g['ctx']) if g['ctx'] else (4,), **f)
        self.c = _ffi.State(net=_dev.ptr(self.net), x=_dev.ptr(self.x), corr=_dev.ptr(self.corr),
                            coords1=_dev.ptr(self.coords1), flow=_dev.ptr(self.flow),
                            delta=_dev.ptr(self.delta), mask=_dev.ptr(self.mask), ws=_dev.ptr(self.ws),
                            ctx=_dev.ptr(self.ctx))
        assert M > 0


class _UpdateBlock:
    variant = 'raft'
    _struct = _ffi.BasicUpdateWeights

    def __init__(self, filters, weights: Optional[Dict[str, np.ndarray]] = None, prefix='update_block', seed=0):
        g = _GEOM[self.variant]
        if filters != g['hdim']:
            raise ValueError(f'{type(self).__name__} is instantiated for filters={g["hdim"]} only (got {filters})')
        self.filters = filters
        self.prefix = prefix
        self._blob = None
        if weights is None:
            weights = {k: v for k, v in weights_mod.init_weights(self.variant, seed).items() if k.startswith(prefix)}
        self.set_weights(weights)

    def _pack(self, weights):
        raise NotImplementedError

    def set_weights(self, weights: Dict[str, np.ndarray]) -> None:
        """Repack Keras-layout weights (``{prefix}/...``) and upload them as one device blob."""
        dev = _dev.require_gpu()
        packed = self._pack(weights)
        chunks, offs, pos = [], {}, 0
        for field, wp, b, npad in packed:
            for tag, arr in (('w', wp), ('b', b)):
                flat = np.ascontiguousarray(arr, dtype=np.float32).ravel()
                offs[(field, tag)] = pos
                chunks.append(flat)
                pad = (-flat.size) % 4          # keep every sub-array 16-byte aligned
                if pad:
                    chunks.append(np.zeros(pad, dtype=np.float32))
                pos += flat.size + pad
        self._blob = torch.from_numpy(np.concatenate(chunks)).to(dev)
        base = self._blob.data_ptr()
        self.c = self._struct()
        for field, _, _, npad in packed:
            setattr(self.c, field, _ffi.ConvWeights(wp=base + 4 * offs[(field, 'w')],
                                                    bias=base + 4 * offs[(field, 'b')], npad=npad))

    # -- reference call signature ------------------------------------------------------------
    def __call__(self, inputs):
        net, inp, corr, flow = [_dev.to_device(t) for t in inputs]
        g = _GEOM[self.variant]
        B, h, w, _ = net.shape
        if net.shape[-1] != g['hdim'] or inp.shape[-1] != g['cdim'] or corr.shape[-1] != g['corr_used'] \
                or flow.shape[-1] != 2:
            raise ValueError('expected [net (..,%d), inp (..,%d), corr (..,%d), flow (..,2)]'
                             % (g['hdim'], g['cdim'], g['corr_used']))
        st = UpdateState(self.variant, B, h, w, net.device)
        st.net.copy_(net)
        st.x.zero_()
        st.x[..., :g['cdim']] = inp
        st.x[..., g['flow_slot']:g['flow_slot'] + 2] = flow
        st.corr.zero_()
        st.corr[..., :g['corr_used']] = corr
        st.flow.copy_(flow)
        # coords1 = coords0 + flow so that the kernel's coords1 += delta keeps flow consistent
        ys, xs = torch.meshgrid(torch.arange(h, device=net.device, dtype=torch.float32),
                                torch.arange(w, device=net.device, dtype=torch.float32), indexing='ij')
        st.coords1.copy_(torch.stack([xs, ys], dim=-1).unsqueeze(0) + flow)
        self.prepare(st)
        self.step(st)
        mask = _dev.wrap(st.mask) if g['mask'] else None
        return _dev.wrap(st.net), mask, _dev.wrap(st.delta)

    def prepare(self, st: UpdateState) -> None:
        """Loop-invariant work that depends on ``inp`` only (once per forward, before the first ``step``)."""

    def step(self, st: UpdateState) -> None:
        raise NotImplementedError


class BasicUpdateBlock(_UpdateBlock):
    """reference update.py:128-153."""
    variant = 'raft'
    _struct = _ffi.BasicUpdateWeights

    def __init__(self, filters=128, **kwargs):
        super().__init__(filters, **kwargs)

    def _pack(self, weights):
        return packing.pack_basic_update(weights, self.prefix)

    def prepare(self, st):
        check(_dev.lib().raft_gru_context_f32(C.byref(self.c), st.B, st.h, st.w, C.byref(st.c),
                                              _dev.stream_ptr()), 'gru_context')

    def step(self, st):
        check(_dev.lib().raft_update_basic_f32(C.byref(self.c), st.B, st.h, st.w, C.byref(st.c),
                                               _dev.stream_ptr()), 'update_basic')


class SmallUpdateBlock(_UpdateBlock):
    """reference update.py:109-125 (returns ``(net, None, delta_flow)``)."""
    variant = 'small'
    _struct = _ffi.SmallUpdateWeights

    def __init__(self, filters=96, **kwargs):
        super().__init__(filters, **kwargs)

    def _pack(self, weights):
        return packing.pack_small_update(weights, self.prefix)

    def step(self, st):
        check(_dev.lib().raft_update_small_f32(C.byref(self.c), st.B, st.h, st.w, C.byref(st.c),
                                               _dev.stream_ptr()), 'update_small')
