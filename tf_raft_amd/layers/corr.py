"""Device mirror of reference ``tf_raft/layers/corr.py``: same names, argument meaning and
shapes, computed by the HIP kernels in ``csrc/corr.hip`` / ``ondemand.hip`` / ``upsample.hip``.

Tensors are fp32 device tensors in the reference's NHWC layouts.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _dev
from .._ffi import check


def _geometry(B, h, w, levels):
    lib = _dev.lib()
    off = (C.c_int64 * (levels + 1))()
    lh = (C.c_int * levels)()
    lw = (C.c_int * levels)()
    check(lib.raft_corr_pyramid_layout(B, h, w, levels, off, lh, lw), 'corr_pyramid_layout')
    return off, list(lh), list(lw)


def _tiles(h, w):
    return (h + 3) // 4, (w + 7) // 8


def tile_maps(maps):
    """(n, h, w) row-major maps -> (n, map_floats) in the library's 4x8-tiled, zero-padded map layout
    (include/raft_hip.h, raft_corr_pyramid_layout)."""
    n, h, w = maps.shape
    ty, tx = _tiles(h, w)
    padded = torch.zeros((n, ty * 4, tx * 8), device=maps.device, dtype=maps.dtype)
    padded[:, :h, :w] = maps
    return padded.view(n, ty, 4, tx, 8).permute(0, 1, 3, 2, 4).reshape(n, ty * tx * 32)


def untile_maps(flat, h, w):
    """Inverse of ``tile_maps``: (n, map_floats) -> (n, h, w)."""
    n = flat.shape[0]
    ty, tx = _tiles(h, w)
    return flat.view(n, ty, tx, 4, 8).permute(0, 1, 3, 2, 4).reshape(n, ty * 4, tx * 8)[:, :h, :w]


def bilinear_sampler(image, coords):
    """reference corr.py:28-69.  image (N, h, w, 1); coords (N, kh, kw, 2) xy -> (N, kh, kw, 1).
    Clamp, then ceil/floor weights: an integer or out-of-range coordinate samples 0."""
    image = _dev.to_device(image)
    coords = _dev.to_device(coords)
    if image.dim() != 4 or image.shape[-1] != 1:
        raise ValueError(f'image must have shape (N, h, w, 1), got {tuple(image.shape)}')
    if coords.dim() != 4 or coords.shape[-1] != 2 or coords.shape[0] != image.shape[0]:
        raise ValueError(f'coords must have shape (N, kh, kw, 2) with N={image.shape[0]}, got {tuple(coords.shape)}')
    n, h, w, _ = image.shape
    _, kh, kw, _ = coords.shape
    out = torch.empty((n, kh, kw, 1), device=image.device, dtype=torch.float32)
    check(_dev.lib().raft_bilinear_sampler_f32(_dev.ptr(image), _dev.ptr(coords), n, h, w, kh, kw,
                                               _dev.ptr(out), _dev.stream_ptr()), 'bilinear_sampler')
    return _dev.wrap(out)


def coords_grid(batch_size, height, width):
    """reference corr.py:72-90: (bs, h, w, 2), [..., 0] = x, [..., 1] = y."""
    dev = _dev.require_gpu()
    out = torch.empty((batch_size, height, width, 2), device=dev, dtype=torch.float32)
    check(_dev.lib().raft_coords_grid_f32(_dev.ptr(out), batch_size, height, width, _dev.stream_ptr()),
          'coords_grid')
    return _dev.wrap(out)


def upflow8(flow, mode='bilinear'):
    """reference corr.py:93-96: 8 * tf.image.resize(flow, (8h, 8w), 'bilinear')."""
    if mode != 'bilinear':
        raise NotImplementedError(f'upflow8 mode {mode!r} is not implemented')
    flow = _dev.to_device(flow)
    if flow.dim() != 4 or flow.shape[-1] != 2:
        raise ValueError(f'flow must have shape (bs, h, w, 2), got {tuple(flow.shape)}')
    bs, h, w, _ = flow.shape
    out = torch.empty((bs, 8 * h, 8 * w, 2), device=flow.device, dtype=torch.float32)
    check(_dev.lib().raft_upflow8_f32(_dev.ptr(flow), bs, h, w, _dev.ptr(out), _dev.stream_ptr()), 'upflow8')
    return _dev.wrap(out)


class CorrBlock:
    """reference corr.py:99-162.

    ``CorrBlock(fmap1, fmap2, num_levels, radius)`` builds the all-pairs correlation volume and
    its pyramid on the device (one allocation, per-query maps 4x8-tiled); ``corr_pyramid`` is the list
    of ``(bs*h*w, h_l, w_l, 1)`` maps (un-tiled copies); ``retrieve(coords)`` returns ``(bs, h, w, levels*(2r+1)^2)``.

    ``alternate=True`` stores no volume: ``retrieve`` computes the footprint correlations on
    demand from ``fmap1`` and the pooled ``fmap2`` pyramid (high-resolution inputs).
    """

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4, alternate=False):
        fmap1 = _dev.to_device(fmap1)
        fmap2 = _dev.to_device(fmap2)
        if fmap1.dim() != 4 or fmap1.shape != fmap2.shape:
            raise ValueError(f'fmap1/fmap2 must both be (bs, h, w, C), got {tuple(fmap1.shape)} / {tuple(fmap2.shape)}')
        self.fmap1 = fmap1
        self.fmap2 = fmap2
        self.num_levels = num_levels
        self.radius = radius
        self.alternate = alternate
        bs, h, w, c = fmap1.shape
        lib = _dev.lib()
        self._off, self._lh, self._lw = _geometry(bs, h, w, num_levels)
        ws_floats = lib.raft_corr_build_workspace_floats(bs, h, w, c, num_levels)
        if ws_floats <= 0:
            raise ValueError('invalid correlation geometry')
        self._f2pyr = torch.empty((ws_floats,), device=fmap1.device, dtype=torch.float32)
        if alternate:
            self._pyr = None
            check(lib.raft_fmap_pyramid_f32(_dev.ptr(fmap2), bs, h, w, c, num_levels, _dev.ptr(self._f2pyr),
                                            _dev.stream_ptr()), 'fmap_pyramid')
        else:
            self._pyr = torch.empty((self._off[num_levels],), device=fmap1.device, dtype=torch.float32)
            check(lib.raft_corr_build_f32(_dev.ptr(fmap1), _dev.ptr(fmap2), bs, h, w, c, num_levels,
                                          _dev.ptr(self._pyr), self._off, _dev.ptr(self._f2pyr),
                                          _dev.stream_ptr()), 'corr_build')

    @property
    def corr_pyramid(self):
        if self._pyr is None:
            raise AttributeError('alternate (on-demand) CorrBlock stores no correlation volume')
        bs, h, w, _ = self.fmap1.shape
        n = bs * h * w
        out = []
        for l in range(self.num_levels):       # un-tiled copies in the reference's (bs*h*w, h_l, w_l, 1) shape
            flat = self._pyr[self._off[l]:self._off[l + 1]].view(n, -1)
            out.append(_dev.wrap(untile_maps(flat, self._lh[l], self._lw[l]).contiguous().unsqueeze(-1)))
        return out

    def untile_pyramid(self, flat):
        """A flat tensor in the pyramid's tiled layout (e.g. the pyramid gradient of ``tf_raft_amd.grad``) as the
        reference's list of ``(bs*h*w, h_l, w_l, 1)`` per-level maps."""
        bs, h, w, _ = self.fmap1.shape
        n = bs * h * w
        return [_dev.wrap(untile_maps(flat[self._off[l]:self._off[l + 1]].view(n, -1), self._lh[l], self._lw[l])
                          .contiguous().unsqueeze(-1)) for l in range(self.num_levels)]

    def _set_level(self, l, maps):
        """Overwrite level ``l`` of the stored volume with (bs*h*w, h_l, w_l[, 1]) maps (parity tests)."""
        maps = _dev.to_device(maps).reshape(-1, self._lh[l], self._lw[l])
        self._pyr[self._off[l]:self._off[l + 1]].copy_(tile_maps(maps).reshape(-1))

    def retrieve(self, coords, out=None, ld_out=None):
        """reference corr.py:116-152.  coords: (bs, h, w, 2) xy.  ``out``/``ld_out`` let the
        forward loop write into its padded (zero-tailed) feature buffer."""
        coords = _dev.to_device(coords)
        bs, h, w, _ = self.fmap1.shape
        if tuple(coords.shape) != (bs, h, w, 2):
            raise ValueError(f'coords must have shape {(bs, h, w, 2)}, got {tuple(coords.shape)}')
        d = 2 * self.radius + 1
        nch = self.num_levels * d * d
        if out is None:
            ld_out = nch
            out = torch.empty((bs, h, w, nch), device=coords.device, dtype=torch.float32)
        lib = _dev.lib()
        if self.alternate:
            check(lib.raft_corr_lookup_ondemand_f32(_dev.ptr(self.fmap1), _dev.ptr(self._f2pyr), _dev.ptr(coords),
                                                    bs, h, w, self.fmap1.shape[-1], self.num_levels, self.radius,
                                                    _dev.ptr(out), ld_out, _dev.stream_ptr()), 'corr_lookup_ondemand')
        else:
            check(lib.raft_corr_lookup_f32(_dev.ptr(self._pyr), self._off, _dev.ptr(coords), bs, h, w,
                                           self.num_levels, self.radius, _dev.ptr(out), ld_out,
                                           _dev.stream_ptr()), 'corr_lookup')
        return _dev.wrap(out)
