"""Backward kernels of the training step, first slice (reference ``tf_raft/model.py:126-144`` differentiates the forward
pass with ``tf.GradientTape``; BASELINE config 5).  What exists: the gradient of ``sequence_loss`` w.r.t. every flow
prediction, the backward of the pyramid lookup (``CorrBlock.retrieve`` + ``bilinear_sampler``) w.r.t. the coordinates and
the correlation pyramid, and the backward of one Keras ``Conv2D`` (+ relu) w.r.t. input, kernel and bias -- each a HIP
kernel behind the C ABI (``csrc/backward.hip``), deterministic, checked against torch autograd of the CPU oracle in
``tests/test_gpu_backward.py``.  The later sections of this module add the rest of ``train_step``: one whole update block,
the loop and its backward through time, the volume build, the encoders with their norms.  torch allocates, slices and
concatenates here; every multiply-add is a HIP kernel.
"""
from __future__ import annotations

import os

import ctypes as C

import numpy as np
import torch

from . import _dev, packing
from ._ffi import check


def _f32(t):
    if type(t) is torch.Tensor and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous():
        return t
    return _dev.to_device(t).as_subclass(torch.Tensor).to(torch.float32).contiguous()


# Packed-weight cache of one training step: the functional path packs Keras-layout kernels for the generic convolution
# kernels; within a step the same kernel is used by every loop iteration (forward, and flipped for the input gradient).
# Keyed by the identity of the source array (kept alive by the entry) AND validated against a version: parameters may be
# device tensors that the optimizer and the batch-norm statistics update IN PLACE, so identity alone would hand back the
# packed copy of the old values.  The version of an entry is (torch's in-place counter of each source tensor, the module
# counter that ``bump_pack_version()`` advances); ``training.AdamW.apply_gradients`` and ``RAFT.train_step``'s moving-statistics
# update call ``bump_pack_version()`` after writing (HIP kernels behind raw pointers do not advance torch's counter), and
# ``RAFT.train_step`` still clears the cache at the start of every step to bound its size.  Any other raw-pointer writer of a
# parameter tensor has to call it too.
_PACK = {}
_PACK_VERSION = [0]


def clear_pack_cache():
    _PACK.clear()


def bump_pack_version():
    """Call after parameters were modified in place outside torch (optimizer / statistics kernels): invalidates every entry."""
    _PACK_VERSION[0] += 1


def _version_of(a):
    return a._version if isinstance(a, torch.Tensor) else 0


def _cached(arr, tag, make, also=None):
    """``make()`` memoised on the identity of ``arr`` (and of ``also``) and on their versions; both are kept alive by the entry."""
    key = (id(arr), id(also), tag)
    if len(_PACK) > 2048:                    # callers outside train_step never clear: keep the cache bounded
        _PACK.clear()
    ver = (_version_of(arr), _version_of(also), _PACK_VERSION[0])
    hit = _PACK.get(key)
    if hit is None or hit[0] is not arr or hit[1] is not also or hit[3] != ver:
        hit = (arr, also, make(), ver)
        _PACK[key] = hit
    return hit[2]


# Parameters may be NumPy arrays in Keras layout (the weight dictionaries of the models) or fp32 DEVICE tensors (the master
# copies ``train_step`` keeps: nothing of a training step then touches the host).  Packing for the convolution kernels is done
# where the parameter lives: ``packing`` on the host, the same index shuffles as torch views / copies on the device.
def _is_t(v):
    return isinstance(v, torch.Tensor)


def _is_dev_f32(v):
    """A torch tensor whose data_ptr() may be handed to a HIP kernel as ``const float *``: on the GPU, fp32, contiguous.  Host or
    other-dtype tensors take the tensor-op packing chain, which converts (or rejects) them in torch."""
    return isinstance(v, torch.Tensor) and v.is_cuda and v.dtype == torch.float32 and v.is_contiguous()


def _cat(parts, axis):
    return torch.cat(list(parts), dim=axis) if _is_t(parts[0]) else np.concatenate(list(parts), axis=axis)


def _params(weights, prefix):
    """Parameters under ``prefix``: device tensors are passed through AS THE SAME OBJECTS (the pack cache is keyed on
    identity), anything else becomes a float32 NumPy array."""
    return {k: (v if type(v) is torch.Tensor else (v.as_subclass(torch.Tensor) if _is_t(v) else np.asarray(v, dtype=np.float32)))
            for k, v in weights.items() if k.startswith(prefix)}


def _dev_param(v):
    """fp32 device tensor of a parameter (no copy when it already is one)."""
    return v.as_subclass(torch.Tensor) if _is_t(v) else _dev.to_device(np.ascontiguousarray(v, dtype=np.float32)).as_subclass(torch.Tensor)


def _pack_conv_any(kernel, bias, sources):
    """``packing.pack_conv`` -> (wp, bias, npad) as device tensors, computed on the device for device parameters."""
    if not _is_t(kernel):
        wp, b, npad = packing.pack_conv(kernel, bias if bias is not None else np.zeros(kernel.shape[3], np.float32), sources)
        return _dev.to_device(wp), _dev.to_device(b), npad
    kh, kw, cin, cout = kernel.shape
    kpad, npad = sum(cp for _, cp in sources), packing.round_up(cout, 64)
    full = torch.zeros((kh * kw, kpad, npad), device=kernel.device, dtype=torch.float32)
    flat = kernel.reshape(kh * kw, cin, cout)
    k_src = k_dst = 0
    for c, cp in sources:
        full[:, k_dst:k_dst + c, :cout] = flat[:, k_src:k_src + c, :]
        k_src += c
        k_dst += cp
    wp = full.reshape(kh * kw, kpad // 4, 4, npad).permute(0, 1, 3, 2).contiguous()
    b = torch.zeros((npad,), device=kernel.device, dtype=torch.float32)
    if bias is not None:
        b[:cout] = _dev_param(bias)
    return wp, b, npad


_G_CTYPES = {}


def _pack_train_device(kernel, bias, wino, dgrad, cpad):
    """``_pack_conv_any([_wino_transform_any]([_dgrad_kernel_any](kernel)), bias, [(K, cpad)])`` for a DEVICE kernel in ONE launch
    (``raft_pack_train_conv_f32``): the master weights change every step, and the tensor-op version of this chain was ~10 small
    kernels per layer and use.  ``wino``: None / '2d' / '1d' (F(4, 5)); returns (wp, bias, npad) like the chain it replaces."""
    import ctypes as C
    kh, kw, cin, cout = kernel.shape
    k_ch, n_ch = (cout, cin) if dgrad else (cin, cout)
    if cpad < k_ch or cpad % 4:
        raise ValueError(f'padded source width {cpad} does not hold {k_ch} channels')
    npad = packing.round_up(n_ch, 64)
    mode = {None: 0, '2d': 1, '1d': 2}[wino]
    taps = kh * kw if mode == 0 else (16 if mode == 1 else 8)
    gkey = {1: '_WINO_G', 2: '_WINO1D4_G'}.get(mode)
    if gkey is not None and gkey not in _G_CTYPES:
        g = np.ascontiguousarray(getattr(packing, gkey), dtype=np.float64).ravel()
        _G_CTYPES[gkey] = (C.c_double * g.size)(*g)
    kernel = kernel.contiguous()
    wp = torch.empty((taps, cpad // 4, npad, 4), device=kernel.device, dtype=torch.float32)
    b = torch.empty((npad,), device=kernel.device, dtype=torch.float32)
    bias_t = None if (bias is None or dgrad) else _dev_param(bias).contiguous()
    check(_dev.lib().raft_pack_train_conv_f32(_dev.ptr(kernel), _dev.ptr(bias_t) if bias_t is not None else None, kh, kw, cin, cout,
                                              1 if dgrad else 0, mode, _G_CTYPES.get(gkey), 8 if mode == 2 else 0, cpad, npad,
                                              _dev.ptr(wp), _dev.ptr(b), _dev.stream_ptr()), 'pack_train_conv')
    return wp, b, npad


# RAFT_TRAIN_PACK=torch restores the tensor-op packing chain (tests compare the two)
FUSED_TRAIN_PACK = os.environ.get('RAFT_TRAIN_PACK', 'hip') != 'torch'


# The training forward and the input gradients run the 3x3 / 1x5 / 5x1 layers on the Winograd kernels of the inference path
# (F(2x2, 3x3): 2.25x fewer multiplies, F(4, 5): 2.5x) -- fp32 throughout, deviation from the direct kernel ~1e-6 relative per
# layer (tests/test_gpu_kernels.py).  False = the direct kernels everywhere (the parity tests check both).
TRAIN_WINOGRAD = True


def _wino_kind(kh, kw):
    if not TRAIN_WINOGRAD:
        return None
    return '2d' if (kh, kw) == (3, 3) else ('1d' if (kh, kw) in ((1, 5), (5, 1)) else None)


_WINO_G_DEV = {}


def _wino_g(name, device):
    """The transform matrices G (float64) on the device, uploaded once (an upload per use would be a blocking copy per layer)."""
    key = (name, str(device))
    if key not in _WINO_G_DEV:
        _WINO_G_DEV[key] = torch.as_tensor(getattr(packing, name), dtype=torch.float64, device=device)
    return _WINO_G_DEV[key]


def _wino_transform_any(kernel):
    """G g G^T (3x3 -> 4x4 taps, F(2x2, 3x3)) or G' g (1x5 / 5x1 -> 8 taps, F(4, 5)) where the parameter lives; float64, one
    rounding (packing.winograd_kernel / winograd1d_kernel)."""
    kh, kw = kernel.shape[:2]
    if not _is_t(kernel):
        return packing.winograd_kernel(kernel) if (kh, kw) == (3, 3) else packing.winograd1d_kernel(kernel, 4)
    k = kernel.to(torch.float64)
    if (kh, kw) == (3, 3):
        g = _wino_g('_WINO_G', kernel.device)
        return torch.einsum('au,bv,uvio->abio', g, g, k).to(torch.float32).contiguous()
    g = _wino_g('_WINO1D4_G', kernel.device)
    return torch.einsum('tk,kio->tio', g, k[0] if kh == 1 else k[:, 0]).to(torch.float32)[:, None].contiguous()


def _conv_launch(x, cpad, kernel_shape, wino, wp_d, b_d, npad, nvalid, act, scale, out, what):
    """One stride-1 'same' convolution of the (B, H, W, cpad) tensor ``x`` through the direct or the Winograd entry point."""
    kh, kw = kernel_shape
    B, H, W, _ = x.shape
    lib = _dev.lib()
    if wino == '2d':
        rc = lib.raft_conv2d_winograd_f32(_dev.ptr(x), cpad, cpad, None, 0, 0, _dev.ptr(wp_d), _dev.ptr(b_d), B, H, W, npad, nvalid,
                                          act, float(scale), _dev.ptr(out), nvalid, _dev.stream_ptr())
    elif wino == '1d':
        rc = lib.raft_conv1d_winograd4_f32(_dev.ptr(x), cpad, cpad, None, 0, 0, _dev.ptr(wp_d), _dev.ptr(b_d), B, H, W, kh, kw, npad,
                                           nvalid, act, float(scale), _dev.ptr(out), nvalid, _dev.stream_ptr())
    else:
        rc = lib.raft_conv2d_f32(_dev.ptr(x), cpad, cpad, None, 0, 0, _dev.ptr(wp_d), _dev.ptr(b_d), B, H, W, kh, kw, npad, nvalid,
                                 act, float(scale), _dev.ptr(out), nvalid, _dev.stream_ptr())
    check(rc, what)


def _dgrad_kernel_any(kernel):
    """``packing.dgrad_kernel``: spatial flip, in / out channels transposed."""
    if not _is_t(kernel):
        return packing.dgrad_kernel(kernel)
    return kernel.flip(0, 1).permute(0, 1, 3, 2).contiguous()


def sequence_loss_grad(y_true, y_pred, gamma=0.8, max_flow=400, upstream=1.0):
    """d ``sequence_loss(y_true, y_pred)`` / d ``y_pred[i]`` for every i (reference losses.py:4-21): list of tensors shaped
    like the predictions."""
    flow_gt, valid = y_true
    flow_gt = _f32(flow_gt)
    valid = torch.as_tensor(np.asarray(valid) if not isinstance(valid, torch.Tensor) else valid)
    valid = (valid != 0).to(device=flow_gt.device, dtype=torch.uint8).contiguous()
    preds = torch.stack([_f32(p) for p in y_pred], dim=0).contiguous()
    n = preds.shape[0]
    if tuple(preds.shape[1:]) != tuple(flow_gt.shape) or tuple(valid.shape) != tuple(flow_gt.shape[:-1]):
        raise ValueError('predictions / flow_gt / valid shapes disagree')
    npix = flow_gt.numel() // 2
    out = torch.empty_like(preds)
    check(_dev.lib().raft_sequence_loss_grad_f32(_dev.ptr(flow_gt), _dev.ptr(valid), _dev.ptr(preds), 2 * npix, n, npix,
                                                 float(gamma), float(max_flow), float(upstream), _dev.ptr(out),
                                                 _dev.stream_ptr()), 'sequence_loss_grad')
    return [_dev.wrap(out[i]) for i in range(n)]


def corr_lookup_backward(corr_block, coords, d_out, d_pyramid=None, want_pyramid_grad=True):
    """Backward of ``corr_block.retrieve(coords)`` (reference corr.py:116-152) for the upstream gradient ``d_out``
    (bs, h, w, levels*(2r+1)^2).  Returns ``(d_coords, d_pyramid)``: ``d_coords`` (bs, h, w, 2) and the gradient w.r.t.
    the stored pyramid as ONE flat tensor in the library's tiled layout (``corr_block.untile_pyramid`` gives the
    reference's per-level ``(bs*h*w, h_l, w_l, 1)`` view).  Pass the previous ``d_pyramid`` to accumulate over the
    iterations of the loop."""
    if corr_block._pyr is None:
        raise ValueError('the on-demand CorrBlock stores no pyramid to differentiate')
    coords = _f32(coords)
    d_out = _f32(d_out)
    bs, h, w, _ = corr_block.fmap1.shape
    nch = corr_block.num_levels * (2 * corr_block.radius + 1) ** 2
    if tuple(coords.shape) != (bs, h, w, 2) or tuple(d_out.shape) != (bs, h, w, nch):
        raise ValueError(f'expected coords {(bs, h, w, 2)} and d_out {(bs, h, w, nch)}')
    d_coords = torch.empty_like(coords)
    if want_pyramid_grad and d_pyramid is None:
        d_pyramid = torch.zeros_like(corr_block._pyr)
    check(_dev.lib().raft_corr_lookup_backward_f32(
        _dev.ptr(corr_block._pyr), corr_block._off, _dev.ptr(coords), _dev.ptr(d_out), nch, bs, h, w,
        corr_block.num_levels, corr_block.radius, _dev.ptr(d_coords),
        _dev.ptr(d_pyramid) if want_pyramid_grad else None, _dev.stream_ptr()), 'corr_lookup_backward')
    return _dev.wrap(d_coords), d_pyramid


class WgradDefer:
    """Weight gradients of layers that are applied several times with the SAME weights (the update block over the iterations
    of the prediction loop, reference model.py:91-109), evaluated ONCE at the end: ``add`` keeps the (input, masked upstream
    gradient) pair of an application, ``finish`` runs one ``raft_conv2d_wgrad_multi_f32`` per layer over all of them -- a single
    pixel reduction instead of one launch + one second-stage sum + one accumulation per application (12 per layer and step at
    the reference's training shape).  Costs the memory of the kept upstream gradients (the inputs are on the tape anyway)."""
    MAX_SEG = 32

    def __init__(self):
        self.jobs = {}

    def add(self, key, x, dy, kshape, trim=None):
        job = self.jobs.setdefault(key, dict(kshape=tuple(kshape), xs=[], dys=[], trim=trim))
        if job['kshape'] != tuple(kshape) or (job['xs'] and job['xs'][0].shape != x.shape):
            raise ValueError(f'{key}: applications of one layer must share their geometry')
        job['xs'].append(x)
        job['dys'].append(dy)

    def finish(self):
        """{key: (d_kernel, d_bias)} -- sums over every application added under the key."""
        lib = _dev.lib()
        out = {}
        for key, job in self.jobs.items():
            kh, kw, cin, cout = job['kshape']
            B, H, W, _ = job['xs'][0].shape
            dev = job['xs'][0].device
            dk = torch.zeros((kh, kw, cin, cout), device=dev, dtype=torch.float32)
            db = torch.zeros((cout,), device=dev, dtype=torch.float32)
            for lo in range(0, len(job['xs']), self.MAX_SEG):
                xs, dys = job['xs'][lo:lo + self.MAX_SEG], job['dys'][lo:lo + self.MAX_SEG]
                n = len(xs)
                ws = torch.empty((int(lib.raft_conv2d_wgrad_workspace_floats(cin, cout, B * n, H, W, kh, kw)),), device=dev,
                                 dtype=torch.float32)
                px = (C.c_void_p * n)(*[_dev.ptr(t) for t in xs])
                pd = (C.c_void_p * n)(*[_dev.ptr(t) for t in dys])
                first = lo == 0
                dk_i = dk if first else torch.empty_like(dk)
                db_i = db if first else torch.empty_like(db)
                check(lib.raft_conv2d_wgrad_multi_f32(px, pd, n, cin, cin, cout, cout, B, H, W, kh, kw, _dev.ptr(dk_i), _dev.ptr(db_i),
                                                      _dev.ptr(ws), _dev.stream_ptr()), 'conv2d_wgrad_multi')
                if not first:
                    dk, db = _axpby(1.0, dk, 1.0, dk_i), _axpby(1.0, db, 1.0, db_i)
            if job['trim'] is not None:
                ci, co = job['trim']
                dk, db = dk[:, :, :ci, :co].contiguous(), db[:co].contiguous()
            out[key] = (dk, db)
        self.jobs = {}
        return out


def conv2d_backward(x, kernel, dy, y=None, defer=None):
    """Backward of ``y = [relu](conv2d(x, kernel) + bias)`` (Keras Conv2D, stride 1, 'same'; reference update.py:10-11,
    91-95, 138-140).  ``x`` (B, H, W, Cin), ``kernel`` (kh, kw, Cin, Cout) NumPy in Keras layout, ``dy`` (B, H, W, Cout);
    pass the forward output ``y`` to apply the relu mask first.  Returns ``(dx, d_kernel, d_bias)`` device tensors.
    Cin and Cout must be multiples of 4 (every layer of the update block except the flow-carrying ones).
    ``defer = (WgradDefer, key[, trim])``: the kernel / bias gradient is left to ``WgradDefer.finish`` (None returned here)."""
    x = _f32(x)
    dy = _f32(dy)
    kernel = (kernel if type(kernel) is torch.Tensor else kernel.as_subclass(torch.Tensor)) if _is_t(kernel) else np.asarray(kernel, dtype=np.float32)
    kh, kw, cin, cout = kernel.shape
    B, H, W, _ = x.shape
    if x.shape[-1] != cin or tuple(dy.shape) != (B, H, W, cout):
        raise ValueError(f'x {tuple(x.shape)} / dy {tuple(dy.shape)} do not match kernel {kernel.shape}')
    lib = _dev.lib()
    if y is not None:
        y = _f32(y)
        masked = torch.empty_like(dy)
        check(lib.raft_relu_backward_f32(_dev.ptr(y), _dev.ptr(dy), _dev.ptr(masked), dy.numel(), _dev.stream_ptr()),
              'relu_backward')
        dy = masked
    # kernel / bias gradient
    if defer is not None:
        x = x.contiguous()
        dy = dy.contiguous()
        defer[0].add(defer[1], x, dy, (kh, kw, cin, cout), defer[2] if len(defer) > 2 else None)
        d_kernel = d_bias = None
    else:
        ws = torch.empty((int(lib.raft_conv2d_wgrad_workspace_floats(cin, cout, B, H, W, kh, kw)),), device=x.device,
                         dtype=torch.float32)
        d_kernel = torch.empty((kh, kw, cin, cout), device=x.device, dtype=torch.float32)
        d_bias = torch.empty((cout,), device=x.device, dtype=torch.float32)
        check(lib.raft_conv2d_wgrad_f32(_dev.ptr(x), cin, cin, _dev.ptr(dy), cout, cout, B, H, W, kh, kw, _dev.ptr(d_kernel),
                                        _dev.ptr(d_bias), _dev.ptr(ws), _dev.stream_ptr()), 'conv2d_wgrad')
    # input gradient: the forward convolution of dy with the flipped, transposed kernel
    cpad = packing.round_up(cout, 32)

    wino = _wino_kind(kh, kw)

    def make_d():
        if FUSED_TRAIN_PACK and _is_dev_f32(kernel):
            return _pack_train_device(kernel, None, wino, True, cpad)
        kd = _dgrad_kernel_any(kernel)
        return _pack_conv_any(_wino_transform_any(kd) if wino else kd, None, [(cout, cpad)])
    wp_d, b_d, npad = _cached(kernel, 'dgrad' + (wino or ''), make_d)
    dyp = dy
    if cpad != cout:
        dyp = torch.zeros((B, H, W, cpad), device=x.device, dtype=torch.float32)
        dyp[..., :cout] = dy
    dx = torch.empty((B, H, W, cin), device=x.device, dtype=torch.float32)
    _conv_launch(dyp, cpad, (kh, kw), wino, wp_d, b_d, npad, cin, 0, 1.0, dx, 'conv2d dgrad')
    if defer is not None:
        return _dev.wrap(dx), None, None
    return _dev.wrap(dx), _dev.wrap(d_kernel), _dev.wrap(d_bias)


# ------------------------------------------------------------------------------------------------------------------------
# second slice: one BasicUpdateBlock call (reference update.py:128-153) forward with saved activations + its backward
# ------------------------------------------------------------------------------------------------------------------------

def _conv_fwd(x, kernel, bias, act=0, scale=1.0):
    """``[relu](conv2d(x, kernel) + bias) * scale`` through ``raft_conv2d_f32`` (stride 1, 'same'); x (B, H, W, Cin)."""
    kernel = (kernel if type(kernel) is torch.Tensor else kernel.as_subclass(torch.Tensor)) if _is_t(kernel) else np.asarray(kernel, dtype=np.float32)
    kh, kw, cin, cout = kernel.shape
    B, H, W, c = x.shape
    cpad = packing.round_up(cin, 32)
    wino = _wino_kind(kh, kw)
    def make():
        if FUSED_TRAIN_PACK and _is_dev_f32(kernel) and (bias is None or _is_dev_f32(bias)):
            return _pack_train_device(kernel, bias, wino, False, cpad)
        return _pack_conv_any(_wino_transform_any(kernel) if wino else kernel, bias, [(cin, cpad)])
    wp_d, b_d, npad = _cached(kernel, 'fwd' + (wino or ''), make, also=bias) if isinstance(bias, (np.ndarray, torch.Tensor)) else make()
    xp = x
    if cpad != c:
        xp = torch.zeros((B, H, W, cpad), device=x.device, dtype=torch.float32)
        xp[..., :c] = x
    out = torch.empty((B, H, W, cout), device=x.device, dtype=torch.float32)
    _conv_launch(xp, cpad, (kh, kw), wino, wp_d, b_d, npad, cout, act, scale, out, 'conv2d')
    return out


def _axpby(alpha, a, beta=0.0, b=None):
    out = torch.empty_like(a)
    check(_dev.lib().raft_axpby_f32(float(alpha), _dev.ptr(a), float(beta), _dev.ptr(b) if b is not None else None, _dev.ptr(out),
                                    a.numel(), _dev.stream_ptr()), 'axpby')
    return out


def _conv_bwd(x, kernel, dy, y=None, defer=None):
    """``conv2d_backward`` for any channel counts: pads Cin / Cout to multiples of 4 around the kernels and trims.
    ``defer = (WgradDefer, key)``: see ``conv2d_backward``."""
    kernel = (kernel if type(kernel) is torch.Tensor else kernel.as_subclass(torch.Tensor)) if _is_t(kernel) else np.asarray(kernel, dtype=np.float32)
    kh, kw, cin, cout = kernel.shape
    ci4, co4 = packing.round_up(cin, 4), packing.round_up(cout, 4)
    if ci4 != cin or co4 != cout:
        def pad():
            kp_ = (torch.zeros((kh, kw, ci4, co4), device=kernel.device, dtype=torch.float32) if _is_t(kernel)
                   else np.zeros((kh, kw, ci4, co4), np.float32))
            kp_[:, :, :cin, :cout] = kernel
            return kp_
        kp = _cached(kernel, 'pad4', pad)
        B, H, W, _ = x.shape
        xp = torch.zeros((B, H, W, ci4), device=x.device, dtype=torch.float32)
        xp[..., :cin] = x
        dyp = torch.zeros((B, H, W, co4), device=x.device, dtype=torch.float32)
        dyp[..., :cout] = dy
        yp = None
        if y is not None:
            yp = torch.zeros((B, H, W, co4), device=x.device, dtype=torch.float32)
            yp[..., :cout] = y
        dx, dk, db = conv2d_backward(xp, kp, dyp, y=yp, defer=None if defer is None else (defer[0], defer[1], (cin, cout)))
        t = lambda v: v.as_subclass(torch.Tensor)
        if defer is not None:
            return t(dx)[..., :cin].contiguous(), None, None
        return t(dx)[..., :cin].contiguous(), t(dk)[:, :, :cin, :cout].contiguous(), t(db)[:cout].contiguous()
    dx, dk, db = conv2d_backward(x, kernel, dy, y=y, defer=defer)
    if defer is not None:
        return dx.as_subclass(torch.Tensor), None, None
    return dx.as_subclass(torch.Tensor), dk.as_subclass(torch.Tensor), db.as_subclass(torch.Tensor)


_UB = {
    # reference update.py:128-153 (BasicUpdateBlock) / 109-125 (SmallUpdateBlock)
    'raft': dict(hdim=128, cdim=128, convc2=True, cor=192, flo=64, cf1=128, mot=126, gru=('1', '2'), mask=True),
    'small': dict(hdim=96, cdim=64, convc2=False, cor=96, flo=32, cf1=64, mot=80, gru=('',), mask=False),
}


def update_block_forward(weights, net, inp, corr, flow, prefix='update_block', variant='raft'):
    """``BasicUpdateBlock`` / ``SmallUpdateBlock`` ``([net, inp, corr, flow])`` (reference update.py:143-153 / 118-125) in
    TRAINING form: every layer a generic HIP convolution with a linear / relu epilogue, the GRU gates as separate kernels,
    every activation the backward needs kept.  Returns ``(net, mask, delta_flow, saved)`` (``mask`` is None for the small
    block).  (The inference path fuses gates and branches into its kernels and keeps nothing; both compute the same
    function: ``tests/test_gpu_backward.py`` checks this one against the oracle.)"""
    lib = _dev.lib()
    p, cfg = prefix, _UB[variant]
    hd = cfg['hdim']
    w = _params(weights, p)
    net, inp, corr, flow = (_f32(t) for t in (net, inp, corr, flow))
    B, H, W, _ = net.shape
    M = B * H * W
    s = {'net0': net, 'inp': inp, 'corr': corr, 'flow': flow, 'variant': variant}
    conv = lambda name, x, act=0, scale=1.0: _conv_fwd(x, w[f'{p}/{name}/kernel'], w[f'{p}/{name}/bias'], act, scale)
    s['cor1'] = conv('encoder/convc1', corr, 1)
    s['cor2'] = conv('encoder/convc2', s['cor1'], 1) if cfg['convc2'] else s['cor1']
    cf1 = cfg['cf1']
    k7 = _cached(w[f'{p}/encoder/convf1/kernel'], 'k7', lambda: _dev_param(w[f'{p}/encoder/convf1/kernel']).reshape(98, cf1).contiguous())
    b7 = _cached(w[f'{p}/encoder/convf1/bias'], 'b7', lambda: _dev_param(w[f'{p}/encoder/convf1/bias']))
    s['flo1'] = torch.empty((B, H, W, cf1), device=net.device, dtype=torch.float32)
    check(lib.raft_conv7x7_c2_f32(_dev.ptr(flow), _dev.ptr(k7), _dev.ptr(b7), cf1, B, H, W, _dev.ptr(s['flo1']), cf1,
                                  _dev.stream_ptr()), 'conv7x7_c2')
    s['flo2'] = conv('encoder/convf2', s['flo1'], 1)
    s['corflo'] = torch.cat([s['cor2'], s['flo2']], dim=-1)                       # update.py:104 / 83
    s['mot'] = conv('encoder/conv', s['corflo'], 1)
    x = torch.cat([inp, s['mot'], flow], dim=-1).contiguous()                      # update.py:106, 146: [inp | motion | flow]
    s['x'] = x
    h = net
    for g in cfg['gru']:                                                           # update.py:51-67 / 26-35
        kzr = _cached(w[f'{p}/gru/convz{g}/kernel'], 'kzr', lambda: _cat(
            [w[f'{p}/gru/convz{g}/kernel'], w[f'{p}/gru/convr{g}/kernel']], 3), also=w[f'{p}/gru/convr{g}/kernel'])
        bzr = _cached(w[f'{p}/gru/convz{g}/bias'], 'bzr', lambda: _cat(
            [w[f'{p}/gru/convz{g}/bias'], w[f'{p}/gru/convr{g}/bias']], 0), also=w[f'{p}/gru/convr{g}/bias'])
        hx = torch.cat([h, x], dim=-1).contiguous()
        a_zr = _conv_fwd(hx, kzr, bzr)
        z, r, rh = (torch.empty_like(h) for _ in range(3))
        check(lib.raft_gru_gate_zr_f32(_dev.ptr(a_zr), _dev.ptr(h), hd, M, _dev.ptr(z), _dev.ptr(r), _dev.ptr(rh),
                                       _dev.stream_ptr()), 'gate_zr')
        rhx = torch.cat([rh, x], dim=-1).contiguous()
        a_q = conv(f'gru/convq{g}', rhx)
        q, hn = torch.empty_like(h), torch.empty_like(h)
        check(lib.raft_gru_gate_q_f32(_dev.ptr(a_q), _dev.ptr(z), _dev.ptr(h), h.numel(), _dev.ptr(q), _dev.ptr(hn),
                                      _dev.stream_ptr()), 'gate_q')
        s[f'h_in{g}'], s[f'hx{g}'], s[f'rhx{g}'], s[f'z{g}'], s[f'r{g}'], s[f'q{g}'] = h, hx, rhx, z, r, q
        h = hn
    s['net'] = h
    s['fh'] = conv('flow_head/conv1', h, 1)
    delta = conv('flow_head/conv2', s['fh'])
    mask = None
    if cfg['mask']:
        s['m0'] = conv('mask/0', h, 1)
        mask = _dev.wrap(conv('mask/2', s['m0'], 0, 0.25))                         # update.py:152
    return _dev.wrap(h), mask, _dev.wrap(delta), s


def basic_update_block_forward(weights, net, inp, corr, flow, prefix='update_block'):
    return update_block_forward(weights, net, inp, corr, flow, prefix, 'raft')


def update_block_backward(weights, saved, d_net, d_mask, d_delta, prefix='update_block', defer=None):
    """Backward of ``update_block_forward``: upstream gradients of its outputs (``d_mask`` None for the small block) ->
    gradients w.r.t. the four inputs (``net``, ``inp``, ``corr``, ``flow``) and w.r.t. every kernel and bias of the block
    (dict under the weight names).  Every arithmetic step is a HIP kernel (convolution dgrad / wgrad, gate and relu backward,
    axpby); torch only concatenates, slices and allocates.
    ``defer`` (a ``WgradDefer``): the kernel / bias gradients of the block's generic convolutions are left to it
    (``deferred_update_grads`` turns its result into the same dict); only convf1's are returned here."""
    lib = _dev.lib()
    p = prefix
    s = saved
    cfg = _UB[s['variant']]
    hd, cd, mot = cfg['hdim'], cfg['cdim'], cfg['mot']
    w = _params(weights, p)
    d_net, d_delta = _f32(d_net), _f32(d_delta)
    grads = {}
    B, H, W, _ = d_net.shape
    n_h = d_net.numel()

    def conv_b(name, x, dy, y=None, kernel=None, key=None):
        k = w[f'{p}/{name}/kernel'] if kernel is None else kernel
        dx, dk, db = _conv_bwd(x, k, dy, y, defer=None if defer is None else (defer, key if key is not None else f'{p}/{name}'))
        if kernel is None and defer is None:
            grads[f'{p}/{name}/kernel'], grads[f'{p}/{name}/bias'] = dk, db
        return dx, dk, db

    dh = d_net
    if cfg['mask']:   # mask = 0.25 * mask.2(relu(mask.0(net)))                   update.py:137-141, 152
        dm = _axpby(0.25, _f32(d_mask))
        dm0, _, _ = conv_b('mask/2', s['m0'], dm)
        dh_a, _, _ = conv_b('mask/0', s['net'], dm0, y=s['m0'])
        dh = _axpby(1.0, dh, 1.0, dh_a)
    # flow head: delta = conv2(relu(conv1(net)))                                  update.py:13-14
    dfh, _, _ = conv_b('flow_head/conv2', s['fh'], d_delta)
    dh_b, _, _ = conv_b('flow_head/conv1', s['net'], dfh, y=s['fh'])
    dh = _axpby(1.0, dh, 1.0, dh_b)
    dx_total = None
    for g in reversed(cfg['gru']):                                                 # SepConvGRU: vertical pass first
        h_in, z, r, q = s[f'h_in{g}'], s[f'z{g}'], s[f'r{g}'], s[f'q{g}']
        dz_pre, dq_pre, dh_in = (torch.empty_like(h_in) for _ in range(3))
        check(lib.raft_gru_gate_q_backward_f32(_dev.ptr(dh), _dev.ptr(z), _dev.ptr(q), _dev.ptr(h_in), n_h, _dev.ptr(dz_pre),
                                               _dev.ptr(dq_pre), _dev.ptr(dh_in), _dev.stream_ptr()), 'gate_q_backward')
        d_rhx, _, _ = conv_b(f'gru/convq{g}', s[f'rhx{g}'], dq_pre)
        d_rh = d_rhx[..., :hd].contiguous()
        dx_q = d_rhx[..., hd:].contiguous()
        dr_pre = torch.empty_like(h_in)
        check(lib.raft_gru_gate_r_backward_f32(_dev.ptr(d_rh), _dev.ptr(r), _dev.ptr(h_in), n_h, _dev.ptr(dr_pre), _dev.ptr(dh_in),
                                               _dev.stream_ptr()), 'gate_r_backward')
        kzr = _cached(w[f'{p}/gru/convz{g}/kernel'], 'kzr', lambda: _cat(
            [w[f'{p}/gru/convz{g}/kernel'], w[f'{p}/gru/convr{g}/kernel']], 3), also=w[f'{p}/gru/convr{g}/kernel'])
        d_zr = torch.cat([dz_pre, dr_pre], dim=-1).contiguous()
        d_hx, dk, db = conv_b(None, s[f'hx{g}'], d_zr, kernel=kzr, key=('zr', p, g, hd))
        if defer is None:
            grads[f'{p}/gru/convz{g}/kernel'], grads[f'{p}/gru/convr{g}/kernel'] = dk[..., :hd].contiguous(), dk[..., hd:].contiguous()
            grads[f'{p}/gru/convz{g}/bias'], grads[f'{p}/gru/convr{g}/bias'] = db[:hd].contiguous(), db[hd:].contiguous()
        dh = _axpby(1.0, dh_in, 1.0, d_hx[..., :hd].contiguous())
        dx_g = _axpby(1.0, dx_q, 1.0, d_hx[..., hd:].contiguous())
        dx_total = dx_g if dx_total is None else _axpby(1.0, dx_total, 1.0, dx_g)
    d_inp = dx_total[..., :cd].contiguous()
    d_mot = dx_total[..., cd:cd + mot].contiguous()
    d_flow_x = dx_total[..., cd + mot:cd + mot + 2].contiguous()
    # motion encoder                                                               update.py:97-106 / 78-85
    d_corflo, _, _ = conv_b('encoder/conv', s['corflo'], d_mot, y=s['mot'])
    d_cor2 = d_corflo[..., :cfg['cor']].contiguous()
    d_flo2 = d_corflo[..., cfg['cor']:].contiguous()
    if cfg['convc2']:
        d_cor1, _, _ = conv_b('encoder/convc2', s['cor1'], d_cor2, y=s['cor2'])
    else:
        d_cor1 = d_cor2
    d_corr, _, _ = conv_b('encoder/convc1', s['corr'], d_cor1, y=s['cor1'])
    d_flo1, _, _ = conv_b('encoder/convf2', s['flo1'], d_flo2, y=s['flo2'])
    masked = torch.empty_like(d_flo1)
    check(lib.raft_relu_backward_f32(_dev.ptr(s['flo1']), _dev.ptr(d_flo1), _dev.ptr(masked), masked.numel(), _dev.stream_ptr()),
          'relu_backward')
    cf1 = cfg['cf1']
    k7 = _cached(w[f'{p}/encoder/convf1/kernel'], 'k7', lambda: _dev_param(w[f'{p}/encoder/convf1/kernel']).reshape(98, cf1).contiguous())
    d_flow_f = torch.empty((B, H, W, 2), device=d_net.device, dtype=torch.float32)
    dk7 = torch.empty((98, cf1), device=d_net.device, dtype=torch.float32)
    db7 = torch.empty((cf1,), device=d_net.device, dtype=torch.float32)
    ws = torch.empty((int(lib.raft_conv7x7_c2_wgrad_workspace_floats(cf1)),), device=d_net.device, dtype=torch.float32)
    check(lib.raft_conv7x7_c2_backward_f32(_dev.ptr(s['flow']), _dev.ptr(masked), cf1, _dev.ptr(k7), cf1, B, H, W, _dev.ptr(d_flow_f),
                                           _dev.ptr(dk7), _dev.ptr(db7), _dev.ptr(ws), _dev.stream_ptr()), 'conv7x7_c2_backward')
    grads[f'{p}/encoder/convf1/kernel'], grads[f'{p}/encoder/convf1/bias'] = dk7.view(7, 7, 2, cf1), db7
    d_flow = _axpby(1.0, d_flow_x, 1.0, d_flow_f)
    return {'net': _dev.wrap(dh), 'inp': _dev.wrap(d_inp), 'corr': _dev.wrap(d_corr), 'flow': _dev.wrap(d_flow)}, \
        {k: _dev.wrap(v) for k, v in grads.items()}


def deferred_update_grads(defer):
    """``WgradDefer.finish()`` of an update block's layers as the weight-name dict ``update_block_backward`` returns (the fused
    z | r gradient split back into convz / convr)."""
    grads = {}
    for key, (dk, db) in defer.finish().items():
        if isinstance(key, tuple) and key[0] == 'zr':
            _, p, g, hd = key
            grads[f'{p}/gru/convz{g}/kernel'], grads[f'{p}/gru/convr{g}/kernel'] = dk[..., :hd].contiguous(), dk[..., hd:].contiguous()
            grads[f'{p}/gru/convz{g}/bias'], grads[f'{p}/gru/convr{g}/bias'] = db[:hd].contiguous(), db[hd:].contiguous()
        else:
            grads[f'{key}/kernel'], grads[f'{key}/bias'] = dk, db
    return grads


def basic_update_block_backward(weights, saved, d_net, d_mask, d_delta, prefix='update_block'):
    return update_block_backward(weights, saved, d_net, d_mask, d_delta, prefix)


# ------------------------------------------------------------------------------------------------------------------------
# third slice: the prediction loop of RAFT.call in training form and its backward through time (reference model.py:91-109)
# ------------------------------------------------------------------------------------------------------------------------

def upsample_flow_backward(flow, mask, d_up):
    """Backward of ``RAFT.upsample_flow(flow, mask)`` (reference model.py:39-66): ``(d_flow, d_mask)``."""
    flow, mask, d_up = _f32(flow), _f32(mask), _f32(d_up)
    B, h, w, _ = flow.shape
    lib = _dev.lib()
    d_flow, d_mask = torch.empty_like(flow), torch.empty_like(mask)
    ws = torch.empty((int(lib.raft_upsample_convex_backward_workspace_floats(B, h, w)),), device=flow.device, dtype=torch.float32)
    check(lib.raft_upsample_convex_backward_f32(_dev.ptr(flow), _dev.ptr(mask), _dev.ptr(d_up), B, h, w, _dev.ptr(d_flow),
                                                _dev.ptr(d_mask), _dev.ptr(ws), _dev.stream_ptr()), 'upsample_convex_backward')
    return d_flow, d_mask


def upflow8_backward(d_up, B, h, w):
    """Backward of ``upflow8`` (reference corr.py:93-96: 8 * half-pixel bilinear resize by 8): ``d_flow`` (B, h, w, 2)."""
    d_up = _f32(d_up)
    out = torch.empty((B, h, w, 2), device=d_up.device, dtype=torch.float32)
    check(_dev.lib().raft_upflow8_backward_f32(_dev.ptr(d_up), B, h, w, _dev.ptr(out), _dev.stream_ptr()), 'upflow8_backward')
    return out


def to_bf16(t):
    """bf16 copy of an fp32 device tensor (round to nearest even, ``raft_f32_to_bf16``)."""
    t = t.as_subclass(torch.Tensor).contiguous()
    out = torch.empty(t.shape, device=t.device, dtype=torch.bfloat16)
    check(_dev.lib().raft_f32_to_bf16(_dev.ptr(t), _dev.ptr(out), t.numel(), _dev.stream_ptr()), 'f32_to_bf16')
    return out


def from_bf16(t):
    out = torch.empty(t.shape, device=t.device, dtype=torch.float32)
    check(_dev.lib().raft_bf16_to_f32(_dev.ptr(t), _dev.ptr(out), t.numel(), _dev.stream_ptr()), 'bf16_to_f32')
    return out


_TAPE_KEEP_F32 = 4096      # tensors smaller than this many elements per pixel-batch stay fp32 (coordinates, flow: 2 channels)


def _tape_narrow(saved):
    """The saved activations of one iteration with every large fp32 tensor stored as bf16 (half the tape memory)."""
    out = {}
    for k, v in saved.items():
        if _is_t(v) and v.dtype == torch.float32 and v.dim() == 4 and v.shape[-1] >= 8:
            out[k] = ('bf16', to_bf16(v))
        else:
            out[k] = v
    return out


def _tape_widen(saved):
    return {k: (from_bf16(v[1]) if isinstance(v, tuple) and v[0] == 'bf16' else v) for k, v in saved.items()}


def dropout_forward(x, rate, seed):
    """Keras Dropout, training mode (reference extractor.py:109-111): ``(y, mask)``."""
    x = x.as_subclass(torch.Tensor).contiguous()
    y = torch.empty_like(x)
    mask = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    check(_dev.lib().raft_dropout_f32(_dev.ptr(x), x.numel(), float(rate), int(seed), _dev.ptr(y), _dev.ptr(mask), _dev.stream_ptr()),
          'dropout')
    return y, (mask, float(rate))


def dropout_backward(dy, mask):
    dy = dy.as_subclass(torch.Tensor).contiguous()
    dx = torch.empty_like(dy)
    check(_dev.lib().raft_dropout_backward_f32(_dev.ptr(dy), _dev.ptr(mask[0]), dy.numel(), mask[1], _dev.ptr(dx), _dev.stream_ptr()),
          'dropout_backward')
    return dx


def loop_forward(weights, corr_block, net0, inp, iters, prefix='update_block', variant='raft', tape_dtype='f32'):
    """The ``for i in range(iters)`` loop of ``RAFT.call`` (reference model.py:91-109) in training form, from the correlation
    volume, ``net0 = tanh(.)`` and ``inp = relu(.)`` on: lookup -> update block -> coords1 += delta -> convex upsampling.
    Returns ``(flow_predictions, tape)``; the tape holds what ``loop_backward`` needs -- with ``tape_dtype='bf16'`` every large
    activation of it is STORED as bf16 and widened again by the backward (all arithmetic stays fp32)."""
    from .layers.corr import coords_grid
    if tape_dtype not in ('f32', 'bf16'):
        raise ValueError(f"tape_dtype must be 'f32' or 'bf16', got {tape_dtype!r}")
    net, inp = _f32(net0), _f32(inp)
    B, h, w, _ = net.shape
    coords0 = coords_grid(B, h, w).as_subclass(torch.Tensor)
    coords1 = coords0.clone()
    preds, tape = [], []
    lib = _dev.lib()
    for _ in range(iters):
        corr = corr_block.retrieve(coords1).as_subclass(torch.Tensor)
        flow = _axpby(1.0, coords1, -1.0, coords0)
        net_n, mask, delta, saved = update_block_forward(weights, net, inp, corr, flow, prefix, variant)
        coords_n = _axpby(1.0, coords1, 1.0, delta.as_subclass(torch.Tensor))
        flow_n = _axpby(1.0, coords_n, -1.0, coords0)
        up = torch.empty((B, 8 * h, 8 * w, 2), device=net.device, dtype=torch.float32)
        if mask is not None:
            mask = mask.as_subclass(torch.Tensor)
            check(lib.raft_upsample_convex_f32(_dev.ptr(flow_n), _dev.ptr(mask), B, h, w, _dev.ptr(up), _dev.stream_ptr()),
                  'upsample_convex')
        else:                                                                       # SmallRAFT: model.py:223
            check(lib.raft_upflow8_f32(_dev.ptr(flow_n), B, h, w, _dev.ptr(up), _dev.stream_ptr()), 'upflow8')
        if tape_dtype == 'bf16':
            saved = _tape_narrow(saved)
            mask_t = ('bf16', to_bf16(mask)) if mask is not None else None
        else:
            mask_t = mask
        tape.append(dict(coords1=coords1, saved=saved, flow_n=flow_n, mask=mask_t))
        preds.append(_dev.wrap(up))
        net, coords1 = net_n.as_subclass(torch.Tensor), coords_n
    return preds, tape


def loop_backward(weights, corr_block, tape, d_preds, prefix='update_block', defer_wgrad=None):
    """Backward through time of ``loop_forward`` for the upstream gradients ``d_preds`` of the flow predictions (e.g.
    ``sequence_loss_grad``).  The reference does not stop the gradient at ``coords1`` (model.py:93-106), so it flows
    through the lookup coordinates of every iteration.  Returns ``(d_net0, d_inp, d_pyramid, weight_grads)``.
    ``defer_wgrad`` (default: on for an fp32 tape, off for a bf16 one, whose point is memory): the weight gradients of the
    update block's convolutions are evaluated once over all iterations (``WgradDefer``) instead of per iteration."""
    iters = len(tape)
    if defer_wgrad is None:
        defer_wgrad = not any(isinstance(v, tuple) for v in tape[0]['saved'].values()) if iters else False
    defer = WgradDefer() if defer_wgrad else None
    d_c = None          # gradient w.r.t. coords1 after the current iteration
    d_net = None
    d_inp = None
    d_pyr = None
    wg = None
    for i in reversed(range(iters)):
        t = tape[i]
        saved = _tape_widen(t['saved'])
        mask_i = from_bf16(t['mask'][1]) if isinstance(t['mask'], tuple) else t['mask']
        if mask_i is not None:
            d_flowlow, d_mask = upsample_flow_backward(t['flow_n'], mask_i, d_preds[i])
        else:
            B_, h_, w_, _ = t['flow_n'].shape
            d_flowlow, d_mask = upflow8_backward(d_preds[i], B_, h_, w_), None
        d_c = d_flowlow if d_c is None else _axpby(1.0, d_c, 1.0, d_flowlow)
        if d_net is None:
            d_net = torch.zeros_like(saved['net'])
        din, dw = update_block_backward(weights, saved, d_net, d_mask, d_c, prefix, defer=defer)
        d_coords, d_pyr = corr_lookup_backward(corr_block, t['coords1'], din['corr'], d_pyramid=d_pyr)
        d_c = _axpby(1.0, d_c, 1.0, din['flow'].as_subclass(torch.Tensor))
        d_c = _axpby(1.0, d_c, 1.0, d_coords.as_subclass(torch.Tensor))
        d_net = din['net'].as_subclass(torch.Tensor)
        d_inp = din['inp'].as_subclass(torch.Tensor) if d_inp is None else _axpby(1.0, d_inp, 1.0, din['inp'].as_subclass(torch.Tensor))
        if wg is None:
            wg = {k: v.as_subclass(torch.Tensor) for k, v in dw.items()}
        else:
            wg = {k: _axpby(1.0, wg[k], 1.0, v.as_subclass(torch.Tensor).contiguous()) for k, v in dw.items()}
    if defer is not None:
        wg.update(deferred_update_grads(defer))
    return _dev.wrap(d_net), _dev.wrap(d_inp), d_pyr, {k: _dev.wrap(v) for k, v in wg.items()}


# ------------------------------------------------------------------------------------------------------------------------
# fourth slice: from the loop back to the encoder outputs
# ------------------------------------------------------------------------------------------------------------------------

def corr_build_backward(corr_block, d_pyramid):
    """Backward of ``CorrBlock(fmap1, fmap2)`` (reference corr.py:100-114, 154-162): the gradient w.r.t. the stored pyramid
    (flat, tiled layout -- what ``corr_lookup_backward`` / ``loop_backward`` accumulate) -> ``(d_fmap1, d_fmap2)``."""
    if corr_block._pyr is None:
        raise ValueError('the on-demand CorrBlock stores no pyramid to differentiate')
    f1 = corr_block.fmap1.as_subclass(torch.Tensor)
    bs, h, w, c = f1.shape
    d_pyr = _f32(d_pyramid)
    d1, d2 = torch.empty_like(f1), torch.empty_like(f1)
    ws = torch.empty_like(corr_block._f2pyr)
    check(_dev.lib().raft_corr_build_backward_f32(_dev.ptr(f1), _dev.ptr(corr_block._f2pyr), _dev.ptr(d_pyr), corr_block._off, bs, h, w, c,
                                                  corr_block.num_levels, _dev.ptr(d1), _dev.ptr(d2), _dev.ptr(ws), _dev.stream_ptr()),
          'corr_build_backward')
    return _dev.wrap(d1), _dev.wrap(d2)


def prepare_state_backward(net0, inp, d_net0, d_inp):
    """Backward of ``net = tanh(cnet[..., :hdim]); inp = relu(cnet[..., hdim:])`` (reference model.py:84-86): ``d_cnet``."""
    net0, inp, d_net0, d_inp = (_f32(t) for t in (net0, inp, d_net0, d_inp))
    hdim, cdim = net0.shape[-1], inp.shape[-1]
    M = net0.numel() // hdim
    out = torch.empty(tuple(net0.shape[:-1]) + (hdim + cdim,), device=net0.device, dtype=torch.float32)
    check(_dev.lib().raft_prepare_state_backward_f32(_dev.ptr(net0), _dev.ptr(inp), _dev.ptr(d_net0), _dev.ptr(d_inp), hdim, cdim, M,
                                                     _dev.ptr(out), _dev.stream_ptr()), 'prepare_state_backward')
    return _dev.wrap(out)


# ------------------------------------------------------------------------------------------------------------------------
# fifth slice: the encoders in training form (reference extractor.py:6-49, 88-130) and their backward
# ------------------------------------------------------------------------------------------------------------------------
# Strided convolutions reuse the stride-1 kernels: a stride-2 'same' convolution is the stride-1 'same' convolution sampled
# at every second position (TensorFlow's asymmetric SAME padding decides which parity), so its backward is the stride-1
# backward of the zero-stuffed upstream gradient; the 1x1 stride-2 'valid' projection subsamples its input first; the 7x7
# stride-2 stem (3 input channels) is a 1x1 convolution over the im2col'ed image.  torch only moves data here (slicing,
# unfold, zero-stuffing, concatenation); every multiply-add is a HIP kernel.  4x wasted work on three layers per encoder is
# the price of a functional path without new convolution kernels.

def _same_pad_before(n, k, stride):
    out = -(-n // stride)
    total = max((out - 1) * stride + k - n, 0)
    return out, total // 2


def _sub_index(n, k, stride):
    """Positions of a stride-`stride` SAME convolution's outputs inside the stride-1 SAME output: (start, count)."""
    out, pb = _same_pad_before(n, k, stride)
    return (k - 1) // 2 - pb, out


def _norm_fwd(x, gamma, beta, per_sample, relu):
    lib = _dev.lib()
    B, H, W, C_ = x.shape
    G, P = (B, H * W) if per_sample else (1, B * H * W)
    g_d, b_d = _dev_param(gamma), _dev_param(beta)
    y = torch.empty_like(x)
    mean, rstd, var = (torch.empty((G, C_), device=x.device, dtype=torch.float32) for _ in range(3))
    ws = torch.empty((int(lib.raft_norm_workspace_doubles(G, C_)),), device=x.device, dtype=torch.float64)
    check(lib.raft_norm_forward_f32(_dev.ptr(x), G, P, C_, _dev.ptr(g_d), _dev.ptr(b_d), 1e-3, 1 if relu else 0, _dev.ptr(y), _dev.ptr(mean),
                                    _dev.ptr(rstd), _dev.ptr(var), _dev.ptr(ws), _dev.stream_ptr()), 'norm_forward')
    return y, dict(x=x, y=y, mean=mean, rstd=rstd, var=var, gamma=g_d, G=G, P=P, relu=relu)


def _norm_bwd(cache, dy):
    lib = _dev.lib()
    x = cache['x']
    C_ = x.shape[-1]
    dy = dy.contiguous()
    if cache['relu']:
        masked = torch.empty_like(dy)
        check(lib.raft_relu_backward_f32(_dev.ptr(cache['y']), _dev.ptr(dy), _dev.ptr(masked), dy.numel(), _dev.stream_ptr()), 'relu_backward')
        dy = masked
    dx = torch.empty_like(x)
    dg, db = torch.empty((C_,), device=x.device, dtype=torch.float32), torch.empty((C_,), device=x.device, dtype=torch.float32)
    ws = torch.empty((int(lib.raft_norm_workspace_doubles(cache['G'], C_)),), device=x.device, dtype=torch.float64)
    check(lib.raft_norm_backward_f32(_dev.ptr(x), _dev.ptr(dy), _dev.ptr(cache['mean']), _dev.ptr(cache['rstd']), _dev.ptr(cache['gamma']),
                                     cache['G'], cache['P'], C_, _dev.ptr(dx), _dev.ptr(dg), _dev.ptr(db), _dev.ptr(ws), _dev.stream_ptr()),
          'norm_backward')
    return dx, dg, db


def _relu_bwd(y, dy):
    out = torch.empty_like(dy)
    check(_dev.lib().raft_relu_backward_f32(_dev.ptr(y), _dev.ptr(dy.contiguous()), _dev.ptr(out), dy.numel(), _dev.stream_ptr()), 'relu_backward')
    return out


def _conv_s(x, kernel, bias, stride):
    """Keras Conv2D(k, stride, 'same') for k in {3} and stride in {1, 2} through the stride-1 kernel (see above)."""
    y1 = _conv_fwd(x, kernel, bias)
    if stride == 1:
        return y1, None
    kh, kw = kernel.shape[:2]
    sy, ny = _sub_index(x.shape[1], kh, stride)
    sx, nx = _sub_index(x.shape[2], kw, stride)
    return y1[:, sy::stride, sx::stride][:, :ny, :nx].contiguous(), (sy, sx, tuple(y1.shape))


def _conv_s_bwd(x, kernel, dy, sub, stride):
    if sub is not None:
        sy, sx, full = sub
        dy1 = torch.zeros(full, device=dy.device, dtype=torch.float32)
        dy1[:, sy::stride, sx::stride][:, :dy.shape[1], :dy.shape[2]] = dy
        dy = dy1
    return _conv_bwd(x, kernel, dy.contiguous())


def _stem_cols(x):
    """im2col of the 7x7 / stride-2 'same' stem over a (B, H, W, 3) image: (B, Ho, Wo, 147) with depth order (ci, ky, kx)."""
    B, H, W, _ = x.shape
    ho, pt = _same_pad_before(H, 7, 2)
    wo, pl = _same_pad_before(W, 7, 2)
    pb, pr = max((ho - 1) * 2 + 7 - H, 0) - pt, max((wo - 1) * 2 + 7 - W, 0) - pl
    xc = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    cols = torch.nn.functional.unfold(xc, 7, stride=2)                     # (B, 3 * 49, Ho * Wo): pure gather
    return cols.transpose(1, 2).reshape(B, ho, wo, 147).contiguous()


def encoder_forward(weights, prefix, x, training=True):
    """``BasicEncoder`` / ``SmallEncoder`` forward (reference extractor.py:113-130 / 158-175) in training form: ``x`` is the
    normalised image batch (B, H, W, 3) (the two frames concatenated along the batch for fnet, extractor.py:116).  Batch
    normalisation uses batch statistics when ``training`` (Keras), instance normalisation is the same in both modes.
    Returns ``(out, tape)``."""
    p = prefix
    w = _params(weights, p + '/')
    x = _f32(x)
    tape = {'layers': []}

    def norm(name, t, relu):
        if f'{p}/{name}/moving_mean' in w:
            if training:
                return _norm_fwd(t, w[f'{p}/{name}/gamma'], w[f'{p}/{name}/beta'], False, relu)
            # inference statistics: an affine map per channel, expressed through the same kernel with fixed moments
            raise NotImplementedError('encoder_forward(training=False) with batch norm: use the inference encoders')
        if f'{p}/{name}/gamma' in w:
            return _norm_fwd(t, w[f'{p}/{name}/gamma'], w[f'{p}/{name}/beta'], True, relu)
        if relu:
            y = torch.empty_like(t)
            check(_dev.lib().raft_axpby_relu_f32(1.0, _dev.ptr(t), 0.0, None, _dev.ptr(y), t.numel(), _dev.stream_ptr()), 'relu')
            return y, dict(identity=True, y=y, relu=True)
        return t, dict(identity=True, y=t, relu=False)

    k1 = w[f'{p}/conv1/kernel']                                             # (7, 7, 3, c0): stem as a 1x1 conv over im2col
    cols = _stem_cols(x)
    kcol = (k1.permute(2, 0, 1, 3).reshape(1, 1, 147, k1.shape[3]).contiguous() if _is_t(k1)
            else np.ascontiguousarray(k1.transpose(2, 0, 1, 3).reshape(1, 1, 147, k1.shape[3])))
    c = _conv_fwd(cols, kcol, w[f'{p}/conv1/bias'])
    y, nc = norm('norm1', c, True)
    tape['stem'] = dict(cols=cols, kcol=kcol, norm=nc)
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        for bi, s in ((0, stride), (1, 1)):
            q = f'layer{li}/{bi}'
            c1, sub1 = _conv_s(y, w[f'{p}/{q}/conv1/kernel'], w[f'{p}/{q}/conv1/bias'], s)
            n1, nc1 = norm(f'{q}/norm1', c1, True)
            c2 = _conv_fwd(n1, w[f'{p}/{q}/conv2/kernel'], w[f'{p}/{q}/conv2/bias'])
            n2, nc2 = norm(f'{q}/norm2', c2, True)
            rec = dict(q=q, x=y, stride=s, sub1=sub1, nc1=nc1, n1=n1, nc2=nc2)
            xs = y
            if s != 1:
                xsub = y[:, ::s, ::s].contiguous()
                d = _conv_fwd(xsub, w[f'{p}/{q}/downsample/0/kernel'], w[f'{p}/{q}/downsample/0/bias'])
                xs, ncd = norm(f'{q}/downsample/1', d, False)
                rec.update(xsub=xsub, ncd=ncd)
            out = torch.empty_like(n2)
            check(_dev.lib().raft_axpby_relu_f32(1.0, _dev.ptr(xs.contiguous()), 1.0, _dev.ptr(n2), _dev.ptr(out), out.numel(), _dev.stream_ptr()),
                  'resblock join')
            rec['out'] = out
            tape['layers'].append(rec)
            y = out
    tape['last'] = y
    out = _conv_fwd(y, w[f'{p}/conv2/kernel'], w[f'{p}/conv2/bias'])
    return _dev.wrap(out), tape


def encoder_backward(weights, prefix, tape, d_out):
    """Backward of ``encoder_forward``: gradient w.r.t. every kernel, bias, gamma and beta of the encoder (dict under the
    weight names; the input image receives none).  Also returns the batch statistics of every batch-norm layer
    (``{name: (mean, biased variance, element count)}``) for the moving-average update."""
    p = prefix
    w = _params(weights, p + '/')
    g, stats = {}, {}

    def norm_b(name, cache, dy):
        if cache.get('identity'):
            return _relu_bwd(cache['y'], dy) if cache['relu'] else dy
        dx, dg, db = _norm_bwd(cache, dy)
        g[f'{p}/{name}/gamma'], g[f'{p}/{name}/beta'] = dg, db
        if f'{p}/{name}/moving_mean' in w:
            stats[f'{p}/{name}'] = (cache['mean'][0], cache['var'][0], float(cache['P']))
        return dx

    dy = _f32(d_out)
    dy, dk, db = _conv_bwd(tape['last'], w[f'{p}/conv2/kernel'], dy)
    g[f'{p}/conv2/kernel'], g[f'{p}/conv2/bias'] = dk, db
    for rec in reversed(tape['layers']):
        q, s = rec['q'], rec['stride']
        m = _relu_bwd(rec['out'], dy)                                       # relu(x + fx): both branches receive m
        d_c2 = norm_b(f'{q}/norm2', rec['nc2'], m)
        d_n1, dk, db = _conv_bwd(rec['n1'], w[f'{p}/{q}/conv2/kernel'], d_c2)
        g[f'{p}/{q}/conv2/kernel'], g[f'{p}/{q}/conv2/bias'] = dk, db
        d_c1 = norm_b(f'{q}/norm1', rec['nc1'], d_n1)
        d_x, dk, db = _conv_s_bwd(rec['x'], w[f'{p}/{q}/conv1/kernel'], d_c1, rec['sub1'], s)
        g[f'{p}/{q}/conv1/kernel'], g[f'{p}/{q}/conv1/bias'] = dk, db
        if s != 1:
            d_d = norm_b(f'{q}/downsample/1', rec['ncd'], m)
            d_xsub, dk, db = _conv_bwd(rec['xsub'], w[f'{p}/{q}/downsample/0/kernel'], d_d)
            g[f'{p}/{q}/downsample/0/kernel'], g[f'{p}/{q}/downsample/0/bias'] = dk, db
            skip = torch.zeros_like(d_x)
            skip[:, ::s, ::s] = d_xsub
        else:
            skip = m
        dy = _axpby(1.0, d_x.contiguous(), 1.0, skip.contiguous())
    st = tape['stem']
    d_c = norm_b('norm1', st['norm'], dy)
    _, dk, db = _conv_bwd(st['cols'], st['kcol'], d_c)
    c0 = dk.shape[-1]
    g[f'{p}/conv1/kernel'] = dk.reshape(3, 7, 7, c0).permute(1, 2, 0, 3).contiguous()      # (ci, ky, kx, co) -> Keras (ky, kx, ci, co)
    g[f'{p}/conv1/bias'] = db
    return {k: _dev.wrap(v) for k, v in g.items()}, stats
