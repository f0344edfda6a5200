"""Backward kernels of the training step, first slice (reference ``tf_raft/model.py:126-144`` differentiates the forward
pass with ``tf.GradientTape``; BASELINE config 5).  What exists: the gradient of ``sequence_loss`` w.r.t. every flow
prediction, the backward of the pyramid lookup (``CorrBlock.retrieve`` + ``bilinear_sampler``) w.r.t. the coordinates and
the correlation pyramid, and the backward of one Keras ``Conv2D`` (+ relu) w.r.t. input, kernel and bias -- each a HIP
kernel behind the C ABI (``csrc/backward.hip``), deterministic, checked against torch autograd of the CPU oracle in
``tests/test_gpu_backward.py``.  The remaining pieces of ``train_step`` (backward of the GRU gates, the convex upsampling,
the encoders and norms; global-norm clipping, AdamW, the RCCL gradient all-reduce) are not built yet:
``RAFT.train_step`` still raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _dev, packing
from ._ffi import check


def _f32(t):
    return _dev.to_device(t).as_subclass(torch.Tensor).to(torch.float32).contiguous()


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


def conv2d_backward(x, kernel, dy, y=None):
    """Backward of ``y = [relu](conv2d(x, kernel) + bias)`` (Keras Conv2D, stride 1, 'same'; reference update.py:10-11,
    91-95, 138-140).  ``x`` (B, H, W, Cin), ``kernel`` (kh, kw, Cin, Cout) NumPy in Keras layout, ``dy`` (B, H, W, Cout);
    pass the forward output ``y`` to apply the relu mask first.  Returns ``(dx, d_kernel, d_bias)`` device tensors.
    Cin and Cout must be multiples of 4 (every layer of the update block except the flow-carrying ones)."""
    x = _f32(x)
    dy = _f32(dy)
    kernel = np.asarray(kernel, dtype=np.float32)
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
    ws = torch.empty((int(lib.raft_conv2d_wgrad_workspace_floats(cin, cout, B, H, W, kh, kw)),), device=x.device,
                     dtype=torch.float32)
    d_kernel = torch.empty((kh, kw, cin, cout), device=x.device, dtype=torch.float32)
    d_bias = torch.empty((cout,), device=x.device, dtype=torch.float32)
    check(lib.raft_conv2d_wgrad_f32(_dev.ptr(x), cin, cin, _dev.ptr(dy), cout, cout, B, H, W, kh, kw, _dev.ptr(d_kernel),
                                    _dev.ptr(d_bias), _dev.ptr(ws), _dev.stream_ptr()), 'conv2d_wgrad')
    # input gradient: the forward convolution of dy with the flipped, transposed kernel
    cpad = packing.round_up(cout, 32)
    wp, b, npad = packing.pack_conv_dgrad(kernel, [(cout, cpad)])
    dyp = dy
    if cpad != cout:
        dyp = torch.zeros((B, H, W, cpad), device=x.device, dtype=torch.float32)
        dyp[..., :cout] = dy
    wp_d, b_d = _dev.to_device(wp), _dev.to_device(b)
    dx = torch.empty((B, H, W, cin), device=x.device, dtype=torch.float32)
    check(lib.raft_conv2d_f32(_dev.ptr(dyp), cpad, cpad, None, 0, 0, _dev.ptr(wp_d), _dev.ptr(b_d), B, H, W, kh, kw, npad,
                              cin, 0, 1.0, _dev.ptr(dx), cin, _dev.stream_ptr()), 'conv2d dgrad')
    return _dev.wrap(dx), _dev.wrap(d_kernel), _dev.wrap(d_bias)
