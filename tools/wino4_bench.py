"""3x3 layers of the update block on the three kernels -- direct, Winograd F(2x2,3x3), Winograd F(4x4,3x3) -- timed alone
with HIP events (back-to-back launches), and the deviation of each from the float64 convolution on a sample of pixels.
  python tools/wino4_bench.py [B ...]            (GPU box)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_raft_amd import _dev, packing          # noqa: E402
from tf_raft_amd._ffi import check             # noqa: E402

H, W = 56, 64
LAYERS = [('fh1_mask0', 128, 512), ('convc2', 256, 192), ('conv', 256, 126), ('fh1', 128, 256), ('convf2', 128, 64)]
rng = np.random.default_rng(0)
lib = _dev.lib()


def timed(run, n=30):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in [int(a) for a in sys.argv[1:]] or [4]:
    print(f'B={B} M={B * H * W}   (us per launch, TFLOP/s of direct-convolution FLOPs, max |err| vs float64 on 64 pixels)')
    for name, cin, cout in LAYERS:
        k = (rng.normal(size=(3, 3, cin, cout)) * (1.0 / np.sqrt(9 * cin))).astype(np.float32)
        bias = rng.normal(size=cout).astype(np.float32)
        xh = np.maximum(rng.normal(size=(B, H, W, cin)), 0).astype(np.float32)
        x = _dev.to_device(xh)
        outs = {}
        flops = 2.0 * B * H * W * 9 * cin * cout
        line = f'  {name:10s} {cin:3d}->{cout:3d}'
        for kind in ('direct', 'wino2', 'wino4', 'wino4ks2'):
            if kind == 'direct':
                wp, b, npad = packing.pack_conv(k, bias)
            elif kind == 'wino2':
                wp, b, npad = packing.pack_conv_winograd(k, bias)
            else:
                wp, b, npad = packing.pack_conv_winograd4(k, bias)
                from tf_raft_amd import _ffi
                _ffi.set_option('RAFT_WINO4_KS', 2 if kind == 'wino4ks2' else 1)
            wp_d, b_d = _dev.to_device(wp), _dev.to_device(b)
            out = torch.empty((B, H, W, cout), device=x.device)
            if kind == 'direct':
                run = lambda: check(lib.raft_conv2d_f32(_dev.ptr(x), cin, cin, None, 0, 0, _dev.ptr(wp_d), _dev.ptr(b_d), B, H, W, 3, 3,
                                                        npad, cout, 1, 1.0, _dev.ptr(out), cout, _dev.stream_ptr()))
            elif kind == 'wino2':
                run = lambda: check(lib.raft_conv2d_winograd_f32(_dev.ptr(x), cin, cin, None, 0, 0, _dev.ptr(wp_d), _dev.ptr(b_d), B, H,
                                                                 W, npad, cout, 1, 1.0, _dev.ptr(out), cout, _dev.stream_ptr()))
            else:
                run = lambda: check(lib.raft_conv2d_winograd4_f32(_dev.ptr(x), cin, cin, None, 0, 0, _dev.ptr(wp_d), _dev.ptr(b_d), B, H,
                                                                  W, npad, cout, 1, 1.0, _dev.ptr(out), cout, _dev.stream_ptr()))
            us = timed(run)
            outs[kind] = out.cpu().numpy()
            line += f' | {kind} {us:6.1f}us {flops / us / 1e6:6.1f}TF'
        # float64 reference on a sample of pixels (border pixels included)
        xp = np.pad(xh[0].astype(np.float64), ((1, 1), (1, 1), (0, 0)))
        ys = [0, 1, 3, 4, 27, 28, 54, 55]
        xs = [0, 1, 3, 4, 31, 32, 62, 63]
        ref = np.zeros((8, 8, cout))
        for iy, y in enumerate(ys):
            for ix, xx in enumerate(xs):
                ref[iy, ix] = np.maximum(np.einsum('uvc,uvco->o', xp[y:y + 3, xx:xx + 3], k.astype(np.float64)) + bias, 0)
        for kind in outs:
            got = outs[kind][0][np.ix_(ys, xs)]
            line += f' | err {kind} {np.abs(got - ref).max():.1e}'
        print(line, flush=True)
