"""Tile micro-benchmark of the fp32-MFMA implicit-GEMM convolution (GPU box):
every layer shape of the BasicUpdateBlock x every output tile, TFLOP/s from HIP-event timing.
  python tools/conv_bench.py [B] > gpurun_out/conv_bench.txt
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_raft_amd import _dev, packing          # noqa: E402
from tf_raft_amd._ffi import check             # noqa: E402


def _ffi_opt(name, value):
    from tf_raft_amd import _ffi
    _ffi.set_option(name, value)   # tuning switch of the library (include/raft_hip.h)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H, W = 56, 64
LAYERS = [  # name, kh, kw, cin(real), cin(pad), cout
    ('convc1', 1, 1, 324, 352, 256), ('convc2', 3, 3, 256, 256, 192), ('convf2', 3, 3, 128, 128, 64),
    ('conv', 3, 3, 256, 256, 126), ('gru_zr', 1, 5, 256, 256, 256), ('gru_q', 1, 5, 256, 256, 128),
    ('gru_zr_v', 5, 1, 256, 256, 256), ('gru_q_v', 5, 1, 256, 256, 128), ('gru_ctx', 1, 5, 128, 128, 384), ('fh1_mask0', 3, 3, 128, 128, 512), ('mask2', 1, 1, 256, 256, 576)]
rng = np.random.default_rng(0)
lib = _dev.lib()
print(f'B={B} M={B*H*W}')
for name, kh, kw, cin, cpad, cout in LAYERS:
    k = (rng.normal(size=(kh, kw, cin, cout)) * 0.05).astype(np.float32)
    wp, b, npad = packing.pack_conv(k, np.zeros(cout, np.float32), [(cin, cpad)])
    x = _dev.to_device(rng.normal(size=(B, H, W, cpad)).astype(np.float32))
    wp_d, b_d = _dev.to_device(wp), _dev.to_device(b)
    out = torch.empty((B, H, W, cout), device=x.device)
    flops = 2.0 * B * H * W * kh * kw * cin * cout
    row = []
    for tile in ('3', '5', '141', '142', '171', '172', '181', '182', 'auto'):
        if tile == 'auto':
            _ffi_opt('RAFT_CONV_TILE', '')
        else:
            if npad % ({'3': 64, '5': 64}.get(tile) or 64 * int(tile[-1])):
                row.append(f'{tile}:   n/a')
                continue
            _ffi_opt('RAFT_CONV_TILE', tile)

        def run():
            check(lib.raft_conv2d_f32(_dev.ptr(x), cpad, cpad, None, 0, 0, _dev.ptr(wp_d), _dev.ptr(b_d), B, H, W,
                                      kh, kw, npad, cout, 1, 1.0, _dev.ptr(out), cout, _dev.stream_ptr()))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        row.append(f'{tile}:{flops / ms / 1e9:6.1f}TF {ms*1e3:6.1f}us')
    if kh == 3 and kw == 3:      # Winograd F(2x2, 3x3) kernel, 32- and 64-channel workgroups (TFLOP/s of the DIRECT algorithm's FLOPs)
        wpw, bw, npw = packing.pack_conv_winograd(k, np.zeros(cout, np.float32), [(cin, cpad)])
        wpw_d, bw_d = _dev.to_device(wpw), _dev.to_device(bw)
        for tnw in ('1', '2'):
            _ffi_opt('RAFT_WINO_TNW', tnw)

            def runw():
                check(lib.raft_conv2d_winograd_f32(_dev.ptr(x), cpad, cpad, None, 0, 0, _dev.ptr(wpw_d), _dev.ptr(bw_d), B, H, W,
                                                   npw, cout, 1, 1.0, _dev.ptr(out), cout, _dev.stream_ptr()))
            for _ in range(3):
                runw()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                runw()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            row.append(f'wino{tnw}:{flops / ms / 1e9:6.1f}TF {ms*1e3:6.1f}us')
        _ffi_opt('RAFT_WINO_TNW', '')
    if kh * kw == 5:             # 1-D Winograd F(2, 5) kernel
        wpw, bw, npw = packing.pack_conv_winograd1d(k, np.zeros(cout, np.float32), [(cin, cpad)])
        wpw_d, bw_d = _dev.to_device(wpw), _dev.to_device(bw)
        for tnw in ('1', '2'):
            _ffi_opt('RAFT_WINO_TNW', tnw)

            def runw():
                check(lib.raft_conv1d_winograd_f32(_dev.ptr(x), cpad, cpad, None, 0, 0, _dev.ptr(wpw_d), _dev.ptr(bw_d), B, H, W,
                                                   kh, kw, npw, cout, 1, 1.0, _dev.ptr(out), cout, _dev.stream_ptr()))
            for _ in range(3):
                runw()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                runw()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            row.append(f'wino{tnw}:{flops / ms / 1e9:6.1f}TF {ms*1e3:6.1f}us')
        _ffi_opt('RAFT_WINO_TNW', '')
    print(f'{name:10s} K={kh*kw*cin:5d} N={cout:4d} | ' + ' | '.join(row), flush=True)
_ffi_opt('RAFT_CONV_TILE', '')
