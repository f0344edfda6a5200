"""Run ONE kernel configuration repeatedly (for rocprofv3 --pmc passes and A/B timing on the GPU box).

  python tools/one_kernel.py conv <layer> <tile|auto> [B] [reps]
  python tools/one_kernel.py lookup staged [B] [reps]
  python tools/one_kernel.py upsample [B] [reps]
Prints the HIP-event average per launch.
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

LAYERS = {  # name: kh, kw, cin(real), cin(pad), cout
    'convc1': (1, 1, 324, 352, 256), 'convc2': (3, 3, 256, 256, 192), 'convf2': (3, 3, 128, 128, 64),
    'conv': (3, 3, 256, 256, 126), 'gru_zr': (1, 5, 256, 256, 256), 'gru_q': (1, 5, 256, 256, 128),
    'gru_zr_v': (5, 1, 256, 256, 256), 'gru_q_v': (5, 1, 256, 256, 128), 'gru_ctx': (1, 5, 128, 128, 384), 'fh1_mask0': (3, 3, 128, 128, 512), 'mask2': (1, 1, 256, 256, 576)}
H, W = 56, 64


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    kind = sys.argv[1]
    rng = np.random.default_rng(0)
    lib = _dev.lib()
    if kind == 'conv':
        name, tile = sys.argv[2], sys.argv[3]
        B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
        reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
        kh, kw, cin, cpad, cout = LAYERS[name]
        if tile != 'auto':
            _ffi_opt('RAFT_CONV_TILE', tile)
        k = (rng.normal(size=(kh, kw, cin, cout)) * 0.05).astype(np.float32)
        wp, b, npad = packing.pack_conv(k, np.zeros(cout, np.float32), [(cin, cpad)])
        x = _dev.to_device(rng.normal(size=(B, H, W, cpad)).astype(np.float32))
        wp_d, b_d = _dev.to_device(wp), _dev.to_device(b)
        out = torch.empty((B, H, W, cout), device=x.device)

        def run():
            check(lib.raft_conv2d_f32(_dev.ptr(x), cpad, cpad, None, 0, 0, _dev.ptr(wp_d), _dev.ptr(b_d), B, H, W,
                                      kh, kw, npad, cout, 1, 1.0, _dev.ptr(out), cout, _dev.stream_ptr()))
        ms = timed(run, reps)
        flops = 2.0 * B * H * W * kh * kw * cin * cout
        print(f'conv {name} tile={tile} B={B}: {ms*1e3:.1f} us  {flops / ms / 1e9:.1f} TFLOP/s')
    elif kind == 'wino':
        name, tnw = sys.argv[2], sys.argv[3]
        B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
        reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
        kh, kw, cin, cpad, cout = LAYERS[name]
        assert kh == 3 and kw == 3
        _ffi_opt('RAFT_WINO_TNW', tnw)
        k = (rng.normal(size=(kh, kw, cin, cout)) * 0.05).astype(np.float32)
        wp, b, npad = packing.pack_conv_winograd(k, np.zeros(cout, np.float32), [(cin, cpad)])
        x = _dev.to_device(rng.normal(size=(B, H, W, cpad)).astype(np.float32))
        wp_d, b_d = _dev.to_device(wp), _dev.to_device(b)
        out = torch.empty((B, H, W, cout), device=x.device)

        def run():
            check(lib.raft_conv2d_winograd_f32(_dev.ptr(x), cpad, cpad, None, 0, 0, _dev.ptr(wp_d), _dev.ptr(b_d), B, H, W,
                                               npad, cout, 1, 1.0, _dev.ptr(out), cout, _dev.stream_ptr()))
        ms = timed(run, reps)
        flops = 2.0 * B * H * W * kh * kw * cin * cout
        print(f'wino {name} tnw={tnw} B={B}: {ms*1e3:.1f} us  {flops / ms / 1e9:.1f} TFLOP/s (direct-algorithm FLOPs)')
    elif kind == 'wino1d':
        # python tools/one_kernel.py wino1d <gru_zr|gru_q|gru_zr_v|gru_q_v> <tnw: 0 auto, 1, 2> [B] [reps] [m: 4 or 2]
        name, tnw = sys.argv[2], sys.argv[3]
        B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
        reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
        m = int(sys.argv[6]) if len(sys.argv) > 6 else 4
        kh, kw, cin, cpad, cout = LAYERS[name]
        assert (kh, kw) in ((1, 5), (5, 1))
        if tnw != '0':
            _ffi_opt('RAFT_WINO_TNW', tnw)
        k = (rng.normal(size=(kh, kw, cin, cout)) * 0.05).astype(np.float32)
        wp, b, npad = packing.pack_conv_winograd1d(k, np.zeros(cout, np.float32), [(cin, cpad)], m=m)
        x = _dev.to_device(rng.normal(size=(B, H, W, cpad)).astype(np.float32))
        wp_d, b_d = _dev.to_device(wp), _dev.to_device(b)
        out = torch.empty((B, H, W, cout), device=x.device)
        fn = lib.raft_conv1d_winograd4_f32 if m == 4 else lib.raft_conv1d_winograd_f32

        def run():
            check(fn(_dev.ptr(x), cpad, cpad, None, 0, 0, _dev.ptr(wp_d), _dev.ptr(b_d), B, H, W, kh, kw,
                     npad, cout, 1, 1.0, _dev.ptr(out), cout, _dev.stream_ptr()))
        ms = timed(run, reps)
        flops = 2.0 * B * H * W * kh * kw * cin * cout
        print(f'wino1d F({m},5) {name} tnw={tnw} B={B}: {ms*1e3:.1f} us  {flops / ms / 1e9:.1f} TFLOP/s (direct-algorithm FLOPs)')
    elif kind == 'lookup':
        ver = sys.argv[2]
        B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
        reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
        assert ver == 'staged', 'the direct-store variant is only the fallback for unaligned / 3-level outputs now'
        from tf_raft_amd.layers.corr import CorrBlock
        f1 = rng.normal(size=(B, H, W, 256)).astype(np.float32)
        f2 = rng.normal(size=(B, H, W, 256)).astype(np.float32)
        corr = CorrBlock(f1, f2, 4, 4)
        ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing='ij')
        coords = np.stack([xs, ys], -1)[None].repeat(B, 0) + rng.normal(scale=6.0, size=(B, H, W, 2)).astype(np.float32)
        coords_d = _dev.to_device(coords)
        out = torch.empty((B, H, W, 352), device=coords_d.device)

        def run():
            corr.retrieve(coords_d, out=out, ld_out=352)
        ms = timed(run, reps)
        bytes_ = B * H * W * (4 * 100 * 4 + 8 + 324 * 4)
        print(f'lookup {ver} B={B}: {ms*1e3:.1f} us  {bytes_ / ms / 1e6:.0f} GB/s algorithmic')
    elif kind == 'ondemand':
        # python tools/one_kernel.py ondemand <sigma> [h] [w] [reps]   (B = 1, C = 256, radius 4; RAFT_ONDEMAND_BLOCK=0/1)
        sigma = float(sys.argv[2])
        h = int(sys.argv[3]) if len(sys.argv) > 3 else 128
        w = int(sys.argv[4]) if len(sys.argv) > 4 else 128
        reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
        from tf_raft_amd.layers.corr import CorrBlock
        f1 = rng.normal(size=(1, h, w, 256)).astype(np.float32)
        f2 = rng.normal(size=(1, h, w, 256)).astype(np.float32)
        corr = CorrBlock(f1, f2, 4, 4, alternate=True)
        ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing='ij')
        coords = np.stack([xs, ys], -1)[None] + rng.normal(scale=sigma, size=(1, h, w, 2)).astype(np.float32)
        coords_d = _dev.to_device(coords)
        out = torch.empty((1, h, w, 352), device=coords_d.device)

        def run():
            corr.retrieve(coords_d, out=out, ld_out=352)
        ms = timed(run, reps)
        print(f'ondemand lookup {h}x{w} sigma={sigma} block={os.environ.get("RAFT_ONDEMAND_BLOCK", "1")}: {ms*1e3:.1f} us')
    elif kind == 'upsample':
        B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
        reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
        flow = _dev.to_device(rng.normal(size=(B, H, W, 2)).astype(np.float32))
        mask = _dev.to_device(rng.normal(size=(B, H, W, 576)).astype(np.float32))
        out = torch.empty((B, 8 * H, 8 * W, 2), device=flow.device)

        def run():
            check(lib.raft_upsample_convex_f32(_dev.ptr(flow), _dev.ptr(mask), B, H, W, _dev.ptr(out), _dev.stream_ptr()))
        ms = timed(run, reps)
        bytes_ = B * H * W * (576 * 4 + 8 + 64 * 2 * 4)
        print(f'upsample_convex B={B}: {ms*1e3:.1f} us  {bytes_ / ms / 1e6:.0f} GB/s algorithmic')
    else:
        raise SystemExit(__doc__)


if __name__ == '__main__':
    main()
