"""Time raft_lookup_convc1_f32 and its two phases alone (GPU box).

    RAFT_BUILD_DEFINES=-DRAFT_FUSED_PROBE python tools/fused_probe.py [B]

The phase-ablated kernels (RAFT_LOOKUP_FUSED = 11 / 12) exist only in a library built with -DRAFT_FUSED_PROBE; the
product library does not contain them (the next ordinary import rebuilds it: the define is part of the build digest)."""
import os
assert '-DRAFT_FUSED_PROBE' in os.environ.get('RAFT_BUILD_DEFINES', ''), __doc__
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_raft_amd import _dev, _ffi, packing            # noqa: E402
from tf_raft_amd.layers.corr import CorrBlock          # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
h, w = 56, 64
rng = np.random.default_rng(0)
dev = torch.device('cuda', 0)
f1 = torch.randn((B, h, w, 256), device=dev)
f2 = torch.randn((B, h, w, 256), device=dev)
corr = CorrBlock(f1, f2, 4, 4)
ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing='ij')
coords = (torch.stack([xs, ys], -1)[None] + torch.randn((B, h, w, 2), device=dev) * 3).contiguous()
kernel = (rng.normal(size=(1, 1, 324, 256)) * 0.1).astype(np.float32)
wp, b, npad = packing.pack_convc1_fused(kernel, np.zeros(256, np.float32))
wp_d, b_d = _dev.to_device(wp), _dev.to_device(b)
out = torch.empty((B, h, w, 256), device=dev)
lk = torch.empty((B, h, w, 352), device=dev)
big = torch.empty(1 << 27, device=dev)                  # 512 MB: evicts L2 / Infinity Cache between launches


def run():
    _ffi.check(_dev.lib().raft_lookup_convc1_f32(_dev.ptr(corr._pyr), corr._off, _dev.ptr(coords), B, h, w, _dev.ptr(wp_d),
                                                 _dev.ptr(b_d), npad, 256, _dev.ptr(out), 256, _dev.stream_ptr()), 'fused')


def run_lookup():
    _ffi.check(_dev.lib().raft_corr_lookup_f32(_dev.ptr(corr._pyr), corr._off, _dev.ptr(coords), B, h, w, 4, 4, _dev.ptr(lk), 352,
                                               _dev.stream_ptr()), 'lookup')


def timed(fn, reps=50, cold=False):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if not cold:
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    for _ in range(10):
        big.zero_()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / 10 * 1e3


for name, v in (('fused', 1), ('no lookup phase', 11), ('no MFMA phase', 12)):
    _ffi.set_option('RAFT_LOOKUP_FUSED', v)
    print(f'B={B} {name:16s}: warm {timed(run):6.1f} us   cold {timed(run, cold=True):6.1f} us')
_ffi.set_option('RAFT_LOOKUP_FUSED', None)
print(f'B={B} stand-alone lookup: warm {timed(run_lookup):6.1f} us   cold {timed(run_lookup, cold=True):6.1f} us')
