"""B = 1 probe (GPU box): per-kernel HIP-event times of the single-stream loop and end-to-end latency under a few tile
switches.   python tools/b1_probe.py"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd                                   # noqa: E402
from tf_raft_amd import _dev, _ffi                   # noqa: E402
from tf_raft_amd import weights as wm                # noqa: E402
from tf_raft_amd.layers.corr import CorrBlock        # noqa: E402

STAGES = ['corr_lookup', 'convc1', 'convc2', 'convf1', 'convf2', 'conv', 'gru_zr1', 'gru_q1', 'gru_zr2',
          'gru_q2', 'fh1_mask0', 'fh2', 'mask2', 'upsample_convex']
dev = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
model = tf_raft_amd.RAFT(weights=wm.init_weights('raft', seed=0), iters_pred=24)
g = torch.Generator(device=dev).manual_seed(1)
i1 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
i2 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255


def latency(n=5):
    for _ in range(2):
        model([i1, i2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        model([i1, i2])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def stages():
    x1, x2 = 2 * (i1 / 255.0) - 1.0, 2 * (i2 / 255.0) - 1.0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    ev[0].record()
    fmap1, fmap2 = model.fnet([x1, x2])
    ev[1].record()
    cnet = model.cnet(x1)
    ev[2].record()
    corr = CorrBlock(fmap1, fmap2, num_levels=4, radius=4)
    ev[3].record()
    torch.cuda.synchronize()
    pre = [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
    st = model._get_state(B, 56, 64, dev)
    model._prepare(cnet, st)
    flow_up = torch.empty((24, B, 448, 512, 2), device=dev)
    buf = (C.c_float * len(STAGES))()
    acc = np.zeros(len(STAGES))
    for _ in range(3):
        model._prepare(cnet, st)
        _ffi.check(_dev.lib().raft_iterate_basic_timed_f32(C.byref(model.update_block.c), _dev.ptr(corr._pyr), corr._off, B, 56, 64,
                                                           24, C.byref(st.c), _dev.ptr(flow_up), _dev.stream_ptr(), buf), 'timed')
        acc += np.array(list(buf))
    return pre, acc / (3 * 24)


pre, st = stages()
print(f'B={B} pre-loop ms: fnet {pre[0]:.3f} cnet {pre[1]:.3f} corr_build {pre[2]:.3f}')
print('per-launch us:', ' '.join(f'{k}={v * 1e3:.1f}' for k, v in zip(STAGES, st)), f' sum/iter={st.sum() * 1e3:.1f} us  x24 = {st.sum() * 24:.2f} ms')
print(f'default                : {latency():.2f} ms')
for name, opts in [('overlap off', None), ('WINO_TNW=2', {'RAFT_WINO_TNW': 2}), ('WINO_TNW=1', {'RAFT_WINO_TNW': 1}),
                   ('CONV_TILE=141', {'RAFT_CONV_TILE': 141}), ('CONV_TILE=171', {'RAFT_CONV_TILE': 171}),
                   ('WINO1D_TM=1', {'RAFT_WINO1D_TM': 1}), ('GRU_WINO4=0 (F(2,5))', {'RAFT_GRU_WINO4': 0}),
                   ]:
    if opts is None:
        model.overlap = False
        print(f'{name:23s}: {latency():.2f} ms')
        model.overlap = True
        continue
    for k, v in opts.items():
        _ffi.set_option(k, v)
    print(f'{name:23s}: {latency():.2f} ms')
    for k in opts:
        _ffi.set_option(k, None)
