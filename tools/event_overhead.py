"""What a HIP-event bracket adds to a kernel's time in bench.py's instrumented replay (GPU box).

The single-stream loop (raft_iterate_basic_f32: the same launches, no events in between) is timed with ONE event pair and
set against the sum of the per-stage event intervals of raft_iterate_basic_timed_f32; the difference divided by the number
of bracketed launches is the bracket's cost.  Run it under `rocprofv3 --kernel-trace --stats` to get the kernels' own
durations of the same single-stream launches (what the per-stage numbers should agree with once the bracket is taken off).

  python tools/event_overhead.py [B] [reps]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd                               # noqa: E402
from tf_raft_amd import _dev, _ffi                # noqa: E402
from tf_raft_amd import weights as wm             # noqa: E402
from tf_raft_amd.layers.corr import CorrBlock     # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5          # the first repetition of each loop is a warm-up
H, W, ITERS = 448, 512, 24
STAGES = ['corr_lookup', 'convc1', 'convc2', 'convf1', 'convf2', 'conv', 'gru_zr1', 'gru_q1', 'gru_zr2', 'gru_q2',
          'fh1_mask0', 'fh2', 'mask2', 'upsample_convex']
dev = torch.device('cuda', 0)
model = tf_raft_amd.RAFT(iters_pred=ITERS, weights=wm.init_weights('raft', seed=0))
g = torch.Generator(device=dev).manual_seed(1000)
img1 = torch.rand((B, H, W, 3), device=dev, generator=g) * 255.0
img2 = torch.rand((B, H, W, 3), device=dev, generator=g) * 255.0
h, w = H // 8, W // 8
x1, x2 = 2 * (img1 / 255.0) - 1.0, 2 * (img2 / 255.0) - 1.0
fmap1, fmap2 = model.fnet([x1, x2])
cnet = model.cnet(x1)
corr = CorrBlock(fmap1, fmap2, num_levels=4, radius=4)
st = model._get_state(B, h, w, dev)
flow_up = torch.empty((ITERS, B, H, W, 2), device=dev)
lib = _dev.lib()
_ffi.set_option('RAFT_LOOKUP_FUSED', 0)           # 14 kernels per iteration, as in bench.py's per-kernel replay

buf = (C.c_float * len(STAGES))()
acc = np.zeros(len(STAGES))
for r in range(reps + 2):                          # two warm-up repetitions (clocks, code objects, caches)
    if r == 2:
        acc[:] = 0
    model._prepare(cnet, st)
    _ffi.check(lib.raft_iterate_basic_timed_f32(C.byref(model.update_block.c), _dev.ptr(corr._pyr), corr._off, B, h, w, ITERS,
                                                C.byref(st.c), _dev.ptr(flow_up), _dev.stream_ptr(), buf), 'timed')
    acc += np.array(list(buf))
stage_us = acc / (reps * ITERS) * 1e3
plain = []
for _ in range(reps + 1):
    model._prepare(cnet, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _ffi.check(lib.raft_iterate_basic_f32(C.byref(model.update_block.c), _dev.ptr(corr._pyr), corr._off, B, h, w, ITERS,
                                          C.byref(st.c), _dev.ptr(flow_up), _dev.stream_ptr()), 'plain')
    e1.record()
    torch.cuda.synchronize()
    plain.append(e0.elapsed_time(e1))
plain_us = float(np.median(plain[1:])) * 1e3 / ITERS
over = (stage_us.sum() - plain_us) / len(STAGES)
print(f'B={B}: sum of the {len(STAGES)} bracketed stages {stage_us.sum():.1f} us / iteration, the same launches without brackets '
      f'{plain_us:.1f} us / iteration -> {over:.2f} us per bracket')
print('stage_us      ', {k: round(float(v), 1) for k, v in zip(STAGES, stage_us)})
print('stage_us - brk', {k: round(float(v - over), 1) for k, v in zip(STAGES, stage_us)})
