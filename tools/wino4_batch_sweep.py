"""RAFT forward at 448x512, iters_pred=24 over the per-GPU batch with the F(4x4,3x3) layer masks (RAFT_CONV_WINO4): which
layers should be on it at which batch.  python tools/wino4_batch_sweep.py [steps]   (GPU box)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd                      # noqa: E402
from tf_raft_amd import _ffi            # noqa: E402
from tf_raft_amd import weights as wm    # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0)
model = tf_raft_amd.RAFT(weights=wm.init_weights('raft', seed=0), iters_pred=24)
for B in (1, 2, 3, 4, 8, 16):
    g = torch.Generator(device=dev).manual_seed(B)
    i1 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
    i2 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
    row = []
    for mask in ('0', '8', '9', '13', 'default'):
        _ffi.set_option('RAFT_CONV_WINO4', None if mask == 'default' else mask)
        for fn in (lambda: model([i1, i2]), lambda: model.predict_step((i1, i2))):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            row.append((time.perf_counter() - t0) / steps * 1e3)
    print(f'B={B:2d} ms/step (all 24 predictions | predict_step):  ' + '   '.join(
        f'mask {m}: {row[2 * i]:6.2f} | {row[2 * i + 1]:6.2f}' for i, m in enumerate(('0', '8', '9', '13', 'default'))), flush=True)
