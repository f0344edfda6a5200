"""The round-3 switches (fused mask + upsampling, F(4x4) encoder stages, rotating loop buffers, F(4x4) update-block layers) one at a
time at batch B, in ONE process: python tools/round3_options.py [B] -> ms per forward (GPU box)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd                      # noqa: E402
from tf_raft_amd import _ffi             # noqa: E402
from tf_raft_amd import weights as wm    # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device('cuda', 0)
model = tf_raft_amd.RAFT(weights=wm.init_weights('raft', seed=0), iters_pred=24)
g = torch.Generator(device=dev).manual_seed(B)
i1 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
i2 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255


def run(label, opts):
    for k, v in opts.items():
        _ffi.set_option(k, v)
    try:
        for _ in range(3):
            model([i1, i2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(12):
            model([i1, i2])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 12 * 1e3
        print(f'B={B} {label:40s} {ms:7.3f} ms  {B / ms * 1e3:7.1f} pairs/s', flush=True)
    finally:
        for k in opts:
            _ffi.set_option(k, None)


run('default', {})
run('MASK_FUSED=0', {'RAFT_MASK_FUSED': '0'})
run('LOOP_ROTATE=0', {'RAFT_LOOP_ROTATE': '0'})
for m in ('0', '1', '3', '7'):
    run(f'ENC_WINO4={m}', {'RAFT_ENC_WINO4': m})
for m in ('0', '8', '9', '13'):
    run(f'CONV_WINO4={m}', {'RAFT_CONV_WINO4': m})
run('default again', {})
