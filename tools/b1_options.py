"""B = 1 (the reference's canonical single-pair call) under the library's tuning switches, in ONE process:
python tools/b1_options.py [B]   -> ms per forward for each option set (GPU box)."""
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
        for _ in range(10):
            model([i1, i2])
        torch.cuda.synchronize()
        print(f'{label:44s} {(time.perf_counter() - t0) / 10 * 1e3:7.3f} ms', flush=True)
    finally:
        for k in opts:
            _ffi.set_option(k, None)


run('default', {})
run('CONV_WINO=0 (direct 3x3 everywhere)', {'RAFT_CONV_WINO': '0'})
for m in (1, 4, 8, 5, 9, 12):
    run(f'CONV_WINO={m}', {'RAFT_CONV_WINO': str(m)})
run('GRU_WINO=0 GRU_WINO4=0 (direct 1x5)', {'RAFT_GRU_WINO': '0', 'RAFT_GRU_WINO4': '0'})
run('GRU_WINO4=15 (F(4,5))', {'RAFT_GRU_WINO4': '15'})
run('LOOKUP_FUSED=0', {'RAFT_LOOKUP_FUSED': '0'})
for t in ('141', '142', '171', '181', '3', '5'):
    run(f'CONV_WINO=0 CONV_TILE={t}', {'RAFT_CONV_WINO': '0', 'RAFT_CONV_TILE': t})
run('WINO_TNW=1', {'RAFT_WINO_TNW': '1'})
run('WINO_CK=1', {'RAFT_WINO_CK': '1'})
run('default again', {})
