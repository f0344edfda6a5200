"""Throughput / latency of RAFT forward prediction at 448x512, iters_pred=24 over the per-GPU batch (GPU box).
  python tools/batch_sweep.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd                      # noqa: E402
from tf_raft_amd import weights as wm    # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device('cuda', 0)
model = tf_raft_amd.RAFT(weights=wm.init_weights('raft', seed=0), iters_pred=24)
from tf_raft_amd import _ffi            # noqa: E402
for B, graph in ((1, '0'), (1, '1'), (2, '0'), (2, '1'), (4, '0'), (4, '1'), (8, '0'), (8, '1'), (16, '')):
    _ffi.set_option('RAFT_LOOP_GRAPH', graph)        # '' = the library's default (graphs for small batches only)
    g = torch.Generator(device=dev).manual_seed(B)
    i1 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
    i2 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
    for _ in range(2):
        model([i1, i2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model([i1, i2])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f'B={B:2d} RAFT_LOOP_GRAPH={graph or "default":7s}: {dt * 1e3:7.2f} ms/step  {B / dt:7.1f} pairs/s', flush=True)
