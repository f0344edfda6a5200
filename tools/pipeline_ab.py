"""Back-to-back inference calls, serial against pipelined (RAFT_PIPELINE), at a few batch sizes; run once per environment:
  python tools/pipeline_ab.py [batches...]        (prints ms per step and pairs/s; median of 3 rounds of 30 steps)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd  # noqa: E402
from tf_raft_amd import weights as wm  # noqa: E402

dev = torch.device('cuda', 0)
batches = [int(a) for a in sys.argv[1:]] or [4, 8]
tag = ' '.join(f'{k}={os.environ[k]}' for k in ('RAFT_PIPELINE', 'RAFT_LOOP_PRIORITY', 'GPU_MAX_HW_QUEUES') if k in os.environ) or 'defaults'
model = tf_raft_amd.RAFT(weights=wm.init_weights('raft', seed=0), iters_pred=24)
for B in batches:
    g = torch.Generator(device=dev).manual_seed(B)
    i1 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
    i2 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
    for _ in range(5):
        model([i1, i2])
    torch.cuda.synchronize()
    rates = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(30):
            model([i1, i2])
        torch.cuda.synchronize()
        rates.append(B * 30 / (time.perf_counter() - t0))
    # latency of ONE call whose result is consumed at once (what a caller that cannot pipeline sees)
    lat = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model([i1, i2])[-1].cpu()
        lat.append((time.perf_counter() - t0) * 1e3)
    print(f'[{tag}] B={B}: {np.median(rates):7.1f} pairs/s (rounds {[round(r, 1) for r in rates]}), {B / np.median(rates) * 1e3:6.2f} ms/step; '
          f'single call + download {np.median(lat):6.2f} ms', flush=True)
