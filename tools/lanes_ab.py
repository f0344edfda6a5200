"""Round 6: several recurrent loops in flight (RAFT(pipeline=True, lanes=D)), against the serial and the one-lane schedule.
One process sweeps (lanes, overlap) settings A/B/A-style; GPU_MAX_HW_QUEUES is a per-process HIP setting, so run once per value:
  GPU_MAX_HW_QUEUES=8 python tools/lanes_ab.py [--batch 4] [--steps 20] [--rounds 5] "serial" "1:1" "2:1" "2:0" "3:0" "4:0"
A setting is  lanes:overlap  (overlap 1 = three-stream loop, 0 = single-stream loop) or  serial  (pipeline=False)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd  # noqa: E402
from tf_raft_amd import weights as wm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--warmup', type=int, default=5)
ap.add_argument('--rounds', type=int, default=5)
ap.add_argument('settings', nargs='*', default=['serial', '1:1', '2:1', '2:0', '3:0', '4:0'])
args = ap.parse_args()
dev = torch.device('cuda', 0)
B = args.batch
wts = wm.init_weights('raft', seed=0)
g = torch.Generator(device=dev).manual_seed(B)
i1 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
i2 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
models = {}
for s in args.settings:
    if s == 'serial':
        models[s] = tf_raft_amd.RAFT(weights=wts, iters_pred=24, pipeline=False)
    else:
        lanes, ov = (int(x) for x in s.split(':'))
        models[s] = tf_raft_amd.RAFT(weights=wts, iters_pred=24, pipeline=True, lanes=lanes, overlap=bool(ov))
rates = {s: [] for s in args.settings}
ref = None
for s in args.settings:           # results must not depend on the schedule
    out = models[s]([i1, i2])[-1].cpu()
    if ref is None:
        ref = out
    if not torch.equal(ref, out):
        print(f'!! {s}: final prediction differs from {args.settings[0]} (max {float((ref - out).abs().max()):.3g})', flush=True)
for r in range(args.rounds):
    for s in args.settings:
        m = models[s]
        for _ in range(args.warmup):
            m([i1, i2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            m([i1, i2])
        torch.cuda.synchronize()
        rates[s].append(B * args.steps / (time.perf_counter() - t0))
tag = ' '.join(f'{k}={os.environ[k]}' for k in ('GPU_MAX_HW_QUEUES', 'RAFT_LOOP_PRIORITY', 'RAFT_EVENT_FENCE') if k in os.environ) or 'defaults'
for s in args.settings:
    v = rates[s]
    print(f'[{tag}] B={B} steps={args.steps} {s:>7}: median {np.median(v):7.1f} pairs/s  (rounds {[round(x, 1) for x in v]})', flush=True)
