"""Does running a batch as several independent chains (own streams) beat one chain over the whole batch?  (GPU box)
  python tools/chains_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd                      # noqa: E402
from tf_raft_amd import weights as wm    # noqa: E402

dev = torch.device('cuda', 0)
wts = wm.init_weights('raft', seed=0)


def bench(total_b, chains, steps=8):
    per = total_b // chains
    models = [tf_raft_amd.RAFT(weights=wts, iters_pred=24) for _ in range(chains)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(chains)]
    g = torch.Generator(device=dev).manual_seed(1)
    ims = [(torch.rand((per, 448, 512, 3), device=dev, generator=g) * 255, torch.rand((per, 448, 512, 3), device=dev, generator=g) * 255)
           for _ in range(chains)]
    torch.cuda.synchronize()

    def step():
        outs = []
        for m, s, (a, b) in zip(models, streams, ims):
            with torch.cuda.stream(s):
                outs.append(m([a, b])[-1])
        return outs
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f'B={total_b} as {chains} chain(s) of {per}: {dt * 1e3:7.2f} ms/step  {total_b / dt:7.1f} pairs/s', flush=True)


for total_b, chains in ((4, 1), (4, 2), (4, 4), (8, 1), (8, 2), (8, 4), (2, 1), (2, 2)):
    bench(total_b, chains)
