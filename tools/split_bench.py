"""Does splitting a batch into P concurrent sub-batches (each its own loop on its own streams) beat one loop over the
whole batch?  GPU box, informational.   python tools/split_bench.py [B] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd                      # noqa: E402
from tf_raft_amd import weights as wm    # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device('cuda', 0)
    w = wm.init_weights('raft', seed=0)
    g = torch.Generator(device=dev).manual_seed(0)
    i1 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
    i2 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
    for parts in (1, 2, 4):
        if B % parts:
            continue
        models = [tf_raft_amd.RAFT(weights=w, iters_pred=24) for _ in range(parts)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
        n = B // parts
        chunks = [(i1[k * n:(k + 1) * n].contiguous(), i2[k * n:(k + 1) * n].contiguous()) for k in range(parts)]

        def step():
            outs = []
            for m, s, (a, b) in zip(models, streams, chunks):
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
        print(f'B={B} as {parts} x {n}: {dt * 1e3:.2f} ms/step  {B / dt:.1f} pairs/s', flush=True)


if __name__ == '__main__':
    main()
