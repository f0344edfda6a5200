"""Throughput of the BASELINE.json configurations other than the bench line (GPU box; informational).

  python tools/config_bench.py [steps]
    config 3 per-GPU shape : RAFT, batch 8 at 448x512, iters_pred=24
    config 4               : RAFT, 1 x 1024x1024, iters_pred=24, alternate (on-demand) correlation -- and the
                             stored-volume path at the same size for comparison (1.43 GB volume)
    SmallRAFT              : batch 4 at 448x512, iters_pred=24
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd                      # noqa: E402
from tf_raft_amd import weights as wm    # noqa: E402


def run(name, model, B, H, W, steps):
    dev = torch.device('cuda', 0)
    g = torch.Generator(device=dev).manual_seed(0)
    i1 = torch.rand((B, H, W, 3), device=dev, generator=g) * 255
    i2 = torch.rand((B, H, W, 3), device=dev, generator=g) * 255
    for _ in range(4):
        model([i1, i2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = model([i1, i2])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f'{name}: {B}x{H}x{W} iters_pred={model.iters_pred}: {dt * 1e3:.2f} ms/step  {B / dt:.2f} pairs/s  '
          f'(peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB, out {tuple(out[-1].shape)})', flush=True)
    torch.cuda.reset_peak_memory_stats()


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    w = wm.init_weights('raft', seed=0)
    ws = wm.init_weights('small', seed=0)
    # every configuration on the serial schedule (a model's default) and on the pipelined one (pipeline=True: several loops in flight)
    for tag, kw in (('serial', dict(pipeline=False)), ('pipelined', dict(pipeline=True))):
        run(f'[{tag}] config2 RAFT (the bench line)', tf_raft_amd.RAFT(weights=w, iters_pred=24, **kw), 4, 448, 512, steps)
        run(f'[{tag}] config3-per-GPU RAFT', tf_raft_amd.RAFT(weights=w, iters_pred=24, **kw), 8, 448, 512, steps)
        run(f'[{tag}] config4 RAFT alternate_corr', tf_raft_amd.RAFT(weights=w, iters_pred=24, alternate_corr=True, **kw), 1, 1024, 1024, steps)
        run(f'[{tag}] config4-size RAFT stored volume', tf_raft_amd.RAFT(weights=w, iters_pred=24, **kw), 1, 1024, 1024, steps)
        run(f'[{tag}] SmallRAFT', tf_raft_amd.SmallRAFT(weights=ws, iters_pred=24, **kw), 4, 448, 512, steps)
        run(f'[{tag}] SmallRAFT config1 shape', tf_raft_amd.SmallRAFT(weights=ws, iters_pred=4, **kw), 1, 256, 256, steps)


if __name__ == '__main__':
    main()
