"""Writes tests/golden/conditioning.json: for each end-to-end parity case, the per-iteration
max-EPE between the CPU oracle run in fp32 and in fp64 (same inputs, same weights).

This measures how well conditioned the reference computation itself is: the sampler's
discontinuity turns rounding noise into O(1) px differences once a clamped tap
coordinate crosses an integer.  GPU parity tests assert 1e-3 only where this file shows the oracle
agreeing with itself to 1e-4.  Run from the repo root:  python tests/golden/make_conditioning.py
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle                                   # noqa: E402
from oracle.losses import max_epe                # noqa: E402
from tf_raft_amd import weights as wm            # noqa: E402

CASES = [('raft', 64, 96, 12, 0), ('raft', 128, 160, 12, 1), ('small', 64, 96, 12, 0),
         ('small', 256, 256, 4, 0), ('raft', 448, 512, 24, 0)]


def run(variant, H, W, iters, seed):
    rng = np.random.default_rng(seed)
    i1 = rng.uniform(0, 255, (1, H, W, 3)).astype(np.float32)
    i2 = rng.uniform(0, 255, (1, H, W, 3)).astype(np.float32)
    wts = wm.init_weights(variant, seed=seed)
    cls = oracle.RAFT if variant == 'raft' else oracle.SmallRAFT
    o32 = cls(wts, iters_pred=iters)([i1, i2])
    o64 = cls(wts, iters_pred=iters, dtype=torch.float64)([i1, i2])
    return dict(epe32v64=[max_epe(a, b) for a, b in zip(o32, o64)],
                max_abs_flow=[float(np.abs(b).max()) for b in o64])


if __name__ == '__main__':
    out = {}
    for variant, H, W, iters, seed in CASES:
        key = f'{variant}_{H}x{W}_seed{seed}_it{iters}'
        out[key] = run(variant, H, W, iters, seed)
        print(key, ' '.join(f'{e:.1e}' for e in out[key]['epe32v64']), flush=True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'conditioning.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote', path)
