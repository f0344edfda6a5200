"""Writes tests/golden/reference_forward_golden.npz: outputs of the reference's UNMODIFIED source
(/root/reference/tf_raft/model.py + layers/*.py) executed under oracle/tfstub on seeded inputs and weights.

The reference tree does not travel to the GPU box, so these vectors are what the ``-m gpu`` tests (and a container without
/root/reference) compare against.  Cases (inputs and weights are those of tests/golden/make_conditioning.py::case_inputs, so the
file holds outputs only):

  * ``{raft,small}_64x96_seed0_it12`` (Keras-default weights, perturb=True set separately below) -- all of iterations 1, 6, 12;
  * ``raft_448x512_seed{0,1}_it24_conditioned``, ``small_448x512_seed0_it24_conditioned``, ``raft_448x512_seed0_it24_jump0``:
    ``flow_predictions[-1]`` of BASELINE's north-star shape (1,448,512,3) x 24 iterations, kept on the pixel grid [1::4, 2::4]
    (114 KB per case) plus float64 checksums of the whole tensor.

Run from the repo root:  python tests/golden/make_reference_forward_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_conditioning import case_inputs, case_key      # noqa: E402
from oracle import reference_runner as rr                # noqa: E402
from tf_raft_amd import weights as wm                    # noqa: E402

SMALL = [('raft', 64, 96, 12, 0), ('small', 64, 96, 12, 0)]
FULL = [('raft', 448, 512, 24, 0, 'conditioned'), ('raft', 448, 512, 24, 1, 'conditioned'),
        ('small', 448, 512, 24, 0, 'conditioned'), ('raft', 448, 512, 24, 0, 'jump0')]
KEEP_ITERS = (0, 5, 11)
GRID = (slice(1, None, 4), slice(2, None, 4))


def small_case(variant, H, W, iters, seed):
    """Perturbed weights (biases / norm parameters away from their identity defaults) + uniform images."""
    rng = np.random.default_rng(100 + seed)
    i1 = rng.uniform(0, 255, (1, H, W, 3)).astype(np.float32)
    i2 = rng.uniform(0, 255, (1, H, W, 3)).astype(np.float32)
    return i1, i2, wm.init_weights(variant, seed=seed, perturb=True)


def subsample(flow):
    return np.ascontiguousarray(flow[:, GRID[0], GRID[1], :])


def checksums(flow):
    f = flow.astype(np.float64)
    return np.array([f.sum(), np.abs(f).sum(), (f * f).sum()])


if __name__ == '__main__':
    out = {}
    for variant, H, W, iters, seed in SMALL:
        i1, i2, wts = small_case(variant, H, W, iters, seed)
        pred = rr.forward(rr.build_model(variant, wts, iters_pred=iters), i1, i2)
        key = f'{variant}_{H}x{W}_seed{seed}_it{iters}_perturbed'
        for k in KEEP_ITERS:
            out[f'{key}/iter{k}'] = pred[k]
        print(key, float(np.abs(pred[-1]).max()), flush=True)
    for variant, H, W, iters, seed, regime in FULL:
        i1, i2, wts = case_inputs(variant, H, W, seed, regime)
        pred = rr.forward(rr.build_model(variant, wts, iters_pred=iters), i1, i2)
        key = case_key(variant, H, W, iters, seed, regime)
        out[f'{key}/last_grid'] = subsample(pred[-1])
        out[f'{key}/last_checksums'] = checksums(pred[-1])
        print(key, float(np.abs(pred[-1]).max()), flush=True)
    np.savez_compressed(os.path.join(HERE, 'reference_forward_golden.npz'), **out)
