"""ONE option sweep for every tuning switch of the library (replaces the round-1..3 family b1_options / b4_options{,2,3,4} /
b8_options / round3_options / mask_bg_options / small_batch_mask_options, whose result files stay under profiles/).

    python tools/option_sweep.py [--batch 4] [--reps 15] [--shape 448x512] [--fresh-model] \
        "label: RAFT_X=1 RAFT_Y=2" "RAFT_Z=0" ...

Each positional argument is one setting: optional "label:" then space-separated NAME=VALUE switches (raft_set_option; names as
in include/raft_hip.h).  The forward pass (RAFT, 24 iterations, Keras-default weights, device-resident random images) is timed
in ONE process, the default setting first and last (box drift shows as the difference between the two).  --fresh-model builds a
new model (and loop context: streams, events) per setting -- needed for switches that are read when the context is created
(RAFT_EVENT_FENCE)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd  # noqa: E402
from tf_raft_amd import _ffi  # noqa: E402
from tf_raft_amd import weights as wm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--reps', type=int, default=15)
    ap.add_argument('--shape', default='448x512')
    ap.add_argument('--fresh-model', action='store_true')
    ap.add_argument('--lanes', type=int, default=0, help='round 6: pipelined forward with this many loops in flight (0: the serial schedule)')
    ap.add_argument('--overlap', type=int, default=-1, help='with --lanes: 1 / 0 forces three-stream / single-stream loops')
    ap.add_argument('settings', nargs='*')
    a = ap.parse_args()
    H, W = (int(v) for v in a.shape.split('x'))
    B = a.batch
    dev = torch.device('cuda', 0)
    wts = wm.init_weights('raft', seed=0)
    g = torch.Generator(device=dev).manual_seed(B)
    i1 = torch.rand((B, H, W, 3), device=dev, generator=g) * 255
    i2 = torch.rand((B, H, W, 3), device=dev, generator=g) * 255
    kw = dict(pipeline=True, lanes=a.lanes) if a.lanes else dict(pipeline=False)
    if a.overlap >= 0:
        kw['overlap'] = bool(a.overlap)
    shared = None if a.fresh_model else tf_raft_amd.RAFT(weights=wts, iters_pred=24, **kw)

    def run(label, opts):
        for k, v in opts.items():
            _ffi.set_option(k, v)
        try:
            model = shared or tf_raft_amd.RAFT(weights=wts, iters_pred=24, **kw)
            for _ in range(3):
                model([i1, i2])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                model([i1, i2])
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.reps * 1e3
            print(f'B={B} {H}x{W} {label:52s} {ms:7.3f} ms  {B / ms * 1e3:7.1f} pairs/s', flush=True)
        finally:
            for k in opts:
                _ffi.set_option(k, None)

    run('default', {})
    for text in a.settings:
        label, _, rest = text.rpartition(':')
        opts = dict(kv.split('=', 1) for kv in rest.split())
        run(label.strip() or rest.strip(), opts)
    run('default', {})


if __name__ == '__main__':
    main()
