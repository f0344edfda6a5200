"""Host-side profile of RAFT.train_step in steady state (cProfile over steps 3..6 of the training probe's setup).
usage: python tools/train_hostprof.py [all|update_block] [f32|bf16]"""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd                                   # noqa: E402
from tf_raft_amd import losses, training             # noqa: E402
from tf_raft_amd import weights as wm                # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'all'
tape = sys.argv[2] if len(sys.argv) > 2 else 'f32'
B, H, W, iters = 4, 368, 496, 12
rng = np.random.default_rng(0)
model = tf_raft_amd.RAFT(weights=wm.init_weights('raft', seed=0), iters=iters, iters_pred=24)
sched = training.CyclicalLearningRate(4e-4, 8e-4, 1000, training.first_cycle_scaler)
model.compile(optimizer=training.AdamW(1e-4, sched), clip_norm=1.0, loss=losses.sequence_loss, epe=losses.end_point_error, trainable=mode,
              tape_dtype=tape)
batch = (rng.uniform(0, 255, (B, H, W, 3)).astype(np.float32), rng.uniform(0, 255, (B, H, W, 3)).astype(np.float32),
         (rng.normal(size=(B, H, W, 2)) * 3).astype(np.float32), np.ones((B, H, W), bool))
for _ in range(3):
    model.train_step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(4):
    model.train_step(batch)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
pr.disable()
t_all = time.perf_counter() - t0
print(f'{mode} {tape}: 4 steps: host enqueue {t_host * 250:.1f} ms per step, with the GPU drained {t_all * 250:.1f} ms per step (under cProfile)')
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:5000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumtime').print_stats(30)
print(s.getvalue()[:5500])
