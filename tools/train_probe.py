"""Time RAFT.train_step at the reference's training shape (configs/train_chairs.yml: batch 4, 368x496 crops, iters 12).
python tools/train_probe.py [B] [all|update_block] [f32|bf16]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd                                   # noqa: E402
from tf_raft_amd import losses, training             # noqa: E402
from tf_raft_amd import weights as wm                # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mode = sys.argv[2] if len(sys.argv) > 2 else 'all'
tape = sys.argv[3] if len(sys.argv) > 3 else 'f32'
H, W, iters = 368, 496, 12
rng = np.random.default_rng(0)
model = tf_raft_amd.RAFT(weights=wm.init_weights('raft', seed=0), iters=iters, iters_pred=24)
sched = training.CyclicalLearningRate(4e-4, 8e-4, 1000, training.first_cycle_scaler)
model.compile(optimizer=training.AdamW(1e-4, sched), clip_norm=1.0, loss=losses.sequence_loss, epe=losses.end_point_error, trainable=mode,
              tape_dtype=tape)
i1 = rng.uniform(0, 255, (B, H, W, 3)).astype(np.float32)
i2 = rng.uniform(0, 255, (B, H, W, 3)).astype(np.float32)
flow = (rng.normal(size=(B, H, W, 2)) * 3).astype(np.float32)
valid = np.ones((B, H, W), bool)
for step in range(int(os.environ.get("TRAIN_PROBE_STEPS", "8"))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = model.train_step((i1, i2, flow, valid))
    torch.cuda.synchronize()
    print(f'B={B} {mode} tape {tape}: step {step}: {B / (time.perf_counter() - t0):6.1f} pairs/s {time.perf_counter() - t0:6.2f} s  loss {float(res["loss"]):.4f} epe {float(res["epe"]):.3f}  '
          f'peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)
