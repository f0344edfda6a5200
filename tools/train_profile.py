"""cProfile of RAFT.train_step (host side).  python tools/train_profile.py"""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd                                   # noqa: E402
from tf_raft_amd import losses, training             # noqa: E402
from tf_raft_amd import weights as wm                # noqa: E402

B, H, W, iters = 4, 368, 496, 12
rng = np.random.default_rng(0)
model = tf_raft_amd.RAFT(weights=wm.init_weights('raft', seed=0), iters=iters, iters_pred=24)
model.compile(optimizer=training.AdamW(1e-4, 4e-4), clip_norm=1.0, loss=losses.sequence_loss, epe=losses.end_point_error)
data = (rng.uniform(0, 255, (B, H, W, 3)).astype(np.float32), rng.uniform(0, 255, (B, H, W, 3)).astype(np.float32),
        (rng.normal(size=(B, H, W, 2)) * 3).astype(np.float32), np.ones((B, H, W), bool))
model.train_step(data)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    model.train_step(data)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
