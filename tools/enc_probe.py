"""fnet (2B images) and cnet (B images) of RAFT at 448x512, one after the other on one stream, HIP-event timed; run it under
rocprofv3 --kernel-trace --stats for the per-kernel durations without the fnet / cnet overlap of the product path.
usage: python tools/enc_probe.py [B] [reps]      (RAFT_ENC_WINO4 / RAFT_ENC_WINO select the 3x3 kernels)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda', 0)
gen = torch.Generator(device=dev)
gen.manual_seed(1000)
x1 = torch.rand((B, 448, 512, 3), device=dev, generator=gen) * 2 - 1
x2 = torch.rand((B, 448, 512, 3), device=dev, generator=gen) * 2 - 1
model = tf_raft_amd.RAFT(iters_pred=1)
for _ in range(2):
    model.fnet([x1, x2])
    model.cnet(x1)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tc = 0.0
for _ in range(reps):
    torch.cuda.synchronize()
    ev[0].record()
    model.fnet([x1, x2])
    ev[1].record()
    model.cnet(x1)
    ev[2].record()
    torch.cuda.synchronize()
    tf += ev[0].elapsed_time(ev[1])
    tc += ev[1].elapsed_time(ev[2])
print(f'B={B} RAFT_ENC_WINO4={os.environ.get("RAFT_ENC_WINO4", "unset")}: fnet {tf / reps:.3f} ms, cnet {tc / reps:.3f} ms')
