"""One RAFT forward loop at batch B, a few calls, under whatever RAFT_LOOP_GRAPH says (run it under rocprofv3 --kernel-trace:
tools/graph_trace.py reads the start / end timestamps of the kernels of the LAST call and says where the time goes).
usage: python tools/graph_probe.py <batch> [calls]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tf_raft_amd  # noqa: E402

B = int(sys.argv[1])
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda', 0)
gen = torch.Generator(device=dev)
gen.manual_seed(1000)
i1 = torch.rand((B, 448, 512, 3), device=dev, generator=gen) * 255.0
i2 = torch.rand((B, 448, 512, 3), device=dev, generator=gen) * 255.0
model = tf_raft_amd.RAFT(iters_pred=24)
for _ in range(3):
    model([i1, i2])
torch.cuda.synchronize()
t = []
for _ in range(calls):
    t0 = time.perf_counter()
    model([i1, i2])
    torch.cuda.synchronize()
    t.append((time.perf_counter() - t0) * 1e3)
print(f'B={B} RAFT_LOOP_GRAPH={os.environ.get("RAFT_LOOP_GRAPH", "unset")}: ms per call {[round(x, 3) for x in t]}')
