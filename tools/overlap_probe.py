"""Three-stream loop on / off at small batches (GPU box): python tools/overlap_probe.py"""
import os, sys, time, torch
sys.path.insert(0, '.')
import tf_raft_amd
from tf_raft_amd import weights as wm
dev = torch.device('cuda', 0)
w = wm.init_weights('raft', seed=0)
for B in (1, 2):
    g = torch.Generator(device=dev).manual_seed(B)
    i1 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
    i2 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
    for ov in (True, False, True, False):
        model = tf_raft_amd.RAFT(weights=w, iters_pred=24, overlap=ov)
        for _ in range(3):
            model([i1, i2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            model([i1, i2])
        torch.cuda.synchronize()
        print(f'B={B} overlap={ov}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms', flush=True)
