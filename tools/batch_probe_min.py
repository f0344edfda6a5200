"""Minimal batch probe (ms per forward at a few batch sizes)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd
from tf_raft_amd import weights as wm
dev = torch.device('cuda', 0)
model = tf_raft_amd.RAFT(weights=wm.init_weights('raft', seed=0), iters_pred=24)
for B in (1, 4):
    g = torch.Generator(device=dev).manual_seed(B)
    i1 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
    i2 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
    for _ in range(3):
        model([i1, i2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        model([i1, i2])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f'  B={B}: {dt * 1e3:6.2f} ms/step {B / dt:6.1f} pairs/s', flush=True)
