"""Background mask branch (RAFT_MASK_BG_WGS = 32) against one workgroup per tile (0) over batch sizes / image sizes, one process."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd
from tf_raft_amd import _ffi
from tf_raft_amd import weights as wm
dev = torch.device('cuda', 0)
model = tf_raft_amd.RAFT(weights=wm.init_weights('raft', seed=0), iters_pred=24)
for B, H, W in ((4, 448, 512), (5, 448, 512), (6, 448, 512), (12, 448, 512), (16, 448, 512), (1, 1024, 1024), (4, 384, 512), (2, 640, 768), (4, 368, 496)):
    g = torch.Generator(device=dev).manual_seed(B)
    i1 = torch.rand((B, H, W, 3), device=dev, generator=g) * 255
    i2 = torch.rand((B, H, W, 3), device=dev, generator=g) * 255
    res = []
    for n in ('0', '32', '0', '32'):
        _ffi.set_option('RAFT_MASK_BG_WGS', n)
        for _ in range(2): model([i1, i2])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6): model([i1, i2])
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 6 * 1e3)
    _ffi.set_option('RAFT_MASK_BG_WGS', None)
    h, w = H // 8, W // 8
    G = B * ((h + 7) // 8) * ((w + 63) // 64) * 8
    print(f'B={B:2d} {H}x{W}: one per tile {res[0]:7.3f} / {res[2]:7.3f} ms, 32 background {res[1]:7.3f} / {res[3]:7.3f} ms  ({(min(res[0], res[2]) / min(res[1], res[3]) - 1) * 100:+.1f} %)  fh1_mask0 grid {G} = {G % 256} mod 256', flush=True)
