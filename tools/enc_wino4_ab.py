"""Encoder F(4x4) stage masks (RAFT_ENC_WINO4) at 4 and 8 pairs with the final loop schedule, one process."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import tf_raft_amd
from tf_raft_amd import _ffi
from tf_raft_amd import weights as wm
dev = torch.device('cuda', 0)
model = tf_raft_amd.RAFT(weights=wm.init_weights('raft', seed=0), iters_pred=24)
for B in (4, 8):
    g = torch.Generator(device=dev).manual_seed(B)
    i1 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
    i2 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
    for label, opts in (('default', {}), ('ENC_WINO4=7', {'RAFT_ENC_WINO4': '7'}), ('ENC_WINO4=3', {'RAFT_ENC_WINO4': '3'}), ('ENC_WINO4=1', {'RAFT_ENC_WINO4': '1'}), ('ENC_WINO4=0', {'RAFT_ENC_WINO4': '0'}), ('default', {})):
        for k, v in opts.items(): _ffi.set_option(k, v)
        for _ in range(3): model([i1, i2])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(12): model([i1, i2])
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 12 * 1e3
        for k in opts: _ffi.set_option(k, None)
        print(f'B={B} {label:16s} {ms:7.3f} ms  {B / ms * 1e3:7.1f} pairs/s', flush=True)
