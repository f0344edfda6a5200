"""8 / 16 pairs: workgroup shapes of convc2 / convf2 on the F(4x4) kernel beside the background mask branch (one process)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd
from tf_raft_amd import _ffi
from tf_raft_amd import weights as wm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0)
model = tf_raft_amd.RAFT(weights=wm.init_weights('raft', seed=0), iters_pred=24)
g = torch.Generator(device=dev).manual_seed(B)
i1 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
i2 = torch.rand((B, 448, 512, 3), device=dev, generator=g) * 255
def run(label, opts):
    for k, v in opts.items(): _ffi.set_option(k, v)
    try:
        for _ in range(3): model([i1, i2])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): model([i1, i2])
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
        print(f'B={B} {label:44s} {ms:7.3f} ms  {B / ms * 1e3:7.1f} pairs/s', flush=True)
    finally:
        for k in opts: _ffi.set_option(k, None)
run('default', {})
run('CONVF2_KS=1', {'RAFT_CONVF2_KS': '1'})
run('CONVF2_KS=1 CONVC2_KS=2', {'RAFT_CONVF2_KS': '1', 'RAFT_CONVC2_KS': '2'})
run('CONVF2_KS=2 CONVC2_KS=2', {'RAFT_CONVF2_KS': '2', 'RAFT_CONVC2_KS': '2'})
run('CONVF2_KS=1 MASK_BG_WGS=64', {'RAFT_CONVF2_KS': '1', 'RAFT_MASK_BG_WGS': '64'})
run('CONV_WINO4=11 (conv on F(2x2))', {'RAFT_CONV_WINO4': '11'})
run('CONV_WINO4=11 CONVF2_KS=1', {'RAFT_CONV_WINO4': '11', 'RAFT_CONVF2_KS': '1'})
run('default', {})
