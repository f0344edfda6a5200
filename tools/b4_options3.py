"""convc2's workgroup shape x convf2's algorithm (direct / F(4x4)) with the mask branch in the background (one process)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd
from tf_raft_amd import _ffi
from tf_raft_amd import weights as wm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
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
        for _ in range(15): model([i1, i2])
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 15 * 1e3
        print(f'B={B} {label:52s} {ms:7.3f} ms  {B / ms * 1e3:7.1f} pairs/s', flush=True)
    finally:
        for k in opts: _ffi.set_option(k, None)
base = 9 if B < 8 else 13
run('default', {})
run('CONVC2_KS=2', {'RAFT_CONVC2_KS': '2'})
run(f'CONV_WINO4={base | 2} (convf2 on F(4x4))', {'RAFT_CONV_WINO4': str(base | 2)})
run(f'CONV_WINO4={base | 2} CONVC2_KS=2', {'RAFT_CONV_WINO4': str(base | 2), 'RAFT_CONVC2_KS': '2'})
run(f'CONV_WINO4={base | 2} CONVC2_KS=2 CONV_WINO=15', {'RAFT_CONV_WINO4': str(base | 2), 'RAFT_CONVC2_KS': '2', 'RAFT_CONV_WINO': '15'})
run('CONVC2_KS=2 CONV_WINO=15 (convf2 on F(2x2))', {'RAFT_CONVC2_KS': '2', 'RAFT_CONV_WINO': '15'})
run('default', {})
