"""Remaining launch-shape switches of the chain at 4 pairs with the final schedule (one process)."""
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
        print(f'B={B} {label:44s} {ms:7.3f} ms  {B / ms * 1e3:7.1f} pairs/s', flush=True)
    finally:
        for k in opts: _ffi.set_option(k, None)
run('default', {})
run('WINO_TNW=2 (64-channel gru_q, conv)', {'RAFT_WINO_TNW': '2'})
run('WINO_TNW=1', {'RAFT_WINO_TNW': '1'})
run('WINO1D_TM=1', {'RAFT_WINO1D_TM': '1'})
run('WINO_CK=1', {'RAFT_WINO_CK': '1'})
run('WINO_SB=0', {'RAFT_WINO_SB': '0'})
run('GRU_WINO4=0 (F(2,5))', {'RAFT_GRU_WINO4': '0'})
run('LOOKUP_FUSED=0', {'RAFT_LOOKUP_FUSED': '0'})
run('CONV_WINO4=15 (conv on F(4x4))', {'RAFT_CONV_WINO4': '15'})
run('default', {})
