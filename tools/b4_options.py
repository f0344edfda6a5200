"""Round-3 option sweep at batch B (one process): convf2 on F(2x2), no K-split F(4x4) workgroups, encoder F(4x4) stage masks;
at 8 pairs also conv off F(4x4); and predict_step with / without the K split.  python tools/b4_options.py [B]"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
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
        print(f'B={B} {label:40s} {ms:7.3f} ms  {B / ms * 1e3:7.1f} pairs/s', flush=True)
    finally:
        for k in opts: _ffi.set_option(k, None)
run('default', {})
run('CONV_WINO=15 (convf2 on F(2x2))', {'RAFT_CONV_WINO': '15'})
run('default', {})
run('CONV_WINO=15 (convf2 on F(2x2))', {'RAFT_CONV_WINO': '15'})
run('WINO4_KS=1', {'RAFT_WINO4_KS': '1'})
run('ENC_WINO4=7', {'RAFT_ENC_WINO4': '7'})
run('ENC_WINO4=3', {'RAFT_ENC_WINO4': '3'})
run('default', {})
if B == 8:
    run('WINO4_KS=1', {'RAFT_WINO4_KS': '1'})
    run('CONV_WINO4=9 (conv on F(2x2))', {'RAFT_CONV_WINO4': '9'})
    run('default', {})
def runp(label, opts):
    for k, v in opts.items(): _ffi.set_option(k, v)
    try:
        for _ in range(3): model.predict_step((i1, i2))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(15): model.predict_step((i1, i2))
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 15 * 1e3
        print(f'B={B} predict_step {label:27s} {ms:7.3f} ms  {B / ms * 1e3:7.1f} pairs/s', flush=True)
    finally:
        for k in opts: _ffi.set_option(k, None)
runp('default', {})
runp('WINO4_KS=1', {'RAFT_WINO4_KS': '1'})
runp('default', {})
