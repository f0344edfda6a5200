"""PCIe-inclusive feed rates: resident inputs vs pageable arrays handed to the model vs tf_raft_amd.prefetch.
python tools/h2d_probe.py [B] [n]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import tf_raft_amd
from tf_raft_amd import weights as wm
from tf_raft_amd.prefetch import prefetch_to_device

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
model = tf_raft_amd.RAFT(iters_pred=24, weights=wm.init_weights('raft', seed=0))
rng = np.random.default_rng(0)
u1 = rng.integers(0, 256, size=(B, 448, 512, 3), dtype=np.uint8)
u2 = rng.integers(0, 256, size=(B, 448, 512, 3), dtype=np.uint8)
f1, f2 = u1.astype(np.float32), u2.astype(np.float32)
d1, d2 = torch.as_tensor(f1).cuda(), torch.as_tensor(f2).cuda()


def run(name, feed, fn):
    for a, b in feed(3):
        fn(a, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for a, b in feed(n):
        fn(a, b)
        marks.append(time.perf_counter())
    cpu_done = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{name:34s} {B * n / dt:8.2f} pairs/s   {dt / n * 1e3:7.3f} ms/step   cpu loop {cpu_done / n * 1e3:7.3f} ms/step')


full = lambda a, b: model([a, b])
last = lambda a, b: model.predict_step((a, b))
rep = lambda x, y: (lambda k: ((x, y) for _ in range(k)))
for label, fn in (('call', full), ('predict_step', last)):
    run(f'{label} resident', rep(d1, d2), fn)
    run(f'{label} pageable fp32', rep(f1, f2), fn)
    run(f'{label} pageable uint8', rep(u1, u2), fn)
    run(f'{label} prefetch fp32', lambda k: prefetch_to_device(rep(f1, f2)(k)), fn)
    run(f'{label} prefetch uint8', lambda k: prefetch_to_device(rep(u1, u2)(k)), fn)
t0 = time.perf_counter()
out = model.predict([np.concatenate([u1] * n), np.concatenate([u2] * n)], batch_size=B)
dt = time.perf_counter() - t0
print(f'model.predict uint8 -> host flows   {B * n / dt:8.2f} pairs/s   {dt / n * 1e3:7.3f} ms/step  {out.shape}')
