"""Stress of the pipelined forward: many back-to-back calls of several models, shapes and entry points interleaved, results consumed
late / out of order / on side streams / dropped, allocator traffic in between; every result is compared bit for bit with a serial
(pipeline=False) model on the same inputs.  usage: python tools/pipeline_stress.py [rounds]"""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_raft_amd  # noqa: E402
from tf_raft_amd import weights as wm  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rnd = random.Random(7)
dev = torch.device('cuda', 0)
models = {}
for name, cls, kw in (('raft', tf_raft_amd.RAFT, {}), ('small', tf_raft_amd.SmallRAFT, {}), ('alt', tf_raft_amd.RAFT, {'alternate_corr': True})):
    w = wm.init_weights('small' if name == 'small' else 'raft', seed=5, perturb=True)
    # round 6: the pipelined models keep `lanes` loops in flight (all three models share the lane streams) and launch them with the
    # kernel shapes of a lanes-times larger batch; the serial reference gets the same launch-shape hint, so only the schedule differs
    lanes = int(os.environ.get('RAFT_LANES', tf_raft_amd.model.DEFAULT_LANES))
    models[name] = (cls(weights=w, iters_pred=6, pipeline=True, lanes=lanes, **kw),
                    cls(weights=w, iters_pred=6, pipeline=False, loop_concurrency=lanes, **kw))
shapes = [(1, 64, 96), (2, 128, 192), (3, 72, 104), (4, 448, 512), (1, 256, 320)]
inputs = {}
for s in shapes:
    g = torch.Generator(device=dev).manual_seed(sum(s))
    inputs[s] = [(torch.rand(s + (3,), device=dev, generator=g) * 255, torch.rand(s + (3,), device=dev, generator=g) * 255) for _ in range(3)]
want = {}


def reference(name, s, k, final):
    key = (name, s, k, final)
    if key not in want:
        a, b = inputs[s][k]
        m = models[name][1]
        out = m.predict_step((a, b)) if final else m([a, b])[-1]
        want[key] = out.cpu().numpy()
    return want[key]


side = torch.cuda.Stream()
checked = 0
for r in range(rounds):
    pending = []
    for _ in range(rnd.randint(3, 8)):
        name = rnd.choice(list(models))
        s = rnd.choice(shapes if name != 'alt' else shapes[:3] + shapes[4:])
        k = rnd.randrange(3)
        final = name == 'raft' and rnd.random() < 0.3
        a, b = inputs[s][k]
        m = models[name][0]
        out = m.predict_step((a, b)) if final else m([a, b])
        if rnd.random() < 0.3:
            junk = torch.full((rnd.randint(1, 1 << 24),), float('nan'), device=dev)   # allocator traffic on the caller's stream
            del junk
        if rnd.random() < 0.2:
            del out                                                                   # dropped while in flight
            continue
        pending.append((name, s, k, final, out))
    rnd.shuffle(pending)
    for name, s, k, final, out in pending:
        last = out if final else out[-1]
        if rnd.random() < 0.5:
            with torch.cuda.stream(side):
                got = last.cpu()
            side.synchronize()
            got = got.numpy()
        else:
            got = last.cpu().numpy()
        np.testing.assert_array_equal(got, reference(name, s, k, final))
        checked += 1
    if r % 5 == 4:
        mname = rnd.choice(list(models))
        models[mname][0].set_weights(models[mname][0].get_weights_dict())      # re-upload while nothing is pending ... or is it
print(f'pipeline stress ({lanes} lanes): {rounds} rounds, {checked} results compared bit for bit: ok')
