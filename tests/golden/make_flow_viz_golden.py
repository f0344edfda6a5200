"""Generates tests/golden/flow_viz_golden.npz by running the REFERENCE's own flow colour coding
(/root/reference/tf_raft/datasets/flow_viz.py -- pure NumPy, importable without TensorFlow) on seeded flows.
Run in the build container only (the reference tree does not exist on the GPU box):
    python tests/golden/make_flow_viz_golden.py
"""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location('ref_flow_viz', '/root/reference/tf_raft/datasets/flow_viz.py')
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rng = np.random.default_rng(0)
flows = {
    'gauss': rng.normal(scale=3.0, size=(24, 32, 2)).astype(np.float32),
    'axes': np.stack(np.meshgrid(np.linspace(-5, 5, 21), np.linspace(-4, 4, 17)), -1).astype(np.float32),
    'tiny': (rng.normal(size=(5, 7, 2)) * 1e-3).astype(np.float32),
}
out = {'colorwheel': ref.make_colorwheel()}
for name, f in flows.items():
    out[f'{name}_flow'] = f
    out[f'{name}_rgb'] = ref.flow_to_image(f)
    out[f'{name}_bgr'] = ref.flow_to_image(f, convert_to_bgr=True)
    out[f'{name}_clip'] = ref.flow_to_image(f, clip_flow=2.0)
np.savez_compressed(os.path.join(HERE, 'flow_viz_golden.npz'), **out)
print('wrote', sorted(out))
