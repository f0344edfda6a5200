"""Execute the reference's UNMODIFIED source under oracle/tfstub (test infrastructure only).

``load_reference()`` imports ``/root/reference/tf_raft`` under the alias package ``_reference_tf_raft`` (this repository has a
``tf_raft`` import shim of its own; the reference only uses relative imports inside its package, model.py:5-7) with the stand-in
``tensorflow`` / ``tensorflow_addons`` of ``oracle/tfstub`` on ``sys.path`` for the duration of the import, and never writes
byte-code into the (read-only) reference tree.  ``build_model`` instantiates the reference's ``RAFT`` / ``SmallRAFT`` and puts
Keras-layout weights (``tf_raft_amd.weights.init_weights``) into its layers by attribute path -- the same dict the oracle and the
HIP path take -- so all three run on identical parameters.

The reference tree only exists in the build container: callers skip when ``reference_available()`` is false and fall back on
the committed outputs (``tests/golden/reference_forward_golden.npz``, written by ``tests/golden/make_reference_forward_golden.py``).
"""
from __future__ import annotations

import contextlib
import importlib
import importlib.util
import os
import sys

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get('RAFT_REFERENCE_ROOT', '/root/reference')
STUB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tfstub')
ALIAS = '_reference_tf_raft'
_STUB_MODULES = ('tensorflow', 'tensorflow_addons')


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'tf_raft', 'model.py'))


@contextlib.contextmanager
def _stub_on_path():
    """Stub importable, byte-code writing off; afterwards the stub leaves ``sys.modules`` again so that nothing else in the
    process mistakes it for TensorFlow (the loaded reference modules keep their own references)."""
    saved = {k: v for k, v in sys.modules.items() if k.split('.')[0] in _STUB_MODULES}
    for k in saved:
        del sys.modules[k]
    dont_write = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    sys.path.insert(0, STUB_DIR)
    try:
        yield
    finally:
        sys.path.remove(STUB_DIR)
        sys.dont_write_bytecode = dont_write
        for k in [k for k in sys.modules if k.split('.')[0] in _STUB_MODULES]:
            del sys.modules[k]
        sys.modules.update(saved)


_loaded = None


def load_reference():
    """-> namespace with the reference's modules: .model, .corr, .update, .extractor, .losses, .tf (the stub they run on)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise FileNotFoundError(f'reference tree not found under {REFERENCE_ROOT}')
    pkg_dir = os.path.join(REFERENCE_ROOT, 'tf_raft')
    with _stub_on_path():
        spec = importlib.util.spec_from_file_location(ALIAS, os.path.join(pkg_dir, '__init__.py'),
                                                      submodule_search_locations=[pkg_dir])
        pkg = importlib.util.module_from_spec(spec)
        sys.modules[ALIAS] = pkg
        spec.loader.exec_module(pkg)
        import types
        ns = types.SimpleNamespace(
            model=importlib.import_module(f'{ALIAS}.model'),
            corr=importlib.import_module(f'{ALIAS}.layers.corr'),
            update=importlib.import_module(f'{ALIAS}.layers.update'),
            extractor=importlib.import_module(f'{ALIAS}.layers.extractor'),
            losses=importlib.import_module(f'{ALIAS}.losses.losses'),
            tf=importlib.import_module('tensorflow'),
            tfa=importlib.import_module('tensorflow_addons'))
    for m in (ns.model, ns.corr, ns.update, ns.extractor, ns.losses):
        assert os.path.realpath(m.__file__).startswith(os.path.realpath(REFERENCE_ROOT)), m.__file__
    _loaded = ns
    return ns


@contextlib.contextmanager
def floatx(dtype):
    """Run the same reference source with ``tf.float32`` rebound (e.g. to torch.float64): the reference names its dtype only
    through that attribute (corr.py:81-82, 133, 162)."""
    tf = load_reference().tf
    old = tf.float32
    tf.float32 = dtype
    try:
        yield
    finally:
        tf.float32 = old


def assign_weights(layer, weights, dtype=torch.float32, prefix=''):
    """Put Keras-layout arrays into the stub layers of a reference model by attribute path.  Every array must be consumed and
    every parameterised layer must be fed: a path mismatch between ``tf_raft_amd.weights`` and the reference's attribute
    names raises here."""
    used = set()
    for path, sub in layer.named_layers(prefix):
        kind = type(sub).__name__
        want = {'Conv2D': ('kernel', 'bias'), 'InstanceNormalization': ('gamma', 'beta'),
                'BatchNormalization': ('gamma', 'beta', 'moving_mean', 'moving_variance')}.get(kind, ())
        for k in want:
            key = f'{path}/{k}'
            if key not in weights:
                raise KeyError(f'no array for {key} ({kind})')
            sub._vars[k] = torch.as_tensor(np.asarray(weights[key])).to(dtype)
            used.add(key)
    missing = [k for k in weights if k not in used and (not prefix or k.startswith(prefix + '/'))]
    if missing:
        raise KeyError(f'arrays not consumed by any reference layer: {missing[:5]} ...')
    return layer


def build_model(variant, weights, dtype=torch.float32, **kwargs):
    """The reference's own ``RAFT(**kwargs)`` / ``SmallRAFT(**kwargs)`` (model.py:10-30, 173-188) carrying ``weights``."""
    ref = load_reference()
    cls = {'raft': ref.model.RAFT, 'small': ref.model.SmallRAFT}[variant]
    return assign_weights(cls(**kwargs), weights, dtype)


def forward(model, image1, image2, training=False, dtype=torch.float32):
    """``model([image1, image2], training=...)`` -> list of numpy arrays (model.py:68-109)."""
    tf = load_reference().tf
    with torch.no_grad():
        out = model([tf.convert_to_tensor(np.asarray(image1), dtype=dtype),
                     tf.convert_to_tensor(np.asarray(image2), dtype=dtype)], training=training)
    return [np.asarray(o.detach().to(torch.float32)) for o in out]
