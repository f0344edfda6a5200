import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The CPU oracle is an eager torch graph of small ops: on the GPU boxes' 256 logical CPUs torch's default (128 threads) runs
    # one (1,448,512,3) forward in 6 s, 16 threads in 0.65 s (bench.py probes the same: cpu_baseline.cores).  The parity tests
    # run it dozens of times.
    try:
        import torch
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1, torch.get_num_threads())))
    except Exception:   # noqa: BLE001
        pass


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:   # noqa: BLE001
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def rng():
    return np.random.default_rng(1234)


def report(name, **vals):
    """Print measured errors so that `pytest -rA` / `-s` logs carry the numbers, not just PASS."""
    print('[parity] ' + name + ' ' + ' '.join(f'{k}={v:.3e}' if isinstance(v, float) else f'{k}={v}'
                                             for k, v in vals.items()))


class _RaftOptions:
    """Tuning switches of the library for the duration of one test (include/raft_hip.h: raft_set_option)."""

    def __init__(self):
        self._touched = []

    def set(self, name, value):
        from tf_raft_amd import _ffi
        _ffi.set_option(name, value)
        self._touched.append(name)

    def restore(self):
        from tf_raft_amd import _ffi
        for name in self._touched:
            _ffi.set_option(name, None)          # back to the load-time (environment) state
        self._touched = []


@pytest.fixture
def raft_opt():
    o = _RaftOptions()
    yield o
    o.restore()
