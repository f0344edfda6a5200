"""CPU oracle for the RAFT forward-prediction hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU / NumPy, fp32 with an fp64 switch) of the
reference's algorithm, op for op, with TensorFlow 2.3 / tensorflow-addons 0.11.1
semantics written out by hand.  It exists so that the HIP path can be checked for
parity.  It is NOT part of the product:

  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
    ``bench.py`` may import it;
  * the product package (``tf_raft_amd``) never imports it and has no CPU fallback.

Why a restatement: the reference is pure Python on top of TensorFlow 2.3.0 and
tensorflow-addons 0.11.1 (reference poetry.lock:568-571, 595-598), neither of which is
installed or installable here (no network, Python 3.10).  The arithmetic therefore
lives in third-party code that is absent from ``/root/reference``; the oracle restates
the published semantics of each op and anchors on the reference's call sites.

PARITY PINNING STATUS
  * pinned against the reference's own known-answer tests: extract_patches /
    depth_to_space ordering arrays (reference tests/test_model.py:16-26), the
    end-point-error / sequence-loss known answers (tests/losses/test_losses.py:7-67)
    and the ``bilinear_sampler == resampler`` property for interior non-integer
    coordinates (tests/layers/test_corr.py:15-27, with torch ``grid_sample`` standing in
    for the absent ``tfa.image.resampler``).  See ``tests/test_oracle_pins.py``.
  * the reference holds NO golden tensor for the whole forward pass (its model tests
    check shapes only, tests/test_model.py:44-77) and the reference cannot be executed
    here, so the end-to-end forward numerics are **parity unpinned**: the oracle is the
    best available statement of "what the reference computes", cross-checked piecewise
    against independent implementations (torch ops) in ``tests/``.
"""
from .model import RAFT, SmallRAFT  # noqa: F401
from .corr import CorrBlock, bilinear_sampler, coords_grid, upflow8  # noqa: F401
from .losses import end_point_error, sequence_loss  # noqa: F401
