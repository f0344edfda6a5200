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
  * pinned to the reference's OWN SOURCE (round 5): ``oracle/reference_runner.py`` executes the unmodified
    ``/root/reference/tf_raft/{model.py, layers/corr.py, layers/update.py, layers/extractor.py, losses/losses.py}`` on the stand-in
    ``tensorflow`` of ``oracle/tfstub`` (its primitives are ``oracle/tf_ops.py``) and ``tests/test_reference_under_stub.py``
    asserts ``oracle.RAFT / SmallRAFT`` equal to it BIT FOR BIT (whole forward incl. (1,448,512,3), training and inference, fp32
    and fp64, and every piece in isolation); the reference's own tests/test_model.py, tests/layers/test_corr.py and
    tests/losses/test_losses.py pass under the same stub.  Outputs of those runs are committed as
    ``tests/golden/reference_forward_golden.npz`` and the GPU tests compare the HIP path with them directly.
  * pinned against the reference's own known-answer tests: extract_patches / depth_to_space ordering arrays (reference
    tests/test_model.py:16-26), the end-point-error / sequence-loss known answers (tests/losses/test_losses.py:7-67) and the
    ``bilinear_sampler == resampler`` property (tests/layers/test_corr.py:15-27).  See ``tests/test_oracle_pins.py``.
  * still recalled, not executed (TensorFlow 2.3 cannot be installed here): the semantics of the TF PRIMITIVES restated in
    ``oracle/tf_ops.py`` -- Conv2D 'same' padding (asymmetric for stride 2), Keras BatchNormalization / tfa
    InstanceNormalization epsilon 1e-3 and biased variance, ``tf.image.resize`` half-pixel bilinear, ``depth_to_space`` DCR order,
    ``extract_patches`` (ky, kx, c) order, ``avg_pool2d`` VALID flooring, ``gather_nd(batch_dims=1)``.  Each is cross-checked
    against an independent torch implementation (``tests/test_oracle_pins.py``); the two ordering ones also against the
    reference's known-answer arrays.  The reference holds no golden tensor produced by real TensorFlow, so "parity unpinned"
    remains true for exactly that list and for nothing above it.
"""
from .model import RAFT, SmallRAFT  # noqa: F401
from .corr import CorrBlock, bilinear_sampler, coords_grid, upflow8  # noqa: F401
from .losses import end_point_error, sequence_loss  # noqa: F401
