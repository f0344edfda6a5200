"""tf_raft_amd -- MI355X (gfx950) native RAFT optical-flow forward prediction.

Drop-in for the forward path of ``daigo0927/tf-raft`` (``tf_raft.model.RAFT`` / ``SmallRAFT``):
Python host code -> ctypes -> ``libraft_hip.so`` (hand-written HIP kernels, C ABI in
``include/raft_hip.h``); PyTorch-ROCm only owns device memory, streams and ``torch.distributed``.
See DESIGN.md.
"""
from .model import RAFT, SmallRAFT  # noqa: F401

__all__ = ['RAFT', 'SmallRAFT']
__version__ = '0.1.0'
