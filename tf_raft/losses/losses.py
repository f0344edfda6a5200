"""Import-path shim for reference ``tf_raft/losses/losses.py``: device reductions of ``tf_raft_amd.losses``."""
from tf_raft_amd.losses import EndPointError, end_point_error, sequence_loss  # noqa: F401
