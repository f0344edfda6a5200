"""Import-path shim for reference ``tf_raft/losses/__init__.py`` (train_sintel.py:9: ``from tf_raft.losses import ...``)."""
from .losses import EndPointError, end_point_error, sequence_loss  # noqa: F401
