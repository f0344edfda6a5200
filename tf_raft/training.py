"""Import-path shim for reference ``tf_raft/training.py`` (train_sintel.py:11: ``from tf_raft.training import VisFlowCallback,
first_cycle_scaler``): the learning-rate scale functions; ``VisFlowCallback`` (Keras callback plumbing) is out of scope."""
from tf_raft_amd.training import first_cycle_scaler, inverse_scaler  # noqa: F401
