from tf_raft_amd.model import RAFT, SmallRAFT  # noqa: F401
