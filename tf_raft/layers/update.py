from tf_raft_amd.layers.update import BasicUpdateBlock, SmallUpdateBlock  # noqa: F401
