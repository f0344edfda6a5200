from tf_raft_amd.layers.extractor import BasicEncoder, SmallEncoder  # noqa: F401
