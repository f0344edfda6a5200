"""Import-path shim: ``from tf_raft.model import RAFT`` resolves to the MI355X implementation
(``tf_raft_amd``), so code written against daigo0927/tf-raft's forward API runs unchanged."""
