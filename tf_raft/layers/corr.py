from tf_raft_amd.layers.corr import CorrBlock, bilinear_sampler, coords_grid, upflow8  # noqa: F401
