"""Workload of the PMC passes (tools/pmc_traffic.sh): ONE full forward (encoders, volume build, state preparation) and
then the SINGLE-STREAM prediction loop, `iters` iterations, at batch B -- first with the product default (lookup fused into
convc1, mask.2 into the upsampling: those kernels are attributed by name), then with RAFT_LOOKUP_FUSED=0 RAFT_MASK_FUSED=0, where the launch order inside an iteration is
fixed (bench.py STAGES), which is how tools/pmc_traffic.py attributes those dispatches to stages.
usage: python tools/pmc_loop.py <batch> <iters>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tf_raft_amd  # noqa: E402

B, iters = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device('cuda', 0)
gen = torch.Generator(device=dev)
gen.manual_seed(1000)
i1 = torch.rand((B, 448, 512, 3), device=dev, generator=gen) * 255.0
i2 = torch.rand((B, 448, 512, 3), device=dev, generator=gen) * 255.0
from tf_raft_amd import _ffi  # noqa: E402
# the launch shapes of the product: a multi-lane pipelined call launches its loops under raft_set_thread_concurrency(lanes)
from tf_raft_amd.model import DEFAULT_LANES  # noqa: E402
conc = int(os.environ.get('RAFT_LOOP_CONCURRENCY', DEFAULT_LANES))
model = tf_raft_amd.RAFT(iters_pred=iters, overlap=False, pipeline=False, loop_concurrency=conc)
out = model([i1, i2])                            # product default: lookup fused into convc1 (attributed by kernel name)
torch.cuda.synchronize()
_ffi.set_option('RAFT_LOOKUP_FUSED', 0)          # then the two-kernel loops: 14 kernels per iteration, attributed by order
_ffi.set_option('RAFT_MASK_FUSED', 0)            # (mask.2 and the upsampling as their own kernels too)
out = model([i1, i2])
torch.cuda.synchronize()
if len(sys.argv) > 3 and sys.argv[3] == 'probe':   # tools/instruction_mix.sh: the MFMA-only probe as the counters' calibration point
    from tf_raft_amd import _dev
    buf = torch.empty(512 * 256, device=dev)
    _ffi.check(_dev.lib().raft_mfma_probe_f32(_dev.ptr(buf), 512, 256, _dev.stream_ptr()), 'mfma_probe')
    torch.cuda.synchronize()
print('done', float(out[-1].abs().max()))
