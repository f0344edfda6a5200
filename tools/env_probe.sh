#!/bin/bash
# HIP runtime knobs that affect launch latency / stream concurrency (GPU box): B=1 and B=4 latency under each setting.
cd ${GRAFT_REPO_ROOT:-.}
run() { echo "== $*"; env "$@" python tools/batch_probe_min.py; }
for i in 1 2 3; do
run GPU_MAX_HW_QUEUES=4
run GPU_MAX_HW_QUEUES=8
done
run GPU_MAX_HW_QUEUES=16
