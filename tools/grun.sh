#!/bin/bash
# Build the library HERE (hipcc cross-compiles; the .so travels with the snapshot), then run a command on the GPU box.
# usage: tools/grun.sh <timeout-seconds> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "from tf_raft_amd import build; build.build_library(verbose=False)"
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
