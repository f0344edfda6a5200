#!/bin/bash
# Per-flag ablation of the F(2x2,3x3) kernel.  usage: wino_abl2.sh build | run
# RAFT_WINO_ABL bits: 1 no in-loop weight traffic, 2 no in-loop LDS patch reads, 4 no halo staging in the loop, 8 no epilogue stores, 16 no input transform
cd "$(dirname "$0")"
ABLS="0 1 2 4 8 16 3 7 15 23 31"
if [ "$1" = build ]; then
  for a in $ABLS; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -w -DRAFT_WINO_ABL=$a -x hip wino_abl.hip ../../tf_raft_amd/csrc/host_util.hip -o ablate_wino_$a & done; wait
else
  for layer in "128 512" "256 192"; do for a in $ABLS; do ./ablate_wino_$a $layer 4 200; done; done
fi
