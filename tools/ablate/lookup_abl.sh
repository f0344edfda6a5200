#!/bin/bash
# Build (here, cross-compiled) or run (GPU box) the lookup ablation set.  usage: lookup_abl.sh build | run [B]
cd "$(dirname "$0")"
ABLS="0 1 2 4 8 3 5 6 7 15 16"
if [ "$1" = build ]; then
  for a in $ABLS; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -w -DRAFT_LOOKUP_ABL=$a -x hip lookup_abl.hip -o ablate_lookup_$a & done; wait
else
  B=${2:-4}
  for a in $ABLS; do ./ablate_lookup_$a $B 300 1 0; done
  ./ablate_lookup_0 $B 300 0 0
  ./ablate_lookup_0 $B 100 1 1; ./ablate_lookup_1 $B 100 1 1; ./ablate_lookup_4 $B 100 1 1
  ./ablate_lookup_0 1 300 1 0; ./ablate_lookup_0 8 300 1 0; ./ablate_lookup_0 16 200 1 0
fi
