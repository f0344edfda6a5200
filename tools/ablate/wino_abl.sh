#!/bin/bash
# Build (here, cross-compiled) or run (GPU box) the F(2x2,3x3) Winograd experiment set.  usage: wino_abl.sh build | run
cd "$(dirname "$0")"
# name : extra defines
VARS="base: pair:-DRAFT_WINO_PAIR=1 xcd:-DRAFT_WINO_XCD=1 pairxcd:-DRAFT_WINO_PAIR=1+-DRAFT_WINO_XCD=1 base_m:-DRAFT_WINO_ABL=15 pair_m:-DRAFT_WINO_PAIR=1+-DRAFT_WINO_ABL=15 base_nw:-DRAFT_WINO_ABL=1 xcd_nw:-DRAFT_WINO_XCD=1+-DRAFT_WINO_ABL=1"
if [ "$1" = build ]; then
  for v in $VARS; do n=${v%%:*}; d=$(echo ${v#*:} | tr '+' ' ');
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -w $d -x hip wino_abl.hip ../../tf_raft_amd/csrc/host_util.hip -o ablate_wino_$n & done; wait
  ls -la ablate_wino_*
else
  for layer in "128 512" "256 192" "256 128" "128 64"; do
    for v in $VARS; do n=${v%%:*}; echo -n "$n: "; ./ablate_wino_$n $layer 4 200; done
  done
  for v in base pair xcd pairxcd; do echo -n "$v B=8: "; ./ablate_wino_$v 128 512 8 100; done
fi
