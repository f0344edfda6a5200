#!/bin/bash
# Build (here, cross-compiled) or run (GPU box) the F(4x4,3x3) Winograd ablation set.  usage: wino4_abl.sh build | run
cd "$(dirname "$0")"
# name : extra defines
VARS="base: nw:-DRAFT_WINO4_ABL=1 ns1:-DRAFT_WINO4_ABL=2 nh:-DRAFT_WINO4_ABL=4 nst:-DRAFT_WINO4_ABL=8 ns2:-DRAFT_WINO4_ABL=16 mfma:-DRAFT_WINO4_ABL=23 mfma_nst:-DRAFT_WINO4_ABL=31 pf3:-DRAFT_WINO4_PF=3 pf6:-DRAFT_WINO4_PF=6 nosnop:-DRAFT_WINO4_NOSNOP nosb:-DRAFT_WINO4_NOSB lag6:-DRAFT_WINO4_LAG=6 lag20:-DRAFT_WINO4_LAG=20 xcd:-DRAFT_WINO4_XCD=1"
if [ "$1" = build ]; then
  for v in $VARS; do n=${v%%:*}; d=$(echo ${v#*:} | tr '+' ' ');
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -w $d -x hip wino4_abl.hip ../../tf_raft_amd/csrc/host_util.hip -o ablate_wino4_$n & done; wait
  ls -la ablate_wino4_*
else
  for layer in "128 512" "128 64" "256 192"; do
    for v in $VARS; do n=${v%%:*}; echo -n "$n: "; ./ablate_wino4_$n $layer 4 100; done
  done
  for v in base mfma nosb; do echo -n "$v B=8: "; ./ablate_wino4_$v 128 512 8 50; done
fi
