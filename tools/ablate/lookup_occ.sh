#!/bin/bash
# Workgroups-per-CU sweep of the lookup kernel through unused dynamic LDS (GPU box).  usage: lookup_occ.sh [B]
cd "$(dirname "$0")"
B=${1:-4}
for pad in 0 4096 7168 11264 16384 24576 37888 65536; do
  for th in 0 1; do echo -n "pad=$pad "; RAFT_LOOKUP_LDS_PAD=$pad ./ablate_lookup_0 $B 200 1 $th; done
done
