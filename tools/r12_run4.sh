cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r12d_option_combos_lanes3.txt
timeout 900 python tools/option_sweep.py --lanes 3 --reps 20 \
  "A: RAFT_WINO_TNW=2 RAFT_CONV_WINO4=15" "B: RAFT_WINO_TNW=2 RAFT_CONV_WINO4=15 RAFT_WINO4_KS=1" "C: RAFT_WINO_TNW=2 RAFT_CONVC2_KS=1" \
  "D: RAFT_WINO_TNW=2 RAFT_WINO4_KS=1" "E: RAFT_WINO_TNW=2 RAFT_CONV_WINO4=15 RAFT_WINO4_KS=1 RAFT_LOOKUP_FUSED=0" "F: RAFT_CONV_WINO4=15 RAFT_WINO4_KS=1" \
  "G: RAFT_WINO_TNW=2 RAFT_CONV_WINO4=15 RAFT_CONVC2_KS=1" "H: RAFT_WINO_TNW=2 RAFT_CONV_WINO4=15 RAFT_WINO4_KS=1 RAFT_MASK_FUSED=0" > $o 2>&1
cat $o
for l in 2 4 5; do timeout 300 python tools/option_sweep.py --lanes $l --reps 20 "B: RAFT_WINO_TNW=2 RAFT_CONV_WINO4=15 RAFT_WINO4_KS=1" 2>&1 | grep pairs | sed "s/^/lanes=$l /" ; done > gpurun_out/r12d_lanes_count_B.txt
cat gpurun_out/r12d_lanes_count_B.txt
for b in 1 2 8; do timeout 300 python tools/option_sweep.py --lanes 3 --batch $b --reps 20 "B: RAFT_WINO_TNW=2 RAFT_CONV_WINO4=15 RAFT_WINO4_KS=1" "RAFT_WINO_TNW=2" "RAFT_CONV_WINO4=15" "RAFT_WINO4_KS=1" "RAFT_GRU_WINO4=15" 2>&1 | grep pairs ; done > gpurun_out/r12d_other_batches.txt
cat gpurun_out/r12d_other_batches.txt
