cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "pipelined or pending or weights_replaced" > gpurun_out/r12c_pipe_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r12c_pipe_tests.log
tail -5 gpurun_out/r12c_pipe_tests.log
o=gpurun_out/r12c_option_sweep_lanes3.txt
timeout 900 python tools/option_sweep.py --lanes 3 --reps 20 \
  "RAFT_CONVC2_KS=1" "RAFT_CONVC2_KS=2" "RAFT_CONVF2_KS=1" "RAFT_CONVF2_KS=2" "RAFT_CONV_WINO4=15" "RAFT_CONV_WINO4=9" "RAFT_CONV_WINO4=8" "RAFT_CONV_WINO4=0" \
  "RAFT_MASK_FUSED=0" "RAFT_LOOKUP_FUSED=0" "RAFT_WINO_TNW=1" "RAFT_WINO_TNW=2" "RAFT_GRU_WINO4=0" "RAFT_WINO4_KS=1" "RAFT_WINO4_KS=2" \
  "RAFT_ENC_WINO4=0" "RAFT_ENC_WINO4=7" "RAFT_CORR_XCD=0" "RAFT_WINO_CK=1" "RAFT_WINO_CK=2" > $o 2>&1
cat $o
for l in 2 3 4 5; do timeout 300 python tools/option_sweep.py --lanes $l --reps 20 2>&1 | grep pairs | sed "s/^/lanes=$l /" ; done > gpurun_out/r12c_lanes_count.txt
cat gpurun_out/r12c_lanes_count.txt
