cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_model.py -q -x -s -k "pipelined or pending or weights_replaced or benchmarked_batches or predict_step" 2>&1 | grep -E "\[parity\]|passed|failed|Error|error|assert" | sed "s/^\.*//" > gpurun_out/r12e_lane_tests.log
tail -25 gpurun_out/r12e_lane_tests.log
o=gpurun_out/r12e_hint_ab.txt
: > $o
for sc in none loop all; do
  for b in 4 8 1 2; do
    RAFT_LANE_SHAPES=$sc timeout 300 python tools/option_sweep.py --lanes 3 --batch $b --reps 20 2>&1 | grep pairs | sed "s/^/shapes=$sc /" >> $o
  done
done
RAFT_LANE_SHAPES=none timeout 300 python tools/option_sweep.py --lanes 3 --batch 4 --reps 20 "B: RAFT_WINO_TNW=2 RAFT_CONV_WINO4=15 RAFT_WINO4_KS=1" 2>&1 | grep pairs | sed "s/^/shapes=none /" >> $o
cat $o
