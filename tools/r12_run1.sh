cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "pipelined or pending or weights_replaced" > gpurun_out/r12a_pipe_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r12a_pipe_tests.log
tail -5 gpurun_out/r12a_pipe_tests.log
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python tools/lanes_ab.py --batch 4 --rounds 4 serial 1:1 2:1 2:0 3:0 4:0 3:1 serial >> gpurun_out/r12a_lanes_ab.txt 2>&1
done
GPU_MAX_HW_QUEUES=8 timeout 600 python tools/lanes_ab.py --batch 8 --rounds 3 serial 1:1 2:1 2:0 3:0 >> gpurun_out/r12a_lanes_ab.txt 2>&1
GPU_MAX_HW_QUEUES=8 timeout 600 python tools/lanes_ab.py --batch 1 --rounds 3 serial 1:1 2:1 4:1 4:0 >> gpurun_out/r12a_lanes_ab.txt 2>&1
cat gpurun_out/r12a_lanes_ab.txt
bash tools/lookup_ta_tcp.sh r12a 8 2>&1 | tail -8
