export TMPDIR=/tmp
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x > $out/r11a_pytest_gpu.log 2>&1; echo "rc=$?" >> $out/r11a_pytest_gpu.log
tail -5 $out/r11a_pytest_gpu.log
for cfg in "RAFT_PIPELINE=0" "RAFT_PIPELINE=1" "RAFT_PIPELINE=1 RAFT_LOOP_PRIORITY=1" "RAFT_PIPELINE=1 GPU_MAX_HW_QUEUES=8"; do
  env $cfg timeout 200 python tools/pipeline_ab.py 4 8 1 2>&1 | grep "B=" | tee -a $out/r11a_pipeline_ab.txt
done
cd /tmp
for p in 0 1; do
  RAFT_PIPELINE=$p timeout 200 rocprofv3 --kernel-trace -f csv -d $GRAFT_REPO_ROOT/$out/r11a_tl$p -o tl -- python $GRAFT_REPO_ROOT/tools/pipeline_ab.py 4 > /dev/null 2>&1
  t=$(ls $GRAFT_REPO_ROOT/$out/r11a_tl$p/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$t" ] && python $GRAFT_REPO_ROOT/tools/pipeline_timeline.py $t "RAFT_PIPELINE=$p B=4" | tee -a $GRAFT_REPO_ROOT/$out/r11a_pipeline_timeline.txt
  rm -rf $GRAFT_REPO_ROOT/$out/r11a_tl$p
done
