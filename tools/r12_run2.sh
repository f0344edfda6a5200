cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r12b_lanes_ab_per_process.txt
: > $o
for q in 4 8 16; do
  for s in serial 1:1 2:1 3:1 2:0 3:0 4:0 6:0; do
    GPU_MAX_HW_QUEUES=$q timeout 300 python tools/lanes_ab.py --batch 4 --rounds 3 $s 2>/dev/null | grep pairs >> $o
  done
done
for s in serial 1:1 2:1 2:0 3:0 4:0; do
  timeout 300 python tools/lanes_ab.py --batch 4 --rounds 3 $s 2>/dev/null | grep pairs >> $o
done
for s in serial 1:1 2:1 2:0 3:0; do
  GPU_MAX_HW_QUEUES=8 timeout 300 python tools/lanes_ab.py --batch 8 --rounds 3 $s 2>/dev/null | grep pairs >> $o
  GPU_MAX_HW_QUEUES=8 timeout 300 python tools/lanes_ab.py --batch 1 --rounds 3 $s 2>/dev/null | grep pairs >> $o
done
cat $o
