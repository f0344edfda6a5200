cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r12i_lanes_by_batch.txt
: > $o
for b in 1 2; do for l in 2 3 4 5 6 8; do timeout 200 python tools/lanes_ab.py --batch $b --rounds 3 $l:0 2>/dev/null | grep pairs >> $o; done; done
for b in 3 6 8; do for l in 2 3 4 5; do timeout 200 python tools/lanes_ab.py --batch $b --rounds 3 $l:0 2>/dev/null | grep pairs >> $o; done; done
for l in 1 2 3; do timeout 300 python tools/lanes_ab.py --batch 16 --rounds 2 --steps 10 $l:0 2>/dev/null | grep pairs >> $o; done
timeout 300 python tools/lanes_ab.py --batch 16 --rounds 2 --steps 10 serial 1:1 2>/dev/null | grep pairs >> $o
RAFT_LOOP_PRIORITY=1 timeout 200 python tools/lanes_ab.py --batch 4 --rounds 3 3:0 2>/dev/null | grep pairs >> $o
timeout 200 python tools/lanes_ab.py --batch 4 --rounds 3 3:0 2>/dev/null | grep pairs >> $o
cat $o
