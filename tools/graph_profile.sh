#!/bin/bash
# Kernel traces of the three-stream loop launched from streams vs replayed as a hipGraph at B = 1 / 2 / 4 (GPU box, repo root).
tag=${1:-r07n}
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp
out=$root/gpurun_out/${tag}_graph_replay_trace.txt
: > $out
for b in 1 2 4; do
  for g in 0 1; do
    RAFT_LOOP_GRAPH=$g timeout 300 rocprofv3 --kernel-trace -f csv -d $root/gpurun_out/${tag}_graph_b${b}_g$g -o gp -- python $root/tools/graph_probe.py $b 4 > $root/gpurun_out/${tag}_graph_b${b}_g$g.log 2>&1
    grep "ms per call" $root/gpurun_out/${tag}_graph_b${b}_g$g.log >> $out
    t=$(ls $root/gpurun_out/${tag}_graph_b${b}_g$g/*kernel_trace.csv 2>/dev/null | head -1)
    [ -n "$t" ] && python $root/tools/graph_trace.py "$t" "B=$b RAFT_LOOP_GRAPH=$g" >> $out 2>&1
    [ -n "$t" ] && [ $b != 2 ] && rm -f "$t"
  done
done
cat $out
