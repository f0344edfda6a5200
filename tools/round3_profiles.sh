#!/bin/bash
# Round-3 profile set (GPU box, repo root): single-stream kernel traces of the prediction loop at B = 4 and B = 8 (kernel
# durations of the stand-alone lookup in the loop at both batches), the in-loop PMC traffic pass at B = 8, and kernel traces of
# the three-stream loop launched from streams vs replayed as a hipGraph at B = 1 / 2.
tag=${1:-r07g}
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp
for b in 4 8; do
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $root/gpurun_out/${tag}_ss_b$b -o ss -- python $root/tools/pmc_loop.py $b 24 > $root/gpurun_out/${tag}_ss_b$b.log 2>&1
  f=$(ls $root/gpurun_out/${tag}_ss_b$b/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $root/gpurun_out/${tag}_single_stream_b${b}_rocprofv3_kernel_stats.csv
  rm -f $root/gpurun_out/${tag}_ss_b$b/*kernel_trace.csv
done
for b in 1 2; do
  for g in 0 1; do
    RAFT_LOOP_GRAPH=$g timeout 300 rocprofv3 --kernel-trace -f csv -d $root/gpurun_out/${tag}_graph_b${b}_g$g -o gp -- python $root/tools/graph_probe.py $b 4 > $root/gpurun_out/${tag}_graph_b${b}_g$g.log 2>&1
    t=$(ls $root/gpurun_out/${tag}_graph_b${b}_g$g/*kernel_trace.csv 2>/dev/null | head -1)
    [ -n "$t" ] && python $root/tools/graph_trace.py "$t" "B=$b RAFT_LOOP_GRAPH=$g" >> $root/gpurun_out/${tag}_graph_replay_trace.txt 2>&1
    grep "ms per call" $root/gpurun_out/${tag}_graph_b${b}_g$g.log >> $root/gpurun_out/${tag}_graph_replay_trace.txt
    rm -f "$t"
  done
done
cd $root
bash tools/pmc_traffic.sh $tag 8 > gpurun_out/${tag}_pmc_traffic.log 2>&1
cat gpurun_out/${tag}_graph_replay_trace.txt
head -20 gpurun_out/${tag}_single_stream_b8_rocprofv3_kernel_stats.csv | cut -c1-150
