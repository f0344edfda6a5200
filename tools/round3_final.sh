#!/bin/bash
# Round-3 closing set (GPU box, repo root): full GPU tests, smoke, the default bench line, rocprofv3 kernel stats of the same
# command, single-stream kernel stats at 4 / 8 pairs, batch sweep, the other configurations, a 2-rank gloo dry run.
tag=${1:-r08z}
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 1500 python -m pytest tests -m gpu -q > $out/${tag}_pytest_gpu.log 2>&1; echo "rc=$?" >> $out/${tag}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1; echo "rc=$?" >> $out/${tag}_smoke.log
timeout 400 python bench.py > $out/${tag}_bench_b4.log 2>&1
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $out/${tag}_prof -o bench -- python $root/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $out/${tag}_bench_b4_under_rocprof.log 2>&1
f=$(ls $out/${tag}_prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_bench_b4_rocprofv3_kernel_stats.csv
rm -f $out/${tag}_prof/*kernel_trace.csv
for b in 4 8; do
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out/${tag}_ss_b$b -o ss -- python $root/tools/pmc_loop.py $b 24 > $out/${tag}_ss_b$b.log 2>&1
  f=$(ls $out/${tag}_ss_b$b/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_single_stream_b${b}_rocprofv3_kernel_stats.csv
  rm -f $out/${tag}_ss_b$b/*kernel_trace.csv
done
timeout 200 rocprofv3 --kernel-trace -f csv -d $out/${tag}_tl -o tl -- python $root/tools/graph_probe.py 4 3 > $out/${tag}_tl.log 2>&1
t=$(ls $out/${tag}_tl/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$t" ]; then python $root/tools/graph_trace.py $t "B=4 final" > $out/${tag}_loop_timeline_b4.txt 2>&1; python $root/tools/step_timeline.py $t all | awk 'NR>=72 && NR<=104' >> $out/${tag}_loop_timeline_b4.txt; rm -f $t; fi
cd $root
timeout 300 python tools/batch_sweep.py 8 2>&1 | grep "^B=" > $out/${tag}_batch_sweep.txt
timeout 400 python tools/config_bench.py > $out/${tag}_config_bench.txt 2>&1
RAFT_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 > $out/${tag}_bench_gpus2_gloo.log 2>&1
timeout 300 python bench.py --train --steps 10 --warmup 3 > $out/${tag}_bench_train.log 2>&1
tail -3 $out/${tag}_pytest_gpu.log; tail -2 $out/${tag}_smoke.log; tail -1 $out/${tag}_bench_b4.log | cut -c1-300; cat $out/${tag}_batch_sweep.txt; tail -4 $out/${tag}_config_bench.txt | cut -c1-200; tail -1 $out/${tag}_bench_gpus2_gloo.log | cut -c1-300; tail -1 $out/${tag}_bench_train.log | cut -c1-200
