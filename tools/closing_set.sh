#!/bin/bash
# Closing set of a state of the tree (GPU box, repo root).  Everything bench.py's evidence fields point to is regenerated here
# from the library that is loaded in the same call:
#   <tag>_pytest_gpu.log, <tag>_smoke.log, <tag>_parity_tests.log   the GPU suite, smoke(), the model parity tests with their errors (skipped with SKIP_TESTS=1)
#   <tag>_bench_b4.log                              the default bench line (BASELINE configs[1])
#   <tag>_bench_b4_rocprofv3_kernel_stats.csv       rocprofv3 --kernel-trace --stats of the same command (product loop)
#   <tag>_single_stream_b{4,8}_rocprofv3_kernel_stats.csv + kernel_durations.json   single-stream loop, per-stage durations
#   <tag>_pmc_per_kernel.csv + pmc_traffic.json     FETCH_SIZE / WRITE_SIZE passes at 8 pairs      (skipped with SKIP_PMC=1)
#   <tag>_loop_timeline_b4.txt                      kernel timeline of the product loop
#   <tag>_lanes_ab.txt                              round 6: serial / one lane / 2-4 lanes, one process per setting; launch-shape hint scopes
#   <tag>_lanes_timeline_b4.txt                     steady-state kernel timeline of the 3-lane schedule
#   <tag>_kernel_concurrent.json                    the dominant kernel on 3 streams at once (rocprofv3 side of bench.py `roofline`)
#   <tag>_config_bench.txt, <tag>_bench_train.log   the other BASELINE configurations, the training step
#   <tag>_instruction_mix.txt                       MFMA / VALU / LDS instructions per wave per kernel (tools/instruction_mix.sh)
# usage: bash tools/closing_set.sh <tag>
tag=${1:-r10a}
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
if [ -z "${SKIP_TESTS:-}" ]; then
  timeout 1800 python -m pytest tests -m gpu -q -x > $out/${tag}_pytest_gpu.log 2>&1; echo "rc=$?" >> $out/${tag}_pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1; echo "rc=$?" >> $out/${tag}_smoke.log
  # the parity tests once more with their measured errors on stdout
  timeout 900 python -m pytest tests/test_gpu_model.py -q -s -k "north_star or jump_regime or mid_regime or pipelined or pending or weights_replaced" 2>&1 | grep -E "\[parity\]|passed|failed" | sed "s/^\.*//" > $out/${tag}_parity_tests.log
fi
cd /tmp
for b in 4 8 16; do
  timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out/${tag}_ss_b$b -o ss -- python $root/tools/pmc_loop.py $b 24 > $out/${tag}_ss_b$b.log 2>&1
  f=$(ls $out/${tag}_ss_b$b/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_single_stream_b${b}_rocprofv3_kernel_stats.csv
done
t4=$(ls $out/${tag}_ss_b4/*kernel_trace.csv 2>/dev/null | head -1); t8=$(ls $out/${tag}_ss_b8/*kernel_trace.csv 2>/dev/null | head -1); t16=$(ls $out/${tag}_ss_b16/*kernel_trace.csv 2>/dev/null | head -1)
cd $root
[ -n "$t4" ] && [ -n "$t8" ] && python tools/kernel_durations.py $out/${tag}_kernel_durations.json 4=$t4 8=$t8 ${t16:+16=$t16} > $out/${tag}_kernel_durations.txt 2>&1
rm -rf $out/${tag}_ss_b4 $out/${tag}_ss_b8 $out/${tag}_ss_b16
if [ -z "${SKIP_PMC:-}" ]; then
  bash tools/pmc_traffic.sh $tag 8 > $out/${tag}_pmc_traffic.txt 2>&1
  rm -rf $out/${tag}_pmc
  # the same passes at the bench line's own batch (4 pairs: a 275 MB volume, which straddles the 256 MiB Infinity Cache)
  bash tools/pmc_traffic.sh ${tag}b4 4 > $out/${tag}_pmc_traffic_b4.txt 2>&1
  rm -rf $out/${tag}b4_pmc
  [ -f $out/${tag}b4_pmc_traffic.json ] && cp $out/${tag}b4_pmc_traffic.json profiles/pmc_traffic_b4.json
fi
# the bench line reads the two summaries just made (the same library is loaded: not stale)
[ -f $out/${tag}_kernel_durations.json ] && cp $out/${tag}_kernel_durations.json profiles/kernel_durations.json
[ -f $out/${tag}_pmc_traffic.json ] && cp $out/${tag}_pmc_traffic.json profiles/pmc_traffic.json
# the dominant kernel with the chip filled the way the product fills it (bench.py `roofline`): rocprofv3 side
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $out/${tag}_conc -o conc -- python $root/tools/concurrent_kernel.py run convc2 4 3 > $out/${tag}_concurrent_convc2.txt 2>&1)
t=$(ls $out/${tag}_conc/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$t" ]; then
  python tools/concurrent_kernel.py summarise $t $out/${tag}_kernel_concurrent.json convc2 4 3 > /dev/null 2>&1 && cp $out/${tag}_kernel_concurrent.json profiles/kernel_concurrent.json
  cp $(ls $out/${tag}_conc/*kernel_stats.csv | head -1) $out/${tag}_concurrent_convc2_rocprofv3_kernel_stats.csv
fi
rm -rf $out/${tag}_conc
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_b4.log 2>&1
# round 6: the schedules of consecutive calls side by side, ONE PROCESS PER SETTING (streams map onto hardware queues in creation
# order: several settings in one process disturb each other) -- serial, one lane on three streams (round 5), 2 / 3 / 4 single-stream lanes
for s in serial 1:1 2:0 3:0 4:0 serial 3:0; do
  timeout 300 python tools/lanes_ab.py --batch 4 --rounds 3 $s 2>/dev/null | grep pairs
done > $out/${tag}_lanes_ab.txt 2>&1
for s in serial 3:0; do
  timeout 300 python tools/lanes_ab.py --batch 8 --rounds 3 $s 2>/dev/null | grep pairs
  timeout 300 python tools/lanes_ab.py --batch 1 --rounds 3 $s 2>/dev/null | grep pairs
done >> $out/${tag}_lanes_ab.txt 2>&1
for sc in none loop all; do
  RAFT_LANE_SHAPES=$sc timeout 300 python tools/lanes_ab.py --batch 4 --rounds 3 3:0 2>/dev/null | grep pairs | sed "s/^/RAFT_LANE_SHAPES=$sc /"
done >> $out/${tag}_lanes_ab.txt 2>&1
# the other BASELINE configurations (configs 1 / 3-per-GPU / 4, SmallRAFT) on both schedules, and the training step
timeout 900 python tools/config_bench.py 10 > $out/${tag}_config_bench.txt 2>&1
timeout 900 python bench.py --train --steps 10 --warmup 3 > $out/${tag}_bench_train.log 2>&1
# steady-state kernel timeline of the multi-lane schedule (queue occupancy, kernels in flight, stretch of each kernel under sharing)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d $out/${tag}_lt -o lt -- python $root/tools/lanes_ab.py --batch 4 --rounds 2 3:0 > /dev/null 2>&1)
t=$(ls $out/${tag}_lt/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$t" ] && python tools/pipeline_timeline.py $t "3 lanes, 4 pairs $tag" > $out/${tag}_lanes_timeline_b4.txt 2>&1
rm -rf $out/${tag}_lt
[ -z "${SKIP_PMC:-}" ] && bash tools/instruction_mix.sh $tag 4 > /dev/null 2>&1 && rm -rf $out/${tag}_mix
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $out/${tag}_prof -o bench -- python $root/bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_b4_under_rocprof.log 2>&1
f=$(ls $out/${tag}_prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $out/${tag}_bench_b4_rocprofv3_kernel_stats.csv
rm -rf $out/${tag}_prof
timeout 200 rocprofv3 --kernel-trace -f csv -d $out/${tag}_tl -o tl -- python $root/tools/graph_probe.py 4 3 > $out/${tag}_tl.log 2>&1
t=$(ls $out/${tag}_tl/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$t" ]; then python $root/tools/graph_trace.py $t "B=4 $tag" > $out/${tag}_loop_timeline_b4.txt 2>&1; python $root/tools/step_timeline.py $t all | awk 'NR>=72 && NR<=104' >> $out/${tag}_loop_timeline_b4.txt; fi
rm -rf $out/${tag}_tl
cd $root
[ -z "${SKIP_TESTS:-}" ] && tail -3 $out/${tag}_pytest_gpu.log && tail -2 $out/${tag}_smoke.log
tail -1 $out/${tag}_bench_b4.log | cut -c1-1500
cat $out/${tag}_kernel_durations.txt 2>/dev/null | head -40
