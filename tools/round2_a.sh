#!/bin/bash
# Round-2 GPU call A: full GPU test suite, bench (B=4), B=1 graph A/B, PMC traffic in the loop at B=8, rocprofv3 traces.
tag=${1:-r05a}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -rA > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_pytest_gpu.log
grep -E "passed|failed|error" gpurun_out/${tag}_pytest_gpu.log | tail -3
grep -E "north-star|alternate corr 1024|stored volume 1024|ondemand-vs-oracle|free-running raft 448|first iteration above" gpurun_out/${tag}_pytest_gpu.log | head -40
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_b4.log 2>&1; echo "bench rc=$?" >> gpurun_out/${tag}_bench_b4.log
tail -2 gpurun_out/${tag}_bench_b4.log | cut -c1-3000
timeout 300 python tools/batch_sweep.py > gpurun_out/${tag}_batch_sweep.txt 2>&1; tail -12 gpurun_out/${tag}_batch_sweep.txt
bash tools/pmc_traffic.sh ${tag} 8 2>&1 | tail -20
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.log 2>&1)
f=$(ls gpurun_out/${tag}_prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -30 "$f"
rm -f gpurun_out/${tag}_prof/*kernel_trace.csv gpurun_out/${tag}_prof/*agent_info.csv
