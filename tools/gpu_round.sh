#!/bin/bash
# One GPU-box round: parity tests, bench line, tile micro-benchmark, rocprofv3 kernel trace of the bench.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag> [steps...]   -> gpurun_out/<tag>_*
tag=${1:-r}; shift
what=${*:-pytest bench convbench prof}
export TMPDIR=/tmp
mkdir -p gpurun_out
for w in $what; do
  case $w in
    pytest)   timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_pytest_gpu.log; tail -4 gpurun_out/${tag}_pytest_gpu.log ;;
    bench)    timeout 600 python bench.py > gpurun_out/${tag}_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/${tag}_bench.log; tail -2 gpurun_out/${tag}_bench.log ;;
    convbench) timeout 300 python tools/conv_bench.py 4 > gpurun_out/${tag}_conv_bench.txt 2>&1; tail -14 gpurun_out/${tag}_conv_bench.txt ;;
    prof)     (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.log 2>&1)
              f=$(ls gpurun_out/${tag}_prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -25 "$f"
              t=$(ls gpurun_out/${tag}_prof/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$t" ] && python tools/rocpd_stats.py "$t" gpurun_out/${tag}_kernel_stats.csv > /dev/null 2>&1 && rm -f "$t" ;;
  esac
done
