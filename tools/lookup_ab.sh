#!/bin/bash
# A/B of the corr_lookup kernels on the GPU box: HIP-event timings over B, rocprofv3 kernel durations, PMC passes.
# usage (repo root on the GPU box): bash tools/lookup_ab.sh <tag> "<versions>" [pmc-version]
tag=${1:-lk}; vers=${2:-"v2 v3 v4"}; pmcv=${3:-}
export TMPDIR=/tmp
out=gpurun_out/${tag}; mkdir -p $out
for v in $vers; do
  for B in 1 4 8; do timeout 120 python tools/one_kernel.py lookup $v $B 200 2>/dev/null | tail -1; done
done | tee $out/events.txt
for v in $vers; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$out/prof_$v -o lk -- python $GRAFT_REPO_ROOT/tools/one_kernel.py lookup $v 4 200 > $GRAFT_REPO_ROOT/$out/prof_$v.log 2>&1)
  f=$(ls $out/prof_$v/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && grep -i "lookup" "$f" | head -3
  rm -f $out/prof_$v/*kernel_trace.csv $out/prof_$v/*agent_info.csv
done | tee $out/rocprof.txt
if [ -n "$pmcv" ]; then
  bash tools/pmc_run.sh $out/pmc 'corr_lookup' -- python tools/one_kernel.py lookup $pmcv 4 50
  python tools/pmc_summary.py $out/pmc $out/pmc_summary.csv; cat $out/pmc_summary.csv
  rm -rf $out/pmc/pmc*/
fi
