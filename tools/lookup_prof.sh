#!/bin/bash
# rocprofv3 kernel durations of one lookup configuration.  usage: bash tools/lookup_prof.sh <outdir> <label> <version> [B]
# (environment variables such as RAFT_LOOKUP_PIPE_GRID pass through)
out=$1; label=$2; v=$3; B=${4:-4}
export TMPDIR=/tmp
mkdir -p $out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$out/prof_$label -o lk -- python $GRAFT_REPO_ROOT/tools/one_kernel.py lookup $v $B 200 > $GRAFT_REPO_ROOT/$out/prof_$label.log 2>&1)
f=$(ls $out/prof_$label/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && echo "$label B=$B: $(grep -i 'lookup' "$f" | head -1)" | tee -a $out/rocprof.txt
rm -rf $out/prof_$label
