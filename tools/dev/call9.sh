export TMPDIR=/tmp; mkdir -p gpurun_out
root=$(pwd)
out=$root/gpurun_out/r10i_corr_stagger.txt; : > $out
for st in 0 31 51 81 32 52 82 0; do
  d=/tmp/cs_$st; rm -rf $d; mkdir -p $d
  (cd /tmp && RAFT_CORR_STAGGER=$st rocprofv3 --kernel-trace --stats -f csv -d $d/kt -o k -- python $root/tools/pmc_loop.py 4 2 > $d/kt.log 2>&1)
  f=$(ls $d/kt/*kernel_stats.csv 2>/dev/null | head -1)
  echo "RAFT_CORR_STAGGER=$st B=4: $(grep corr_gemm $f | awk -F, '{print "calls", $2, "avg_ns", $4, "min_ns", $6, "max_ns", $7}')" >> $out
done
cat $out
