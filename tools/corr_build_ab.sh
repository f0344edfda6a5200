#!/bin/bash
# Volume build A/B (GPU box, repo root): rocprofv3 kernel duration and FETCH_SIZE / WRITE_SIZE of corr_gemm_kernel with the
# plain tile grid (RAFT_CORR_XCD=0) and the XCD-aware tile order (default), one forward at 4 and 8 pairs.
# usage: bash tools/corr_build_ab.sh <outfile>
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=${1:-$root/gpurun_out/corr_build_ab.txt}
: > $out
for x in ${XCD_VALUES:-0 1}; do
  for b in 4 8; do
    xx=$x
    d=/tmp/cb_${x}_$b; rm -rf $d; mkdir -p $d
    (cd /tmp && RAFT_CORR_XCD=$xx rocprofv3 --kernel-trace --stats -f csv -d $d/kt -o k -- python $root/tools/pmc_loop.py $b 2 > $d/kt.log 2>&1
     RAFT_CORR_XCD=$xx rocprofv3 --pmc FETCH_SIZE -f csv -d $d/f -o p -- python $root/tools/pmc_loop.py $b 2 > $d/f.log 2>&1
     RAFT_CORR_XCD=$xx rocprofv3 --pmc WRITE_SIZE -f csv -d $d/w -o p -- python $root/tools/pmc_loop.py $b 2 > $d/w.log 2>&1)
    f=$(ls $d/kt/*kernel_stats.csv 2>/dev/null | head -1)
    echo "RAFT_CORR_XCD=$x B=$b: $(grep corr_gemm $f | awk -F, '{print "calls", $2, "avg_ns", $4, "min_ns", $6, "max_ns", $7}')" >> $out
    python3 - $d >> $out <<'PY'
import csv, glob, sys
d = sys.argv[1]
for c, sub in (('FETCH_SIZE', 'f'), ('WRITE_SIZE', 'w')):
    v = [float(r['Counter_Value']) for p in glob.glob(f'{d}/{sub}/**/*counter_collection.csv', recursive=True)
         for r in csv.DictReader(open(p)) if 'corr_gemm' in r['Kernel_Name'] and r['Counter_Name'] == c]
    if v:
        print(f'    {c} raw KiB per launch {sum(v)/len(v):.0f}' + ('  (x2 = %.1f MB read)' % (2 * sum(v) / len(v) * 1024 / 1e6) if c == 'FETCH_SIZE' else '  (%.1f MB written)' % (sum(v) / len(v) * 1024 / 1e6)))
PY
  done
done
cat $out
