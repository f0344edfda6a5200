#!/bin/bash
# Per-kernel durations of the two encoders at B pairs under RAFT_ENC_WINO4 = 0 / 7 (GPU box, repo root).
tag=${1:-r07u}; b=${2:-4}
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp
out=$root/gpurun_out/${tag}_encoder_kernels_b$b.txt
: > $out
for m in 0 7; do
  RAFT_ENC_WINO4=$m timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $root/gpurun_out/${tag}_enc_$m -o enc -- python $root/tools/enc_probe.py $b 5 > $root/gpurun_out/${tag}_enc_$m.log 2>&1
  grep "fnet" $root/gpurun_out/${tag}_enc_$m.log >> $out
  f=$(ls $root/gpurun_out/${tag}_enc_$m/*kernel_stats.csv | head -1)
  python - "$f" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].replace('void ', '').replace('(anonymous namespace)::', '')
    if float(r['Percentage']) < 0.3: continue
    print(f"  {float(r['AverageNs'])/1e3:8.1f} us x{int(r['Calls']):4d}  {float(r['Percentage']):5.1f} %  {n[:90]}")
PY
  rm -f $root/gpurun_out/${tag}_enc_$m/*kernel_trace.csv
done
cat $out
