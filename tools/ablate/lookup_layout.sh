#!/bin/bash
# Tile-shape study of the correlation maps.  usage: lookup_layout.sh build   (here, cross-compiled)
#                                                                       lookup_layout.sh run <outfile-prefix>   (GPU box)
cd "$(dirname "$0")"
SHAPES="2:3 3:2 1:4 2:2 3:3 1:3 0:3 0:5"     # log2(h):log2(w): 4x8 (product), 8x4, 2x16, 4x4, 8x8, 2x8, 1x8 (row-major, 32 B), 1x32
if [ "$1" = build ]; then
  for s in $SHAPES; do
    hl=${s%:*}; wl=${s#*:}
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -w -DRAFT_TILE_H_LOG2=$hl -DRAFT_TILE_W_LOG2=$wl -x hip lookup_layout.hip -o ablate_layout_$((1<<hl))x$((1<<wl)) &
  done; wait; ls -la ablate_layout_*
  exit 0
fi
out=${2:-/tmp/layout}
export TMPDIR=/tmp
{
for s in $SHAPES; do
  hl=${s%:*}; wl=${s#*:}; n=$((1<<hl))x$((1<<wl)); exe=$(pwd)/ablate_layout_$n
  for B in 4 8 16; do $exe $B 100 0; done
  $exe 4 50 1
done
} > ${out}_events.txt 2>&1
# rocprofv3: kernel durations (B = 8, 16) and the memory-side counters (B = 8: a 550 MB volume, beyond the Infinity Cache)
for s in $SHAPES; do
  hl=${s%:*}; wl=${s#*:}; n=$((1<<hl))x$((1<<wl)); exe=$(pwd)/ablate_layout_$n
  d=/tmp/layout_$n; rm -rf $d; mkdir -p $d
  (cd /tmp && for B in 8 16; do rocprofv3 --kernel-trace --stats -f csv -d $d/kt$B -o k -- $exe $B 50 0 > $d/kt$B.log 2>&1; done
   rocprofv3 --pmc FETCH_SIZE -f csv -d $d/fetch -o p -- $exe 8 20 0 > $d/fetch.log 2>&1
   rocprofv3 --pmc WRITE_SIZE -f csv -d $d/write -o p -- $exe 8 20 0 > $d/write.log 2>&1
   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum -f csv -d $d/tcc -o p -- $exe 8 20 0 > $d/tcc.log 2>&1
   rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr -f csv -d $d/tcp -o p -- $exe 8 20 0 > $d/tcp.log 2>&1)
  echo "== tile $n"
  for B in 8 16; do f=$(ls $d/kt$B/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && grep -E "corr_lookup|corr_gemm" $f | awk -F, -v B=$B '{print "B=" B, $1, "calls", $2, "avg_ns", $4}' | cut -c1-160; done
  python3 - $d <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        k = 'lookup' if 'corr_lookup' in r['Kernel_Name'] else ('gemm' if 'corr_gemm' in r['Kernel_Name'] else None)
        if k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    print(k, ' '.join(f'{c}={sum(v)/len(v):.1f}' for c, v in sorted(acc[k].items())))
PY
done > ${out}_rocprof.txt 2>&1
cat ${out}_events.txt; cat ${out}_rocprof.txt
