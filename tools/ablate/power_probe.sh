#!/bin/bash
# usage (GPU box, repo root, after tools/ablate/lookup_layout.sh build and the ablate_gemm_<bits> builds): bash tools/ablate/power_probe.sh
# average power / clocks while one kernel variant loops for a few seconds (rocm-smi sampled every 0.2 s in the background)
export TMPDIR=/tmp
cd "$(dirname "$0")"
for a in 0 3 5 4; do
  ( ./ablate_gemm_$a 4 24000 0 > /tmp/gemm_$a.txt 2>&1 ) &
  pid=$!
  sleep 0.8
  : > /tmp/smi_$a.txt
  while kill -0 $pid 2>/dev/null; do /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" >> /tmp/smi_$a.txt; sleep 0.2; done
  wait $pid
  echo "== GEMM_ABL=$a: $(sed 's/.*build/build/' /tmp/gemm_$a.txt | cut -c1-18)"
  python3 - /tmp/smi_$a.txt <<'PY'
import re, sys, collections
acc = collections.defaultdict(list)
for line in open(sys.argv[1]):
    m = re.search(r'Power \(W\): ([0-9.]+)', line)
    if m:
        acc['socket power W'].append(float(m.group(1)))
    m = re.search(r'(sclk|mclk|fclk) clock level: \d+: \(([0-9]+)Mhz\)', line)
    if m:
        acc[m.group(1) + ' MHz'].append(float(m.group(2)))
for k, v in acc.items():
    print(f'   {k:28s} n={len(v):3d} mean {sum(v)/len(v):8.1f} min {min(v):8.1f} max {max(v):8.1f}')
PY
done
/opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | head -4
