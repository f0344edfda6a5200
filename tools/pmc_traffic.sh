#!/bin/bash
# HBM traffic per launch of every kernel of the forward pass, IN THE LOOP, at B = 8 (550 MB volume > the 256 MiB Infinity
# Cache): two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC
# slots") over tools/pmc_loop.py, then tools/pmc_traffic.py -> profiles/pmc_traffic.json.
# usage (GPU box, repo root):  bash tools/pmc_traffic.sh <tag> [batch]
set -u
tag=${1:-pmc}; batch=${2:-8}
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${tag}_pmc
mkdir -p "$out"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -f csv -d "$out/$c" -o pmc -- python "$root/tools/pmc_loop.py" "$batch" 3 > "$out/$c.log" 2>&1
  echo "$c rc=$?" >> "$out/$c.log"
done
cd "$root"
python tools/pmc_traffic.py "$out" "$batch" gpurun_out/${tag}_pmc_traffic.json "gpurun_out/${tag}_pmc_per_kernel.csv"
