#!/bin/bash
# Vector-pipe instruction mix of the loop kernels at B pairs (default 4): two rocprofv3 --pmc passes over tools/pmc_loop.py plus the
# MFMA-only probe (calibration: does SQ_INSTS_VALU count MFMAs?), summarised by tools/instruction_mix.py.
# usage (GPU box, repo root):  bash tools/instruction_mix.sh <tag> [batch]
set -u
tag=${1:-mix}; batch=${2:-4}
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${tag}_mix
mkdir -p "$out"
cd /tmp
groups=(
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
)
i=0
for g in "${groups[@]}"; do
  timeout 600 rocprofv3 --pmc $g -f csv -d "$out/pmc$i" -o pmc -- python "$root/tools/pmc_loop.py" "$batch" 3 probe > "$out/pmc$i.log" 2>&1
  echo "pmc$i rc=$?" >> "$out/pmc$i.log"
  i=$((i+1))
done
cd "$root"
python tools/pmc_summary.py "$out" "gpurun_out/${tag}_instruction_mix_raw.csv"
python tools/instruction_mix.py "gpurun_out/${tag}_instruction_mix_raw.csv" > "gpurun_out/${tag}_instruction_mix.txt"
cat "gpurun_out/${tag}_instruction_mix.txt"
