#!/bin/bash
# PMC passes for one command (GPU box).  usage: tools/pmc_run.sh <outdir> <kernel-regex> -- <cmd...>
# Each counter group is collected in its own rocprofv3 run (--pmc must not be mixed with tracing domains).
set -u
out=$1; regex=$2; shift 3
export TMPDIR=/tmp
mkdir -p "$out"
groups=(
 "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"
 "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for g in "${groups[@]}"; do
  rocprofv3 --pmc $g --kernel-include-regex "$regex" -f csv -d "$out/pmc$i" -o pmc -- "$@" > "$out/pmc$i.log" 2>&1
  echo "pmc$i rc=$?" >> "$out/pmc$i.log"
  i=$((i+1))
done
