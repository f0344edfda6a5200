#!/bin/bash
# Round 6: texture-addresser / vector-L1 counters of the stand-alone pyramid lookup beside the L2 ones --
# is the gap to the moved-bytes floor address processing (TA), outstanding misses (TCP) or the fabric (TCC)?
# Each group is its own rocprofv3 pass (--pmc only, no tracing domains).  usage: bash tools/lookup_ta_tcp.sh <tag> [pairs=8]
tag=${1:-r12a}; B=${2:-8}
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${tag}_ta_tcp
mkdir -p $out
cd /tmp
rocprofv3 -L > $out/counters_all.txt 2>&1
grep -o -E "\b(TA|TCP|TD|TCC)_[A-Z0-9_]+\b" $out/counters_all.txt | sort -u > $out/counters_ta_tcp_td_tcc.txt
groups=(
 "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr"
 "GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
 "GRBM_GUI_ACTIVE TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum"
 "GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum"
 "GRBM_GUI_ACTIVE TA_FLAT_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum"
 "GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
 "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
 "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum"
 "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum"
 "GRBM_GUI_ACTIVE TCP_TA_TCP_STATE_READ_sum TCP_TD_TCP_STALL_CYCLES_sum"
 "GRBM_GUI_ACTIVE TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum"
 "GRBM_GUI_ACTIVE TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum"
 "GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_TC_STALL_sum"
 "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
 "GRBM_GUI_ACTIVE TCC_REQ_sum TCC_READ_sum"
 "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
 "GRBM_GUI_ACTIVE TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum"
 "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"
 "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU"
)
i=0
for g in "${groups[@]}"; do
  timeout 240 rocprofv3 --pmc $g --kernel-include-regex "corr_lookup" -f csv -d $out/pmc$i -o pmc -- python $root/tools/one_kernel.py lookup staged $B 30 > $out/pmc$i.log 2>&1
  echo "group $i [$g] rc=$?" >> $out/passes.txt
  i=$((i+1))
done
cd $root
python tools/pmc_summary.py $out $root/gpurun_out/${tag}_lookup_ta_tcp_b$B.csv
timeout 120 python tools/one_kernel.py lookup staged $B 200 > $root/gpurun_out/${tag}_lookup_standalone_b$B.txt 2>&1
cp $out/passes.txt $root/gpurun_out/${tag}_lookup_ta_tcp_passes.txt
cp $out/counters_ta_tcp_td_tcc.txt $root/gpurun_out/${tag}_counter_names.txt
tail -3 $out/pmc0.log
rm -rf $out/pmc*/ 
cat $root/gpurun_out/${tag}_lookup_ta_tcp_b$B.csv | head -5
