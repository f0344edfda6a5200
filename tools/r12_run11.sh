cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
o=gpurun_out/r12k_concurrency_by_layer.txt
: > $o
for spec in "convc2 4 1" "convc2 4 2" "convc2 4 3" "convf2 4 3" "convf2 4 9" "conv 4 2" "conv 4 4" "fh1_mask0 4 1" "fh1_mask0 4 2" "convc2 8 1" "convc2 8 2"; do
  timeout 200 python tools/concurrent_kernel.py run $spec 2>/dev/null | grep instances >> $o
done
cat $o
# L2 hit / miss of convc2 alone against three instances sharing the chip
for n in 1 3; do
  for g in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $g --kernel-include-regex conv_wino4 -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r12k_pmc_n$n -o pmc -- python $GRAFT_REPO_ROOT/tools/concurrent_kernel.py run convc2 4 $n > /dev/null 2>&1)
  done
  python tools/pmc_summary.py gpurun_out/r12k_pmc_n$n gpurun_out/r12k_convc2_instances${n}_pmc.csv
  rm -rf gpurun_out/r12k_pmc_n$n
  cat gpurun_out/r12k_convc2_instances${n}_pmc.csv
done
