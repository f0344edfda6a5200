export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "lookup or corr_build or sampler" 2>&1 | tail -3
{ for i in 1 2; do for v in noreuse reuse; do for B in 4 8 16; do tools/ablate/ablate_layout_4x8_$v $B 100 0; done; tools/ablate/ablate_layout_4x8_$v 8 50 1; done; done; } > gpurun_out/r10c_lookup_row_reuse_ab.txt 2>&1; cat gpurun_out/r10c_lookup_row_reuse_ab.txt
timeout 400 bash tools/corr_build_ab.sh gpurun_out/r10c_corr_build_ab.txt
bash tools/pmc_run.sh gpurun_out/r10c_gemm_pmc 'corr_gemm' -- tools/ablate/ablate_layout_4x8_reuse 4 20 0 > /dev/null 2>&1; python tools/pmc_summary.py gpurun_out/r10c_gemm_pmc gpurun_out/r10c_gemm_pmc_summary.csv; cat gpurun_out/r10c_gemm_pmc_summary.csv; rm -rf gpurun_out/r10c_gemm_pmc
RAFT_EVENT_FENCE=0 timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "stream or rotating or graph or north_star_benchmarked or fused_mask" 2>&1 | tail -3
timeout 300 python tools/option_sweep.py --batch 4 --fresh-model "RAFT_EVENT_FENCE=0" "RAFT_EVENT_FENCE=1" 2>&1 | grep "B="
timeout 300 python tools/option_sweep.py --batch 8 --fresh-model "RAFT_EVENT_FENCE=0" "RAFT_EVENT_FENCE=1" 2>&1 | grep "B="
