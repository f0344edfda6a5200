cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r12g_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r12g_pytest_gpu.log
tail -8 gpurun_out/r12g_pytest_gpu.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r12g_conc -o conc -- python $GRAFT_REPO_ROOT/tools/concurrent_kernel.py run convc2 4 3 > $GRAFT_REPO_ROOT/gpurun_out/r12g_concurrent_convc2.txt 2>&1)
t=$(ls gpurun_out/r12g_conc/*kernel_trace.csv | head -1)
python tools/concurrent_kernel.py summarise $t gpurun_out/r12g_kernel_concurrent.json convc2 4 3
cp gpurun_out/r12g_kernel_concurrent.json profiles/kernel_concurrent.json
cp $(ls gpurun_out/r12g_conc/*kernel_stats.csv | head -1) gpurun_out/r12g_concurrent_convc2_rocprofv3_kernel_stats.csv
rm -rf gpurun_out/r12g_conc
grep -v amdgpu.ids gpurun_out/r12g_concurrent_convc2.txt | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r12g_bench_b4.log 2>gpurun_out/r12g_bench_b4.err
tail -1 gpurun_out/r12g_bench_b4.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('value','ms_per_step','pairs_per_s_by_regime','final_iter_epe_conditioned','batch1_pairs_per_s','predict_step_pairs_per_s','pairs_per_s_at_8_pairs_per_gpu')})
print(d.get('roofline'))
"
tail -3 gpurun_out/r12g_bench_b4.err
