cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r12f_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r12f_pytest_gpu.log
tail -6 gpurun_out/r12f_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r12f_smoke.log 2>&1; echo "rc=$?" >> gpurun_out/r12f_smoke.log; tail -2 gpurun_out/r12f_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r12f_bench_b4.log 2>gpurun_out/r12f_bench_b4.err
tail -1 gpurun_out/r12f_bench_b4.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('value','ms_per_step','schedule','pairs_per_s_by_regime','final_iter_epe_conditioned','batch1_pairs_per_s','predict_step_pairs_per_s','pairs_per_s_at_8_pairs_per_gpu')})
print(d.get('roofline'))
print(d.get('stage_ms'))
"
tail -3 gpurun_out/r12f_bench_b4.err
