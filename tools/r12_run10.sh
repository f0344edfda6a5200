cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
RAFT_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 6 --warmup 2 > gpurun_out/r12j_bench_n2_gloo_one_gpu.log 2> gpurun_out/r12j_bench_n2_gloo_one_gpu.err
echo "rc=$?"
tail -1 gpurun_out/r12j_bench_n2_gloo_one_gpu.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('value','n_gpus','ms_per_step','schedule','scaling_efficiency','scaling_efficiency_basis','rank_step_ms','all_gather_us','one_gpu_same_shape_pairs_per_s','ranks_seen','final_iter_epe_conditioned','backend')})
"
tail -5 gpurun_out/r12j_bench_n2_gloo_one_gpu.err
