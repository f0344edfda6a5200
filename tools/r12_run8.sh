cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r12h_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r12h_pytest_gpu.log
tail -8 gpurun_out/r12h_pytest_gpu.log
