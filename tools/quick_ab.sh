export TMPDIR=/tmp
python tools/batch_probe_min.py 2>&1 | grep "B="
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/q_b4.log 2>&1; python tools/print_bench.py gpurun_out/q_b4.log | head -2
python bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/q_b1.log 2>&1; python tools/print_bench.py gpurun_out/q_b1.log | head -2
