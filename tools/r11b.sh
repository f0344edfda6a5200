export TMPDIR=/tmp
out=gpurun_out; mkdir -p $out
timeout 120 tools/ablate/ablate_pipe_overlap > $out/r11b_pipe_overlap.txt 2>&1; cat $out/r11b_pipe_overlap.txt
timeout 900 python -m pytest tests -m gpu -q -x > $out/r11b_pytest_gpu.log 2>&1; echo "rc=$?" >> $out/r11b_pytest_gpu.log
tail -3 $out/r11b_pytest_gpu.log
for p in 0 1 0 1; do
  RAFT_PIPELINE=$p timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $out/r11b_bench_p$p.log 2>&1
  python - <<PY
import json
d=json.loads([l for l in open('$out/r11b_bench_p$p.log') if l.startswith('{')][-1])
print('RAFT_PIPELINE=$p', 'value', d['value'], 'ms/step', d['ms_per_step'], 'mfma', d.get('mfma_fp32_tflops_measured',{}).get('short_launch_from_idle'), d.get('mfma_fp32_tflops_measured',{}).get('sustained'), 'copy', d.get('hbm_copy_gbs_measured'), 'roofline', d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('frac_of_sustained_mfma'))
PY
done 2>&1 | tee $out/r11b_bench_ab.txt
rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | head -5
