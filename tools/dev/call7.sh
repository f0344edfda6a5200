export TMPDIR=/tmp; mkdir -p gpurun_out
root=$(pwd)
cd /tmp
RAFT_OVERLAP=0 timeout 200 rocprofv3 --kernel-trace -f csv -d /tmp/tl0 -o tl -- python $root/tools/graph_probe.py 4 3 > /tmp/tl0.log 2>&1
t=$(ls /tmp/tl0/*kernel_trace.csv 2>/dev/null | head -1)
cd $root
python tools/step_timeline.py $t > gpurun_out/r10g_preloop_single_stream_b4.txt 2>&1
cd /tmp
timeout 200 rocprofv3 --kernel-trace -f csv -d /tmp/tl1 -o tl -- python $root/tools/graph_probe.py 4 3 > /tmp/tl1.log 2>&1
t=$(ls /tmp/tl1/*kernel_trace.csv 2>/dev/null | head -1)
cd $root
python tools/step_timeline.py $t > gpurun_out/r10g_preloop_product_b4.txt 2>&1
cat gpurun_out/r10g_preloop_single_stream_b4.txt | cut -c1-120
tail -3 gpurun_out/r10g_preloop_product_b4.txt
