export TMPDIR=/tmp
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_backward.py -q -x 2>&1 | tail -4
for p in torch hip torch hip; do RAFT_TRAIN_PACK=$p timeout 300 python bench.py --train --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('RAFT_TRAIN_PACK=$p', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step loss', d.get('loss_last_step'))"; done | tee $out/r11q_train_pack_ab.txt
