"""Where one RAFT.train_step spends its time, from a rocprofv3 --kernel-trace CSV of tools/train_probe.py: the LAST step (the kernel-name sequence
repeats from step to step) -- wall time, union busy time, and the kernels ranked by summed duration.
usage: python tools/train_trace.py <kernel_trace.csv> [top]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
names = [r[2] for r in rows]
period = next((P for P in range(50, len(names) // 2) if names[-P:] == names[-2 * P:-P]), None)   # steps launch the same kernels
if period is None:
    print('no repeating step found in', len(names), 'kernels')
    sys.exit(1)
step = rows[-period:]
t0, t1 = step[0][0], max(e for _, e, _ in step)
ev = sorted([(s, 1) for s, _, _ in step] + [(e, -1) for _, e, _ in step])
busy = depth = 0
prev = t0
for t, d in ev:
    if depth >= 1:
        busy += t - prev
    depth += d
    prev = t
print(f'last step: {len(step)} kernels, first-to-last kernel {(t1 - t0) / 1e6:.2f} ms, union busy {busy / 1e6:.2f} ms, '
      f'sum of durations {sum(e - s for s, e, _ in step) / 1e6:.2f} ms')
agg = defaultdict(lambda: [0, 0])
for s, e, n in step:
    k = n.replace('void ', '').replace('(anonymous namespace)::', '')[:100]
    agg[k][0] += e - s
    agg[k][1] += 1
for k, (ns, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f'{ns / 1e6:8.3f} ms  x{c:5d}  avg {ns / c / 1e3:8.1f} us  {k}')
