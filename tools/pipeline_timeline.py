"""Kernel timeline of pipelined back-to-back calls from a rocprofv3 --kernel-trace CSV of tools/pipeline_ab.py-style runs:
per queue, which kernels ran; how much of the wall time had the loop queue busy; where encoder kernels sat relative to loop kernels.
usage: python tools/pipeline_timeline.py <kernel_trace.csv> [label]"""
import csv
import sys
from collections import Counter, defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60], r.get('Queue_Id', '')))
rows.sort()
label = sys.argv[2] if len(sys.argv) > 2 else ''
# steady state: the last 40 % of the trace
t_lo = rows[0][0] + int(0.6 * (rows[-1][1] - rows[0][0]))
ks = [r for r in rows if r[0] >= t_lo]
wall = (max(e for _, e, _, _ in ks) - ks[0][0]) / 1e3
ev = sorted([(s, 1) for s, _, _, _ in ks] + [(e, -1) for _, e, _, _ in ks])
busy = over = depth = 0
prev = ks[0][0]
for t, d in ev:
    if depth >= 1:
        busy += t - prev
    if depth >= 2:
        over += t - prev
    depth += d
    prev = t
calls = sum(1 for r in ks if 'enc_prep_kernel' in r[2]) / 2
print(f'{label}: steady-state window {wall:.1f} us, {calls:.1f} calls, {wall / max(calls, 1):.1f} us per call; union busy {busy / 1e3:.1f} us, '
      f'>= 2 kernels in flight {over / 1e3:.1f} us, sum of durations {sum(e - s for s, e, _, _ in ks) / 1e3:.1f} us')
per_q = defaultdict(Counter)
dur_q = Counter()
for s, e, n, q in ks:
    per_q[q][n] += 1
    dur_q[q] += e - s
for q in sorted(per_q):
    top = ', '.join(f'{n} x{c}' for n, c in per_q[q].most_common(6))
    print(f'  queue {q}: busy {dur_q[q] / 1e3:.1f} us ({dur_q[q] / 1e3 / wall:.2f} of the window): {top}')
# average duration per kernel name in the window (compare with the serial trace: stretch under overlap)
dur = defaultdict(list)
for s, e, n, q in ks:
    dur[n].append((e - s) / 1e3)
print('  avg us per kernel (count):', ', '.join(f'{n[:40]} {sum(v) / len(v):.1f} ({len(v)})' for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]))
