"""Where a forward call spends its time, from a rocprofv3 --kernel-trace CSV of tools/graph_probe.py: the kernels of the LAST
call (found as the last gap > 0.5 ms between kernels... the calls are separated by host synchronisation), their union
busy time, the idle time between them, and how much of the time two or more kernels overlap.
usage: python tools/graph_trace.py <kernel_trace.csv> [label]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '')))
rows.sort()
# split into calls at idle gaps > 300 us
calls, cur, last_end = [], [], None
for s, e, n, q in rows:
    if last_end is not None and s - last_end > 300_000 and cur:
        calls.append(cur)
        cur = []
    cur.append((s, e, n, q))
    last_end = e if last_end is None else max(last_end, e)
if cur:
    calls.append(cur)
call = max(calls[-3:], key=len) if len(calls) >= 3 else calls[-1]
t0, t1 = call[0][0], max(e for _, e, _, _ in call)
ev = sorted([(s, 1) for s, _, _, _ in call] + [(e, -1) for _, e, _, _ in call])
busy = over = 0
depth, prev = 0, t0
for t, d in ev:
    if depth >= 1:
        busy += t - prev
    if depth >= 2:
        over += t - prev
    depth += d
    prev = t
label = sys.argv[2] if len(sys.argv) > 2 else ''
print(f'{label}: {len(calls)} calls in the trace; last call {len(call)} kernels on queues {sorted({q for *_, q in call})}: '
      f'wall {(t1 - t0) / 1e6:.3f} ms, union busy {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms, '
      f'>= 2 kernels in flight {over / 1e6:.3f} ms, sum of durations {sum(e - s for s, e, _, _ in call) / 1e6:.3f} ms')
# the 8 largest idle gaps and what follows them
gaps, end = [], call[0][1]
for s, e, n, q in call[1:]:
    if s > end:
        gaps.append((s - end, n[:60]))
    end = max(end, e)
gaps.sort(reverse=True)
print('  largest idle gaps (us, next kernel):', [(round(g / 1e3, 1), n) for g, n in gaps[:6]], ' total gaps', len(gaps),
      ' median gap us', round(sorted(g for g, _ in gaps)[len(gaps) // 2] / 1e3, 2) if gaps else 0)
