"""Where one forward call spends its time, from a rocprofv3 --kernel-trace CSV of tools/graph_probe.py.
Calls are delimited by their first kernel (`enc_prep_kernel`: two per forward, the first one opens a call); the LAST complete
call is analysed: wall time, union busy time, idle time, time with >= 2 kernels in flight -- for the whole call and for the
prediction loop alone (from the first lookup kernel on) -- the idle-gap distribution, and which kernels ran on which queue.
usage: python tools/graph_trace.py <kernel_trace.csv> [label]"""
import csv
import sys
from collections import Counter

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '')))
rows.sort()
label = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [i for i, r in enumerate(rows) if 'enc_prep_kernel' in r[2]][::2]
if len(starts) < 2:
    print(label, ': fewer than two calls in the trace')
    sys.exit(0)
call = rows[starts[-2]:starts[-1]]                      # the last COMPLETE call


def stats(ks):
    t0, t1 = ks[0][0], max(e for _, e, _, _ in ks)
    ev = sorted([(s, 1) for s, _, _, _ in ks] + [(e, -1) for _, e, _, _ in ks])
    busy = over = depth = 0
    prev = t0
    for t, d in ev:
        if depth >= 1:
            busy += t - prev
        if depth >= 2:
            over += t - prev
        depth += d
        prev = t
    gaps, end = [], ks[0][1]
    for s, e, n, q in ks[1:]:
        if s > end:
            gaps.append(s - end)
        end = max(end, e)
    gaps.sort()
    return dict(wall=(t1 - t0) / 1e6, busy=busy / 1e6, idle=(t1 - t0 - busy) / 1e6, over=over / 1e6,
                sum=sum(e - s for s, e, _, _ in ks) / 1e6, n=len(ks), gaps=len(gaps),
                gap_med=gaps[len(gaps) // 2] / 1e3 if gaps else 0.0, gap_p90=gaps[int(len(gaps) * 0.9)] / 1e3 if gaps else 0.0,
                gap_sum=sum(gaps) / 1e6)


first_loop = next(i for i, r in enumerate(call) if 'lookup' in r[2])
for name, ks in (('whole call', call), ('prediction loop', call[first_loop:])):
    s = stats(ks)
    print(f'{label} {name}: {s["n"]} kernels, wall {s["wall"]:.3f} ms, union busy {s["busy"]:.3f}, idle {s["idle"]:.3f} '
          f'({s["gaps"]} gaps, median {s["gap_med"]:.2f} us, p90 {s["gap_p90"]:.2f} us), >= 2 kernels in flight {s["over"]:.3f} ms, '
          f'sum of kernel durations {s["sum"]:.3f} ms')
# which kernel boundaries the idle gaps sit at: (kernel that ended last -> kernel that starts after the gap)
short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:34]   # noqa: E731
ctx, end, last = Counter(), call[first_loop][1], call[first_loop]
dur = {}
for r in call[first_loop + 1:]:
    if r[0] > end:
        key = (short(last[2]) + ' q' + last[3], short(r[2]) + ' q' + r[3])
        ctx[key] += 1
        dur[key] = dur.get(key, 0) + (r[0] - end)
    if r[1] > end:
        end, last = r[1], r
for key, c in ctx.most_common(8):
    print(f'  idle x{c:3d} mean {dur[key] / c / 1e3:5.2f} us  after {key[0]}  before {key[1]}')
q = Counter((r[3], r[2].split('(')[0].replace('void ', '')[:40]) for r in call[first_loop:])
by_queue = {}
for (queue, kn), c in q.items():
    by_queue.setdefault(queue, []).append(f'{kn} x{c}')
for queue in sorted(by_queue):
    print(f'  queue {queue}:', ', '.join(sorted(by_queue[queue]))[:600])
