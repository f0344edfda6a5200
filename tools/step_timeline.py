"""Kernel-by-kernel timeline of the part of one forward call BEFORE the prediction loop (encoders, correlation pyramid, state
preparation), from a rocprofv3 --kernel-trace CSV of tools/graph_probe.py: start offset, duration, gap to the previous end, queue.
usage: python tools/step_timeline.py <kernel_trace.csv> [all]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '')))
rows.sort()
starts = [i for i, r in enumerate(rows) if 'enc_prep_kernel' in r[2]]
# two encoder calls (fnet, cnet) per forward, possibly on two streams: calls are separated by > 2 ms of loop kernels
calls = [starts[0]]
for i in starts[1:]:
    if rows[i][0] - rows[calls[-1]][0] > 3_000_000:
        calls.append(i)
call = rows[calls[-2]:calls[-1]]
t0 = call[0][0]
end = t0
whole = len(sys.argv) > 2
busy = {}
for s, e, n, q in call:
    if 'lookup' in n and not whole:
        break
    short = n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:58]
    print(f'{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - end) / 1e3:6.1f}  q{q}  {short}')
    busy[q] = busy.get(q, 0) + (e - s)
    end = max(end, e)
print('pre-loop wall', round((end - t0) / 1e3, 1), 'us; busy per queue', {q: round(v / 1e3, 1) for q, v in busy.items()})
