"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv files under a directory) per kernel:
mean of each counter per dispatch.   python tools/pmc_summary.py <dir> [out.csv]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r['Kernel_Name'][:120]
            a = acc[k][r['Counter_Name']]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
rows = []
names = sorted({c for k in acc for c in acc[k]})
for k in sorted(acc):
    rows.append([k] + [round(acc[k][c][0] / acc[k][c][1], 1) if c in acc[k] else '' for c in names] +
                [max(v[1] for v in acc[k].values())])
out = open(sys.argv[2], 'w', newline='') if len(sys.argv) > 2 else sys.stdout
w = csv.writer(out)
w.writerow(['kernel'] + names + ['dispatches'])
w.writerows(rows)
