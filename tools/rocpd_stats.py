"""Summarise a rocprofv3 kernel trace (rocpd sqlite ``*_results.db`` or ``*_kernel_trace.csv``) into the
per-kernel statistics table that is committed under ``profiles/``.

    python tools/rocpd_stats.py <results.db | kernel_trace.csv> [out.csv]

Columns: name, calls, total_us, avg_us, min_us, max_us, pct, vgpr, lds_bytes, grid, workgroup
(the same quantities ``rocprofv3 --stats`` prints; kernel names are truncated to 160 characters).
"""
import csv
import sqlite3
import sys
from collections import defaultdict


def from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        'select name, end - start, vgpr_count + accum_vgpr_count, lds_size, grid_x * grid_y * grid_z, '
        'workgroup_x * workgroup_y * workgroup_z from kernels')
    return [(r[0], float(r[1]) / 1e3, r[2], r[3], r[4], r[5]) for r in rows]


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            grid = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0)
            wg = int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 0)) or 0)
            out.append((r['Kernel_Name'], dur, int(r.get('VGPR_Count', 0) or 0) + int(r.get('Accum_VGPR_Count', 0) or 0),
                        int(r.get('LDS_Block_Size', 0) or 0), grid, wg))
    return out


def summarise(rows):
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0, 0, 0, 0, 0])
    for name, dur, vgpr, lds, grid, wg in rows:
        a = agg[name]
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
        a[4], a[5], a[6], a[7] = vgpr, lds, grid, wg
    total = sum(a[1] for a in agg.values()) or 1.0
    table = []
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        table.append([name[:160], a[0], round(a[1], 1), round(a[1] / a[0], 2), round(a[2], 2), round(a[3], 2),
                      round(100 * a[1] / total, 2), a[4], a[5], a[6], a[7]])
    return table


def main():
    src = sys.argv[1]
    rows = from_db(src) if src.endswith('.db') else from_csv(src)
    table = summarise(rows)
    out = open(sys.argv[2], 'w', newline='') if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(['name', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct', 'vgpr', 'lds_bytes', 'grid', 'workgroup'])
    w.writerows(table)
    if out is not sys.stdout:
        out.close()


if __name__ == '__main__':
    main()
