"""rocprofv3 --kernel-trace CSVs of the single-stream loop (tools/pmc_loop.py at B = 4 and B = 8) -> average kernel duration per
stage -> profiles/kernel_durations.json, the rocprofv3 side of bench.py's roofline objects (bench.py measures the same launches
live with HIP events and reports how far the two are apart).
usage: python tools/kernel_durations.py <out.json> <batch>=<kernel_trace.csv> [<batch>=<kernel_trace.csv> ...]

Attribution as in tools/pmc_traffic.py: pmc_loop.py runs one forward with the product defaults (lookup fused into convc1,
mask.2 into the upsampling: attributed by kernel name) and one with the two-kernel stages, whose 14 launches per iteration come
in bench.py's STAGES order.  The file records the digest of the HIP sources it was measured on (_meta.source_digest)."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import STAGES, attribute, meta  # noqa: E402


def read_trace(path):
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
    rows.sort()
    return rows


def main():
    out_json = sys.argv[1]
    res = {'_comment': 'average rocprofv3 kernel durations (us) per stage of the SINGLE-STREAM prediction loop (tools/pmc_loop.py), '
                       'per batch size; regenerate with tools/closing_set.sh', '_meta': meta([a.split('=', 1)[1] for a in sys.argv[2:]])}
    res['_meta']['raw_files'] = [os.path.basename(f) for f in res['_meta']['raw_files']]
    for arg in sys.argv[2:]:
        batch, path = arg.split('=', 1)
        att = attribute(read_trace(path))
        res[f'b{int(batch)}'] = {st: {'avg_us': round(sum(v) / len(v), 3), 'min_us': round(min(v), 3), 'max_us': round(max(v), 3), 'launches': len(v)}
                                 for st, v in att.items() if v}
        for st in STAGES + ['lookup_convc1_fused', 'mask_upsample_fused', 'corr_build']:
            if st in res[f'b{int(batch)}']:
                print(batch, st, res[f'b{int(batch)}'][st])
    with open(out_json, 'w') as fh:
        json.dump(res, fh, indent=1)


if __name__ == '__main__':
    main()
