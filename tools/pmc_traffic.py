"""rocprofv3 --pmc CSVs of tools/pmc_traffic.sh -> HBM bytes per launch per stage (profiles/pmc_traffic.json).
usage: python tools/pmc_traffic.py <dir with FETCH_SIZE/ and WRITE_SIZE/> <batch> <out.json> [per_kernel.csv]

hbm_bytes_per_launch = 2 * FETCH_SIZE + WRITE_SIZE (both reported in KiB): MI355X_MICROARCH.md section HBM -- on gfx950
FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled; WRITE_SIZE matched the algorithmic write bytes of the
lookup / upsample / convolution kernels within 2 % in round 1 and is taken as reported.
Loop kernels are attributed by dispatch order: the single-stream loop launches bench.py's STAGES in order, starting at the
first corr_lookup dispatch; everything before it is the pre-loop part and is attributed by kernel name."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

STAGES = ['corr_lookup', 'convc1', 'convc2', 'convf1', 'convf2', 'conv', 'gru_zr1', 'gru_q1', 'gru_zr2',
          'gru_q2', 'fh1_mask0', 'fh2', 'mask2', 'upsample_convex']


def algorithmic_bytes_per_pair(h=56, w=64):
    M = h * w
    f = 4
    return {
        'corr_lookup': M * (4 * 100 * f + 8 + 324 * f),
        'upsample_convex': M * (576 * f + 8 + 64 * 2 * f),
        'convc1': M * (324 + 256) * f, 'convc2': M * (256 + 192) * f, 'convf1': M * (2 + 128) * f,
        'convf2': M * (128 + 64) * f, 'conv': M * (256 + 126) * f,
        'gru_zr1': M * (256 + 128 + 256 + 256) * f, 'gru_zr2': M * (256 + 128 + 256 + 256) * f,   # h, x, ctx in; z, r*h out
        'gru_q1': M * (256 + 128 + 128 + 128 + 128) * f, 'gru_q2': M * (256 + 128 + 128 + 128 + 128) * f,
        'fh1_mask0': M * (128 + 512) * f, 'fh2': M * (256 + 6) * f, 'mask2': M * (256 + 576) * f,
        'corr_build': 2 * M * 256 * f + sum((h >> l) * (w >> l) for l in range(4)) * M * f,
        'lookup_convc1_fused': M * (4 * 100 * f + 8 + 256 * f),      # footprints + coords in, cor1 out
        'mask_upsample_fused': M * (256 * f + 8 + 64 * 2 * f),       # mask.0's output + flow in, 8x8x2 flow out (the mask is never stored)
    }


def meta(raw_files):
    """Where and on what the numbers were taken: digest of the HIP sources + flags of the library that ran (bench.py compares it
    with its own and flags the file as stale otherwise), the commit, the raw files the summary was made from."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from tf_raft_amd import build
    try:
        head = subprocess.run(['git', '-C', root, 'rev-parse', '--short=12', 'HEAD'], capture_output=True, text=True).stdout.strip()
    except OSError:
        head = ''
    return {'source_digest': build.source_digest(), 'git_head': head or os.environ.get('RAFT_GIT_HEAD', ''), 'raw_files': raw_files}


def read(root, counter):
    rows = []
    for path in glob.glob(os.path.join(root, counter, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                if r['Counter_Name'] == counter:
                    rows.append((int(r['Dispatch_Id']), r['Kernel_Name'], float(r['Counter_Value'])))
    rows.sort()
    return rows


def attribute(rows):
    """{stage: [values]} -- loop stages by order after the first lookup dispatch, pre-loop kernels by name."""
    out = defaultdict(list)
    first = next((i for i, r in enumerate(rows) if 'corr_lookup' in r[1]), None)
    if first is None:
        raise SystemExit('no corr_lookup dispatch found')
    for _, name, v in rows[:first]:
        if 'corr_gemm' in name:
            out['corr_build'].append(v)
        elif 'fmap_' in name:
            out['corr_build_fmap_pyramid'].append(v)
        elif 'lookup_convc1' in name:
            out['lookup_convc1_fused'].append(v)
        elif 'mask_upsample' in name:
            out['mask_upsample_fused'].append(v)
    loop = rows[first:]
    last = max(i for i, r in enumerate(loop) if 'upsample_convex' in r[1])     # torch reductions of the caller follow
    loop = loop[:last + 1]
    assert len(loop) % len(STAGES) == 0, (len(loop), 'dispatches in the loop is not a multiple of the stage count')
    for i, (_, name, v) in enumerate(loop):
        st = STAGES[i % len(STAGES)]
        if st == 'corr_lookup':
            assert 'corr_lookup' in name, (i, name)
        if st == 'upsample_convex':
            assert 'upsample' in name, (i, name)
        out[st].append(v)
    return out


def main():
    root, batch, out_json = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    fetch = attribute(read(root, 'FETCH_SIZE'))
    write = attribute(read(root, 'WRITE_SIZE'))
    alg = algorithmic_bytes_per_pair()
    res = {'_comment': 'HBM bytes per launch from rocprofv3 --pmc passes on MI355X: tools/pmc_traffic.sh (FETCH_SIZE and '
                       f'WRITE_SIZE in separate passes over tools/pmc_loop.py: one forward at batch {batch} x 448x512, '
                       'single-stream loop, every kernel measured IN the loop; counters in KiB). hbm_bytes_per_launch = '
                       '2 * FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md HBM: gfx950 tallies 128-byte read requests at '
                       '64 bytes). Regenerate with: bash tools/pmc_traffic.sh <tag> 8'}
    res['_meta'] = meta([os.path.basename(sys.argv[4])] if len(sys.argv) > 4 else [])
    rows = []
    for st in STAGES + ['lookup_convc1_fused', 'mask_upsample_fused', 'corr_build', 'corr_build_fmap_pyramid']:
        if st not in fetch or st not in write:
            continue
        f = sum(fetch[st]) / len(fetch[st])
        wv = sum(write[st]) / len(write[st])
        hbm = (2 * f + wv) * 1024
        res[st] = {'batch': batch, 'in_loop': st in STAGES or st in ('lookup_convc1_fused', 'mask_upsample_fused'), 'fetch_size_kib_raw': round(f, 1), 'write_size_kib': round(wv, 1),
                   'hbm_bytes_per_launch': int(round(hbm)), 'launches_averaged': len(fetch[st]),
                   'algorithmic_bytes_per_pair': alg.get(st),
                   'algorithmic_bytes_per_launch': alg[st] * batch if st in alg else None,
                   'source': f'{os.path.basename(sys.argv[4]) if len(sys.argv) > 4 else ""} (tools/pmc_traffic.sh, in-loop --pmc passes at B={batch})'}
        rows.append([st, batch, round(f, 1), round(wv, 1), int(round(hbm)), alg[st] * batch if st in alg else '',
                     round(hbm / (alg[st] * batch), 3) if st in alg else ''])
    with open(out_json, 'w') as fh:
        json.dump(res, fh, indent=1)
    if len(sys.argv) > 4:
        with open(sys.argv[4], 'w', newline='') as fh:
            w = csv.writer(fh)
            w.writerow(['stage', 'batch', 'FETCH_SIZE_KiB_raw', 'WRITE_SIZE_KiB', 'hbm_bytes_per_launch',
                        'algorithmic_bytes_per_launch', 'traffic_over_algorithmic'])
            w.writerows(rows)
    for r in rows:
        print(*r)


if __name__ == '__main__':
    main()
