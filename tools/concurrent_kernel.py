"""Round 6: one F(4x4,3x3) layer of the update block launched on N streams at once under the product's launch-shape hint -- the
measurement behind bench.py's `roofline` when several loops are in flight (bench.concurrent_wino4_us).  Under rocprofv3
--kernel-trace the same run gives the kernel's average duration WHILE the chip is shared; with a second argument the trace is
turned into profiles/kernel_concurrent.json (the rocprofv3 side of that roofline object).
  python tools/concurrent_kernel.py run [stage=convc2] [pairs=4] [instances=3]
  python tools/concurrent_kernel.py summarise <kernel_trace.csv> <out.json> [stage] [pairs] [instances]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def run(stage, B, conc):
    import torch
    import bench
    import tf_raft_amd
    from tf_raft_amd import _dev, _ffi
    from tf_raft_amd import weights as wm
    model = tf_raft_amd.RAFT(iters_pred=2, weights=wm.init_weights('raft', seed=0))
    cin, cout, field = bench.W44_FIELDS[stage]
    h, w = 56, 64
    with _ffi.thread_concurrency(conc):
        us = bench.concurrent_wino4_us(model, _dev, _ffi, field, cin, cout, B, h, w, conc)
        one = bench.concurrent_wino4_us(model, _dev, _ffi, field, cin, cout, B, h, w, 1) if conc > 1 else us
    flops = 2.0 * B * h * w * 9 * cin * cout / 4.0
    print(f'{stage} pairs={B} instances={conc}: {us:.2f} us per launch (slowest stream) = {conc * flops / us / 1e6:.1f} TFLOP/s executed '
          f'({conc * flops / us / 1e6 / 157.3:.3f} of 157.3); one instance alone in the same shape {one:.2f} us = {flops / one / 1e6:.1f} TFLOP/s', flush=True)


def summarise(trace, out_json, stage, B, conc):
    from pmc_traffic import meta
    rows = []
    with open(trace) as fh:
        for r in csv.DictReader(fh):
            if 'conv_wino4_kernel' in r['Kernel_Name']:
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
    rows.sort()
    # the concurrent phase = the first 3 * conc warm-up + 30 * conc timed launches; the single-instance phase follows
    n_conc = (3 + 30) * conc
    durs = [(e - s) / 1e3 for s, e in rows[3 * conc:n_conc]]
    single = [(e - s) / 1e3 for s, e in rows[n_conc + 3:n_conc + 33]]
    wall = (max(e for _, e in rows[3 * conc:n_conc]) - min(s for s, _ in rows[3 * conc:n_conc])) / 1e3
    res = {'_comment': 'rocprofv3 --kernel-trace of tools/concurrent_kernel.py run: the F(4x4) layer on `instances` streams at once under '
                       'the launch-shape hint of the multi-lane schedule', '_meta': meta([os.path.basename(trace)]),
           stage: {'batch': B, 'instances': conc, 'avg_us': round(sum(durs) / len(durs), 3), 'min_us': round(min(durs), 3),
                   'max_us': round(max(durs), 3), 'launches': len(durs), 'wall_us_per_launch_per_stream': round(wall / 30, 3),
                   'single_instance_avg_us': round(sum(single) / len(single), 3) if single else None}}
    with open(out_json, 'w') as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res[stage]))


if __name__ == '__main__':
    a = sys.argv[1:]
    if a and a[0] == 'summarise':
        summarise(a[1], a[2], a[3] if len(a) > 3 else 'convc2', int(a[4]) if len(a) > 4 else 4, int(a[5]) if len(a) > 5 else 3)
    else:
        a = a[1:] if a and a[0] == 'run' else a
        run(a[0] if a else 'convc2', int(a[1]) if len(a) > 1 else 4, int(a[2]) if len(a) > 2 else 3)
