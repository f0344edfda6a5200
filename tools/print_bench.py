"""Print the headline numbers of a bench.py log.  usage: python tools/print_bench.py <log>"""
import json
import sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('pairs/s', d['value'], 'ms/step', d['ms_per_step'], 'B=1 ms', d.get('batch1_ms_per_pair'), 'predict_step', d.get('predict_step_pairs_per_s'))
print('stage_us', {k: round(v * 1e3, 1) for k, v in d['stage_ms'].items()})
print('pre_loop_ms', d.get('pre_loop_ms'), 'update executed TF', d.get('update_block_executed_tflops'))
for k in ('roofline', 'roofline_corr_lookup', 'roofline_corr_build'):
    r = d.get(k, {})
    print(k, r.get('kernel'), r.get('achieved'), r.get('unit'), 'frac', r.get('frac'), 'traffic', r.get('traffic'))
if 'event_bracket_us' in d:
    print('event bracket us', d['event_bracket_us'], d.get('single_stream_iteration_ms'),
          'lookup frac of copy', d['roofline_corr_lookup'].get('frac_of_measured_copy'),
          'upsample frac of copy', d.get('roofline_upsample_convex', {}).get('frac_of_measured_copy'))
