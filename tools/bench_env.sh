#!/bin/bash
# Run bench.py once per value of an environment variable and print value / ms per step / selected stage times.
# usage: bash tools/bench_env.sh VAR "v1 v2 ..." [stage names...]
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v timeout 300 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms']
print('$var=$v', d['value'], d['ms_per_step'], ' '.join('%s=%.4f' % (k, s[k]) for k in sys.argv[1:]))" "$@"
done
