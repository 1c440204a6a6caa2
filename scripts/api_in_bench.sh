#!/bin/bash
# api_call of bench.py, repeated, next to the stand-alone script on the same box.
for i in 1 2 3; do
  python bench.py --steps 5 --warmup 2 --extras api --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
a = d['api_call']['formats']
print('bench.py: step %.3f ms  api total %.3f gas %.3f all %.3f' % (d['ms_per_step'], a['total']['ms_per_call'], a['gas']['ms_per_call'], a['all']['ms_per_call']))"
  python scripts/perf_api.py 4 2>/dev/null | head -1
done
