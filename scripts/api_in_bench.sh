#!/bin/bash
# api_call of bench.py (torch initialised in the process) under different numbers of HIP hardware queues.
for q in "" 2 8 16 24; do
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; fi
  python bench.py --steps 5 --warmup 2 --extras api,farfield --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
a = d['api_call']['formats']
print('GPU_MAX_HW_QUEUES=%-3s step %.3f ms  api total %.3f gas %.3f all %.3f  farfield %.3f / %.3f' % ('$q', d['ms_per_step'], a['total']['ms_per_call'], a['gas']['ms_per_call'], a['all']['ms_per_call'], d['farfield_option']['plain']['ms_per_step'], d['farfield_option']['remove_pedestal']['ms_per_step']))"
done
