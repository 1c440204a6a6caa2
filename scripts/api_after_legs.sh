#!/bin/bash
# api_call of bench.py after different preceding legs.
for legs in api sustained,api pedestal,api atmosphere,api banded,api small,api farfield,api continuum,api sustained,pedestal,atmosphere,banded,small,farfield,api; do
  python bench.py --steps 20 --warmup 5 --extras $legs --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
a = d['api_call']['formats']
print('%-70s api total %.3f gas %.3f all %.3f' % ('$legs', a['total']['ms_per_call'], a['gas']['ms_per_call'], a['all']['ms_per_call']))"
done
