#!/bin/bash
# Lanes the asynchronous calls rotate over (engine option lanes), on the legs that use them.
for round in 1 2; do
for mode in "--pedestal" "--farfield --pedestal" "--levels-per-gpu 8 --profile standard --pedestal"; do
for lanes in 2 4 8; do
  python bench.py --steps 20 --warmup 5 --no-extras $mode --engine-option lanes=$lanes 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('round $round %-50s lanes $lanes ms/step %.4f' % ('$mode', d['ms_per_step']))"
done
done
done
for lanes in 2 4 8; do
python bench.py --steps 5 --warmup 2 --extras small --no-cpu-baseline --engine-option lanes=$lanes 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())['small_grid_options']
print('lanes $lanes in flight: config0 %.4f ms %.4g evals/s   config1 %.4f ms %.4g evals/s' % (d['config0']['ms_per_step'], d['config0']['value'], d['config1']['ms_per_step'], d['config1']['value']))"
done
