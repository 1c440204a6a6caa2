#!/bin/bash
# configs[0] / configs[1] rings of four asynchronous calls (bench.py's small_grid_options) under
# engine options, interleaved.  Usage on the GPU box: scripts/ab_small_lanes.sh "<opts A>" "<opts B>" ...
for round in 1 2; do
  for opts in "$@"; do
    args=""
    for o in $opts; do args="$args --engine-option $o"; done
    python bench.py --extras small --no-cpu-baseline --steps 3 --warmup 1 $args 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
s = d['small_grid_options']
print('round $round [%s] config0 %.2f us/call (%.3g evals/s)  config1 %.4f ms/step (%.3g)' % ('${opts:-defaults}', s['config0']['ms_per_step']*1e3, s['config0']['value'], s['config1']['ms_per_step'], s['config1']['value']))"
  done
done
