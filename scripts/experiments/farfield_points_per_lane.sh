#!/bin/bash
# Far-field legs and the user-facing call with 4 against 8 points per lane (tiles of 250 against 500
# points on the 0.001 cm-1 grid), interleaved: bash scripts/experiments/farfield_points_per_lane.sh
for round in 1 2 3; do
for args in "--farfield --pedestal" "--farfield"; do
for p in 8 4; do
  python bench.py --steps 20 --warmup 5 --no-extras $args --points-per-lane $p 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('P=$p round $round %-24s ms/step %.4f' % ('$args', d['ms_per_step']))"
done
done
done
for p in 8 4 8 4; do
  POINTS_PER_LANE=$p python scripts/perf_api.py 4 2>/dev/null | head -1 | sed "s/^/P=$p api: /"
done
