#!/bin/bash
# Tiling choices on the same box: points per lane x cell-aligned tiles, direct and far-field.
for round in 1 2; do
for mode in "" "--farfield" "--config 1" "--config 2"; do
for opts in "--points-per-lane 4" "--points-per-lane 8" "--points-per-lane 4 --engine-option aligned_tiles=1" "--points-per-lane 8 --engine-option aligned_tiles=1" "--points-per-lane 2"; do
  python bench.py --steps 10 --warmup 3 --no-extras $mode $opts 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('round $round %-12s %-58s ms/step %.4f' % ('$mode', '$opts', d['ms_per_step']))"
done
done
done
