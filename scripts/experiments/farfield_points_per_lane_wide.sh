#!/bin/bash
# Far-field + pedestal with 4 against 8 points per lane over more shapes of the job.
for round in 1 2; do
for args in "--levels-per-gpu 8 --profile standard" "--banded" "--config 2" "--levels-per-gpu 4 --output total"; do
for p in 8 4; do
  python bench.py --steps 10 --warmup 4 --no-extras --farfield --pedestal $args --points-per-lane $p 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('P=$p round $round %-44s ms/step %.4f' % ('$args', d['ms_per_step']))"
done
done
done
