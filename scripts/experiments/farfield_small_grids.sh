#!/bin/bash
# configs[1] (0.01 cm-1) and configs[0] (0.1 cm-1) with the far-field series, per points per lane
# (0 = what pick_tiling chooses), next to the direct kernel.
for config in 1 0; do
  for mode in "" "--pedestal"; do
    python bench.py --steps 20 --warmup 5 --no-extras --config $config $mode 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('config $config direct        %-12s ms/step %.4f' % ('$mode', d['ms_per_step']))"
    for p in 0 1 2 4 8; do
      python bench.py --steps 20 --warmup 5 --no-extras --config $config --farfield $mode --points-per-lane $p 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('config $config far-field P=$p %-12s ms/step %.4f' % ('$mode', d['ms_per_step']))"
    done
  done
done
