#!/bin/bash
# Which part of the configs[1] step (500 k points, tiles 2.56 cm-1 wide) is what: the diagnostic
# switches of accumulate_tile (1 general ranges off, 2 fast ranges off, 4 clipping lines off,
# 8 left-overs of the fast ranges off, 16 core lines off, 32 inner points off), launches run alone.
for ablate in 0 1 2 3 4 8 16 32 0; do
  python bench.py --config 1 --steps 50 --warmup 5 --no-extras --no-cpu-baseline --ablate $ablate 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('ablate %-3s ms/step %.4f  accumulate launch alone %.4f ms' % ('$ablate', d['ms_per_step'], r['avg_launch_ms']))"
done
