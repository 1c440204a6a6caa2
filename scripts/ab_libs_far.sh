#!/bin/bash
# Far-field leg with 4 and 8 points per lane, libraries interleaved: scripts/ab_libs_far.sh <name> ...
cp pylbl_amd/liblbl_amd.so /tmp/orig.so
for round in 1 2; do
for args in "--farfield --points-per-lane 8" "--farfield --points-per-lane 4"; do
for name in "$@"; do
  cp build/liblbl_$name.so pylbl_amd/liblbl_amd.so
  python bench.py --steps 20 --warmup 3 --no-extras $args 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-8s round $round %-42s ms/step %.4f accumulate ms/step %.4f' % ('$name', '$args', d['ms_per_step'], d['kernel_ms_per_step']['accumulate']))"
done
done
done
cp /tmp/orig.so pylbl_amd/liblbl_amd.so
