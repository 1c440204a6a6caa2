#!/bin/bash
# Legs with the pedestal removed, two builds of the library interleaved (build/liblbl_<name>.so):
#   bash scripts/experiments/ab_chain_forms.sh masks range
# (round 4: the relaxation with bit masks over the previous 256 runs against the form that walks the
# stretch of earlier runs itself)
cp pylbl_amd/liblbl_amd.so /tmp/orig.so
for round in 1 2 3; do
for args in "--farfield --pedestal" "--pedestal" "--banded --pedestal --farfield" "--config 0 --pedestal" "--config 1 --pedestal --farfield" "--farfield --pedestal --levels-per-gpu 8 --profile standard"; do
for name in "$@"; do
  cp build/liblbl_$name.so pylbl_amd/liblbl_amd.so
  python bench.py --steps 20 --warmup 5 --no-extras $args 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-8s round $round %-62s ms/step %.4f' % ('$name', '$args', d['ms_per_step']))"
done
done
done
for name in "$@" "$@"; do
  cp build/liblbl_$name.so pylbl_amd/liblbl_amd.so
  python scripts/perf_api.py 4 2>/dev/null | head -1 | sed "s/^/$name api: /"
done
cp /tmp/orig.so pylbl_amd/liblbl_amd.so
