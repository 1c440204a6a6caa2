#!/bin/bash
# Legs with the pedestal removed, libraries interleaved: scripts/ab_libs_pedestal.sh <name> ...
cp pylbl_amd/liblbl_amd.so /tmp/orig.so
for round in 1 2 3; do
for args in "--farfield --pedestal" "--pedestal" "--banded --pedestal"; do
for name in "$@"; do
  cp build/liblbl_$name.so pylbl_amd/liblbl_amd.so
  python bench.py --steps 20 --warmup 5 --no-extras $args 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-8s round $round %-52s ms/step %.4f' % ('$name', '$args', d['ms_per_step']))"
done
done
done
for name in "$@"; do
  cp build/liblbl_$name.so pylbl_amd/liblbl_amd.so
  python scripts/perf_api.py 4 2>/dev/null | head -1 | sed "s/^/$name api: /"
done
cp /tmp/orig.so pylbl_amd/liblbl_amd.so
