#!/bin/bash
# Far-field parameter sweep: swaps prebuilt libraries in (build/liblbl_far_<terms>.so).
cp pylbl_amd/liblbl_amd.so /tmp/orig.so
for t in 21 27 17 33; do
  cp build/liblbl_far_$t.so pylbl_amd/liblbl_amd.so
  for p in 8 4; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --farfield --points-per-lane $p 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('terms $t P=$p', 'evals/s %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], {k: round(v, 3) for k, v in d['kernel_ms_per_step'].items()})"
  done
done
cp /tmp/orig.so pylbl_amd/liblbl_amd.so
