#!/bin/bash
# A/B of prebuilt engine libraries (build/liblbl_<name>.so) on the default bench, interleaved.
cp pylbl_amd/liblbl_amd.so /tmp/orig.so
for round in 1 2 3; do
for name in "$@"; do
  cp build/liblbl_$name.so pylbl_amd/liblbl_amd.so
  python bench.py --steps 10 --warmup 3 --no-extras $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name round $round', 'evals/s %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'acc/launch %.3f' % d['roofline']['avg_launch_ms'])"
done
done
cp /tmp/orig.so pylbl_amd/liblbl_amd.so
