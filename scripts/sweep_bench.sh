#!/bin/bash
# Points-per-lane sweep of the default bench workload (one process per setting).
for p in 2 4 8; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --points-per-lane $p | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('P=$p', 'evals/s %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'acc ms/launch %.3f' % d['roofline']['avg_launch_ms'], d['kernel_ms_per_step'])"
done
