#!/bin/bash
# A/B of engine options through bench.py flags on several workloads, interleaved.
# Usage on the GPU box: scripts/ab_options.sh "--polyquads 0" "--polyquads 1"
for round in $(seq 1 ${ROUNDS:-2}); do
for args in "" "--pedestal" "--config 1" "--config 2" "--levels-per-gpu 8 --profile standard" "--config 4 --levels-per-gpu 2 --profile standard"; do
for opt in "$@"; do
  python bench.py --steps ${STEPS:-10} --warmup 3 --no-extras $args $opt 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernel_ms_per_step']
print('%-16s round $round %-50s ms/step %.4f accumulate %.4f prep+sched %.4f' % ('$opt', '$args', d['ms_per_step'], k['accumulate'], k['prepare'] + k['schedule']))"
done
done
done
