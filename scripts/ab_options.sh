#!/bin/bash
# A/B of bench.py flags (e.g. engine options) on several workloads, interleaved.
# Usage on the GPU box: scripts/ab_options.sh "--engine-option item_floor=512" "--engine-option item_floor=6000"
# WORKLOADS (newline-separated bench arguments) overrides the default list.
DEFAULT_WORKLOADS=$'\n--pedestal\n--banded\n--config 1\n--levels-per-gpu 8 --profile standard'
IFS=$'\n' read -r -d '' -a workloads <<< "${WORKLOADS-$DEFAULT_WORKLOADS}"
for round in $(seq 1 ${ROUNDS:-2}); do
for args in "${workloads[@]}"; do
for opt in "$@"; do
  python bench.py --steps ${STEPS:-10} --warmup 3 --no-extras $args $opt 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernel_ms_per_step']
print('%-34s round $round %-42s ms/step %.4f accumulate %.4f' % ('$opt', '$args', d['ms_per_step'], k['accumulate']))"
done
done
done
