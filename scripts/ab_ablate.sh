#!/bin/bash
# A/B of one diagnostic switch on the same library: scripts/ab_ablate.sh <mask> [rounds]
# prints ms/step and accumulate ms/step with --ablate 0 and --ablate <mask>, interleaved.
# (parts of the kernel can only be switched off in the diagnostics build: python -m pylbl_amd.build ablate)
export PYLBL_AMD_LIBRARY=$(pwd)/pylbl_amd/liblbl_amd_ablate.so
MASK=${1:-64}
for round in $(seq 1 ${2:-2}); do
for args in "--config target" "--config 1" "--config 2" "--levels-per-gpu 8 --profile standard" "--farfield"; do
for ablate in 0 $MASK; do
  python bench.py --steps ${STEPS:-10} --warmup 3 --no-extras $args --ablate $ablate 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ablate %-3s round $round %-42s ms/step %.4f accumulate ms/step %.4f' % ('$ablate', '$args', d['ms_per_step'], d['kernel_ms_per_step']['accumulate']))"
done
done
done
