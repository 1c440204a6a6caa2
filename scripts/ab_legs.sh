#!/bin/bash
# Interleaved A/B of two builds of the engine library on bench legs (GPU box, from the repo root):
#   scripts/ab_legs.sh <out> <variant .so> [rounds]     with EXTRAS="api,farfield,pedestal"
# A = pylbl_amd/liblbl_amd.so as it travelled, B = the variant (loaded through $PYLBL_AMD_LIBRARY).
# Prints, per round and library, the compact line's legs.
OUT=$1; VARIANT=$2; ROUNDS=${3:-3}
: > $OUT
for round in $(seq 1 $ROUNDS); do
  for which in main variant; do
    if [ $which = main ]; then unset PYLBL_AMD_LIBRARY; else export PYLBL_AMD_LIBRARY=$(realpath $VARIANT); fi
    python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --extras ${EXTRAS:-api,farfield,pedestal} --full-record /tmp/ab_full.json 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
legs = d.get('legs', {})
show = {k: (round(v['ms_per_step'], 4) if 'ms_per_step' in v else {a: round(b, 4) for a, b in v.items()}) for k, v in legs.items()}
print('round $round [$which] step %.4f ms' % d['ms_per_step'], show)" | tee -a $OUT
  done
done
unset PYLBL_AMD_LIBRARY
